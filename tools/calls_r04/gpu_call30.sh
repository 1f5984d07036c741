#!/bin/bash
# r04 GPU call 30: random shapes with the selectable paths drawn as well (arithmetic forms, scheduling switches, lanes, NTT kernels)
mkdir -p gpurun_out/c30
H2HIP_FUZZ_KNOBS=1 timeout 300 python tools/fuzz_shapes.py 200 7 > gpurun_out/c30/fuzz_knobs.log 2>&1
H2HIP_FUZZ_KNOBS=1 timeout 200 python tools/fuzz_shapes.py 100 8 13 16 > gpurun_out/c30/fuzz_knobs_mid.log 2>&1
tail -2 gpurun_out/c30/fuzz_knobs.log; tail -2 gpurun_out/c30/fuzz_knobs_mid.log
