#!/bin/bash
# r06 GPU call 29: a longer fuzz of random shapes against the oracle prover with NEW seeds, the knob fuzz now drawing the r06 switches too
set -u
O=$PWD/gpurun_out/r06c29; mkdir -p $O
timeout 500 python tools/fuzz_shapes.py 400 11 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
H2HIP_FUZZ_KNOBS=1 timeout 700 python tools/fuzz_shapes.py 600 12 > $O/fuzz_knobs.log 2>&1; tail -1 $O/fuzz_knobs.log
H2HIP_FUZZ_KNOBS=1 timeout 500 python tools/fuzz_shapes.py 400 14 13 16 > $O/fuzz_knobs_mid.log 2>&1; tail -1 $O/fuzz_knobs_mid.log
timeout 500 python tools/fuzz_shapes.py 400 13 13 17 > $O/fuzz_mid.log 2>&1; tail -1 $O/fuzz_mid.log
