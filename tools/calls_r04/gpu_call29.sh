#!/bin/bash
# r04 GPU call 29: lanes of the batch MSM at k = 19 once more, on the final scheduling
mkdir -p gpurun_out/c29; O=gpurun_out/c29
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=msm_lanes:3,2 2>&1 | grep -E "create_proof min" > $O/l32.log
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=msm_lanes:3,4 2>&1 | grep -E "create_proof min" > $O/l34.log
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=msm_chunk:0,48 2>&1 | grep -E "create_proof min" > $O/chunk.log
cat $O/l32.log $O/l34.log $O/chunk.log
