#!/bin/bash
# r04 GPU call 6: cheap proof-level A/Bs with the tail overlap in place
mkdir -p gpurun_out/c6; O=gpurun_out/c6
for ab in msm_lanes:3,2 msm_lanes:3,4 msm_chunk:0,24 msm_chunk:0,48 msm_seg:4,8 msm_defer_reduce:1,0; do
  timeout 200 python tools/prove_time.py 19 1 1 1 0 18 4 --ab=$ab 2>&1 | grep "create_proof min" >> $O/ab19.log
done
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 10 --param=msm_window_bits=16 2>&1 | grep "create_proof rep" | tail -6 > $O/c16.log
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 10 2>&1 | grep "create_proof rep" | tail -6 > $O/c15.log
cat $O/ab19.log; echo c16; cat $O/c16.log; echo c15; cat $O/c15.log
