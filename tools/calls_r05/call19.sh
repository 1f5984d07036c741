#!/bin/bash
# r05 GPU call 19: transform times with warm clocks (tools/ntt_r04.py now runs 20 untimed repetitions first: the bench's block measured 0.52 ms for the 2^22
# transform the tool had reported at 0.58) — the tile kernel and the wave-owned radix-8 pass
set -u
O=$PWD/gpurun_out/r05c19; mkdir -p $O
timeout 600 python tools/ntt_r04.py ntt_w8=0:1:2 > $O/ntt_times_warm.log 2>&1; cat $O/ntt_times_warm.log | grep -v amdgpu
timeout 300 python tools/msm_r03.py 19,20 > $O/msm_breakdown.log 2>&1; tail -3 $O/msm_breakdown.log
