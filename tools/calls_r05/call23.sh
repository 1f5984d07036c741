#!/bin/bash
# r05 GPU call 23: the accumulation without the per-boundary wait (next2 requested outside the branch), sorted entries one group ahead, and the variants that
# request the next TABLE entry one addition ahead (msm_accum_prefetch = 1: two waves per SIMD, 2: three)
set -u
O=$PWD/gpurun_out/r05c23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "msm" > $O/pytest_msm.log 2>&1; tail -2 $O/pytest_msm.log
for pf in 0 1 2; do timeout 300 python tools/msm_r03.py 19,20 pre:msm_accum_prefetch=$pf > $O/msm_pf$pf.log 2>&1; tail -2 $O/msm_pf$pf.log; done
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 4 - "msm_accum_prefetch=1" "msm_accum_prefetch=2" > $O/ab_k19.log 2>&1; tail -14 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "msm_accum_prefetch=1" "msm_accum_prefetch=2" > $O/ab_k21.log 2>&1; tail -8 $O/ab_k21.log
timeout 600 python tools/proof_configs_ab.py 17 4 1 1 16 3 - "msm_accum_prefetch=1" "msm_accum_prefetch=2" > $O/ab_k17.log 2>&1; tail -11 $O/ab_k17.log
