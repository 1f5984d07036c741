#!/bin/bash
# r05 GPU call 26: the accumulation with its products issued as side-by-side pairs (msm_accum_pair = 1: two waves per SIMD, 2: three)
set -u
O=$PWD/gpurun_out/r05c26; mkdir -p $O
for pf in 0 1 2; do timeout 300 python tools/msm_r03.py 19,20 pre:msm_accum_pair=$pf > $O/msm_pair$pf.log 2>&1; tail -2 $O/msm_pair$pf.log; done
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 4 - "msm_accum_pair=1" "msm_accum_pair=2" > $O/ab_k19.log 2>&1; tail -14 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "msm_accum_pair=1" "msm_accum_pair=2" > $O/ab_k21.log 2>&1; tail -8 $O/ab_k21.log
