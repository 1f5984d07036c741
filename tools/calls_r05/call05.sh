#!/bin/bash
# (msm_accum_waves was removed after this call: slower, profiles/r05_accum_two_waves_ab.log) r05 GPU call 5: the accumulation at two waves per SIMD (205 registers, no spills) with the sorts' LDS / workgroup size chosen so that another lane's
# histogram / scatter workgroups can co-reside with it — whole proofs at k = 19 and k = 21, batches of 4 MSMs
set -u
O=$PWD/gpurun_out/r05c05; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 3 - "msm_accum_waves=2" "msm_accum_waves=2,msm_scatter_full_lds=0" "msm_accum_waves=2,msm_scatter_full_lds=0,msm_sort_threads=512" "msm_scatter_full_lds=0" "msm_scatter_full_lds=0,msm_sort_threads=512" > $O/ab_k19.log 2>&1; cat $O/ab_k19.log | tail -20
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "msm_accum_waves=2" "msm_accum_waves=2,msm_scatter_full_lds=0,msm_sort_threads=512" > $O/ab_k21.log 2>&1; cat $O/ab_k21.log | tail -8
timeout 300 python tools/msm_r03.py 19,20 msm_accum_waves=3:2 > $O/msm_waves.log 2>&1; grep "2^" $O/msm_waves.log | cut -c1-200
