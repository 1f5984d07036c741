#!/bin/bash
# (records a REMOVED switch: plonk_split_single_msm — slower, profiles/r05_split_single_msm_ab.log) r05 GPU call 8: SHPLONK's two lone commitments as 2 / 4 point-range parts through the batch API (plonk_split_single_msm) — whole proofs, k = 19 / 21 / 17
set -u
O=$PWD/gpurun_out/r05c08; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 3 - "plonk_split_single_msm=2" "plonk_split_single_msm=4" > $O/ab_k19.log 2>&1; tail -11 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "plonk_split_single_msm=2" "plonk_split_single_msm=4" > $O/ab_k21.log 2>&1; tail -8 $O/ab_k21.log
timeout 600 python tools/proof_configs_ab.py 18 2 1 1 17 2 - "plonk_split_single_msm=2" "plonk_split_single_msm=4" > $O/ab_k18.log 2>&1; tail -8 $O/ab_k18.log
