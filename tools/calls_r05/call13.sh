#!/bin/bash
# r05 GPU call 13: two schedule switches inside whole proofs — the grand products' lagrange_to_coeff in front of round 3's commitments (plonk_early_intt)
# and the lanes' first sorts one behind the other (msm_stagger_sorts)
set -u
O=$PWD/gpurun_out/r05c13; mkdir -p $O
timeout 700 python tools/proof_configs_ab.py 19 1 1 1 18 4 - "plonk_early_intt=1" "msm_stagger_sorts=1" "plonk_early_intt=1,msm_stagger_sorts=1" > $O/ab_k19.log 2>&1; tail -18 $O/ab_k19.log
timeout 500 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "plonk_early_intt=1" "msm_stagger_sorts=1" "plonk_early_intt=1,msm_stagger_sorts=1" > $O/ab_k21.log 2>&1; tail -10 $O/ab_k21.log
timeout 300 python tools/proof_configs_ab.py 17 4 1 1 16 4 - "plonk_early_intt=1" "msm_stagger_sorts=1" "plonk_early_intt=1,msm_stagger_sorts=1" > $O/ab_k17.log 2>&1; tail -18 $O/ab_k17.log
