#!/bin/bash
# r06 GPU call 11: EVAL_J = 32 (the batched evaluation's tile tree amortised over 4x the coefficients) and kate_coeffs_per_lane 4 vs 8 / 16 in whole proofs
set -u
O=$PWD/gpurun_out/r06c11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py -m gpu -x -q -k "eval or kate or k19 or full_range_field_inputs_pointwise" > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
bash tools/quotient_pmc.sh > $O/quotient_pmc.md 2>&1; grep "fr_eval" $O/quotient_pmc.md
for k in "19 1 1 1 0 18 5" "21 2 1 1 0 20 3" "15 17 3 1 0 14 5"; do
  tag=$(echo $k | cut -d' ' -f1)
  timeout 500 python tools/prove_time.py $k --ab=kate_coeffs_per_lane:4,8 > $O/ab_kate8_k$tag.log 2>&1; grep "kate_coeffs_per_lane=" $O/ab_kate8_k$tag.log
done
timeout 500 python tools/prove_time.py 19 1 1 1 0 18 5 --ab=kate_coeffs_per_lane:0,8 > $O/ab_kate08_k19.log 2>&1; grep "kate_coeffs_per_lane=" $O/ab_kate08_k19.log
