#!/bin/bash
# r05 GPU call 7: the permutation's and the lookups' grand products in one pass (plonk_merge_products) — whole proofs at k = 19 (1 + 1 columns) and
# k = 16 / 17 single-column variants; plus the prover parity tests on the final code
set -u
O=$PWD/gpurun_out/r05c07; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 3 - "plonk_merge_products=0" > $O/ab_k19.log 2>&1; tail -8 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 17 1 1 1 16 3 - "plonk_merge_products=0" > $O/ab_k17.log 2>&1; tail -8 $O/ab_k17.log
timeout 900 python -m pytest tests/test_plonk_prover.py tests/test_reference_shapes_golden.py -m gpu -x -q -k "not pairing-22 and not ecdsa-11" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
