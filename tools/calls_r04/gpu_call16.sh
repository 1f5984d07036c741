#!/bin/bash
# r04 GPU call 16: the sort's offsets in one launch (msm_offsets_kernel): MSM parity tests, per-kernel times, whole proofs
mkdir -p gpurun_out/c16; O=gpurun_out/c16
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py tests/test_reference_shapes_golden.py -m gpu -q -x -k "msm or k19 or k16 or gpu0 or shape3 or wide or reference" > $O/pytest.log 2>&1
timeout 300 python tools/msm_r03.py 19,20 > $O/msm_breakdown.log 2>&1
timeout 300 python tools/prove_time.py 19 1 1 1 0 18 8 2>&1 | grep -E "create_proof rep|  [a-z_0-9]+ +[0-9.]+ ms" > $O/prove_k19.log
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 4 2>&1 | grep -E "create_proof rep" > $O/prove_k21.log
timeout 200 python tools/prove_time.py 15 17 3 1 0 14 6 2>&1 | grep -E "create_proof rep" > $O/prove_k15.log
tail -3 $O/pytest.log; cat $O/msm_breakdown.log | tail -30; cat $O/prove_k19.log $O/prove_k21.log $O/prove_k15.log
