#!/bin/bash
# r04 GPU call 12: range-sharded SHPLONK / grand products on the GPU (one rank forced), kate apply kernel without scratch: tile length sweep
mkdir -p gpurun_out/c12; O=gpurun_out/c12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_single_rank.py tests/test_reference_shapes_golden.py tests/test_plonk_prover.py -m gpu -q -x > $O/pytest.log 2>&1
for j in 0 4 2; do
  timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --param=kate_coeffs_per_lane=$j 2>&1 | grep -E "create_proof rep|multiopen|evaluations" | sed "s/^/k19 J=$j: /" >> $O/kate_j.log
done
for j in 0 4; do
  timeout 300 python tools/prove_time.py 21 2 1 1 0 20 4 --param=kate_coeffs_per_lane=$j 2>&1 | grep -E "create_proof rep|multiopen|evaluations" | sed "s/^/k21 J=$j: /" >> $O/kate_j.log
done
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=kate_coeffs_per_lane:8,4 2>&1 | grep "create_proof min" > $O/kate_ab.log
timeout 200 python tools/invert_sweep.py > $O/invert_sweep.log 2>&1
tail -3 $O/pytest.log; cat $O/kate_j.log $O/kate_ab.log $O/invert_sweep.log
