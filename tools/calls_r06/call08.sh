#!/bin/bash
# r06 GPU call 8: the packed (16-bit pairs) LDS histogram A/B in whole proofs, MSM parity, the per-kernel table of one proof
set -u
O=$PWD/gpurun_out/r06c08; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py -m gpu -x -q -k "msm or k19 or random_shapes" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for k in "19 1 1 1 0 18 5" "20 3 1 1 0 19 3" "17 4 1 1 0 16 5" "21 2 1 1 0 20 3"; do
  tag=$(echo $k | cut -d' ' -f1)
  timeout 500 python tools/prove_time.py $k --ab=msm_hist_packed:0,1 > $O/ab_hist_k$tag.log 2>&1; grep "msm_hist_packed=" $O/ab_hist_k$tag.log
done
timeout 300 python tools/msm_r03.py 19,20 > $O/msm_breakdown.log 2>&1; cat $O/msm_breakdown.log
