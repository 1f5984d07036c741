#!/bin/bash
# r06 GPU call 14: the packed histogram at 2^21 points (chunks of 65535): parity at k = 21 / 22 shapes, A/B in whole proofs
set -u
O=$PWD/gpurun_out/r06c14; mkdir -p $O
timeout 900 python -m pytest tests/test_reference_shapes_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "pairing-21 or pairing-22 or msm-L7 or msm_g1 or msm_matches or closed_form" > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
for k in "21 2 1 1 0 20 3" "22 1 1 1 0 21 2" "20 3 1 1 0 19 3"; do
  tag=$(echo $k | cut -d' ' -f1)
  timeout 600 python tools/prove_time.py $k --ab=msm_hist_packed:0,1 > $O/ab_hist_k$tag.log 2>&1; grep "msm_hist_packed=" $O/ab_hist_k$tag.log
done
