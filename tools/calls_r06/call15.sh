#!/bin/bash
# r06 GPU call 15: advice columns uploaded inside round 1's commitment batch (plonk_lazy_upload): parity on the golden shapes, A/B with host-resident advice
set -u
O=$PWD/gpurun_out/r06c15; mkdir -p $O
timeout 1200 python -m pytest tests/test_reference_shapes_golden.py tests/test_plonk_prover.py -m gpu -x -q > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
for k in "21 2 1 1 0 20 3" "18 13 2 1 0 17 5" "20 3 1 1 0 19 3" "15 105 14 1 0 14 5" "16 8 2 1 0 15 5"; do
  tag=$(echo $k | cut -d' ' -f1-2 | tr ' ' '_')
  timeout 600 python tools/prove_time.py $k --ab=plonk_lazy_upload:0,1 > $O/ab_lazy_k$tag.log 2>&1; grep "plonk_lazy_upload=" $O/ab_lazy_k$tag.log
done
