#!/bin/bash
# r05 GPU call 9: the whole -m gpu suite on the round's code (incl. the new golden shapes), then the big golden shapes (H2HIP_GOLDEN_BIG=1: the
# 24-column k = 21 lines and whatever k = 23 / 24 entries the file holds), smoke
set -u
O=$PWD/gpurun_out/r05c09; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
( time H2HIP_GOLDEN_BIG=1 timeout 1500 python -m pytest tests/test_reference_shapes_golden.py -m gpu -q -k "msm-L12 or fixed_msm-L11 or msm-L8 or msm-L9 or msm-L13" --durations=6 ) > $O/pytest_big.log 2>&1; tail -12 $O/pytest_big.log
