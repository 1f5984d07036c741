#!/bin/bash
# r06 GPU call 23: the auto chunk count (msm_sort_groups = 0) against r05's 32 in whole proofs and synchronous MSMs; 2^21: 64 chunks against 33 / 48
set -u
O=$PWD/gpurun_out/r06c23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_shapes_golden.py -m gpu -x -q -k "msm or ecdsa-19 or ecdsa-15 or pairing-21 or pairing-14" > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
timeout 400 python tools/msm_r03.py 19,20,21 msm_sort_groups=32:0 2>&1 | grep "2^" | cut -c1-120
timeout 400 python tools/msm_r03.py 21 msm_sort_groups=33:48:64 2>&1 | grep "2^" | cut -c1-120
run() { n=$(echo "$2" | tr ':,' '__'); tag=$(echo $1 | cut -d' ' -f1); timeout 600 python tools/prove_time.py $1 --ab=$2 > $O/k${tag}_$n.log 2>&1; echo "k=$tag"; grep "create_proof min" $O/k${tag}_$n.log | head -4; }
run "19 1 1 1 0 18 5" msm_sort_groups:32,0
run "21 2 1 1 0 20 3" msm_sort_groups:33,0
run "22 1 1 1 0 21 2" msm_sort_groups:65,0
run "17 4 1 1 0 16 5" msm_sort_groups:32,0
run "20 3 1 1 0 19 3" msm_sort_groups:32,0
run "18 2 1 1 0 17 5" msm_sort_groups:32,0
run "15 17 3 1 0 14 5" msm_sort_groups:32,0
