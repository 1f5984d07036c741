#!/bin/bash
# r06 GPU call 26: the histogram over bucket sub-ranges (msm_hist_split: 1 = whole window, 0 = auto: 2 at c = 16) in whole proofs at k = 20 / 21 / 22 (k = 19: auto = 1)
set -u
O=$PWD/gpurun_out/r06c26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_shapes_golden.py -m gpu -x -q -k "msm or pairing-21 or pairing-20 or pairing-22" > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
run() { n=$(echo "$2" | tr ':,' '__'); tag=$(echo $1 | cut -d' ' -f1); timeout 600 python tools/prove_time.py $1 --ab=$2 > $O/k${tag}_$n.log 2>&1; echo "k=$tag"; grep "create_proof min" $O/k${tag}_$n.log | head -4; }
run "21 2 1 1 0 20 3" msm_hist_split:1,0
run "20 3 1 1 0 19 3" msm_hist_split:1,0
run "22 1 1 1 0 21 2" msm_hist_split:1,0
run "21 2 1 1 0 20 3" msm_hist_split:1,4
run "19 1 1 1 0 18 5" msm_hist_split:1,2
timeout 400 python tools/msm_r03.py 20,21 msm_hist_split=1:0:1:0 2>&1 | grep "2^" | cut -c1-200
