#!/bin/bash
# r06 GPU call 20: entries per lane, pinning the rule: k = 20 (24 / 32 / 48), k = 19 (28 / 32), k = 17 (24 / 28), k = 18, k = 21 (56)
set -u
O=$PWD/gpurun_out/r06c20; mkdir -p $O
run() { n=$(echo "$2" | tr ':,' '__'); tag=$(echo $1 | cut -d' ' -f1); timeout 600 python tools/prove_time.py $1 --ab=$2 > $O/k${tag}_$n.log 2>&1; echo "k=$tag"; grep "create_proof min" $O/k${tag}_$n.log | head -4; }
run "20 3 1 1 0 19 3" msm_chunk:24,32
run "20 3 1 1 0 19 3" msm_chunk:24,48
run "19 1 1 1 0 18 5" msm_chunk:28,32
run "19 1 1 1 0 18 5" msm_chunk:28,24
run "17 4 1 1 0 16 5" msm_chunk:24,28
run "18 2 1 1 0 17 5" msm_chunk:0,24
run "18 2 1 1 0 17 5" msm_chunk:0,28
run "21 2 1 1 0 20 3" msm_chunk:0,56
run "16 8 2 1 0 15 5" msm_chunk:0,24
run "15 17 3 1 0 14 5" msm_chunk:0,24
