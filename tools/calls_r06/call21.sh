#!/bin/bash
# r06 GPU call 21: lone MSMs (SHPLONK's W, W') on r05's long lanes (msm_chunk_lone = -1, default) against the batch rule (0): whole proofs
set -u
O=$PWD/gpurun_out/r06c21; mkdir -p $O
for k in "19 1 1 1 0 18 7" "17 4 1 1 0 16 7" "20 3 1 1 0 19 3"; do
  tag=$(echo $k | cut -d' ' -f1)
  timeout 500 python tools/prove_time.py $k --ab=msm_chunk_lone:0,-1 > $O/ab_k$tag.log 2>&1; echo "k=$tag"; grep "msm_chunk_lone=" $O/ab_k$tag.log
done
timeout 300 python tools/msm_r03.py 19,20 > $O/msm_breakdown.log 2>&1; grep "2^" $O/msm_breakdown.log
