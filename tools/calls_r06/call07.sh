#!/bin/bash
# r06 GPU call 7: can the scatter slip in beside the accumulations?  With its full 128 KiB of LDS a scatter workgroup needs a CU without any accumulation
# workgroup (3 x 36 KiB + 128 > 160); with only its cursors (B / S counters: 32 KiB at S = 2) it fits into the slot one retiring accumulation workgroup leaves
set -u
O=$PWD/gpurun_out/r06c07; mkdir -p $O
for k in "19 1 1 1 0 18 5" "21 2 1 1 0 20 3" "17 4 1 1 0 16 5"; do
  tag=$(echo $k | cut -d' ' -f1)
  timeout 500 python tools/prove_time.py $k --ab=msm_scatter_full_lds:1,0 > $O/ab_full_k$tag.log 2>&1; grep "msm_scatter_full_lds=" $O/ab_full_k$tag.log
  timeout 500 python tools/prove_time.py $k --param=msm_scatter_split=2 --ab=msm_scatter_full_lds:1,0 > $O/ab_full_s2_k$tag.log 2>&1; grep "msm_scatter_full_lds=" $O/ab_full_s2_k$tag.log
done
