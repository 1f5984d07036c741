#!/bin/bash
# r06 GPU call 9: workgroup size of the sort kernels (1024 / 512 / 256 lanes) in whole proofs, now that their LDS footprints fit beside accumulations
set -u
O=$PWD/gpurun_out/r06c09; mkdir -p $O
for k in "19 1 1 1 0 18 5" "21 2 1 1 0 20 3" "17 4 1 1 0 16 5"; do
  tag=$(echo $k | cut -d' ' -f1)
  timeout 500 python tools/prove_time.py $k --ab=msm_sort_threads:1024,512 > $O/ab_t512_k$tag.log 2>&1; grep "msm_sort_threads=" $O/ab_t512_k$tag.log
  timeout 500 python tools/prove_time.py $k --ab=msm_sort_threads:1024,256 > $O/ab_t256_k$tag.log 2>&1; grep "msm_sort_threads=" $O/ab_t256_k$tag.log
done
