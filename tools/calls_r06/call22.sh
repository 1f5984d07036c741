#!/bin/bash
# r06 GPU call 22: chunks per window of the counting sort: 17 windows x 32 chunks = 544 workgroups of 1024 lanes against 512 slots (2 per CU) is 1.06 rounds;
# 30 chunks = 510 workgroups is one
set -u
O=$PWD/gpurun_out/r06c22; mkdir -p $O
timeout 400 python tools/msm_r03.py 19,20 msm_sort_groups=32:30:28:24:16 2>&1 | grep "2^\|groups"
for ab in msm_sort_groups:0,30 msm_sort_groups:0,24 msm_sort_groups:0,15; do
  n=$(echo $ab | tr ':,' '__')
  timeout 400 python tools/prove_time.py 19 1 1 1 0 18 5 --ab=$ab > $O/k19_$n.log 2>&1; grep "create_proof min" $O/k19_$n.log | head -4
done
