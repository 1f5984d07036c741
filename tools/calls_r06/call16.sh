#!/bin/bash
# r06 GPU call 16: do r05's scheduling choices still hold now that the sort kernels fit beside accumulations?  staggered first sorts, lanes, early iNTT
set -u
O=$PWD/gpurun_out/r06c16; mkdir -p $O
for ab in msm_stagger_sorts:-1,1 msm_stagger_sorts:-1,0 msm_lanes:0,2 msm_lanes:0,4 plonk_early_intt:1,0 msm_seg:4,8; do
  n=$(echo $ab | tr ':,' '__')
  timeout 400 python tools/prove_time.py 19 1 1 1 0 18 5 --ab=$ab > $O/k19_$n.log 2>&1; grep "create_proof min" $O/k19_$n.log | head -4
done
for ab in msm_stagger_sorts:-1,0 msm_lanes:0,3; do
  n=$(echo $ab | tr ':,' '__')
  timeout 600 python tools/prove_time.py 21 2 1 1 0 20 3 --ab=$ab > $O/k21_$n.log 2>&1; grep "create_proof min" $O/k21_$n.log | head -4
done
