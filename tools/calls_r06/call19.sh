#!/bin/bash
# r06 GPU call 19: entries per lane (msm_chunk), finer: k = 19 / 21 / 17 / 20
set -u
O=$PWD/gpurun_out/r06c19; mkdir -p $O
for ab in msm_chunk:0,16 msm_chunk:0,20 msm_chunk:0,28 msm_chunk:24,20; do
  n=$(echo $ab | tr ':,' '__')
  timeout 400 python tools/prove_time.py 19 1 1 1 0 18 5 --ab=$ab > $O/k19_$n.log 2>&1; grep "create_proof min" $O/k19_$n.log | head -4
done
for ab in msm_chunk:0,48 msm_chunk:0,32; do
  n=$(echo $ab | tr ':,' '__')
  timeout 600 python tools/prove_time.py 21 2 1 1 0 20 3 --ab=$ab > $O/k21_$n.log 2>&1; grep "create_proof min" $O/k21_$n.log | head -4
done
for ab in msm_chunk:0,24 msm_chunk:0,16; do
  n=$(echo $ab | tr ':,' '__')
  timeout 400 python tools/prove_time.py 17 4 1 1 0 16 5 --ab=$ab > $O/k17_$n.log 2>&1; grep "create_proof min" $O/k17_$n.log | head -4
  timeout 400 python tools/prove_time.py 20 3 1 1 0 19 3 --ab=$ab > $O/k20_$n.log 2>&1; grep "create_proof min" $O/k20_$n.log | head -4
done
