#!/bin/bash
# r06 GPU call 18: entries per lane (msm_chunk) and the window size re-checked in whole proofs now that the sorts run beside the accumulations
set -u
O=$PWD/gpurun_out/r06c18; mkdir -p $O
for ab in msm_chunk:0,24 msm_chunk:0,48 msm_chunk:0,68; do
  n=$(echo $ab | tr ':,' '__')
  timeout 400 python tools/prove_time.py 19 1 1 1 0 18 5 --ab=$ab > $O/k19_$n.log 2>&1; grep "create_proof min" $O/k19_$n.log | head -4
done
for c in 15 16 14; do
  timeout 400 python tools/prove_time.py 19 1 1 1 0 18 7 --param=msm_window_bits=$c > $O/k19_c$c.log 2>&1; echo "k=19 c=$c: $(grep 'create_proof rep' $O/k19_c$c.log | awk '{print $4}' | sort -n | head -4 | tr '\n' ' ')"
done
for c in 16 15; do
  timeout 600 python tools/prove_time.py 21 2 1 1 0 20 4 --param=msm_window_bits=$c > $O/k21_c$c.log 2>&1; echo "k=21 c=$c: $(grep 'create_proof rep' $O/k21_c$c.log | awk '{print $4}' | sort -n | head -3 | tr '\n' ' ')"
done
