#!/bin/bash
# r06 GPU call 13: r05's final tree (commit 7ac01be, prebuilt under _cmp/r05, not committed) against this tree ON THE SAME BOX, alternating processes:
# tools/prove_time.py at the k = 19 ECDSA, k = 21 pairing and k = 17 shapes
set -u
O=$PWD/gpurun_out/r06c13; mkdir -p $O; R6=$PWD; R5=$PWD/_cmp/r05
for k in "19 1 1 1 0 18 7" "21 2 1 1 0 20 4" "17 4 1 1 0 16 7"; do
  tag=$(echo $k | cut -d' ' -f1)
  for round in 1 2 3; do
    for tree in r05 r06; do
      if [ $tree = r05 ]; then cd $R5; else cd $R6; fi
      timeout 400 python tools/prove_time.py $k > $O/${tree}_k${tag}_$round.log 2>&1
      echo "$tree k=$tag round $round: $(grep 'create_proof rep' $O/${tree}_k${tag}_$round.log | awk '{print $4}' | sort -n | head -4 | tr '\n' ' ')"
    done
  done
done
cd $R6
