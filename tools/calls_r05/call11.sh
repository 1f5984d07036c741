#!/bin/bash
# r05 GPU call 11: batch-inversion run length inside the k = 19 proof now that the permutation set and the lookup share ONE inversion of 2^20 elements
# (fr_invert_run: elements per lane = per inversion; auto gives 16 there)
set -u
O=$PWD/gpurun_out/r05c11; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 3 - "fr_invert_run=4" "fr_invert_run=8" "fr_invert_run=32" > $O/ab_k19.log 2>&1; tail -13 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "fr_invert_run=8" "fr_invert_run=16" > $O/ab_k21.log 2>&1; tail -7 $O/ab_k21.log
timeout 200 python tools/invert_sweep.py > $O/invert_sweep.log 2>&1; tail -12 $O/invert_sweep.log
