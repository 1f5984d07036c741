#!/bin/bash
# r05 GPU call 16: the quotient's gate identities in front of the join of the grand products' transforms (plonk_gate_before_join)
set -u
O=$PWD/gpurun_out/r05c16; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 6 - "plonk_gate_before_join=1" > $O/ab_k19.log 2>&1; tail -13 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 3 - "plonk_gate_before_join=1" > $O/ab_k21.log 2>&1; tail -7 $O/ab_k21.log
timeout 600 python tools/proof_configs_ab.py 17 4 1 1 16 6 - "plonk_gate_before_join=1" > $O/ab_k17.log 2>&1; tail -13 $O/ab_k17.log
timeout 600 python tools/proof_configs_ab.py 15 17 3 1 14 6 - "plonk_gate_before_join=1" > $O/ab_k15.log 2>&1; tail -13 $O/ab_k15.log
