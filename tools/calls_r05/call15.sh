#!/bin/bash
# r05 GPU call 15: msm_stagger_sorts once more, with plonk_early_intt on (the default now), more rounds and more shapes: k = 21 / 20 (two lanes),
# k = 19 / 18 (three lanes; pairing-19 and pairing-18 have 6 / 13 advice columns: several rounds of columns per batch)
set -u
O=$PWD/gpurun_out/r05c15; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 4 - "msm_stagger_sorts=1" > $O/ab_k21.log 2>&1; tail -9 $O/ab_k21.log
timeout 600 python tools/proof_configs_ab.py 20 3 1 1 19 4 - "msm_stagger_sorts=1" > $O/ab_k20.log 2>&1; tail -9 $O/ab_k20.log
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 6 - "msm_stagger_sorts=1" > $O/ab_k19.log 2>&1; tail -13 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 19 6 1 1 18 4 - "msm_stagger_sorts=1" > $O/ab_p19.log 2>&1; tail -9 $O/ab_p19.log
timeout 600 python tools/proof_configs_ab.py 18 2 1 1 17 6 - "msm_stagger_sorts=1" > $O/ab_k18.log 2>&1; tail -13 $O/ab_k18.log
