#!/bin/bash
# r05 GPU call 20: HIP stream priority of the batch MSM's lanes (which also carry the side transforms): lowest / highest against the default
set -u
O=$PWD/gpurun_out/r05c20; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 4 - "msm_lane_priority=1" "msm_lane_priority=-1" > $O/ab_k19.log 2>&1; tail -14 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 17 4 1 1 16 4 - "msm_lane_priority=1" "msm_lane_priority=-1" > $O/ab_k17.log 2>&1; tail -14 $O/ab_k17.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "msm_lane_priority=1" "msm_lane_priority=-1" > $O/ab_k21.log 2>&1; tail -8 $O/ab_k21.log
