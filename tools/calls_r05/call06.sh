#!/bin/bash
# (msm_sort_first was removed after this call: neutral-to-slower, profiles/r05_msm_sort_first_ab.log) r05 GPU call 6: sort-first scheduling of the batch MSM (msm_sort_first: the sorts of a round of columns are queued on all lanes before any of their
# accumulations) — whole proofs at k = 19 / 21 / 15 (17 + 3 columns), with 3 and 4 lanes; MSM parity tests under the switch
set -u
O=$PWD/gpurun_out/r05c06; mkdir -p $O
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 3 - "msm_sort_first=1" "msm_sort_first=1,msm_lanes=4" "msm_lanes=4" > $O/ab_k19.log 2>&1; tail -14 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "msm_sort_first=1" "msm_sort_first=1,msm_lanes=3" > $O/ab_k21.log 2>&1; tail -8 $O/ab_k21.log
timeout 600 python tools/proof_configs_ab.py 15 17 3 1 14 2 - "msm_sort_first=1" > $O/ab_k15.log 2>&1; tail -6 $O/ab_k15.log
timeout 600 python tools/proof_configs_ab.py 18 2 1 1 17 2 - "msm_sort_first=1" > $O/ab_k18.log 2>&1; tail -6 $O/ab_k18.log
