#!/bin/bash
# pipelined MSM time vs accumulate variant / lanes at 2^19 (batch 7) and 2^20 (batch 8)
for cfg in "19 7" "20 8"; do set -- $cfg; for v in 3 2; do for l in 2 3 4; do
  timeout 150 python bench.py --no-cpu-baseline --no-replay --steps 28 --log-n $1 --batch $2 --lanes $l --param msm_accum_variant=$v 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=2^$1 variant=$v lanes=$l batch=$2', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3))"
done; done; done
