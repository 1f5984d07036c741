#!/bin/bash
# deferred joint bucket reduction (batch API) on / off
for cfg in "20 4" "20 8" "19 5" "19 7" "18 7" "21 4"; do set -- $cfg; for d in 1 0; do for l in 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-replay --steps $(( $2 * 3 )) --log-n $1 --batch $2 --lanes $l --param msm_defer_reduce=$d 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=2^$1 batch=$2 defer=$d lanes=$l', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3))"
done; done; done
