#!/bin/bash
for cfg in "14 7" "17 7" "18 7"; do set -- $cfg; for f in 8 4 1; do
  timeout 150 python bench.py --no-cpu-baseline --no-replay --steps 21 --log-n $1 --batch $2 --param msm_fuse_cols=$f 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=2^$1 batch=$2 fuse=$f', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3))"
done; done
