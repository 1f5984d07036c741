#!/bin/bash
# usage (on the GPU box): bash tools/lanes_sweep.sh
for l in 1 2 3 4; do for b in 4 12; do
  timeout 300 python bench.py --lanes $l --batch $b --steps 24 --no-cpu-baseline --no-replay 2>/dev/null > /tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('lanes', $l, 'batch', d['batch'], 'ms/step', round(d['ms_per_step'],3), 'sync', round(d['sync_ms_per_msm'],3))"
done; done
