#!/bin/bash
for h in 1; do for v in 3 2; do for l in 2 3; do for b in 8; do
  timeout 150 python bench.py --no-cpu-baseline --no-replay --steps 24 --batch $b --lanes $l --param msm_accum_variant=$v --param msm_heavy_stream=$h 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('heavy=$h variant=$v lanes=$l batch=$b', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3), {k:round(v,2) for k,v in d['kernel_ms_per_msm'].items()})"
done; done; done; done
