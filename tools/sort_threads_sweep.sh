#!/bin/bash
# pipelined (batch 4/8) MSM time vs the sort kernels' workgroup size and lane count
for t in 1024 512 256; do for l in 3 4; do for b in 4 8; do
  timeout 150 python bench.py --no-cpu-baseline --no-replay --steps 24 --batch $b --lanes $l --param msm_sort_threads=$t 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('threads=$t lanes=$l batch=$b', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3))"
done; done; done
