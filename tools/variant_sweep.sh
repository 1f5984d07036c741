#!/bin/bash
# pipelined MSM time vs accumulate variant (waves per SIMD) and chunk size
for v in 3 2; do for k in 0 32 16; do for b in 4 8; do
  timeout 150 python bench.py --no-cpu-baseline --no-replay --steps 24 --batch $b --param msm_accum_variant=$v --param msm_chunk=$k 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant=$v K=$k batch=$b', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3), 'accum_iso', round(d['roofline']['avg_launch_ms_isolated'],3))"
done; done; done
