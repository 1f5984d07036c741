#!/bin/bash
# fused multi-column MSM: time per MSM vs columns per fused group (msm_fuse_cols) and batch size
for cfg in "20 4" "20 8" "19 5" "19 7" "16 7"; do set -- $cfg; for f in 8 4 2 1; do
  timeout 150 python bench.py --no-cpu-baseline --no-replay --steps $(( $2 * 3 )) --log-n $1 --batch $2 --param msm_fuse_cols=$f 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=2^$1 batch=$2 fuse=$f', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3), 'accum', d['kernel_ms_per_msm'].get('msm_accum_kernel'), 'launches', d['roofline']['launches'])"
done; done
