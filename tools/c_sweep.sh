#!/bin/bash
# window size sweep at 2^19 / 2^18 / 2^16 (precomputed tables): synchronous and pipelined
for n in 16 18 19; do for c in 0 13 14 15 16; do
  timeout 150 python bench.py --no-cpu-baseline --no-replay --steps 21 --log-n $n --batch 7 --param msm_window_bits=$c 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=2^$n c=$c', 'pipelined', round(d['ms_per_step'],3), 'sync', round(d.get('sync_ms_per_msm',0),3), 'W', d['config']['windows'])"
done; done
