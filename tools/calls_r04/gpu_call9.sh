#!/bin/bash
# r04 GPU call 9: hook-after-reduction; overlap 1 vs 2; bench with in-run PMC traffic
mkdir -p gpurun_out/c9; O=gpurun_out/c9
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=plonk_tail_overlap:0,1 2>&1 | grep "create_proof min" > $O/prove19.log
for ov in 1 2 1 2; do
  timeout 300 python bench.py --no-sweep --no-replay --no-cpu-baseline --no-pmc-traffic --param plonk_tail_overlap=$ov 2> $O/b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('overlap=$ov ms %.3f host_adv %.3f array %.3f' % (d['ms_per_step'], d['seconds_per_proof_host_advice']*1e3, d['seconds_per_proof_with_rng']['predrawn_array_outside_the_call']*1e3), d['stage_ms'])" >> $O/ov.log
done
timeout 600 python bench.py --no-sweep --no-replay --no-cpu-baseline > $O/bench_pmc.json 2> $O/bench_pmc.err
cat $O/prove19.log $O/ov.log; tail -2 $O/bench_pmc.err; python -c "
import json
d=json.load(open('gpurun_out/c9/bench_pmc.json'))
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','traffic_measurement','avg_launch_ms')})"
