#!/bin/bash
# r04 GPU call 26: the bench's per-launch timing through hipExtLaunchKernelGGL's start / stop events instead of separate event records
mkdir -p gpurun_out/c26; O=gpurun_out/c26
timeout 300 python tools/rng_ab.py 19 > $O/rng_ab.log 2>&1
timeout 600 python bench.py --no-sweep --no-pmc-traffic 2>/dev/null | tail -1 > $O/bench.json
timeout 600 python bench.py --no-sweep --no-pmc-traffic 2>/dev/null | tail -1 > $O/bench2.json
grep -E "^(device|array)" $O/rng_ab.log; python - <<'PY'
import json
for f in ("bench","bench2"):
    d=json.load(open("gpurun_out/c26/%s.json"%f))
    r=d["roofline"]
    print(f, d["ms_per_step"], r["achieved"], r["frac"], {k:v for k,v in r.items() if "us" in k or "launch" in k or "duration" in k})
PY
