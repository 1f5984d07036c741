#!/bin/bash
# r06 GPU call 1: the default bench line as the driver runs it (the compact last line must parse: VERDICT r05 #1), then the partition state (read-only)
set -u
O=$PWD/gpurun_out/r06c01; mkdir -p $O; REPO=$PWD
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.stdout 2>$O/bench.err; echo "bench rc=$?"
tail -1 $O/bench.stdout > $O/bench_line.json; wc -c $O/bench_line.json; wc -l $O/bench.stdout
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06c01/bench_line.json"))
print(d["ms_per_step"], d["roofline"], d["cpu_baseline"].get("value"), d.get("create_proof_k21_pairing_shape"))
PY
cp bench_extra.json $O/ 2>/dev/null
tail -3 $O/bench.err
bash tools/rccl_cpx_try.sh $O/rccl_partition_state.log
