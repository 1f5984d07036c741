#!/bin/bash
# r05 GPU call 1: the multi-rank shared-GPU golden-digest test (2 and 3 ranks on GPU 0 over gloo), the default bench line, the 2-rank bench with the
# sharded pairing-21 block, and the per-kernel account of one k = 19 proof as this round's starting point
set -u
O=$PWD/gpurun_out/r05c01; mkdir -p $O; REPO=$PWD
( time timeout 1200 python -m pytest tests/test_sharded_multirank_gpu.py -m gpu -x -q --durations=5 ) > $O/multirank.log 2>&1
tail -15 $O/multirank.log
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device --dist-backend gloo --steps 5 --warmup 2 --no-sweep --shard-ntt-columns on 2> $O/bench_2rank.err | tail -1 > $O/bench_2rank_shared_gpu_gloo.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p19 -o t -- python $REPO/tools/prove_time.py 19 1 1 1 0 18 5 > $O/prove_time_k19.log 2>&1
cd $REPO
python tools/rocprof_proof.py $(find $O/p19 -name "*.db" | head -1) > $O/create_proof_k19_kernels.md 2>&1
rm -rf $O/p19
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c01/bench.json"))
print("N=1", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_binding_roof"), d["roofline_proof"]["int"]["frac"], d.get("speedup_vs_cpu_port"))
print("witness", {k:(v.get("ms_per_proof_median") if isinstance(v,dict) else v) for k,v in d.get("witness_distribution",{}).items()})
k21=d.get("create_proof_k21_pairing_shape",{}); print("k21", k21.get("seconds_median"), k21.get("roofline_proof",{}).get("int",{}).get("frac"), k21.get("error"))
try:
    e=json.load(open("gpurun_out/r05c01/bench_2rank_shared_gpu_gloo.json"))
    print("N=2", e["ms_per_step"], e.get("sharded_bytes_equal_unsharded"), "cpu_baseline" in e, e.get("create_proof_k21_pairing_shape",{}).get("sharded",e.get("create_proof_k21_pairing_shape")))
except Exception as ex:
    print("2rank:", ex)
PY
tail -3 $O/bench.err $O/bench_2rank.err; head -40 $O/create_proof_k19_kernels.md
