#!/bin/bash
# r05 GPU call 4: the sharded prover after the side-stream change (first-round transforms + all-gather next to round 2's commitments) and the measured
# column-dealing decision: multi-rank shared-GPU golden test, the forced single-rank sharded tests over RCCL and the callback, the 2-rank bench whose
# pairing-21 block must reproduce the committed oracle digest
set -u
O=$PWD/gpurun_out/r05c04; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_sharded_multirank_gpu.py tests/test_sharded_single_rank.py -m gpu -x -q -s --durations=5 ) > $O/sharded.log 2>&1
grep -E "passed|failed|measured|error" $O/sharded.log | tail -8
timeout 600 python -m pytest tests/test_reference_shapes_golden.py -m gpu -x -q -k "ecdsa-19 or pairing-21" > $O/golden_forced.log 2>&1; tail -2 $O/golden_forced.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device --dist-backend gloo --steps 5 --warmup 2 --no-sweep --shard-ntt-columns on 2> $O/bench_2rank.err | tail -1 > $O/bench_2rank_shared_gpu_gloo_pairing21.json
python - <<'PY'
import json
e=json.load(open("gpurun_out/r05c04/bench_2rank_shared_gpu_gloo_pairing21.json"))
b=e.get("create_proof_k21_pairing_shape",{})
print("N=2 headline ms", e["ms_per_step"], "sharded==unsharded", e.get("sharded_bytes_equal_unsharded"), "cpu_baseline" in e, "roofline" in e)
print("pairing-21 one GPU: digest==golden", b.get("equals_committed_oracle_prover_digest"), "sharded:", {k:v for k,v in b.get("sharded",{}).items() if k in ("seconds","sharded_bytes_equal_unsharded","ranks_emit_identical_bytes","equals_committed_oracle_prover_digest","proof_sha256")}, b.get("error"))
PY
tail -3 $O/bench_2rank.err
