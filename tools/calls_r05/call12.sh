#!/bin/bash
# r05 GPU call 12: the round's last code — the whole -m gpu suite, the k = 23 / 24 golden entries (H2HIP_GOLDEN_BIG=1), the bench line (N = 1) and the 2-rank shared-GPU bench
set -u
O=$PWD/gpurun_out/r05c12; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
( time H2HIP_GOLDEN_BIG=1 timeout 1500 python -m pytest tests/test_reference_shapes_golden.py -m gpu -q -k "msm-L8 or msm-L9 or msm-L13" --durations=4 ) > $O/pytest_big.log 2>&1; tail -9 $O/pytest_big.log
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device --dist-backend gloo --steps 5 --warmup 2 --no-sweep --shard-ntt-columns on 2> $O/bench_2rank.err | tail -1 > $O/bench_2rank_shared_gpu_gloo_pairing21.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c12/bench.json"))
print("N=1", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_of_binding_roof"], d["roofline_proof"]["int"]["frac"], "msm", d["msm_2_20"]["ms_per_msm"], d["msm_2_20"]["roofline_int"]["frac_whole_msm"], "ntt", d["ntt_2_22"]["ntt_ms"], d["ntt_2_22"]["roofline_int"]["frac"])
e=json.load(open("gpurun_out/r05c12/bench_2rank_shared_gpu_gloo_pairing21.json"))
b=e["create_proof_k21_pairing_shape"]
print("N=2", e["ms_per_step"], e["sharded_bytes_equal_unsharded"], "cpu_baseline" in e, b["equals_committed_oracle_prover_digest"], b["sharded"]["equals_committed_oracle_prover_digest"], "msm blk", e.get("msm_2_20",{}).get("ms_per_msm"), e.get("msm_2_20",{}).get("error"))
PY
for f in $O/bench.err $O/bench_2rank.err; do tail -n 2 $f; done   # (as run, this line read `tail -2 a b`, which tail rejects: the call reported rc=1 for that alone)
