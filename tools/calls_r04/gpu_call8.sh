#!/bin/bash
# r04 GPU call 8: random-commit-ahead A/B + correctness on the RNG paths
mkdir -p gpurun_out/c8; O=gpurun_out/c8
timeout 600 python -m pytest tests/test_rng_chacha.py tests/test_plonk_prover.py -m gpu -q -x -k "rng or k16 or k19 or gpu0 or shape3 or repeatable" > $O/pytest.log 2>&1
timeout 300 python bench.py --no-sweep --no-replay --param plonk_tail_overlap=1 > $O/bench_ov1.json 2> $O/bench1.err
timeout 300 python bench.py --no-sweep --no-replay --param plonk_tail_overlap=2 > $O/bench_ov2.json 2> $O/bench2.err
timeout 300 python bench.py --no-sweep --no-replay --param plonk_tail_overlap=1 --no-cpu-baseline > $O/bench_ov1b.json 2> $O/bench1b.err
timeout 300 python bench.py --no-sweep --no-replay --param plonk_tail_overlap=2 --no-cpu-baseline > $O/bench_ov2b.json 2> $O/bench2b.err
tail -3 $O/pytest.log; tail -2 $O/bench2.err
python - <<'PY'
import json
for f in ("bench_ov1","bench_ov2","bench_ov1b","bench_ov2b"):
    try:
        d=json.load(open("gpurun_out/c8/%s.json"%f))
        print(f, "ms", round(d["ms_per_step"],3), "host_adv", round(d["seconds_per_proof_host_advice"]*1e3,3), "array", round(d["seconds_per_proof_with_rng"]["predrawn_array_outside_the_call"]*1e3,3), "proof frac", round(d["roofline_proof"]["int"]["frac"],4), {k:v for k,v in d["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
