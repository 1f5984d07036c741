#!/bin/bash
# r04 GPU call 5: tail overlap A/B + correctness
mkdir -p gpurun_out/c5; O=gpurun_out/c5
timeout 900 python -m pytest tests/test_plonk_prover.py tests/test_rng_chacha.py tests/test_reference_shapes_golden.py tests/test_host_cpp.py tests/test_virtual_region.py -m gpu -q -x > $O/pytest.log 2>&1
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 8 --ab=plonk_tail_overlap > $O/prove19.log 2>&1
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 5 --ab=plonk_tail_overlap > $O/prove21.log 2>&1
timeout 200 python tools/prove_time.py 15 17 3 1 0 14 8 --ab=plonk_tail_overlap > $O/prove15.log 2>&1
timeout 300 python bench.py --no-sweep --no-replay > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest.log; grep "plonk_tail_overlap=" $O/prove19.log $O/prove21.log $O/prove15.log; tail -14 $O/prove19.log; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/c5/bench.json"))
for k in ("value","ms_per_step","seconds_per_proof_host_advice","seconds_per_proof_with_rng","cold_first_proof_seconds","stage_ms","speedup_vs_cpu_port"):
    v=d.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a!="note"}
    print(k, v)
print("int", d["roofline_int"]["frac"], "proof", d["roofline_proof"]["int"]["frac"], "cpu ok", d.get("cpu_baseline",{}).get("proof_bytes_equal_to_gpu"))
PY
