#!/bin/bash
# r04 GPU call 4: bench line with the device RNG / warm keygen, NTT numbers after the grid + LDS fix, RNG tests
mkdir -p gpurun_out/c4; O=gpurun_out/c4
timeout 300 python -m pytest tests/test_rng_chacha.py tests/test_gpu_parity.py -m gpu -q -x -k "rng or chacha or ntt or coset" > $O/pytest.log 2>&1
timeout 300 python tools/ntt_r04.py ntt_tile_kernel=1:0 > $O/ntt.log 2>&1
timeout 600 python bench.py --no-sweep > $O/bench.json 2> $O/bench.err
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 8 > $O/prove19.log 2>&1
tail -3 $O/pytest.log; cat $O/ntt.log | cut -c1-330; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/c4/bench.json"))
for k in ("value","ms_per_step","seconds_per_proof","seconds_per_proof_host_advice","seconds_per_proof_with_rng","cold_first_proof_seconds","keygen_seconds","stage_ms","speedup_vs_cpu_port","speedup_vs_cpu_port_advice_resident_in_hbm"):
    print(k, d.get(k))
print("roofline", {k:v for k,v in d["roofline"].items() if k in ("achieved","frac","avg_launch_ms")})
print("int", d["roofline_int"]["frac"], "proof", d["roofline_proof"]["int"]["frac"])
print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k in ("value","seconds","proof_bytes_equal_to_gpu","verifying_keys_equal","gpu_proof_verified_by_oracle","error")})
print("ntt", d.get("ntt_2_22")); print("k21", {k:v for k,v in d.get("create_proof_k21_pairing_shape",{}).items() if "ms" in k or "seconds" in k})
PY
tail -16 $O/prove19.log
