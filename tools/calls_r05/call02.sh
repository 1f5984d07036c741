#!/bin/bash
# r05 GPU call 2: host round trips through the host-mapped flag (host_poll) A/B on whole proofs, the zero-row skip of the coset transforms' first pass
# (NTT times against profiles/archive/r04_ntt_times.log), the NTT parity tests on the GPU, and the kernel trace / timeline of ONE TIMED bench proof
# (advice resident, device RNG, no stage laps: the r04 account was taken from tools/prove_time.py's lapped proof)
set -u
O=$PWD/gpurun_out/r05c02; mkdir -p $O; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py -m gpu -x -q -k "ntt or coset or fft or k19 or plonk" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/prove_time.py 19 1 1 1 0 18 3 --ab=host_poll:0,1 > $O/poll_ab_k19.log 2>&1; grep "host_poll=" $O/poll_ab_k19.log
timeout 300 python tools/prove_time.py 15 17 3 1 0 14 3 --ab=host_poll:0,1 > $O/poll_ab_k15.log 2>&1; grep "host_poll=" $O/poll_ab_k15.log
timeout 300 python tools/ntt_r04.py > $O/ntt_times.log 2>&1; cat $O/ntt_times.log
timeout 600 python bench.py --no-sweep 2>$O/bench.err | tail -1 > $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python $REPO/bench.py --pmc-child --steps 6 --warmup 2 > $O/trace.log 2>&1
cd $REPO
DB=$(find $O/tr -name "*.db" | head -1)
python tools/rocprof_proof.py $DB > $O/bench_proof_k19_kernels.md 2>&1
python tools/rocprof_timeline.py $DB modmul_bench 9999 16000 > $O/bench_proof_k19_timeline.md 2>&1
rm -rf $O/tr
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c02/bench.json"))
print("N=1 ms/step", d["ms_per_step"], "host_advice", d["seconds_per_proof_host_advice"]*1e3, "int frac", d["roofline"].get("frac_of_binding_roof"), "proof frac", d["roofline_proof"]["int"]["frac"])
print("ntt", {k:(v if not isinstance(v,dict) else {a:round(b,4) if isinstance(b,float) else b for a,b in v.items()}) for k,v in d["ntt_2_22"]["k19_workhorses"].items()}, d["ntt_2_22"]["ntt_ms"])
k21=d.get("create_proof_k21_pairing_shape",{}); print("k21", k21.get("seconds_median"), k21.get("equals_committed_oracle_prover_digest"), k21.get("golden",{}).get("verifying_keys_equal"), k21.get("error"))
PY
head -5 $O/bench_proof_k19_kernels.md; tail -2 $O/bench.err
