#!/bin/bash
# r05 GPU call 21: bench.py with the proofs-in-flight block (default flags, as the driver runs it), wall time of the whole command
set -u
O=$PWD/gpurun_out/r05c21; mkdir -p $O
( time timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json ) 2> $O/time.log; tail -4 $O/time.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05c21/bench.json"))
print(d["ms_per_step"], d["roofline"]["frac_of_binding_roof"], d.get("create_proof_in_flight"))
print(d["ntt_2_22"]["ntt_ms"], d["ntt_2_22"]["roofline_int"]["frac"], d["msm_2_20"]["ms_per_msm"])
PY
