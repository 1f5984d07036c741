#!/bin/bash
# r06 GPU call 12: out-of-place inversion in the grand products: parity, stage times, bench
set -u
O=$PWD/gpurun_out/r06c12; mkdir -p $O
timeout 1200 python -m pytest tests/test_plonk_prover.py tests/test_gpu_parity.py tests/test_reference_shapes_golden.py -m gpu -x -q -k "not msm" > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
timeout 400 python tools/prove_time.py 19 1 1 1 0 18 7 > $O/proof_k19.log 2>&1; tail -14 $O/proof_k19.log
timeout 600 python bench.py --no-sweep 2>$O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['create_proof_k21_pairing_shape']['seconds'], d['msm_2_20']['ms_per_msm'], d['roofline_proof']['int']['frac'])"
