#!/bin/bash
# r06 GPU call 6: 16-bit digit codes in the sort (digits / hist / scatter move half the bytes): parity, per-kernel MSM times, whole proofs
set -u
O=$PWD/gpurun_out/r06c06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $O/pytest_msm.log 2>&1; tail -3 $O/pytest_msm.log
timeout 300 python tools/msm_r03.py 19,20,21 > $O/msm_breakdown.log 2>&1; cat $O/msm_breakdown.log
timeout 400 python tools/prove_time.py 19 1 1 1 0 18 7 > $O/proof_k19.log 2>&1; grep "create_proof min" $O/proof_k19.log
timeout 600 python tools/prove_time.py 21 2 1 1 0 20 4 > $O/proof_k21.log 2>&1; grep "create_proof min" $O/proof_k21.log
timeout 600 python bench.py --no-sweep 2>$O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['create_proof_k21_pairing_shape']['seconds'], d['msm_2_20']['ms_per_msm'])"
