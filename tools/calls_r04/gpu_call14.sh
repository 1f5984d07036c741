#!/bin/bash
# r04 GPU call 14: quotient kernels on 9 x 29-bit limbs (fr29.cuh) vs the saturated kernels: parity tests, same-process A/B in whole proofs
mkdir -p gpurun_out/c14; O=gpurun_out/c14
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py -m gpu -q -x -k "quotient or k19 or k16 or gpu0 or shape3 or wide or prover_steps" > $O/pytest.log 2>&1
timeout 300 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=quotient_29:0,1 2>&1 | grep -E "create_proof min|quotient_terms" > $O/q29_k19.log
timeout 400 python tools/prove_time.py 21 2 1 1 0 20 4 --ab=quotient_29:0,1 2>&1 | grep -E "create_proof min|quotient_terms" > $O/q29_k21.log
timeout 300 python tools/prove_time.py 20 1 1 1 0 19 4 --ab=msm_lanes:2,3 2>&1 | grep "create_proof min" > $O/lanes_k20.log
timeout 300 python tools/prove_time.py 15 17 3 1 0 14 6 --ab=quotient_29:0,1 2>&1 | grep -E "create_proof min" > $O/q29_k15.log
tail -3 $O/pytest.log; for f in q29_k19 q29_k21 q29_k15 lanes_k20; do echo "== $f"; cat $O/$f.log; done
