#!/bin/bash
# r04 GPU call 15: kate division / evaluation / product-factor kernels on 9 x 29-bit limbs: GPU parity suite, A/B in whole proofs
mkdir -p gpurun_out/c15; O=gpurun_out/c15
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 ) > $O/pytest.log 2>&1
timeout 300 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=kate_29:0,1 2>&1 | grep -E "create_proof min|multiopen|evaluations" > $O/k29_k19.log
timeout 300 python tools/prove_time.py 19 1 1 1 0 18 6 --param=kate_29=0 --ab=quotient_29:0,1 2>&1 | grep -E "create_proof min" > $O/q29_k19.log
timeout 400 python tools/prove_time.py 21 2 1 1 0 20 4 --ab=kate_29:0,1 2>&1 | grep -E "create_proof min|multiopen|evaluations" > $O/k29_k21.log
tail -14 $O/pytest.log; for f in k29_k19 q29_k19 k29_k21; do echo "== $f"; cat $O/$f.log; done
