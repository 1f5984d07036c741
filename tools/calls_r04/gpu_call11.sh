#!/bin/bash
# r04 GPU call 11: side work on lanes vs own contexts, overlap 1 vs 2, same process alternating
mkdir -p gpurun_out/c11; O=gpurun_out/c11
timeout 300 python -m pytest tests/test_rng_chacha.py tests/test_plonk_prover.py -m gpu -q -x -k "rng or k16 or k19 or gpu0 or shape3 or repeatable or wide" > $O/pytest.log 2>&1
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=plonk_side_on_lanes:0,1 2>&1 | grep "create_proof min" > $O/lanes_ov2.log
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --param=plonk_tail_overlap=1 --ab=plonk_side_on_lanes:0,1 2>&1 | grep "create_proof min" > $O/lanes_ov1.log
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=plonk_tail_overlap:1,2 2>&1 | grep "create_proof min" > $O/ov12_lanes.log
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=plonk_tail_overlap:0,2 2>&1 | grep "create_proof min" > $O/ov02_lanes.log
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 5 --ab=plonk_tail_overlap:0,2 2>&1 | grep "create_proof min" > $O/ov02_k21.log
timeout 200 python tools/prove_time.py 15 17 3 1 0 14 6 --ab=plonk_tail_overlap:0,2 2>&1 | grep "create_proof min" > $O/ov02_k15.log
tail -2 $O/pytest.log; for f in lanes_ov2 lanes_ov1 ov12_lanes ov02_lanes ov02_k21 ov02_k15; do echo "== $f"; cat $O/$f.log; done
