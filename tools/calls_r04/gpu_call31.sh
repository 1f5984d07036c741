#!/bin/bash
# r04 GPU call 31: round 2 with the random polynomial's MSM first and the grand products inside the batch (plonk_random_first), device generator
mkdir -p gpurun_out/c31; O=gpurun_out/c31
timeout 400 python -m pytest tests/test_rng_chacha.py tests/test_plonk_prover.py -m gpu -q -x -k "rng or k19 or k16 or gpu0" > $O/pytest.log 2>&1
timeout 300 python tools/rng_ab.py 19 --ab=plonk_random_first:0,1 2>&1 | grep -E "plonk_random_first" > $O/ab_k19.log
timeout 400 python tools/rng_ab.py 21 --ab=plonk_random_first:0,1 2>&1 | grep -E "plonk_random_first" > $O/ab_k21.log
timeout 300 python tools/rng_ab.py 17 --ab=plonk_random_first:0,1 2>&1 | grep -E "plonk_random_first" > $O/ab_k17.log
tail -2 $O/pytest.log; cat $O/ab_k19.log $O/ab_k21.log $O/ab_k17.log
