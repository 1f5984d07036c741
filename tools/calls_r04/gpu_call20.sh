#!/bin/bash
# r04 GPU call 20: round 1's lookup permutation inside the commitment batch (advice MSMs start first): parity, same-process A/B, bench line
mkdir -p gpurun_out/c20; O=$PWD/gpurun_out/c20; REPO=$PWD
timeout 600 python -m pytest tests/test_plonk_prover.py tests/test_reference_shapes_golden.py tests/test_virtual_region.py -m gpu -q -x > $O/pytest.log 2>&1
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=plonk_permute_in_commit:0,1 2>&1 | grep -E "create_proof min" > $O/ab_k19.log
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 4 --ab=plonk_permute_in_commit:0,1 2>&1 | grep -E "create_proof min" > $O/ab_k21.log
timeout 200 python tools/prove_time.py 15 17 3 1 0 14 6 --ab=plonk_permute_in_commit:0,1 2>&1 | grep -E "create_proof min" > $O/ab_k15.log
timeout 200 python tools/prove_time.py 18 2 1 1 0 17 6 --ab=plonk_permute_in_commit:0,1 2>&1 | grep -E "create_proof min" > $O/ab_k18.log
timeout 600 python bench.py --no-sweep --no-pmc-traffic 2>/dev/null | tail -1 | head -c 500 > $O/bench_head.json
tail -2 $O/pytest.log; for f in ab_k19 ab_k21 ab_k15 ab_k18; do echo "== $f"; cat $O/$f.log; done; cat $O/bench_head.json; echo
