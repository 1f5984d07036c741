#!/bin/bash
# r04 GPU call 28: SHPLONK's sets through one division call (one carry launch): parity, proofs with stage laps, bench line
mkdir -p gpurun_out/c28; O=gpurun_out/c28
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py tests/test_reference_shapes_golden.py -m gpu -q -x -k "kate or prover_steps or k19 or k16 or gpu0 or shape3 or wide or reference" > $O/pytest.log 2>&1
timeout 300 python tools/rng_ab.py 19 > $O/rng_ab.log 2>&1
timeout 600 python bench.py --no-sweep --no-pmc-traffic 2>/dev/null | tail -1 | head -c 420 > $O/bench_head.json
tail -2 $O/pytest.log; grep -E "^(device|array)|multiopen" $O/rng_ab.log | cut -c1-330; cat $O/bench_head.json; echo
