#!/bin/bash
# r04 GPU call 3: NTT lock-step hypothesis (start stagger), regression of the pruned MSM / NTT code
mkdir -p gpurun_out/c3; O=gpurun_out/c3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_plonk_prover.py -m gpu -q -x -k "msm or ntt or coset or k16 or k19 or random_shapes" > $O/pytest.log 2>&1
timeout 300 python tools/ntt_r04.py ntt_stagger_mode=0:1:2:3 ntt_stagger=2:4:8 > $O/ntt_stagger10.log 2>&1
timeout 300 python tools/ntt_r04.py ntt_grid_full=1 ntt_stagger_mode=0:1:2:3 ntt_stagger=4 > $O/ntt_stagger10_full.log 2>&1
timeout 300 python tools/ntt_r04.py ntt_tile_bits=11 ntt_stagger_mode=0:1:2:3 ntt_stagger=4:8 > $O/ntt_stagger11.log 2>&1
tail -3 $O/pytest.log; cat $O/ntt_stagger10.log $O/ntt_stagger10_full.log $O/ntt_stagger11.log | cut -c1-400
