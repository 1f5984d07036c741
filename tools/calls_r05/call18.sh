#!/bin/bash
# r05 GPU call 18: the batch inversion on 9 x 29-bit limbs with loads ahead (fr_batch_invert29_kernel): parity, the sweep over run lengths, whole proofs
set -u
O=$PWD/gpurun_out/r05c18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "batch_invert or inverse" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/invert_sweep.py > $O/invert_sweep.log 2>&1; cat $O/invert_sweep.log | grep -v amdgpu
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 4 - "fr_invert_29=0" "fr_invert_run=8" "fr_invert_run=32" > $O/ab_k19.log 2>&1; tail -18 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 17 4 1 1 16 4 - "fr_invert_29=0" "fr_invert_run=8" > $O/ab_k17.log 2>&1; tail -14 $O/ab_k17.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 - "fr_invert_29=0" > $O/ab_k21.log 2>&1; tail -6 $O/ab_k21.log
