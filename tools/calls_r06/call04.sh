#!/bin/bash
# r06 GPU call 4: the whole -m gpu suite (all-to-all routing through the multi-rank and single-rank RCCL tests, the flex_gate KATs, the NTT layouts),
# then whole-proof A/Bs of the NTT tile layouts (0 = r05's 48-byte elements, 1 = planes + matrix swizzle, 2 = planes without a swizzle)
set -u
O=$PWD/gpurun_out/r06c04; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
for pair in 0,2 0,1 2,1; do
  timeout 400 python tools/prove_time.py 19 1 1 1 0 18 5 --ab=ntt_lds_planes:$pair > $O/proof_ab_k19_$pair.log 2>&1; grep "ntt_lds_planes=" $O/proof_ab_k19_$pair.log
done
timeout 600 python tools/prove_time.py 21 2 1 1 0 20 3 --ab=ntt_lds_planes:0,2 > $O/proof_ab_k21_0,2.log 2>&1; grep "ntt_lds_planes=" $O/proof_ab_k21_0,2.log
timeout 400 python tools/prove_time.py 17 4 1 1 0 16 5 --ab=ntt_lds_planes:0,2 > $O/proof_ab_k17_0,2.log 2>&1; grep "ntt_lds_planes=" $O/proof_ab_k17_0,2.log
