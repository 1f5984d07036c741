#!/bin/bash
# r05 GPU call 10: the wave-owned radix-8 NTT pass (ntt_w8 = 1: three waves per SIMD, 2: two, 0: the tile kernel) — parity tests, ms per transform
# 2^16 .. 2^24 and the prover's workhorses, whole proofs at k = 19 / 21
set -u
O=$PWD/gpurun_out/r05c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or coset or fft or full_range" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 400 python tools/ntt_r04.py ntt_w8=0:1:2:0:1:2 > $O/ntt_times.log 2>&1; cat $O/ntt_times.log | cut -c1-330
timeout 600 python tools/proof_configs_ab.py 19 1 1 1 18 2 "ntt_w8=0" "ntt_w8=1" "ntt_w8=2" > $O/ab_k19.log 2>&1; tail -8 $O/ab_k19.log
timeout 600 python tools/proof_configs_ab.py 21 2 1 1 20 2 "ntt_w8=0" "ntt_w8=1" "ntt_w8=2" > $O/ab_k21.log 2>&1; tail -8 $O/ab_k21.log
