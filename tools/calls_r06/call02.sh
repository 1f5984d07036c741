#!/bin/bash
# r06 GPU call 2: the tile kernel's new LDS layout (limb planes at a swizzled index, skewed twiddles: ntt_lds_planes=1) against r05's 48-byte
# elements — parity on the GPU, transform times A/B in one process, the LDS counters of both, whole proofs A/B
set -u
O=$PWD/gpurun_out/r06c02; mkdir -p $O; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or coset or fft" > $O/pytest_ntt.log 2>&1; tail -3 $O/pytest_ntt.log
timeout 600 python tools/ntt_r04.py ntt_lds_planes=0:1:0:1 > $O/ntt_times_ab.log 2>&1; cat $O/ntt_times_ab.log
NTT_PARAMS="ntt_lds_planes=0" bash tools/ntt_pmc.sh > $O/ntt_pmc_layout48.log 2>&1
NTT_PARAMS="ntt_lds_planes=1" bash tools/ntt_pmc.sh > $O/ntt_pmc_planes.log 2>&1
paste $O/ntt_pmc_layout48.log $O/ntt_pmc_planes.log
timeout 400 python tools/prove_time.py 19 1 1 1 0 18 3 --ab=ntt_lds_planes:0,1 > $O/proof_ab_k19.log 2>&1; grep "ntt_lds_planes=" $O/proof_ab_k19.log
timeout 600 python tools/prove_time.py 21 2 1 1 0 20 2 --ab=ntt_lds_planes:0,1 > $O/proof_ab_k21.log 2>&1; grep "ntt_lds_planes=" $O/proof_ab_k21.log
