#!/bin/bash
# r04 GPU call 2: issue probe, NTT tile-size A/B, PMC of both tile kernels, proofs with both
mkdir -p gpurun_out/c2; O=gpurun_out/c2
timeout 200 python tools/issue_probe.py > $O/issue_probe.log 2>&1
timeout 300 python tools/ntt_r04.py ntt_tile_bits=10:11:10:11 > $O/ntt_ab.log 2>&1
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 8 --ab=ntt_tile_bits:10,11 > $O/prove19.log 2>&1
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 5 --ab=ntt_tile_bits:10,11 > $O/prove21.log 2>&1
NTT_PARAMS="ntt_tile_bits=10" timeout 400 bash tools/ntt_pmc.sh > $O/ntt_pmc_tile10.log 2>&1
NTT_PARAMS="ntt_tile_bits=11" timeout 400 bash tools/ntt_pmc.sh > $O/ntt_pmc_tile11.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ntt or coset" > $O/pytest_ntt.log 2>&1
cat $O/issue_probe.log; cat $O/ntt_ab.log; grep "ntt_tile_bits=" $O/prove19.log $O/prove21.log; cat $O/ntt_pmc_tile10.log; echo ---; cat $O/ntt_pmc_tile11.log; tail -3 $O/pytest_ntt.log
