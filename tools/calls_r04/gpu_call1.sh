#!/bin/bash
# r04 GPU call 1: new parity tests + NTT tile-kernel A/B + proof timings
mkdir -p gpurun_out/c1; O=gpurun_out/c1
python -c "import torch" 2>/dev/null
( time timeout 1500 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest.log 2>&1
timeout 300 python tools/ntt_r04.py ntt_tile_kernel=0:1:0:1 > $O/ntt_ab.log 2>&1
timeout 300 python tools/prove_time.py 19 1 1 1 0 18 12 --ab=ntt_tile_kernel > $O/prove19.log 2>&1
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 6 --ab=ntt_tile_kernel > $O/prove21.log 2>&1
timeout 600 bash tools/ntt_pmc.sh > $O/ntt_pmc.log 2>&1
tail -30 $O/pytest.log; cat $O/ntt_ab.log; tail -12 $O/prove19.log; tail -8 $O/prove21.log; cat $O/ntt_pmc.log | tail -20
