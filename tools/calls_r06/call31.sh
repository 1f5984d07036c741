#!/bin/bash
# r06 GPU call 31: the NTT tile passes' butterfly products in Shoup form (ntt_shoup 1 / 0) — parity, ms per transform, whole proofs
set -u
O=$PWD/gpurun_out/r06c31; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_shapes_golden.py -m gpu -x -q -k "ntt or fft or coset or ecdsa-19 or ecdsa-15 or pairing-19 or pairing-21 or msm-L9 or ec_add" > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
timeout 600 python tools/ntt_r04.py ntt_shoup=0:1:0:1 > $O/ntt_times.log 2>&1; cat $O/ntt_times.log | cut -c1-400
run() { n=$(echo "$2" | tr ':,' '__'); tag=$(echo $1 | cut -d' ' -f1); timeout 600 python tools/prove_time.py $1 --ab=$2 > $O/k${tag}_$n.log 2>&1; echo "k=$tag"; grep "create_proof min" $O/k${tag}_$n.log | head -4; }
run "19 1 1 1 0 18 5" ntt_shoup:0,1
run "17 4 1 1 0 16 5" ntt_shoup:0,1
run "21 2 1 1 0 20 3" ntt_shoup:0,1
run "15 17 3 1 0 14 5" ntt_shoup:0,1
# (the build this call measured — f29_mul_shoup in fr29.cuh, ntt_tile_kernel<.., SHOUP>, the knob ntt_shoup — showed no gain and was not kept: profiles/r06_ntt_shoup.log)
