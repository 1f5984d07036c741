#!/bin/bash
# r06 GPU call 27: small job tables as kernel arguments (upload_poke: byte threshold, 0 = the pinned ring + hipMemcpyAsync) — parity, then same-process A/Bs of whole proofs
set -u
O=$PWD/gpurun_out/r06c27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_shapes_golden.py -m gpu -x -q -k "kate or eval or ecdsa-19 or ecdsa-15 or pairing-19 or shplonk or multiopen" > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
run() { n=$(echo "$2" | tr ':,' '__'); tag=$(echo $1 | cut -d' ' -f1); timeout 600 python tools/prove_time.py $1 --ab=$2 > $O/k${tag}_$n.log 2>&1; echo "k=$tag"; grep "create_proof min" $O/k${tag}_$n.log | head -4; }
run "19 1 1 1 0 18 5" upload_poke:0,15360
run "19 1 1 1 0 18 5" upload_poke:0,32768
run "19 1 1 1 0 18 5" upload_poke:0,3840
run "19 1 1 1 0 18 5" upload_poke:15360,32768
run "17 4 1 1 0 16 5" upload_poke:0,32768
run "15 17 3 1 0 14 5" upload_poke:0,32768
run "21 2 1 1 0 20 3" upload_poke:0,32768
# (the patch this call measured — poke_kernel<W> in capi.hip's upload_jobs — showed no difference and was not kept: profiles/r06_upload_poke_ab.log)
