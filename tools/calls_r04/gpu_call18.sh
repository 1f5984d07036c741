#!/bin/bash
# r04 GPU call 18: the batch's bucket zero-fill queued after the tail hook's work and on lane 0 (it sat in front of the side transforms in a shared
# hardware queue): timeline of one proof again, whole proofs, bench line
mkdir -p gpurun_out/c18; O=$PWD/gpurun_out/c18; REPO=$PWD
timeout 300 python -m pytest tests/test_plonk_prover.py tests/test_gpu_parity.py -m gpu -q -x -k "k19 or k16 or gpu0 or shape3 or wide or msm_batch or repeatable" > $O/pytest.log 2>&1
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=clean_on_lane:0,1 2>&1 | grep -E "create_proof min" > $O/ab_k19.log
timeout 200 python tools/prove_time.py 19 1 1 1 0 18 6 --ab=plonk_tail_overlap:0,1 2>&1 | grep -E "create_proof min" > $O/ov_k19.log
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 4 --ab=plonk_tail_overlap:0,1 2>&1 | grep -E "create_proof min" > $O/ov_k21.log
timeout 200 python tools/prove_time.py 15 17 3 1 0 14 6 --ab=plonk_tail_overlap:0,1 2>&1 | grep -E "create_proof min" > $O/ov_k15.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/p19 -o t -- python $REPO/tools/prove_time.py 19 1 1 1 0 18 5 > $O/prove.log 2>&1
cd $REPO
python tools/rocprof_timeline.py $(find $O/p19 -name "*.db" | head -1) modmul_bench 99 20000 > $O/timeline_k19.md 2>&1
rm -rf $O/p19
timeout 600 python bench.py --no-sweep --no-pmc-traffic 2>/dev/null | tail -1 | head -c 600 > $O/bench_head.json
tail -2 $O/pytest.log; for f in ab_k19 ov_k19 ov_k21 ov_k15; do echo "== $f"; cat $O/$f.log; done; cat $O/bench_head.json; echo; grep "create_proof rep" $O/prove.log
