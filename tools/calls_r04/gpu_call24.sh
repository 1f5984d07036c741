#!/bin/bash
# r04 GPU call 24: with a third lane context for the side work at 2^20+ points: k = 20 / 21 / 22 proofs, bench-mode timeline at k = 21
mkdir -p gpurun_out/c24; O=$PWD/gpurun_out/c24; REPO=$PWD
timeout 300 python tools/prove_time.py 21 2 1 1 0 20 5 2>&1 | grep -E "create_proof rep" > $O/k21.log
timeout 300 python tools/prove_time.py 21 1 1 1 0 20 5 2>&1 | grep -E "create_proof rep" > $O/k21e.log
timeout 300 python tools/prove_time.py 20 3 1 1 0 19 5 2>&1 | grep -E "create_proof rep" > $O/k20.log
timeout 300 python tools/prove_time.py 22 1 1 1 0 21 4 2>&1 | grep -E "create_proof rep" > $O/k22.log
timeout 200 python -m pytest tests/test_plonk_prover.py -m gpu -q -x -k "k21 or k19" > $O/pytest.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/b -o t -- python $REPO/bench.py --pmc-child --k 21 --steps 4 --warmup 1 > $O/bench_child.log 2>&1
cd $REPO
python tools/rocprof_timeline.py $(find $O/b -name "*.db" | head -1) lk_keys_kernel 4 70000 > $O/timeline_bench_k21.md 2>&1
rm -rf $O/b
for f in k21 k21e k20 k22; do echo "== $f"; cat $O/$f.log | tr '\n' ' '; echo; done; tail -2 $O/pytest.log
