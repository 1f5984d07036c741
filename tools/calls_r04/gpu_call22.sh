#!/bin/bash
# r04 GPU call 22: quad-lane bucket reduction above 32768 segments (k = 21 round 1: 5 columns x 2^16 buckets / 8), bench-mode timeline at k = 21
mkdir -p gpurun_out/c22; O=$PWD/gpurun_out/c22; REPO=$PWD
timeout 400 python tools/prove_time.py 21 2 1 1 0 20 4 --ab=msm_quad_seg_max:32768,131072 2>&1 | grep -E "create_proof min" > $O/quad_k21.log
timeout 400 python tools/prove_time.py 20 3 1 1 0 19 4 --ab=msm_quad_seg_max:32768,131072 2>&1 | grep -E "create_proof min" > $O/quad_k20.log
timeout 400 python tools/prove_time.py 22 1 1 1 0 21 3 --ab=msm_quad_seg_max:32768,131072 2>&1 | grep -E "create_proof min" > $O/quad_k22.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/b -o t -- python $REPO/bench.py --pmc-child --k 21 --steps 4 --warmup 1 > $O/bench_child.log 2>&1
cd $REPO
python tools/rocprof_timeline.py $(find $O/b -name "*.db" | head -1) lk_keys_kernel 4 70000 > $O/timeline_bench_k21.md 2>&1
rm -rf $O/b
for f in quad_k21 quad_k20 quad_k22; do echo "== $f"; cat $O/$f.log; done; grep -c "|" $O/timeline_bench_k21.md; tail -2 $O/bench_child.log
