#!/bin/bash
# r04 GPU call 19: dispatch timeline of one proof of the BENCH's timed loop (device RNG, advice resident): rocprofv3 kernel trace of bench.py --pmc-child
mkdir -p gpurun_out/c19; O=$PWD/gpurun_out/c19; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/b -o t -- python $REPO/bench.py --pmc-child --steps 6 --warmup 2 > $O/bench_child.log 2>&1
cd $REPO
python tools/rocprof_timeline.py $(find $O/b -name "*.db" | head -1) lk_keys_kernel 6 16000 > $O/timeline_bench_k19.md 2>&1
rm -rf $O/b
grep -c "|" $O/timeline_bench_k19.md; tail -2 $O/bench_child.log
