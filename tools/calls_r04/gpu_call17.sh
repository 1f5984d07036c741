#!/bin/bash
# r04 GPU call 17: dispatch timeline of ONE k = 19 proof (rocprofv3 kernel trace of tools/prove_time.py; t = 0 at the modmul marker kernel before the last proof);
# time of the default `python bench.py`
mkdir -p gpurun_out/c17; O=$PWD/gpurun_out/c17; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/p19 -o t -- python $REPO/tools/prove_time.py 19 1 1 1 0 18 5 > $O/prove.log 2>&1
cd $REPO
python tools/rocprof_timeline.py $(find $O/p19 -name "*.db" | head -1) modmul_bench 99 20000 > $O/timeline_k19.md 2>&1
rm -rf $O/p19
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.log
tail -3 $O/bench_time.log; grep -c "|" $O/timeline_k19.md; grep "create_proof rep" $O/prove.log
