#!/bin/bash
# r04 GPU call 21: dispatch timelines of one k = 21 proof and one k = 15 (17 + 3 columns) proof
mkdir -p gpurun_out/c21; O=$PWD/gpurun_out/c21; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/p21 -o t -- python $REPO/tools/prove_time.py 21 2 1 1 0 20 4 > $O/prove21.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/p15 -o t -- python $REPO/tools/prove_time.py 15 17 3 1 0 14 5 > $O/prove15.log 2>&1
cd $REPO
python tools/rocprof_timeline.py $(find $O/p21 -name "*.db" | head -1) modmul_bench 99 80000 > $O/timeline_k21.md 2>&1
python tools/rocprof_timeline.py $(find $O/p15 -name "*.db" | head -1) modmul_bench 99 20000 > $O/timeline_k15.md 2>&1
rm -rf $O/p21 $O/p15
grep -c "|" $O/timeline_k21.md $O/timeline_k15.md; grep "create_proof rep" $O/prove21.log $O/prove15.log | tail -4
