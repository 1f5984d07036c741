#!/bin/bash
# r04 GPU call 23: why do the grand products wait for the side transforms at k = 21 (own side context) and not at k = 19 (last lane's context)?
mkdir -p gpurun_out/c23; O=$PWD/gpurun_out/c23; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/a -o t -- python $REPO/tools/prove_time.py 19 1 1 1 0 18 4 --param=plonk_side_on_lanes=0 > $O/a.log 2>&1
timeout 400 rocprofv3 --kernel-trace -d $O/b -o t -- python $REPO/tools/prove_time.py 21 1 1 1 0 20 3 --param=msm_lanes=3 > $O/b.log 2>&1
cd $REPO
python tools/rocprof_timeline.py $(find $O/a -name "*.db" | head -1) modmul_bench 99 20000 > $O/timeline_k19_ownside.md 2>&1
python tools/rocprof_timeline.py $(find $O/b -name "*.db" | head -1) modmul_bench 99 80000 > $O/timeline_k21_lanes3.md 2>&1
rm -rf $O/a $O/b
grep "create_proof rep" $O/a.log $O/b.log | tail -4
