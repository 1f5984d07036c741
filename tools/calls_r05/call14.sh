#!/bin/bash
# r05 GPU call 14: dispatch timeline of one timed bench proof with plonk_early_intt on (default now) — where the grand products' lagrange_to_coeff runs
set -u
O=$PWD/gpurun_out/r05c14; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $REPO/bench.py --pmc-child --steps 20 --warmup 3 > $O/trace.log 2>&1
python $REPO/tools/rocprof_proof.py $(find $O/trace -name "*.db" | head -1) > $O/bench_proof_k19_kernels.md 2>&1
python $REPO/tools/rocprof_timeline.py $(find $O/trace -name "*.db" | head -1) modmul_bench 9999 20000 > $O/bench_proof_k19_timeline.md 2>&1
rm -rf $O/trace
head -3 $O/bench_proof_k19_kernels.md
