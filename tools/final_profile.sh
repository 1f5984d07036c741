#!/bin/bash
# Evidence run (on the GPU box, from the repo root): GPU test suite, smoke, bench, rocprofv3 kernel trace and the two PMC
# passes; everything lands under gpurun_out/final/.  The kernel trace runs bench.py with --no-replay so that every
# msm_accum_kernel launch in it is a 2^20-point launch (the replay's 2^19 launches have the same grid and cannot be told apart).
set -u
OUT=$PWD/gpurun_out/final
mkdir -p $OUT
REPO=$PWD
SKIP_TESTS=${SKIP_TESTS:-0}
if [ "$SKIP_TESTS" != "1" ]; then
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1
fi
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --no-cpu-baseline --no-replay --steps 64 > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python $REPO/bench.py --no-cpu-baseline --no-replay --steps 8 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python $REPO/bench.py --no-cpu-baseline --no-replay --steps 8 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ptrace -o t -- python $REPO/tools/prove_time.py 19 1 1 1 0 18 4 > $OUT/prove_time.log 2>&1
cd $REPO
python tools/rocprof_proof.py $(find $OUT/ptrace -name "*.db" | head -1) > $OUT/create_proof_kernels.md 2>&1
rm -rf $OUT/ptrace
# the reference's 18 benchmark shapes (halo2-ecc/configs/{bn254/bench_pairing,secp256k1/bench_ecdsa}.config) and the kernel account of the widest one
timeout 600 python tools/config_sweep.py all 5 > $OUT/config_sweep.md 2> $OUT/config_sweep.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/wtrace -o t -- python $REPO/tools/prove_time.py 14 211 27 1 0 13 3 > $OUT/prove_time_k14_wide.log 2>&1
cd $REPO
python tools/rocprof_proof.py $(find $OUT/wtrace -name "*.db" | head -1) > $OUT/create_proof_k14_wide_kernels.md 2>&1
rm -rf $OUT/wtrace
python tools/rocprof_summary.py $(ls $OUT/trace/*.db | head -1) > $OUT/kernel_trace.md 2>&1
python tools/rocprof_pmc.py $(ls $OUT/pmc_fetch/*.db | head -1) $(ls $OUT/pmc_write/*.db | head -1) $OUT/pmc_hbm.md $OUT/pmc_hbm.json > /dev/null 2>&1
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pytest_gpu.log $OUT/smoke.log 2>/dev/null | tail -6; head -c 900 $OUT/bench.json; echo; head -8 $OUT/kernel_trace.md; head -8 $OUT/pmc_hbm.md
