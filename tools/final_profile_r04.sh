#!/bin/bash
# r04 evidence run (on the GPU box, from the repo root); everything lands under gpurun_out/final/ and is copied to profiles/archive/r04_* by hand:
#   pytest_gpu.log            the whole -m gpu suite + smoke()
#   bench.json                `python bench.py` with default flags (the driver's command without --steps / --warmup), incl. the in-run PMC traffic
#   kernel_trace.md           rocprofv3 --kernel-trace --stats of the same command's timed workload (k = 19 create_proof x 20)
#   pmc_hbm.{md,json}, pmc_accum.json   FETCH_SIZE / WRITE_SIZE passes (separate) of that workload: HBM traffic per launch, every kernel
#   create_proof_k19_kernels.md, create_proof_k21_kernels.md   per-kernel account of ONE proof (tools/prove_time.py under rocprofv3)
#   config_sweep.md           the reference's 18 benchmark shapes
#   bench_2rank_shared_gpu_gloo_k21.json   bench.py --gpus 2 --share-device --dist-backend gloo --k 21 (functional evidence of the sharded path at config #5's size)
set -u
OUT=$PWD/gpurun_out/final
mkdir -p $OUT
REPO=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > $OUT/pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $OUT/pytest_gpu.log 2>&1
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --pmc-child --steps 20 --warmup 3 > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python $REPO/bench.py --pmc-child --steps 4 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python $REPO/bench.py --pmc-child --steps 4 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p19 -o t -- python $REPO/tools/prove_time.py 19 1 1 1 0 18 5 > $OUT/prove_time_k19.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/p21 -o t -- python $REPO/tools/prove_time.py 21 2 1 1 0 20 4 > $OUT/prove_time_k21.log 2>&1
cd $REPO
python tools/rocprof_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_trace.md 2>&1
python tools/rocprof_pmc.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) $OUT/pmc_hbm.md $OUT/pmc_hbm.json > /dev/null 2>&1
python tools/rocprof_proof.py $(find $OUT/p19 -name "*.db" | head -1) > $OUT/create_proof_k19_kernels.md 2>&1
python tools/rocprof_proof.py $(find $OUT/p21 -name "*.db" | head -1) > $OUT/create_proof_k21_kernels.md 2>&1
python - <<'PY'
import json, os
out = os.path.join(os.getcwd(), "gpurun_out", "final")
try:
    d = json.load(open(os.path.join(out, "pmc_hbm.json")))
    k = [n for n in d if n.startswith("msm_accum_kernel")][0]
    json.dump({"kernel": k, **d[k], "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python bench.py --pmc-child --steps 4 --warmup 1` "
               "(tools/final_profile_r04.sh); bytes = (2 * FETCH_SIZE_KB + WRITE_SIZE_KB) * 1024: FETCH_SIZE tallies 128-byte requests as 64 B on gfx950 "
               "(MI355X_MICROARCH.md, profiles/archive/r02_hbm_counter_calibration.md); mean over the run's launches (keygen's and the proofs' 2^19-point MSMs); every kernel "
               "of the run: profiles/archive/r04_bench_pmc_hbm.md"}, open(os.path.join(out, "pmc_accum.json"), "w"), indent=1)
except Exception as e:
    print("pmc_accum:", e)
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/p19 $OUT/p21
timeout 300 python tools/ntt_r04.py ntt_tile_kernel=1:0 > $OUT/ntt_times.log 2>&1
timeout 300 python tools/msm_r03.py 19,20 > $OUT/msm_breakdown.log 2>&1
timeout 900 python tools/config_sweep.py all 5 > $OUT/config_sweep.md 2> $OUT/config_sweep.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device --dist-backend gloo --k 21 --steps 3 --warmup 1 --no-replay --no-sweep --no-pmc-traffic --shard-ntt-columns on 2> $OUT/bench_2rank.err | tail -1 > $OUT/bench_2rank_shared_gpu_gloo_k21.json
tail -24 $OUT/pytest_gpu.log; head -c 700 $OUT/bench.json; echo; head -14 $OUT/kernel_trace.md; head -8 $OUT/pmc_hbm.md; head -30 $OUT/create_proof_k19_kernels.md; head -c 600 $OUT/bench_2rank_shared_gpu_gloo_k21.json; tail -3 $OUT/bench_2rank.err
