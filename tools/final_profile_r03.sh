#!/bin/bash
# r03 evidence run (on the GPU box, from the repo root): bench.py, a rocprofv3 kernel trace of the same command's timed workload (k = 19
# create_proof: every msm_accum_kernel launch in it is a 2^19-point launch), the two PMC passes for HBM traffic, the kernel account of one
# proof and the config sweep; everything lands under gpurun_out/final/.
set -u
OUT=$PWD/gpurun_out/final
mkdir -p $OUT
REPO=$PWD
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --no-cpu-baseline --no-replay --no-sweep --steps 20 > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python $REPO/bench.py --no-cpu-baseline --no-replay --no-sweep --steps 4 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python $REPO/bench.py --no-cpu-baseline --no-replay --no-sweep --steps 4 > $OUT/pmc_write.log 2>&1
cd $REPO
python tools/rocprof_summary.py $(ls $OUT/trace/*.db | head -1) > $OUT/kernel_trace.md 2>&1
python tools/rocprof_proof.py $(ls $OUT/trace/*.db | head -1) > $OUT/create_proof_kernels.md 2>&1
python tools/rocprof_pmc.py $(ls $OUT/pmc_fetch/*.db | head -1) $(ls $OUT/pmc_write/*.db | head -1) $OUT/pmc_hbm.md $OUT/pmc_hbm.json > /dev/null 2>&1
python - <<'PY'
import json, os
out = os.path.join(os.getcwd(), "gpurun_out", "final")
d = json.load(open(os.path.join(out, "pmc_hbm.json")))
k = [n for n in d if n.startswith("msm_accum_kernel")][0]
json.dump({"kernel": k, **d[k]}, open(os.path.join(out, "pmc_accum.json"), "w"), indent=1)   # -> profiles/archive/r03_create_proof_k19_pmc_hbm.json (+ the "how" note)
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
head -c 600 $OUT/bench.json; echo; head -12 $OUT/kernel_trace.md; head -8 $OUT/pmc_hbm.md
