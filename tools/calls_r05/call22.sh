#!/bin/bash
# r05 GPU call 22: what the accumulation's waves wait for — instruction-fetch and wait-reason counters (whichever of them gfx950 exposes)
set -u
O=$PWD/gpurun_out/r05c22; mkdir -p $O; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" | sed 's/Name:[[:space:]]*//' | sort -u > $O/counters.txt
grep -i -E "ICACHE|IFETCH|SQ_WAIT|SQ_INST_LEVEL|SQ_LEVEL|INSTS_BRANCH|SQ_INSTS_SMEM|SQ_BUSY_CU|SQ_VALU_MFMA|SQ_INST_CYCLES|SQ_ACTIVE_INST" $O/counters.txt | tr '\n' ' ' > $O/picked.txt; cat $O/picked.txt; echo
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_VALU" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o p -- python $REPO/tools/msm_breakdown.py 19 msm_accum_kernel > $O/p$i.log 2>&1
  python - <<PY
import sqlite3,glob
p=glob.glob("$O/p$i/*.db")
if p:
    db=sqlite3.connect(p[0])
    try:
        for r in db.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%msm_accum%' group by counter_name"):
            print(r[0], r[1], f"{r[2]:.4g}")
    except Exception as e: print("err", e)
else:
    print("group $i: no db"); print(open("$O/p$i.log").read()[-300:])
PY
  rm -rf $O/p$i
done 2>&1 | tee $O/accum_wait_pmc.log
