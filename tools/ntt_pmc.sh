#!/bin/bash
# PMC view of the NTT pass kernels for a 2^22 transform (separate passes per counter group); NTT_PARAMS="name=value ..." sets context parameters
OUT=$PWD/gpurun_out/nttpmc; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p -- python $REPO/tools/ntt_one.py 22 $NTT_PARAMS > $OUT/p$i.log 2>&1
done
cd $REPO
for i in 1 2 3 4; do python - <<PY
import sqlite3,glob
p=glob.glob("$OUT/p$i/*.db")
if p:
    db=sqlite3.connect(p[0])
    try:
        for r in db.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%ntt_%kernel%' and kernel_name not like '%ntt_twiddle_kernel%' and kernel_name not like '%direct_twiddle%' group by counter_name"):
            print(r[0], r[1], f"{r[2]:.4g}")
    except Exception as e: print("err", e)
PY
done
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
