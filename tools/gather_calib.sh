#!/bin/bash
# PMC passes over tools/gather_calib.py --msm: FETCH_SIZE and the L2 memory-side request counters for the calibration probes and
# msm_accum_kernel in one command (separate passes per counter group, --kernel-trace only).  Output: gpurun_out/calib/*.txt
OUT=$PWD/gpurun_out/calib; mkdir -p $OUT; REPO=$PWD
python tools/gather_calib.py > $OUT/timing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p -- python $REPO/tools/gather_calib.py --msm > $OUT/p$i.log 2>&1
  python - <<PY > $OUT/pass$i.txt 2>&1
import sqlite3,glob
p=glob.glob("$OUT/p$i/**/*.db", recursive=True)
print("# counters: $grp")
if p:
    db=sqlite3.connect(p[0])
    try:
        for r in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%probe%' or kernel_name like '%msm_accum%' or kernel_name like '%msm_digits%' group by kernel_name, counter_name order by kernel_name"):
            print(r[0].split("(")[0].replace("void ","").replace("h2::",""), r[1], r[2], "%.6g" % r[3])
    except Exception as e: print("err", e)
else:
    print("no db (counter group unavailable?)"); print(open("$OUT/p$i.log").read()[-600:])
PY
  rm -rf $OUT/p$i
done
cd $REPO
cat $OUT/timing.txt $OUT/pass*.txt
