#!/bin/bash
# r05 GPU call 3: 128-byte table entries pre-split into 29-bit limbs (msm_table_split) against the packed 64-byte entries: per-kernel MSM times at 2^19 / 2^20
# (tables built under each setting, alternated) and whole k = 19 / k = 21 proofs; the accumulation's VALU instruction count under both (PMC)
set -u
O=$PWD/gpurun_out/r05c03; mkdir -p $O; REPO=$PWD
for rep in 1 2; do for v in 0 1; do
  timeout 300 python tools/msm_r03.py 19,20 pre:msm_table_split=$v >> $O/msm_split_ab.log 2>&1
done; done
grep "2^" $O/msm_split_ab.log | cut -c1-260
for rep in 1 2; do for v in 0 1; do
  echo "msm_table_split=$v" >> $O/prove_split_ab.log
  timeout 300 python tools/prove_time.py 19 1 1 1 0 18 12 --param=msm_table_split=$v 2>&1 | grep "create_proof rep" | sort -t: -k2 -n | awk '{print $4}' | sort -n | head -8 | tr '\n' ' ' >> $O/prove_split_ab.log; echo >> $O/prove_split_ab.log
done; done
cat $O/prove_split_ab.log
for v in 0 1; do
  echo "k21 msm_table_split=$v" >> $O/prove21_split_ab.log
  timeout 300 python tools/prove_time.py 21 2 1 1 0 20 6 --param=msm_table_split=$v 2>&1 | grep "create_proof rep" | awk '{print $4}' | sort -n | head -4 | tr '\n' ' ' >> $O/prove21_split_ab.log; echo >> $O/prove21_split_ab.log
done
cat $O/prove21_split_ab.log
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $O/pytest_msm.log 2>&1; tail -2 $O/pytest_msm.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $O/pmc$v -o p -- python $REPO/tools/msm_r03.py 19 pre:msm_table_split=$v > $O/pmc$v.log 2>&1
done
cd $REPO
python - <<'PY'
import glob, sqlite3
for v in (0,1):
    dbs = glob.glob("gpurun_out/r05c03/pmc%d/**/*.db" % v, recursive=True)
    if not dbs: print("no db", v); continue
    db = sqlite3.connect(dbs[0])
    for row in db.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%msm_accum_kernel%' group by counter_name"):
        print("msm_table_split=%d" % v, row)
PY
rm -rf $O/pmc0 $O/pmc1
