#!/bin/bash
# Are the quotient / grand-product identity kernels HBM-bound or multiplier-bound?  PMC of one k = 19 proof's launches: executed VALU
# instructions, busy cycles and (separate passes) FETCH_SIZE / WRITE_SIZE; derived per kernel: HBM TB/s, VALU issue time (wave-instructions x 4 cycles / 1024
# SIMDs / clock) against the launch duration, and the algorithmic products per row the kernel's formulas need.
OUT=$PWD/gpurun_out/quotpmc; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p -- python $REPO/bench.py --pmc-child --steps 3 --warmup 1 > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import sqlite3, glob
def q(i, sql):
    p = glob.glob("$OUT/p%d/*.db" % i)
    return sqlite3.connect(p[0]).execute(sql).fetchall() if p else []
pat = "(kernel_name like '%quotient_%' or kernel_name like '%perm_product_terms%' or kernel_name like '%lookup_product_terms%' or kernel_name like '%fr_eval_tile%' or kernel_name like '%divide_by_vanishing%')"
rows = {}
for i in (1, 2, 3):
    try:
        for name, cname, cnt, avg in q(i, "select kernel_name, counter_name, count(*), avg(value) from counters_collection where %s group by kernel_name, counter_name" % pat):
            rows.setdefault(name.split("(")[0].replace("h2::", "").replace("void ", ""), {})[cname] = avg
    except Exception as e:
        print("err", i, e)
dur = {}
try:
    for name, d in q(1, "select name, avg(end - start) from kernels where %s group by name" % pat.replace("kernel_name", "name")):
        dur[name.split("(")[0].replace("h2::", "").replace("void ", "")] = d * 1e-3
except Exception as e:
    print("dur err", e)
# algorithmic products per extended-domain row of the k = 19 shape (1 advice + 1 lookup + 1 constants column, one permutation set of 3 columns): DESIGN.md §3
alg = {"quotient_permutation29_kernel": 18.5, "quotient_lookup_batch29_kernel": 13.5, "quotient_flex_gate_batch29_kernel": 2.5}
print("| kernel | us (under PMC) | VALU wave-instr | VALU issue us (x4 cyc / 1024 SIMDs / 2.3 GHz) | HBM MB | TB/s | alg. products/row | products/s at that duration |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(rows.items()):
    us = dur.get(k, 0)
    vi = v.get("SQ_INSTS_VALU", 0)
    mb = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / 1e6
    a = alg.get(k)
    print("| %s | %.1f | %.3g | %.1f | %.0f | %.2f | %s | %s |" % (k, us, vi, vi * 4 / 1024 / 2.3e3, mb, mb / us if us else 0, a if a else "", "%.3g" % (a * (1 << 21) / (us * 1e-6)) if a and us else ""))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
