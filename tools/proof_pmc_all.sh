#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES and the duration of EVERY kernel of the bench's timed proofs (one PMC pass): a table sorted by total time with the VALU issue
# time beside the duration — where a kernel's duration is far above its issue time it waits (latency, residency); where the instructions per wave look
# large for what the kernel computes, look at its code (how r06 found the omega^i0 chain start).
OUT=$PWD/gpurun_out/pmcall; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/p1 -o p -- python $REPO/bench.py --pmc-child --steps 3 --warmup 1 > $OUT/p1.log 2>&1
cd $REPO
python - <<PY
import sqlite3, glob, collections
p = glob.glob("$OUT/p1/*.db")
db = sqlite3.connect(p[0])
rows = collections.defaultdict(dict)
for name, cname, cnt, avg, tot in db.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"):
    k = name.split("(")[0].replace("h2::", "").replace("void ", "")
    rows[k][cname] = (cnt, avg, tot)
dur = {}
for name, cnt, avg, tot in db.execute("select name, count(*), avg(end - start), sum(end - start) from kernels group by name"):
    dur[name.split("(")[0].replace("h2::", "").replace("void ", "")] = (cnt, avg * 1e-3, tot * 1e-3)
print("| kernel | launches | total us | avg us | VALU wave-instr / launch | instr / wave | VALU issue us / launch | issue / duration |")
print("|---|---|---|---|---|---|---|---|")
for k, (cnt, avg, tot) in sorted(dur.items(), key=lambda kv: -kv[1][2])[:45]:
    v = rows.get(k, {})
    vi = v.get("SQ_INSTS_VALU", (0, 0, 0))[1]
    wv = v.get("SQ_WAVES", (0, 1, 0))[1] or 1
    issue = vi * 4 / 1024 / 2.3e3
    print("| %s | %d | %.0f | %.1f | %.3g | %.0f | %.1f | %.2f |" % (k[:60], cnt, tot, avg, vi, vi / wv, issue, issue / avg if avg else 0))
PY
rm -rf $OUT/p1
