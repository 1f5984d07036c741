#!/usr/bin/env python3
"""Per-kernel account of ONE create_proof from a rocprofv3 --kernel-trace results .db of tools/prove_time.py: the dispatches between the
last `perm_product_terms`-less marker pair are not tagged, so the proof is delimited by its first kernel after the previous proof's last
kernel: proofs are separated by > 50 us of host time with no kernel running only at their boundaries ... in practice: take the LAST
occurrence of lk_keys_kernel (lookup sort, once per lookup column pair), walk back to the preceding quiet gap > `gap_us`, forward to the end.
usage: rocprof_proof.py results.db [--timeline]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(n.split("(")[0].replace("void ", "").replace("h2::", ""), s, e) for n, s, e in rows]
    # a proof's first kernel is the q_lookup * advice product (fr_binop_kernel<2>) right before the lookup's key extraction (lk_keys_kernel)
    marks = [i for i, r in enumerate(rows) if "modmul_bench_kernel" in r[0]]
    if marks and any("lk_keys_kernel" in r[0] for r in rows[marks[-1] + 1:]):   # tools/prove_time.py launches the multiplier probe right before its last proof
        sel = rows[marks[-1] + 1:]
    else:
        last_keys = max(i for i, r in enumerate(rows) if "lk_keys_kernel" in r[0])
        first = max(i for i, r in enumerate(rows[:last_keys]) if "fr_binop_kernel<2>" in r[0])
        end = min([i for i, r in enumerate(rows) if i > last_keys and "modmul" in r[0]] + [len(rows)])   # bench.py runs the multiplier probes after its last proof
        sel = rows[first:end]
    t0, t1 = sel[0][1], max(r[2] for r in sel)
    agg = {}
    for n, s, e in sel:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    # union of busy time
    busy, cur_s, cur_e = 0.0, None, None
    for n, s, e in sorted(sel, key=lambda r: r[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("# last create_proof in %s: %d dispatches, span %.1f us, GPU busy (union) %.1f us, idle %.1f us" % (
        sys.argv[1].split("/")[-1], len(sel), (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3))
    print("| kernel | calls | total_us | avg_us |")
    print("|---|---|---|---|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.1f | %.1f |" % (n, c, t, t / c))
    if "--timeline" in sys.argv:
        print("\n| start_us | dur_us | gap_before_us | kernel |")
        print("|---|---|---|---|")
        prev_end = t0
        for n, s, e in sel:
            print("| %.1f | %.1f | %.1f | %s |" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
            prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
