#!/usr/bin/env python3
"""Prints the dispatch timeline (start, end, queue, kernel) of a window of a rocprofv3 --kernel-trace results .db:
used to see how the kernels of pipelined MSMs actually overlap.
usage: rocprof_timeline.py results.db [anchor-kernel-substring] [n-th occurrence] [window_us]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else "msm_accum"
    nth = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    window = float(sys.argv[4]) if len(sys.argv) > 4 else 5000.0
    cur = db.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"select name, start, end, {qcol or '0'} from kernels order by start"
    rows = cur.execute(sel).fetchall()
    anchors = [r for r in rows if anchor in r[0]]
    if not anchors:
        print("anchor not found; columns:", cols)
        return
    t0 = anchors[min(nth, len(anchors) - 1)][1]
    print(f"# columns of kernels view: {cols}")
    print(f"# t = 0 at the start of occurrence {nth} of *{anchor}*; times in us")
    print("| start | end | dur | queue | kernel |")
    print("|---|---|---|---|---|")
    for name, st, en, q in rows:
        rel = (st - t0) / 1e3
        if -200.0 <= rel <= window:
            short = name.split("(")[0].replace("void ", "").replace("h2::", "")
            print(f"| {rel:9.1f} | {(en - t0) / 1e3:9.1f} | {(en - st) / 1e3:8.1f} | {q} | {short} |")


if __name__ == "__main__":
    main()
