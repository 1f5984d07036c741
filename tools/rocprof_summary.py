#!/usr/bin/env python3
"""Turns a rocprofv3 results .db (rocpd sqlite, from `rocprofv3 --kernel-trace --stats`) into the per-kernel
summary table committed under profiles/ (name, calls, total/avg/min/max duration, %).  With --pmc, also prints
the mean of every collected counter per kernel."""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print("| kernel | calls | total_ms | avg_us | min_us | max_us | % | vgpr | sgpr | lds_B | grid_x | wg_x |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0]
        if len(name) > 90:
            name = name[:87] + "..."
        print(f"| {name} | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | {100*r[2]/total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")
    # the accumulation kernel runs at several problem sizes in one bench command (2^20 bench steps, 2^19 replay): split by grid
    sub = cur.execute("select grid_x, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%msm_accum%' "
                      "group by grid_x order by grid_x desc").fetchall()
    if sub:
        print("\n# msm_accum_kernel by launch size (grid_x = lanes = sorted entries / K; 262144 lanes = the 2^20-point bench workload)")
        print("| grid_x | calls | avg_us | min_us | max_us |")
        print("|---|---|---|---|---|")
        for r in sub:
            print(f"| {r[0]} | {r[1]} | {r[2]/1e3:.1f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} |")
    if "--pmc" in sys.argv:
        try:
            cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
            print("\n# counters_collection columns:", cols)
            q = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
            print("| kernel | counter | samples | mean | sum |")
            print("|---|---|---|---|---|")
            for r in q:
                print(f"| {r[0][:80]} | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} |")
        except Exception as e:   # schema differs between rocprofv3 versions
            print("pmc query failed:", e)


if __name__ == "__main__":
    main()
