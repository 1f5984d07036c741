#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each with --kernel-trace only).
Applies the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) counts 128-B read
requests as 64 B for 16 B/lane loads -> x2 (calibrated here: msm_digits_kernel streams exactly 32 MiB of scalars and
reports 16 MiB); WRITE_SIZE (KB) needs no correction (msm_digits_kernel writes exactly 64 MiB and reports 64 MiB).
usage: rocprof_pmc.py fetch.db write.db out.md out.json"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    lines = ["# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; KB -> bytes, FETCH x2 per guide)",
             "", "| kernel | launches | FETCH_SIZE raw KB | WRITE_SIZE KB | traffic bytes/launch (2*F+W)*1024 |", "|---|---|---|---|---|"]
    for name in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1])):
        fv, wv = f.get(name, (0, 0.0)), w.get(name, (0, 0.0))
        traffic = (2 * fv[1] + wv[1]) * 1024
        short = name.split("(")[0].replace("void ", "").replace("h2::", "")
        out[short] = {"fetch_kb_raw": fv[1], "write_kb": wv[1], "traffic_bytes_per_launch": traffic, "launches": max(fv[0], wv[0])}
        lines.append(f"| {short} | {max(fv[0], wv[0])} | {fv[1]:.0f} | {wv[1]:.0f} | {traffic:.3e} |")
    open(sys.argv[3], "w").write("\n".join(lines) + "\n")
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
