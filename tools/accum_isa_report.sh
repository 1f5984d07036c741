#!/bin/bash
# Where do msm_accum_kernel's spills sit?  (VERDICT r05 next 7 asked for vgpr_spill_count 0.)  Compiles csrc/msm.hip to gfx950 assembly and reports, for both
# instantiations: the register / spill / scratch metadata, the line range of the hot loop (the loop that holds the v_mad_u64_u32 stream) and every scratch_
# instruction with the loop it belongs to.  CPU only (hipcc cross-compiles).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o $T/msm.s $R/halo2-lib_amd/csrc/msm.hip 2>/dev/null
python3 - $T/msm.s <<'PY'
import re, sys
src = open(sys.argv[1]).read()
for inst in ("ILb1E", "ILb0E"):
    m = re.search(r"^(_ZN2h216msm_accum_kernel%s[^:\n]*):.*?s_endpgm" % inst, src, re.S | re.M)
    body = m.group(0).splitlines()
    name = m.group(1)
    meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(name), src, re.S)
    md = re.search(r"(\.private_segment_fixed_size:\s+\d+).*?(\.sgpr_spill_count:\s+\d+).*?(\.vgpr_count:\s+\d+).*?(\.vgpr_spill_count:\s+\d+)", src[src.index(".name:           " + name) - 1500: src.index(".name:           " + name) + 1500], re.S)
    print("== msm_accum_kernel<%s>: %d lines of assembly" % ("true (pre-split tables)" if inst == "ILb1E" else "false (packed tables)", len(body)))
    blk = src[src.index(".name:           " + name) - 1200: src.index(".name:           " + name) + 800]
    for key in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size"):
        mm = re.search(r"\.%s:\s+(\d+)" % key, blk)
        print("   %-28s %s" % (key, mm.group(1) if mm else "?"))
    # loops: label -> (first line, last line that branches back to it)
    labels = {l[:-1].split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if mm:
            t = mm.group(1) or mm.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i, t))
    # outermost loops only
    outer = [lp for lp in loops if not any(o[0] <= lp[0] and lp[1] <= o[1] and o != lp for o in loops)]
    merged = {}
    for a, b, t in outer:
        merged[t] = (a, max(b, merged.get(t, (a, b))[1]))
    for t, (a, b) in sorted(merged.items(), key=lambda kv: kv[1][0]):
        seg = body[a:b + 1]
        mads = sum("v_mad_u64_u32" in l for l in seg)
        scr = [a + i for i, l in enumerate(seg) if "scratch_" in l]
        print("   loop %-10s lines %5d-%5d: %4d v_mad_u64_u32, %d scratch instructions%s" % (t, a, b, mads, len(scr), "  <-- the accumulation loop" if mads > 1000 else ""))
    inloop = lambda i: any(a <= i <= b for a, b in merged.values())
    scr_all = [i for i, l in enumerate(body) if "scratch_" in l]
    print("   scratch instructions in all: %d, inside a loop: %d, outside (prologue / the wave-level merge after the loop): %d" % (len(scr_all), sum(inloop(i) for i in scr_all), sum(not inloop(i) for i in scr_all)))
PY
rm -rf $T
