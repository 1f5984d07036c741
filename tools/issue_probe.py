"""How many waves per SIMD does the 9x29 Montgomery product stream need to keep the VALU busy?  h2hip_bench_modmul29 with `blocks` = w x (CUs)
256-lane workgroups (one wave per SIMD and workgroup: w waves per SIMD) and 1 or 2 independent products interleaved per lane.
Prints products/s and the fraction of the best rate seen.   usage: python tools/issue_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H

ctx = H.Context(0)
cus = 256
rows = []
for chains in (1, 2):
    for w in (1, 2, 3, 4, 5, 6, 8):
        best = 0.0
        for _ in range(3):
            ms, n = ctx.bench_modmul(cus * w, 2048, chains, unsaturated=True)
            best = max(best, n / (ms * 1e-3))
        rows.append((chains, w, best))
top = max(r[2] for r in rows)
for chains, w, rate in rows:
    print("chains=%d waves/SIMD=%d: %.4g products/s (%.2f of the best)" % (chains, w, rate, rate / top), flush=True)
# the NTT pass kernel's radix-4 round in a loop, its parts switched on one at a time (fr_ops.hip: ntt_round_probe_kernel)
names = {16: "round, registers only", 17: "+ LDS round trip (conflict-free)", 18: "+ block barrier", 19: "+ first round's 4-way LDS conflicts"}
for mode in (16, 17, 18, 19):
    for w in (1, 2, 3) + ((4, 6) if mode == 16 else ()):
        best = 0.0
        for _ in range(3):
            ms, n = ctx.bench_modmul(cus * w, 512, mode, unsaturated=True)
            best = max(best, n / (ms * 1e-3))
        print("probe %-36s waves/SIMD=%d: %.4g products/s (%.2f of the multiplier peak)" % (names[mode], w, best, best / top), flush=True)
