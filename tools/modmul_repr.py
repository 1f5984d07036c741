"""the three 254-bit Montgomery multiplier representations side by side: saturated 8x32, unsaturated 9x29 (v_mad_u64_u32),
— products/s on this GPU (h2hip_bench_modmul / _modmul29).  The third representation of r03, 5x52 FP64-FMA (0.70x of 9x29), is a standalone
probe since r06: tools/probes/modmul52_probe.hip"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H

ctx = H.Context(0)
print("| representation | chains/lane | ms | products/s |")
print("|---|---|---|---|")
for name, fn, chains in (("saturated 8x32 (field.cuh fe_mul)", lambda c: ctx.bench_modmul(16384, 256, c), (1, 2)),
                         ("unsaturated 9x29 (fq29.cuh f29_mul)", lambda c: ctx.bench_modmul(16384, 256, c, unsaturated=True), (1, 2))):
    for c in chains:
        best = min((fn(c) for _ in range(3)), key=lambda t: t[0])
        print("| %s | %d | %.2f | %.3e |" % (name, c, best[0], best[1] / (best[0] * 1e-3)))
