import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from bench import synthetic_bases, synthetic_scalars
ctx = H.Context(0)
names = ("msm_accum_kernel", "msm_merge")
for log_n in (19, 20):
    n = 1 << log_n
    bases_h = synthetic_bases(n, 1); s = synthetic_scalars(n, 2); ds = ctx.to_device(s)
    b = ctx.bases_upload(bases_h, 1)
    for k1 in (0, 21, 22, 32, 40, 43, 44, 48, 64, 86, 88, 128):
        ctx.set_param("msm_chunk", k1)
        ctx.msm_dev(b, ds, n)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.timer_start()
        for _ in range(8): ctx.msm_dev(b, ds, n)
        ms = ctx.timer_stop() / 8
        parts = {nm.replace("msm_", "").replace("_kernel", ""): round(ctx.profile_get(nm)[0] / 8, 3) for nm in names}
        ctx.profile_enable(False)
        print(f"n=2^{log_n} K={k1}: {ms:.3f} ms", parts, flush=True)
    b.free(); ctx.free(ds)
