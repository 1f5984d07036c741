import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from bench import synthetic_bases, synthetic_scalars
ctx = H.Context(0)
names = ("msm_accum_kernel", "msm_merge", "msm_presum", "msm_seg", "msm_winsum", "msm_fold")
for log_n in (16, 19, 20):
    n = 1 << log_n
    bases_h = synthetic_bases(n, 1); s = synthetic_scalars(n, 2); ds = ctx.to_device(s)
    for pre in (1, 0):
        b = ctx.bases_upload(bases_h, pre)
        ref = None
        for quad in (0, 1):
            ctx.set_param("msm_quad_tails", quad)
            out = ctx.msm_dev(b, ds, n, H.POINT_AFFINE)
            if ref is None: ref = out
            assert np.array_equal(out, ref)
            ctx.profile_enable(True); ctx.profile_reset(); ctx.timer_start()
            for _ in range(5): ctx.msm_dev(b, ds, n)
            ms = ctx.timer_stop() / 5
            parts = {nm.replace("msm_", "").replace("_kernel", ""): round(ctx.profile_get(nm)[0] / 5, 3) for nm in names}
            ctx.profile_enable(False)
            print(f"n=2^{log_n} pre={pre} quad={quad}: {ms:.3f} ms", parts, flush=True)
        b.free()
    ctx.free(ds)
