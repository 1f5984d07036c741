"""bucket-reduction segment length (msm_seg) sweep: synchronous MSM time and the seg / winsum kernels"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import synthetic_bases, synthetic_scalars
ctx = H.Context(0)
for log_n in (16, 19, 20):
    n = 1 << log_n
    s = synthetic_scalars(n, 2); ds = ctx.to_device(s)
    b = ctx.bases_upload(synthetic_bases(n, 1), 1)
    for L in (2, 4, 8, 16, 32):
        ctx.set_param("msm_seg", L)
        ctx.msm_dev(b, ds, n)
        ctx.timer_start()
        for _ in range(8): ctx.msm_dev(b, ds, n)
        ms = ctx.timer_stop() / 8
        ctx.profile_enable(True); ctx.profile_reset()
        for _ in range(4): ctx.msm_dev(b, ds, n)
        parts = {nm: round(ctx.profile_get(nm)[0] / 4, 3) for nm in ("msm_seg_kernel", "msm_winsum_kernel")}
        ctx.profile_enable(False)
        print(f"n=2^{log_n} L={L}: {ms:.3f} ms", parts, flush=True)
    b.free(); ctx.free(ds)
