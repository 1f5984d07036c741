"""MSM parameter sweep on the GPU.  usage: msm_sweep.py [log_n] ; prints per-kernel ms for a few (window, chunk) settings."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from bench import synthetic_bases, synthetic_scalars
from tests.util import circuit_like_fr
ctx = H.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
bases_h = synthetic_bases(n, 1)
names = ("msm_digits", "msm_hist_kernel", "msm_hist_scan", "scan_kernels", "msm_scatter", "msm_accum_kernel", "msm_merge", "msm_presum", "msm_seg", "msm_winsum", "msm_fold", "point_finish")
for kind in ("uniform", "circuit"):
    s = synthetic_scalars(n, 2) if kind == "uniform" else circuit_like_fr(n, 3)
    ds = ctx.to_device(s)
    for c in (13, 14, 15, 16):
        ctx.set_param("msm_window_bits", c)
        b = ctx.bases_upload(bases_h, 1)
        for k1 in (0,):
            ctx.set_param("msm_chunk", k1)
            ctx.msm_dev(b, ds, n)
            ctx.profile_enable(True); ctx.profile_reset()
            ctx.timer_start()
            for _ in range(5): ctx.msm_dev(b, ds, n)
            ms = ctx.timer_stop() / 5
            parts = {nm.replace("msm_", "").replace("_kernel", ""): round(ctx.profile_get(nm)[0] / 5, 3) for nm in names}
            ctx.profile_enable(False)
            print(f"n=2^{log_n} {kind} c={c} K={k1}: {ms:.3f} ms", parts, flush=True)
        b.free()
    ctx.free(ds)
