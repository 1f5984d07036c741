"""Scale check on the GPU: SRS generation, precomputed tables, MSM and NTT at k = 20..23 via the size-independent identity
commit_lagrange(values) == commit(lagrange_to_coeff(values)) (both sides computed by the device, compared bit-exactly)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from bench import synthetic_scalars
ctx = H.Context(0)
for k in [int(a) for a in sys.argv[1:]] or [20, 21, 22]:
    n = 1 << k
    t = time.time(); params = HP.ParamsKZG.setup(ctx, k, 0xABCDEF0123456789ABCDEF + k, precompute=True); ctx.sync(); t_setup = time.time() - t
    dom = HP.EvaluationDomain(ctx, 5, k)
    vals = synthetic_scalars(n, k)
    dv = ctx.to_device(vals)
    ctx.timer_start(); c1 = ctx.msm_dev(params.g_lagrange, dv, n, H.POINT_AFFINE); t_msm = ctx.timer_stop()
    ctx.timer_start(); ctx.ifft_dev(dv, dom.omega_inv, k, dom.ifft_divisor); t_ntt = ctx.timer_stop()
    c2 = ctx.msm_dev(params.g, dv, n, H.POINT_AFFINE)
    ok = np.array_equal(c1, c2) and c1.any()
    print(f"k={k} setup+tables {t_setup:.2f}s  msm(affine out) {t_msm:.2f} ms  ifft {t_ntt:.2f} ms  commit_lagrange==commit: {ok}", flush=True)
    ctx.free(dv); params.free()
    assert ok
