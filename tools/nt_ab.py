"""A/B of msm_table_nontemporal on synchronous MSMs: accumulate kernel time at 2^19 / 2^20"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import synthetic_scalars
from halo2_lib_amd import halo2_proofs as HP

ctx = H.Context(0)
for log_n in (19, 20):
    n = 1 << log_n
    params = HP.ParamsKZG.setup(ctx, log_n, 0x1234567, precompute=True)
    ds = ctx.to_device(synthetic_scalars(n, 2))
    for nt in (0, 1, 0, 1):
        ctx.set_param("msm_table_nontemporal", nt)
        ctx.msm_dev(params.g, ds, n)
        ctx.profile_enable(True); ctx.profile_reset(); ctx.timer_start()
        for _ in range(8): ctx.msm_dev(params.g, ds, n)
        ms = ctx.timer_stop() / 8
        t, c = ctx.profile_get("msm_accum_kernel")
        ctx.profile_enable(False)
        print("2^%d nontemporal=%d: %.3f ms per MSM, accumulate %.3f ms" % (log_n, nt, ms, t / c), flush=True)
    params.free(); ctx.free(ds)
