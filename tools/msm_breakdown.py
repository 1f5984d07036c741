"""per-kernel time of one synchronous MSM (ctx profiling events), precomputed bases"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import synthetic_scalars
from halo2_lib_amd import halo2_proofs as HP

NAMES = sys.argv[2].split(",") if len(sys.argv) > 2 else None
ctx = H.Context(0)
for kv in os.environ.get("H2_PARAMS", "").split():   # name=value context parameters
    ctx.set_param(kv.split("=")[0], int(kv.split("=")[1]))
for log_n in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["19", "20"])]:
    n = 1 << log_n
    s = synthetic_scalars(n, 2); ds = ctx.to_device(s)
    params = HP.ParamsKZG.setup(ctx, log_n, 0x1234567, precompute=True)
    b = params.g
    ctx.msm_dev(b, ds, n)
    ctx.profile_enable(True); ctx.profile_reset(); ctx.timer_start()
    reps = 8
    for _ in range(reps): ctx.msm_dev(b, ds, n)
    ms = ctx.timer_stop() / reps
    tot = 0.0
    print(f"n=2^{log_n}: {ms:.3f} ms per MSM (sync)")
    for nm in NAMES or H.h2hip.MSM_PROFILE_NAMES:
        t, c = ctx.profile_get(nm)
        if c:
            print(f"   {nm:28s} {t / reps:7.3f} ms  ({c // reps} launches)")
            tot += t / reps
    print(f"   {'sum of kernels':28s} {tot:7.3f} ms")
    ctx.profile_enable(False)
    params.free(); ctx.free(ds)
