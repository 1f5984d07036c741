"""h2hip_msm_g1 (host scalars, the unmodified-prover path) vs h2hip_msm_g1_dev (resident scalars)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from bench import synthetic_bases, synthetic_scalars
ctx = H.Context(0)
for log_n in (16, 19, 20):
    n = 1 << log_n
    s = synthetic_scalars(n, 2); ds = ctx.to_device(s)
    b = ctx.bases_upload(synthetic_bases(n, 1), 1)
    for name, fn in (("dev ", lambda: ctx.msm_dev(b, ds, n)), ("host", lambda: ctx.msm(b, s, H.POINT_JACOBIAN))):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(8): fn()
        print(f"n=2^{log_n} {name}: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms", flush=True)
    cols = [s] * 6
    dps = [ds] * 6
    for name, fn in (("batch of 6, dev ", lambda: ctx.msm_batch_dev(b, dps, n)), ("batch of 6, host", lambda: ctx.msm_batch(b, cols))):
        fn()
        t0 = time.perf_counter()
        for _ in range(4): fn()
        print(f"n=2^{log_n} {name}: {(time.perf_counter() - t0) / 4 / 6 * 1e3:.3f} ms per MSM", flush=True)
    t0 = time.perf_counter()
    for _ in range(8): ctx.free(ctx.to_device(s))
    print(f"n=2^{log_n} upload+free only: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms", flush=True)
    b.free(); ctx.free(ds)
