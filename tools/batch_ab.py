"""2^19 / 2^20: synchronous MSM vs h2hip_msm_g1_batch_dev of 4 distinct uniform columns, over msm_lanes / msm_defer_reduce settings"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import synthetic_scalars
from halo2_lib_amd import halo2_proofs as HP

ctx = H.Context(0)
for log_n in (19, 20):
    n = 1 << log_n
    params = HP.ParamsKZG.setup(ctx, log_n, 0x1234567, precompute=True)
    cols = [ctx.to_device(synthetic_scalars(n, 10 + j)) for j in range(4)]
    def timed(f, reps=10):
        f(); ctx.sync()
        t = time.perf_counter()
        for _ in range(reps): f()
        ctx.sync()
        return (time.perf_counter() - t) / reps * 1e3
    print("2^%d sync: %.3f ms per MSM" % (log_n, timed(lambda: ctx.msm_dev(params.g, cols[0], n))), flush=True)
    ctx.set_param("msm_split_streams", 0)
    for lanes in (1, 2, 3):
        ctx.set_param("msm_lanes", lanes)
        ms = timed(lambda: ctx.msm_batch_dev(params.g, cols, n))
        print("2^%d batch4 lanes=%d (no split): %.3f ms per MSM" % (log_n, lanes, ms / 4), flush=True)
    ctx.set_param("msm_lanes", 0)
    for split in (1, 0, 1, 0):
        ctx.set_param("msm_split_streams", split)
        for cnt in (2, 4):
            ms = timed(lambda: ctx.msm_batch_dev(params.g, cols[:cnt], n))
            print("2^%d batch%d split_streams=%d: %.3f ms per MSM" % (log_n, cnt, split, ms / cnt), flush=True)
    ctx.set_param("msm_split_streams", 0)
    for c in cols: ctx.free(c)
    params.free()
