"""same-run A/B of the MSM's bucket-major sort (msm_fold_windows) and sweeps of msm_seg / msm_chunk / msm_scatter_split under it:
2^19 / 2^20 points, synchronous MSM and h2hip_msm_g1_batch_dev of 4 distinct uniform columns"""
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
    def both(tag):
        s = timed(lambda: ctx.msm_dev(params.g, cols[0], n))
        b = timed(lambda: ctx.msm_batch_dev(params.g, cols, n)) / 4
        print("2^%d %-28s sync %.3f ms   batch4 %.3f ms per MSM" % (log_n, tag, s, b), flush=True)
    for fold in (0,):
        ctx.set_param("msm_fold_windows", fold)
        both("fold_windows=%d" % fold)
    ctx.set_param("msm_fold_windows", int(os.environ.get("H2_FOLD", "0")))
    for seg in (4, 8, 16, 32, 64, 128):
        ctx.set_param("msm_seg", seg)
        both("seg=%d" % seg)
    ctx.set_param("msm_seg", int(os.environ.get("H2_SEG_DEFAULT", "4")))
    for chunk in (0, 24, 32, 43, 48, 64, 86, 128):
        ctx.set_param("msm_chunk", chunk)
        both("chunk=%d" % chunk)
    ctx.set_param("msm_chunk", 0)
    for split in (0, 1, 2, 4, 8):
        ctx.set_param("msm_scatter_split", split)
        both("scatter_split=%d" % split)
    ctx.set_param("msm_scatter_split", 0)
    for c in cols: ctx.free(c)
    params.free()
