"""h2hip_fr_batch_invert_dev: time vs elements per inversion"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import synthetic_scalars
ctx = H.Context(0)
for log_n in (16, 19, 21):
    n = 1 << log_n
    d = ctx.to_device(synthetic_scalars(n, 3))
    for run in (2, 4, 8, 16, 32):
        ctx.set_param("fr_invert_run", run)
        ctx._chk(ctx.lib.h2hip_fr_batch_invert_dev(ctx.handle, d, n)); ctx.sync()
        ctx.timer_start()
        for _ in range(10): ctx._chk(ctx.lib.h2hip_fr_batch_invert_dev(ctx.handle, d, n))
        print(f"n=2^{log_n} run={run}: {ctx.timer_stop() / 10:.4f} ms", flush=True)
    ctx.free(d)
