"""r03 MSM tuning: per-kernel breakdown of a synchronous MSM and ms per MSM in batches of 4, over accumulate variants / chunk lengths / lanes.
usage: python tools/msm_r03.py [log_n,...] [name=v1:v2:...] ..."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import synthetic_scalars
from halo2_lib_amd import halo2_proofs as HP

sizes = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["19", "20"])]
pre = [a[4:] for a in sys.argv[2:] if a.startswith("pre:")]      # pre:name=value — set before the SRS tables are built (msm_window_bits)
sweeps = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(":")]) for a in sys.argv[2:] if not a.startswith("pre:")]
ctx = H.Context(0)
for kv in pre:
    ctx.set_param(kv.split("=")[0], int(kv.split("=")[1]))


def timed(f, reps=10):
    f(); ctx.sync()
    t = time.perf_counter()
    for _ in range(reps): f()
    ctx.sync()
    return (time.perf_counter() - t) / reps * 1e3


for log_n in sizes:
    n = 1 << log_n
    params = HP.ParamsKZG.setup(ctx, log_n, 0x1234567, precompute=True)
    cols = [ctx.to_device(synthetic_scalars(n, 10 + j)) for j in range(4)]
    for _ in range(20): ctx.msm_dev(params.g, cols[0], n)   # r06: the clocks have dropped during the set-up above — without this the FIRST configuration of a sweep reads 5 - 10 % slow
    ctx.sync()
    for combo in itertools.product(*[v for _, v in sweeps]) if sweeps else [()]:
        for (name, _), v in zip(sweeps, combo):
            ctx.set_param(name, v)
        tag = " ".join("%s=%d" % (nm, v) for (nm, _), v in zip(sweeps, combo))
        sync = timed(lambda: ctx.msm_dev(params.g, cols[0], n))
        b4 = timed(lambda: ctx.msm_batch_dev(params.g, cols, n)) / 4
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(4): ctx.msm_dev(params.g, cols[0], n)
        ctx.profile_enable(False)
        acc = ctx.profile_dump()
        parts = " ".join("%s=%.3f" % (k.replace("msm_", "").replace("_kernel", ""), v[0] / 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]))
        print("2^%d %s %s: sync %.3f ms, batch4 %.3f ms/MSM | %s" % (log_n, " ".join(pre), tag, sync, b4, parts), flush=True)
    for c in cols: ctx.free(c)
    params.free()
