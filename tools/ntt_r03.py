"""r03 NTT A/B: radix-8 pass kernel (ntt_radix8 = 1) vs the radix-4 kernel, forward transforms and the k = 19 workhorses, ms (HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from bench import synthetic_scalars

ctx = H.Context(0)
sweeps = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(":")]) for a in sys.argv[1:]]
import itertools
for combo in itertools.product(*[v for _, v in sweeps]) if sweeps else [()]:
    for (name, _), v in zip(sweeps, combo):
        ctx.set_param(name, v)
    tag = " ".join("%s=%d" % (nm, v) for (nm, _), v in zip(sweeps, combo))
    out = []
    for log_n in (16, 19, 20, 21, 22, 24):
        n = 1 << log_n
        dom = HP.EvaluationDomain(ctx, 2, log_n)
        dp = ctx.to_device(synthetic_scalars(n, 1))
        ctx.best_fft_dev(dp, dom.omega, log_n); ctx.sync(); ctx.timer_start()
        for _ in range(10): ctx.best_fft_dev(dp, dom.omega, log_n)
        out.append("2^%d %.4f" % (log_n, ctx.timer_stop() / 10)); ctx.free(dp)
    d19 = HP.EvaluationDomain(ctx, 5, 19)
    src = ctx.to_device(synthetic_scalars(1 << 19, 78)); ext = ctx.malloc(32 << 21)
    for name, fn in (("intt19", lambda: ctx.ifft_dev(src, d19.omega_inv, 19, d19.ifft_divisor)),
                     ("coset19to21", lambda: ctx.coeff_to_extended_dev(src, 19, ext, 21, d19.extended_omega, d19.g_coset)),
                     ("cosetintt21", lambda: ctx.extended_to_coeff_dev(ext, 21, d19.extended_omega_inv, d19.extended_ifft_divisor, d19.g_coset_inv))):
        fn(); ctx.sync(); ctx.timer_start()
        for _ in range(10): fn()
        out.append("%s %.4f" % (name, ctx.timer_stop() / 10))
    ctx.free(src); ctx.free(ext)
    print(tag, "|", " | ".join(out), flush=True)
