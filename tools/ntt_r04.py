"""r04 NTT A/B: same-process sweep of context parameters (e.g. ntt_tile_kernel=0:1) over forward transforms 2^16..2^24 and the k = 19 / k = 21
workhorses, ms per transform (HIP events; 20 untimed repetitions — the clocks drop during the host-side setup of every size — then the better of two loops
of 20) and the fraction of the multiplier roof ((n/2) log2 n products against
h2hip_bench_modmul29's rate measured in the same run).   usage: python tools/ntt_r04.py name=v0:v1 [name=...]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from bench import synthetic_scalars

ctx = H.Context(0)
ms, nprod = ctx.bench_modmul(16384, 256, 2, unsaturated=True)
peak = nprod / (ms * 1e-3)
print("multiplier peak (9x29, this box): %.4g products/s" % peak, flush=True)
sweeps = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(":")]) for a in sys.argv[1:]]
REPS = 20


def timed(fn):
    for _ in range(REPS): fn()
    best = 1e9
    for _ in range(2):
        ctx.timer_start()
        for _ in range(REPS): fn()
        best = min(best, ctx.timer_stop() / REPS)
    return best


for combo in itertools.product(*[v for _, v in sweeps]) if sweeps else [()]:
    for (name, _), v in zip(sweeps, combo):
        ctx.set_param(name, v)
    tag = " ".join("%s=%d" % (nm, v) for (nm, _), v in zip(sweeps, combo))
    out = []
    for log_n in (16, 19, 20, 21, 22, 23, 24):
        n = 1 << log_n
        dom = HP.EvaluationDomain(ctx, 2, log_n)
        dp = ctx.to_device(synthetic_scalars(n, 1))
        t = timed(lambda: ctx.best_fft_dev(dp, dom.omega, log_n))
        out.append("2^%d %.4f (%.2f)" % (log_n, t, (n / 2 * log_n) / (t * 1e-3) / peak)); ctx.free(dp)
    for k, ek in ((19, 21), (21, 23)):
        d = HP.EvaluationDomain(ctx, 5 if ek - k == 2 and k == 19 else 4, k)
        src = ctx.to_device(synthetic_scalars(1 << k, 78)); ext = ctx.malloc(32 << ek)
        for name, fn in (("intt%d" % k, lambda: ctx.ifft_dev(src, d.omega_inv, k, d.ifft_divisor)),
                         ("coset%dto%d" % (k, ek), lambda: ctx.coeff_to_extended_dev(src, k, ext, ek, d.extended_omega, d.g_coset)),
                         ("cosetintt%d" % ek, lambda: ctx.extended_to_coeff_dev(ext, ek, d.extended_omega_inv, d.extended_ifft_divisor, d.g_coset_inv))):
            out.append("%s %.4f" % (name, timed(fn)))
        ctx.free(src); ctx.free(ext)
    print(tag, "|", " | ".join(out), flush=True)
