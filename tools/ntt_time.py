import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from tests.util import rand_fr, domain_consts
ctx = H.Context(0)
for cbits in (3, 2, 1):
    ctx.set_param("ntt_min_col_bits", cbits)
    for log_n in (16, 19, 20, 21, 22, 23, 24):
        n = 1 << log_n; a = rand_fr(n, 1); w, wi, d = domain_consts(log_n); dp = ctx.to_device(a)
        ctx.best_fft_dev(dp, w, log_n); ctx.sync(); ctx.timer_start()
        for _ in range(10): ctx.best_fft_dev(dp, w, log_n)
        print("min_col_bits", cbits, "ntt", log_n, round(ctx.timer_stop() / 10, 4), "ms", flush=True); ctx.free(dp)
