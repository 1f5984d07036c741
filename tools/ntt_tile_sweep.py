import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from tests.util import rand_fr, domain_consts
ctx = H.Context(0)
for tb in (8, 9, 10):
    try:
        ctx.set_param("ntt_tile_bits", tb)
    except Exception as e:
        print("tile_bits", tb, "rejected:", e); continue
    for log_n in (19, 21, 22):
        n = 1 << log_n; a = rand_fr(n, 1); w, wi, d = domain_consts(log_n); dp = ctx.to_device(a)
        ctx.best_fft_dev(dp, w, log_n); ctx.sync(); ctx.timer_start()
        for _ in range(10): ctx.best_fft_dev(dp, w, log_n)
        print("tile_bits", tb, "ntt", log_n, round(ctx.timer_stop() / 10, 4), "ms", flush=True); ctx.free(dp)
