import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import halo2_lib_amd as H
from tests.util import rand_fr, domain_consts
ctx=H.Context(0)
for skip in (0, 1, 2, 3):
    ctx.set_param("ntt_debug_skip", skip)
    for log_n in (20, 22):
        n=1<<log_n; a=rand_fr(n,1); w,wi,d=domain_consts(log_n); dp=ctx.to_device(a)
        ctx.best_fft_dev(dp,w,log_n); ctx.sync(); ctx.timer_start()
        for _ in range(10): ctx.best_fft_dev(dp,w,log_n)
        print("skip",skip,"ntt",log_n, round(ctx.timer_stop()/10,4),"ms", flush=True); ctx.free(dp)
