import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from tests.util import rand_fr, domain_consts
ctx = H.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
for a_ in sys.argv[2:]:   # name=value context parameters
    ctx.set_param(a_.split('=')[0], int(a_.split('=')[1]))
n = 1 << log_n; a = rand_fr(n, 1); w, wi, d = domain_consts(log_n); dp = ctx.to_device(a)
for _ in range(3): ctx.best_fft_dev(dp, w, log_n)
ctx.sync()
