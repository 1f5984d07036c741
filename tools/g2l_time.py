"""time h2hip_g1_to_lagrange and check it against the setup's Lagrange basis"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
ctx = H.Context(0)
for k in [int(a) for a in sys.argv[1:]] or [12, 16, 19]:
    params = HP.ParamsKZG.setup(ctx, k, 0xABCDEF0123 + k, precompute=False)
    ctx.sync(); t = time.time()
    gl = ctx.g1_to_lagrange(params.g, k, 0)
    ctx.sync(); dt = time.time() - t
    ok = np.array_equal(ctx.bases_download(gl), ctx.bases_download(params.g_lagrange))
    print(f"k={k}: g_to_lagrange {dt * 1e3:.1f} ms, equals setup's g_lagrange: {ok}", flush=True)
    gl.free(); params.free()
    assert ok
