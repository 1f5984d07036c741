"""time h2hip_lookup_permute_dev (and the phases inside it) on the GPU"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP


def _fr_from_ints(vals):
    """canonical integers -> (n,4) u64 Montgomery limbs"""
    R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (v << 256) % R_MOD
        out[i] = [(m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
    return out


ctx = H.Context()
g = np.random.default_rng(3)
for k in (16, 19, 20):
    n = 1 << k
    u = n - 20
    a = _fr_from_ints([int(v) for v in g.integers(0, 1 << (k - 1), size=n)])
    s = _fr_from_ints([i if i < (1 << (k - 1)) else 0 for i in range(n)])
    da, ds = ctx.to_device(a), ctx.to_device(s)
    oa, os_ = ctx.malloc(n * 32), ctx.malloc(n * 32)
    call = lambda: ctx._chk(ctx.lib.h2hip_lookup_permute_dev(ctx.handle, da, ds, u, oa, os_))
    call(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    ctx.sync()
    print(f"k={k}: lookup_permute {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)
