"""HBM-counter calibration: runs the stream / 64-byte-gather / 128-byte-gather probes with known useful byte counts
and one synchronous 2^20-point MSM with precomputed tables, so that rocprofv3 PMC passes over THIS command see the probes and
msm_accum_kernel side by side.  Prints the probes' timing (GB/s of useful bytes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H

ctx = H.Context()
GiB = 1 << 30
for kind, label in ((0, "stream"), (64, "gather64"), (128, "gather128")):
    # 2^24 gathers (1 GiB of 64-byte entries = the volume of the accumulation's table reads at 2^20 points) out of a 1 GiB table
    ms, nb = ctx.bench_gather(kind, GiB, 1 << 18, 64)
    print("%-10s useful %.3f GB in %.3f ms -> %.1f GB/s useful" % (label, nb / 1e9, ms, nb / ms / 1e6), flush=True)
if "--msm" in sys.argv:
    from halo2_lib_amd import halo2_proofs as HP

    k = 20
    params = HP.ParamsKZG.setup(ctx, k, 0x1234567, precompute=True)
    g = np.random.default_rng(3)
    s = g.integers(0, 2**63, size=(1 << k, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.to_device(s)
    for _ in range(4):   # (r02 compared plain against non-temporal table gathers here: profiles/archive/r02_hbm_counter_calibration.md; only the latter is left)
        ctx.msm_dev(params.g, d, 1 << k)
    print("msm done")
