"""Random circuit shapes against the oracle prover on the GPU: for every drawn (k, advice, lookup advice, fixed, instance, lookup bits) the HIP
create_proof's bytes must equal the oracle prover's on the same SRS / witness / RNG stream, and both verifiers must accept (the checks of
tests/test_plonk_prover.py::_check).  Shapes are drawn over the whole range the small oracle finishes in about a second — narrow and wide,
with and without lookups / instances / precomputed bases — so that batching boundaries the fixed test list does not name are crossed too.

    python tools/fuzz_shapes.py [seconds=120] [seed=1] [kmin kmax]      (kmin kmax: draw k uniformly from that range instead, e.g. 13 16)
    H2HIP_FUZZ_KNOBS=1: every shape also draws the selectable kernel paths (NTT kernel / tile, fused columns, lanes, deferred reduction, the
    pointwise kernels' arithmetic form, the prover's scheduling switches) — the non-default code on the real GPU, where a race shows
"""
import os, random, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from tests.test_plonk_prover import _check

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
krange = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None
ctx = H.Context()
t0, done, ks = time.time(), 0, {}
while time.time() - t0 < budget:
    k = rnd.randint(*krange) if krange else rnd.choice([6, 7, 8, 8, 9, 9, 10, 10, 11, 12])
    wide = rnd.random() < 0.25
    na = rnd.randint(1, 40 if wide and k <= 9 else 6)
    nl = rnd.choice([0, 1, 1, 2, 3]) if not wide else rnd.randint(0, 36 if k <= 8 else 4)
    nf = rnd.randint(1, 3)
    ni = rnd.choice([0, 0, 1, 2])
    lb = None if nl == 0 else rnd.randint(max(1, k - 4), k - 1)
    pre = rnd.random() < 0.5
    seed = rnd.randint(1, 1 << 20)
    shape = (k, na, nl, nf, ni, lb)
    knobs = {}
    if os.environ.get("H2HIP_FUZZ_KNOBS"):
        knobs = {"ntt_tile_kernel": rnd.choice([1, 1, 0]), "ntt_tile_bits": rnd.choice([10, 10, 8, 6]), "msm_fuse_cols": rnd.choice([0, 1, 4]),
                 "msm_lanes": rnd.choice([0, 1, 2]), "msm_defer_reduce": rnd.choice([1, 1, 0]),
                 # r04: the pointwise kernels' arithmetic form and the prover's scheduling switches
                 "quotient_29": rnd.choice([1, 1, 0]), "kate_29": rnd.choice([1, 1, 0]), "plonk_tail_overlap": rnd.choice([1, 1, 0]),
                 "plonk_permute_in_commit": rnd.choice([1, 1, 0]), "plonk_side_on_lanes": rnd.choice([1, 1, 0]), "clean_on_lane": rnd.choice([1, 0]),
                 # r05: host round trips through the mapped flag, the table-entry format (read when the SRS is set up), one-pass grand products,
                 # the wave-owned NTT pass (2^12+ points only)
                 "host_poll": rnd.choice([1, 1, 0]), "msm_table_split": rnd.choice([1, 1, 0]), "plonk_merge_products": rnd.choice([1, 1, 0]),
                 # r05, last: the grand products' lagrange_to_coeff in front of round 3's commitments; the lanes' first sorts one behind the other
                 "plonk_early_intt": rnd.choice([1, 1, 0]), "msm_stagger_sorts": rnd.choice([-1, 0, 1]),
                 # r06: the sort's histogram / scatter variants, the entries-per-lane rules, host advice uploaded inside round 1's commitment batch
                 "msm_hist_packed": rnd.choice([1, 1, 0]), "msm_scatter_full_lds": rnd.choice([0, 0, 1]), "msm_hist_split": rnd.choice([0, 1, 2, 4]),
                 "msm_sort_groups": rnd.choice([0, 0, 8, 30]), "msm_chunk_lone": rnd.choice([-1, -1, 0, 1]), "msm_chunk": rnd.choice([0, 0, 8, 24, 64]),
                 "plonk_lazy_upload": rnd.choice([1, 1, 0])}
        for name, val in knobs.items():
            ctx.set_param(name, val)
    try:
        out = _check(ctx, *shape, seed=seed, threads=16 if krange else 8, oracle_prover=True, precompute=pre, second_proof=rnd.random() < 0.3)
    except Exception as e:   # noqa: BLE001 — report the shape, then fail
        print("FAIL shape", shape, "seed", seed, "precompute", pre, "knobs", knobs, "->", repr(e)[:400], flush=True)
        sys.exit(1)
    out[6].free()
    out[7].free()
    done += 1
    ks[k] = ks.get(k, 0) + 1
    if done % 10 == 0:
        print("%d shapes ok, %.0f s (last %s precompute=%s)" % (done, time.time() - t0, shape, pre), flush=True)
print("fuzz_shapes: %d random shapes, proof bytes equal to the oracle prover's for every one; per k: %s; %.0f s" % (done, dict(sorted(ks.items())), time.time() - t0))
