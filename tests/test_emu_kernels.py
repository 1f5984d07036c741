"""CPU-only: runs the product's *kernel sources* through the host emulation shim (tests/emu/) against the
oracle, so index math / carries / sort / segmented accumulation are exercised in the GPU-less container.
This is NOT the product path (the product only ever loads the hipcc-built libh2hip.so)."""
import numpy as np
import pytest

import halo2_lib_amd as H
from oracle import bn254 as O
from oracle import c_oracle as CO
from tests.emu_util import emu_context
from tests.util import R, circuit_like_fr, domain_consts, fr, jac_to_affine_ints, rand_fr


@pytest.fixture(scope="module")
def ctx():
    c = emu_context()
    yield c
    c.close()


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 10, 11, 13, 15])
def test_ntt_matches_oracle(ctx, log_n):
    a = rand_fr(1 << log_n, log_n)
    w, winv, div = domain_consts(log_n)
    got = ctx.best_fft(a, w, log_n)
    assert np.array_equal(got, CO.best_fft(a, log_n, w, threads=4))
    assert np.array_equal(ctx.ifft(got, winv, log_n, div), a)


@pytest.mark.parametrize("tile", [4, 6, 8])
def test_ntt_small_tiles_force_many_passes(ctx, tile):
    """the generic pass kernel with tiles small enough that a 2^12 transform takes up to six passes: every (stages per pass, columns per
    tile, leftover-stage) combination, the fused coset scalings included"""
    ctx.set_param("ntt_tile_bits", tile)
    try:
        for log_n in (7, 12):
            a = rand_fr(1 << log_n, 3)
            w, winv, div = domain_consts(log_n)
            got = ctx.best_fft(a, w, log_n)
            assert np.array_equal(got, CO.best_fft(a, log_n, w))
            assert np.array_equal(ctx.ifft(got, winv, log_n, div), a)
        a = rand_fr(1 << 9, 8)
        we, weinv, ediv = domain_consts(11)
        z, zinv = fr([O.ZETA]), fr([O.ZETA * O.ZETA % R])
        ext = ctx.coeff_to_extended(a, 9, 11, we, z)
        assert np.array_equal(ext, CO.coeff_to_extended(a, 9, 11, we, z, threads=4))
        assert np.array_equal(ctx.extended_to_coeff(ext, 11, weinv, ediv, zinv)[: 1 << 9], a)
    finally:
        ctx.set_param("ntt_tile_bits", 10)


@pytest.mark.parametrize("k,ek", [(0, 2), (3, 5), (9, 11), (12, 14)])
def test_coset_extension(ctx, k, ek):
    a = rand_fr(1 << k, k)
    we, weinv, ediv = domain_consts(ek)
    z, zinv = fr([O.ZETA]), fr([O.ZETA * O.ZETA % R])
    ext = ctx.coeff_to_extended(a, k, ek, we, z)
    assert np.array_equal(ext, CO.coeff_to_extended(a, k, ek, we, z, threads=4))
    back = ctx.extended_to_coeff(ext, ek, weinv, ediv, zinv)
    assert np.array_equal(back[: 1 << k], a) and not back[1 << k:].any()


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 257, 3000])
@pytest.mark.parametrize("kind", ["uniform", "circuit"])
def test_msm_matches_oracle(ctx, n, kind):
    bases = CO.known_dlog_bases(n, fr([777 + n]), fr([13]))
    s = rand_fr(n, n) if kind == "uniform" else circuit_like_fr(n, n)
    b = ctx.bases_upload(bases)
    want = CO.best_multiexp(s, bases, threads=4)
    assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), want)
    assert [jac_to_affine_ints(ctx.msm(b, s, H.POINT_JACOBIAN))] == O.limbs_to_points(want)
    b.free()


@pytest.mark.parametrize("c,k1,seg", [(4, 2, 1), (5, 4, 2), (9, 16, 8), (13, 32, 16)])
def test_msm_parameter_sweep(ctx, c, k1, seg):
    n = 1500
    bases = CO.known_dlog_bases(n, fr([5]), fr([3]))
    b = ctx.bases_upload(bases)
    for name, v in (("msm_window_bits", c), ("msm_chunk", k1), ("msm_seg", seg)):
        ctx.set_param(name, v)
    try:
        for s in (rand_fr(n, c), circuit_like_fr(n, c)):
            assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases, threads=4))
    finally:
        for name, v in (("msm_window_bits", 0), ("msm_chunk", 0), ("msm_seg", 4)):
            ctx.set_param(name, v)
        b.free()


def test_msm_reference_edge_cases(ctx):
    # halo2-ecc/src/bn254/tests/msm_sum_infinity.rs:16-69 (+ identity base, zero scalars, n = 0, prefix of bases)
    P = O.g1_mul(O.G1_GEN, 0xDEADBEEF)
    cases = [
        ([1, 1, R - 2], [P, P, P]),
        ([1, 1, R - 1], [P, P, O.g1_add(P, P)]),
        ([1, 1, 1, R - 1], [P, P, P, O.g1_mul(P, 3)]),
        ([1, 1, 1, R - 1], [O.G1_GEN] * 3 + [O.g1_mul(O.G1_GEN, 3)]),
        ([R - 1, R - 1, 1, 1], [P, P, P, O.g1_add(P, P)]),
        ([5, 7], [None, P]),
        ([0, 0], [P, P]),
        ([R - 1], [P]),
    ]
    for sc, bs in cases:
        b = ctx.bases_upload(O.points_to_limbs(bs))
        assert O.limbs_to_points(ctx.msm(b, fr(sc), H.POINT_AFFINE)) == [O.msm_naive(sc, bs)]
        b.free()
    b = ctx.bases_upload(O.points_to_limbs([P, P, P]))
    assert O.limbs_to_points(ctx.msm(b, np.zeros((0, 4), dtype=np.uint64), H.POINT_AFFINE)) == [None]
    assert O.limbs_to_points(ctx.msm(b, fr([2, 3]), H.POINT_AFFINE)) == [O.g1_mul(P, 5)]
    with pytest.raises(H.H2HipError):
        ctx.msm(b, fr([1, 2, 3, 4]), H.POINT_AFFINE)   # more scalars than bases
    b.free()


def test_fr_batches(ctx):
    a, b, c = rand_fr(700, 1), rand_fr(700, 2), rand_fr(700, 3)
    assert np.array_equal(ctx.fr_mul(a, b), CO.fr_mul(a, b))
    assert np.array_equal(ctx.fr_add(a, b), CO.fr_add(a, b))
    assert np.array_equal(ctx.fr_sub(a, b), CO.fr_sub(a, b))
    assert np.array_equal(ctx.fr_mul_add(a, b, c), CO.fr_add(CO.fr_mul(a, b), c))
    # polynomial linear combinations of the multiopen argument: y + s*x and y*s
    sc = rand_fr(1, 9)
    rep = np.repeat(sc, len(a), 0)
    assert np.array_equal(ctx.fr_axpy(a, sc, b), CO.fr_add(a, CO.fr_mul(rep, b)))
    assert np.array_equal(ctx.fr_scale(a, sc), CO.fr_mul(a, rep))


def test_msm_multi_chunk_sort_and_deep_merge(ctx):
    # n > 4096 gives several counting-sort chunks per window (G > 1); msm_chunk=2 makes the partial list long
    # enough for three merge levels; the skewed half produces runs that span many workgroups.
    n = 9000
    bases = CO.known_dlog_bases(n, fr([11]), fr([7]))
    s = np.concatenate([rand_fr(n // 2, 5), circuit_like_fr(n - n // 2, 6)])
    b = ctx.bases_upload(bases)
    ctx.set_param("msm_window_bits", 12)
    ctx.set_param("msm_chunk", 2)
    try:
        assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases, threads=8))
    finally:
        ctx.set_param("msm_window_bits", 0)
        ctx.set_param("msm_chunk", 0)
        b.free()


@pytest.mark.parametrize("c", [0, 5, 11])
def test_msm_precomputed_bases(ctx, c):
    # H2HIP_BASES_PRECOMPUTE: table of 2^(c*w) multiples, shared bucket set, no window fold
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    n = 700
    P = O.g1_mul(O.G1_GEN, 99)
    pts = O.limbs_to_points(CO.known_dlog_bases(n - 3, fr([3]), fr([9]))) + [None, P, P]   # identity + duplicate bases
    bases = O.points_to_limbs(pts)
    ctx.set_param("msm_window_bits", c)
    try:
        b = ctx.bases_upload(bases, BASES_PRECOMPUTE)
        for s in (rand_fr(n, c), circuit_like_fr(n, c + 1)):
            assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases, threads=4))
        # a prefix of the table still works (n < table size)
        s = rand_fr(100, 3)
        assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases[:100], threads=2))
        b.free()
    finally:
        ctx.set_param("msm_window_bits", 0)


@pytest.mark.parametrize("n", [1, 2, 31, 257, 5000])
def test_batch_invert_and_grand_product(ctx, n):
    a = rand_fr(n, n)
    if n > 2:
        a[1] = 0   # 0 -> 0
        a[n - 1] = 0
    assert np.array_equal(ctx.fr_batch_invert(a), CO.fr_batch_invert(a))
    num, den = rand_fr(n, n + 1), rand_fr(n, n + 2)
    assert np.array_equal(ctx.fr_grand_product(num, den), CO.fr_grand_product(num, den))
    pp = ctx.fr_prefix_product(num)
    assert np.array_equal(pp, CO.fr_grand_product(num, np.repeat(fr([1]), n, axis=0))[1:])


@pytest.mark.parametrize("n", [1, 2, 9, 2048, 2049, 20000])
def test_eval_polynomial_and_kate_division(ctx, n):
    c = rand_fr(n, n)
    x = rand_fr(1, 99)
    assert np.array_equal(ctx.fr_eval_polynomial(c, x), CO.fr_eval_polynomial(c, x))
    if n >= 2:
        assert np.array_equal(ctx.fr_kate_division(c, x), CO.fr_kate_division(c, x))


@pytest.mark.parametrize("t,r_p", [(3, 57), (5, 60)])
def test_poseidon_batch_matches_reference_kat(ctx, t, r_p):
    from oracle.poseidon import Spec

    spec = Spec(t, 8, r_p)
    ctx.poseidon_set_spec(t, 8, r_p, fr([c for row in spec.constants for c in row]), fr([m for row in spec.mds for m in row]))
    # the reference's golden vector (halo2-base/src/poseidon/hasher/tests/state.rs:29-33,55-61) as instance 0
    states = [list(range(t))] + [O.random_scalars(t, 100 + i) for i in range(70)]
    inputs = [[0] * (t - 1)] + [O.random_scalars(t - 1, 200 + i) for i in range(70)]
    got = ctx.poseidon_permute(np.stack([fr(s) for s in states]), np.stack([fr(i) for i in inputs]))
    want = [spec.absorb_and_permute(s, i) for s, i in zip(states, inputs)]
    assert [O.limbs_to_ints(g, R) for g in got] == want
    kat3 = [7853200120776062878684798364095072458815029376092732009249414926327459813530,
            7142104613055408817911962100316808866448378443474503659992478482890339429929,
            6549537674122432311777789598043107870002137484850126429160507761192163713804]
    if t == 3:
        assert O.limbs_to_ints(got[0], R) == kat3
    # fewer inputs than RATE: padding 1 after the last input
    got1 = ctx.poseidon_permute(np.stack([fr(s) for s in states[:5]]), np.stack([fr(i[:1]) for i in inputs[:5]]))
    assert [O.limbs_to_ints(g, R) for g in got1] == [spec.absorb_and_permute(s, i[:1]) for s, i in zip(states[:5], inputs[:5])]


def test_quotient_flex_gate(ctx):
    k, ek = 6, 8
    ne, step = 1 << ek, 1 << (ek - k)
    acc, q, a, y = rand_fr(ne, 1), rand_fr(ne, 2), rand_fr(ne, 3), rand_fr(1, 4)
    got = ctx.quotient_flex_gate(acc, q, a, ek, k, y)
    rot = lambda v, r: np.roll(v, -r * step, axis=0)
    gate = CO.fr_mul(q, CO.fr_sub(CO.fr_add(a, CO.fr_mul(rot(a, 1), rot(a, 2))), rot(a, 3)))
    want = CO.fr_add(CO.fr_mul(acc, np.repeat(y, ne, axis=0)), gate)
    assert np.array_equal(got, want)


def test_msm_batch_pipelined(ctx):
    n = 600
    bases = CO.known_dlog_bases(n, fr([21]), fr([4]))
    b = ctx.bases_upload(bases)
    cols = [rand_fr(n, 1), circuit_like_fr(n, 2), rand_fr(n, 3), np.zeros((n, 4), dtype=np.uint64), rand_fr(n, 5)]
    dptrs = [ctx.to_device(c) for c in cols]
    got = ctx.msm_batch_dev(b, dptrs, n, H.POINT_AFFINE)
    for j, c in enumerate(cols):
        assert np.array_equal(got[j:j + 1], CO.best_multiexp(c, bases, threads=2))
    gotj = ctx.msm_batch_dev(b, dptrs[:2], n, H.POINT_JACOBIAN)
    assert [jac_to_affine_ints(gotj[0])] == O.limbs_to_points(got[0:1])
    assert np.array_equal(ctx.msm_batch(b, cols, H.POINT_AFFINE), got)   # host columns: staged per lane
    # precomputed tables: the columns are fused into multi-column MSMs (msm_fuse_cols per group, groups over the lanes)
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    bp = ctx.bases_upload(bases, BASES_PRECOMPUTE)
    many = dptrs + dptrs[:4]                       # 9 columns: groups of 5 + 4 at the default, 3 x 3 at fuse = 3
    for fuse in (16, 8, 3, 1):                     # 16: all nine columns in one fused MSM
        ctx.set_param("msm_fuse_cols", fuse)
        gotf = ctx.msm_batch_dev(bp, many, n, H.POINT_AFFINE)
        for j in range(len(many)):
            assert np.array_equal(gotf[j:j + 1], got[j % 5:j % 5 + 1]), (fuse, j)
    ctx.set_param("msm_fuse_cols", 0)              # auto: groups of 4 at this size
    assert np.array_equal(ctx.msm_batch_dev(bp, dptrs[:1], n, H.POINT_AFFINE), got[0:1])
    gota = ctx.msm_batch_dev(bp, many, n, H.POINT_AFFINE)
    assert all(np.array_equal(gota[j:j + 1], got[j % 5:j % 5 + 1]) for j in range(len(many)))
    bp.free()
    for d in dptrs:
        ctx.free(d)
    b.free()


def test_unsaturated_arithmetic_against_saturated(tmp_path):
    """fq29.cuh / ec29.cuh vs field.cuh / ec.cuh on the host (tests/emu/fq29_selftest.cpp), with limb-bound asserts."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fq29_selftest")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++17", "-O2", "-I", os.path.join(root, "tests", "emu"),
                           "-Wno-unused-value", "-o", exe, os.path.join(root, "tests", "emu", "fq29_selftest.cpp"), "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fq29 selftest OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_division_step_inversion_matches_fermat(tmp_path):
    """modinv.cuh (Bernstein-Yang division steps, the product's fe_inv) vs Fermat's a^(p-2) on the host, both fields"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "modinv_selftest")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++17", "-O2", "-I", os.path.join(root, "tests", "emu"),
                           "-Wno-unused-value", "-o", exe, os.path.join(root, "tests", "emu", "modinv_selftest.cpp"), "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "modinv selftest OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _quotient_identity_checks(ctx, k, ek, edge_patterns=False):
    ne, step = 1 << ek, 1 << (ek - k)
    rs = lambda seed: O.random_scalars(ne, seed)
    if edge_patterns:
        # field elements whose STORED limb patterns (Montgomery form) are the edge patterns of tests/util.edge_fr_values — 0, 1, r - 1, around 2^252 and
        # 2^253, the Montgomery constants — in every combination the cyclic offsets produce: the unsaturated kernels' bounds at their extremes
        from tests.util import edge_fr_values
        rinv = O.inv_mod(pow(2, 256, R), R)
        edge = [e * rinv % R for e in edge_fr_values()]
        rs = lambda seed: [edge[(i * (2 * seed + 1) + seed) % len(edge)] for i in range(ne)]
    acc, z, a, s, ap, sp, l0, ll, lb = [rs(i) for i in range(1, 10)]
    beta, gamma, y = O.random_scalars(3, 77)
    got = ctx.quotient_lookup(fr(acc), fr(z), fr(a), fr(s), fr(ap), fr(sp), fr(l0), fr(ll), fr(lb), ek, k, fr([beta]), fr([gamma]), fr([y]))
    assert O.limbs_to_ints(got, R) == O.quotient_lookup_terms(acc, z, a, s, ap, sp, l0, ll, lb, step, beta, gamma, y)
    cols, sigmas = [rs(20 + j) for j in range(3)], [rs(30 + j) for j in range(3)]
    zp = rs(40)
    we = O.omega_for(ek)
    # single-set call, a last set with the chaining term, and the separate loops upstream uses with several sets
    for (terms, prev, j0, rot) in ((1 | 8, None, 0, 0), (2 | 4 | 8, zp, 3, -7), (1 | 2 | 8, None, 0, 0), (1, None, 0, 0), (2, None, 0, 0), (4, zp, 0, -7), (8, None, 3, 0)):
        got = ctx.quotient_permutation_set(fr(acc), fr(z), None if prev is None else fr(prev), [fr(c) for c in cols], [fr(c) for c in sigmas], j0,
                                           fr(l0), fr(ll), fr(lb), ek, k, terms, rot, fr([beta]), fr([gamma]), fr([O.DELTA]), fr([O.ZETA]),
                                           fr([we]), fr([y]))
        want = O.quotient_permutation_set_terms(acc, z, prev, cols, sigmas, j0, l0, ll, lb, step, terms, rot % (1 << k), beta, gamma,
                                                O.DELTA, O.ZETA, we, y)
        assert O.limbs_to_ints(got, R) == want


@pytest.mark.parametrize("unsaturated", [1, 0])
def test_quotient_lookup_and_permutation_identities(ctx, unsaturated):
    """both arithmetic forms of the quotient kernels (fr29.cuh's 9 x 29-bit limbs: the default; the saturated kernels) against big-int arithmetic"""
    ctx.set_param("quotient_29", unsaturated)
    try:
        _quotient_identity_checks(ctx, 4, 6)
        _quotient_identity_checks(ctx, 4, 6, edge_patterns=True)
    finally:
        ctx.set_param("quotient_29", 1)


def test_msm_quad_and_serial_tails_agree(ctx):
    """bucket reduction / fold on quad-lane arithmetic (quad29.cuh) vs the one-lane kernels, incl. duplicate-heavy input
    (equal bucket sums -> the doubling branch) and the identity result"""
    n = 900
    P = O.g1_mul(O.G1_GEN, 31337)
    bases = np.concatenate([CO.known_dlog_bases(n - 300, fr([17]), fr([2])), np.repeat(O.points_to_limbs([P]), 300, axis=0)])
    cols = [rand_fr(n, 1), circuit_like_fr(n, 2), np.repeat(fr([0x1234567]), n, axis=0), np.zeros((n, 4), dtype=np.uint64)]
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    for flags in (0, BASES_PRECOMPUTE):
        b = ctx.bases_upload(bases, flags)
        for s in cols:
            want = CO.best_multiexp(s, bases, threads=4)
            for quad in (1, 0):
                ctx.set_param("msm_quad_tails", quad)
                assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), want)
        ctx.set_param("msm_quad_tails", 1)
        b.free()


def _lookup_permute_checks(ctx, sizes, big=True, presort=False):
    """big: a few full-size field elements among the keys (forces the bitonic network); without them every key is small and the counting
    sort runs.  presort: the table's keys are sorted once up front (h2hip_lookup_table_sort_dev), as the prover does at keygen."""
    rng = np.random.default_rng(5)
    for u, tbits in sizes:
        table = list(range(1 << tbits)) + [0] * max(0, u - (1 << tbits))   # range table padded with zeros, like halo2-base's
        table = table[:u] if len(table) >= u else table
        if len(table) < u:
            table += [0] * (u - len(table))
        vals = [int(x) for x in rng.integers(0, 1 << tbits, size=u)]
        vals[: u // 4] = [0] * (u // 4)            # many repeats of one value
        if u >= 8 and big:
            bigv = O.random_scalars(3, 9)
            vals[-3:], table[-3:] = bigv, bigv[::-1]  # a few full-size field elements
        tset = set(table)
        vals = [v if v in tset else 0 for v in vals]   # every input must occur in the (truncated / overwritten) table
        want_a, want_s = O.permute_expression_pair(vals, table)
        a = np.concatenate([fr(vals), rand_fr(7, 1)])   # rows beyond `usable` must be ignored
        s = np.concatenate([fr(table), rand_fr(7, 2)])
        got_a, got_s = ctx.lookup_permute(a, s, u, presort_table=presort)
        assert O.limbs_to_ints(got_a, R) == want_a
        assert O.limbs_to_ints(got_s, R) == want_s
    with pytest.raises(H.H2HipError):
        ctx.lookup_permute(fr([1, 2, 3, 99]), fr([1, 2, 3, 4]), 4)


def test_lookup_permute_expression_pair(ctx):
    _lookup_permute_checks(ctx, [(1, 1), (5, 2), (300, 5), (1024, 8), (3001, 9)])
    _lookup_permute_checks(ctx, [(1, 1), (300, 5), (1024, 8), (3001, 9), (3001, 11)], big=False)                 # counting sort (small keys)
    _lookup_permute_checks(ctx, [(5, 2), (3001, 9)], big=False, presort=True)
    _lookup_permute_checks(ctx, [(3001, 9)], big=True, presort=True)
    ctx.set_param("lookup_big_tile_bits", 12)      # the 4096-key LDS tile (default: from 2^19 keys) on 4096 / 8192 padded keys
    try:
        _lookup_permute_checks(ctx, [(4000, 9), (5000, 10)])
    finally:
        ctx.set_param("lookup_big_tile_bits", 19)


def test_lookup_permute_batch_emulated(ctx):
    from tests.golden_checks import check_lookup_permute_batch

    check_lookup_permute_batch(ctx)


def test_emulated_kernels_match_committed_golden_fixtures(ctx):
    from tests.golden_checks import check_backend_against_golden

    check_backend_against_golden(ctx)


def test_msm_randomized_shapes_emulated(ctx):
    """small-size twin of the GPU suite's randomized differential test (sizes, base kinds, distributions, batch shapes)"""
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    rng = np.random.default_rng(77)
    nmax = 1500
    bases_all = CO.known_dlog_bases(nmax, fr([99]), fr([5]))
    for case in range(4):
        n = int(rng.integers(1, nmax))
        flags = BASES_PRECOMPUTE if case % 2 else 0
        cols = [rand_fr(n, 10 * case), circuit_like_fr(n, 10 * case + 1), rand_fr(n, 10 * case + 2)]
        cols[2][: n // 2] = cols[2][0]
        b = ctx.bases_upload(bases_all[:n], flags)
        want = [CO.best_multiexp(s, bases_all[:n], threads=4) for s in cols]
        dptrs = [ctx.to_device(s) for s in cols]
        for fuse, defer in ((0, 1), (1, 1), (1, 0), (3, 1), (3, 0)):
            ctx.set_param("msm_fuse_cols", fuse)
            ctx.set_param("msm_defer_reduce", defer)
            got = ctx.msm_batch_dev(b, dptrs, n, H.POINT_AFFINE)
            for j in range(len(cols)):
                assert np.array_equal(got[j:j + 1], want[j]), (case, n, flags, fuse, defer, j)
        ctx.set_param("msm_fuse_cols", 0)
        ctx.set_param("msm_defer_reduce", 1)
        for d in dptrs:
            ctx.free(d)
        b.free()


@pytest.mark.parametrize("k,ek", [(3, 5), (2, 6), (0, 3)])
def test_divide_by_vanishing_poly(ctx, k, ek):
    """a[i] / ((zeta * w_ext^i)^n - 1): the few-inverses path (ext_k - k <= 3, kernel arguments) and the table path"""
    ne, n = 1 << ek, 1 << k
    a = O.random_scalars(ne, 3)
    we = O.omega_for(ek)
    got = ctx.divide_by_vanishing_poly(fr(a), ek, k, fr([we]), fr([O.ZETA]))
    want = [v * O.inv_mod((pow(O.ZETA * pow(we, i, R) % R, n, R) - 1) % R, R) % R for i, v in enumerate(a)]
    assert O.limbs_to_ints(got, R) == want


@pytest.mark.parametrize("n", [1, 37, 2500])
def test_prover_steps_emulated(ctx, n):
    from tests.golden_checks import check_prover_steps

    check_prover_steps(ctx, n)
    if n > 1:   # the multi-point division's saturated kernels (the default runs on unsaturated limbs: fr29.cuh)
        ctx.set_param("kate_29", 0)
        try:
            check_prover_steps(ctx, n)
        finally:
            ctx.set_param("kate_29", 1)


@pytest.mark.parametrize("unsaturated", [1, 0])
def test_quotient_batches_emulated(ctx, unsaturated):
    from tests.golden_checks import check_quotient_batches

    ctx.set_param("quotient_29", unsaturated)
    try:
        check_quotient_batches(ctx)
    finally:
        ctx.set_param("quotient_29", 1)


def test_ntt_batches_emulated(ctx):
    from tests.golden_checks import check_ntt_batches

    check_ntt_batches(ctx)


def _g2_msm_checks(ctx, sizes):
    """MSM over G2 against the oracle's G2 arithmetic (oracle/pairing.py): bases with known discrete logs Q_i = (k0 + i*d)*G2 built by repeated
    affine addition, so that the expected result is ONE scalar multiplication (sum_i s_i*(k0 + i*d))*G2; plus the identity / duplicate / P-P
    cases of the reference's G1 MSM tests (halo2-ecc/src/bn254/tests/msm_sum_infinity.rs:16-69) carried over to G2"""
    from oracle import pairing as PR

    Q = O.Q_MOD

    def limbs(points):   # (x.c0, x.c1, y.c0, y.c1) Montgomery limbs per point; None -> zeros
        vals = []
        for P_ in points:
            vals += [0, 0, 0, 0] if P_ is None else [P_[0][0], P_[0][1], P_[1][0], P_[1][1]]
        return O.ints_to_limbs(vals, Q).reshape(-1, 16)

    def from_limbs(a):
        v = O.limbs_to_ints(np.asarray(a).reshape(-1, 4), Q)
        return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))

    for n in sizes:
        k0, d = 31337 + n, 7
        D, cur, pts = PR.g2_mul(PR.G2_GEN, d), PR.g2_mul(PR.G2_GEN, k0), []
        for _ in range(n):
            pts.append(cur)
            cur = PR.g2_add(cur, D)
        s = O.limbs_to_ints(rand_fr(n, 60 + n), R)
        if n >= 4:
            s[1], s[2] = 0, 1
            pts[3] = None                                      # an identity base
        total = sum(si * (k0 + i * d) for i, (si, P_) in enumerate(zip(s, pts)) if P_ is not None) % R
        got = from_limbs(ctx.msm_g2(limbs(pts), fr(s)))
        assert got == PR.g2_mul(PR.G2_GEN, total), n
    P_ = PR.g2_mul(PR.G2_GEN, 0xDEADBEEF)
    for scal, bases in (([1, R - 1], [P_, P_]), ([1, 1, 1, R - 1], [P_, P_, P_, PR.g2_mul(P_, 3)]), ([0, 0], [P_, P_]), ([], [])):
        assert from_limbs(ctx.msm_g2(limbs(bases), fr(scal))) is None
    assert from_limbs(ctx.msm_g2(limbs([P_, P_]), fr([1, 1]))) == PR.g2_add(P_, P_)


def test_msm_g2_emulated(ctx):
    _g2_msm_checks(ctx, [1, 2, 37, 300])


def test_msm_batch_pipelined_and_mixed_bases(ctx):
    """the per-column lane pipeline with the deferred joint reduction for 2..5 columns incl. an all-zero one; and h2hip_msm_g1_multi_dev:
    columns over two different base sets in one call"""
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    n = 700
    bases_a = CO.known_dlog_bases(n, fr([21]), fr([4]))
    bases_b = CO.known_dlog_bases(n, fr([99]), fr([9]))
    ctx.set_param("msm_window_bits", 6)
    ba, bb = ctx.bases_upload(bases_a, BASES_PRECOMPUTE), ctx.bases_upload(bases_b, BASES_PRECOMPUTE)
    ctx.set_param("msm_window_bits", 0)
    cols = [rand_fr(n, 1), circuit_like_fr(n, 2), np.zeros((n, 4), dtype=np.uint64), rand_fr(n, 4), rand_fr(n, 5)]
    dptrs = [ctx.to_device(c) for c in cols]
    want_a = [CO.best_multiexp(c, bases_a, threads=2) for c in cols]
    want_b = [CO.best_multiexp(c, bases_b, threads=2) for c in cols]
    ctx.set_param("msm_fuse_cols", 1)          # per-column pipeline + deferred joint reduction (what large sizes use)
    try:
        for count in (2, 3, 5):
            got = ctx.msm_batch_dev(ba, dptrs[:count], n, H.POINT_AFFINE)
            assert all(np.array_equal(got[j:j + 1], want_a[j]) for j in range(count)), count
        sets = [ba, bb, bb, ba, bb]
        got = ctx.msm_multi_dev(sets, dptrs, n, H.POINT_AFFINE)
        for j, s in enumerate(sets):
            assert np.array_equal(got[j:j + 1], (want_a if s is ba else want_b)[j]), j
    finally:
        ctx.set_param("msm_fuse_cols", 0)
    # automatic fusing: runs of columns over the same set become fused multi-column MSMs (a group never spans two sets), reduced together
    many = [ba] * 7 + [bb] * 2 + [ba] * 3
    mptrs = [dptrs[j % 5] for j in range(len(many))]
    for defer in (1, 0):
        ctx.set_param("msm_defer_reduce", defer)
        got = ctx.msm_multi_dev(many, mptrs, n, H.POINT_AFFINE)
        for j, s2 in enumerate(many):
            assert np.array_equal(got[j:j + 1], (want_a if s2 is ba else want_b)[j % 5]), (defer, j)
    ctx.set_param("msm_defer_reduce", 1)
    for d in dptrs:
        ctx.free(d)
    ba.free()
    bb.free()


def test_msm_sort_degenerate_inputs(ctx):
    """the counting sort on degenerate inputs: several chunks per window, all scalars equal (every entry of a window lands in ONE key),
    windows with no entries at all, at the automatic and at a small window size"""
    n = 20000
    bases = CO.known_dlog_bases(n, fr([3]), fr([11]))
    b = ctx.bases_upload(bases)
    ones = np.repeat(fr([1]), n, axis=0)
    big = np.repeat(fr([R - 5]), n, axis=0)
    mixed = np.concatenate([ones[: n // 2], rand_fr(n - n // 2, 91)])
    try:
        for c in (0, 12):
            ctx.set_param("msm_window_bits", c)
            for s in (ones, big, mixed):
                assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases, threads=8)), c
    finally:
        ctx.set_param("msm_window_bits", 0)
        b.free()


def test_msm_sort_variants_r06(ctx):
    """r06's sort: the LDS histogram as 16-bit counter pairs or plain words, over 1 / 2 / 4 bucket sub-ranges per window, the scatter with its cursors
    only or the whole LDS, other chunk counts — the same point as the oracle's for uniform, all-equal (one bucket takes a whole chunk) and 0 / 1 scalars"""
    n = 20000
    bases = CO.known_dlog_bases(n, fr([5]), fr([13]))
    b = ctx.bases_upload(bases)
    cols = [rand_fr(n, 191), np.repeat(fr([R - 7]), n, axis=0), np.concatenate([np.repeat(fr([1]), n // 2, axis=0), np.repeat(fr([0]), n - n // 2, axis=0)])]
    want = [CO.best_multiexp(s, bases, threads=8) for s in cols]
    names = ("msm_hist_packed", "msm_hist_split", "msm_scatter_full_lds", "msm_sort_groups")
    try:
        for vals in ((1, 0, 0, 0), (1, 2, 0, 0), (1, 4, 1, 0), (0, 2, 0, 3), (0, 1, 1, 5), (1, 1, 0, 2)):
            for nm, v in zip(names, vals):
                ctx.set_param(nm, v)
            for s, w in zip(cols, want):
                assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), w), vals
    finally:
        for nm, v in zip(names, (1, 0, 0, 0)):
            ctx.set_param(nm, v)
        b.free()


def test_full_range_field_inputs(ctx):
    """limb patterns from the whole of [0, r) plus the edge patterns through every kernel family (tests/full_range_checks.py); the GPU suite
    runs the same checks at BASELINE sizes"""
    from tests import full_range_checks as F

    F.check_ntt(ctx, [3, 9, 11, 13], threads=4)
    F.check_coset(ctx, [(5, 7), (10, 12)], threads=4)
    F.check_msm_scalars(ctx, [33, 3000], threads=4, flags_list=(0, 1))
    F.check_pointwise(ctx, 2000)
    F.check_inverse_and_products(ctx, [1, 37, 3000])
    F.check_eval_and_division(ctx, [1, 9, 2049])


@pytest.mark.parametrize("tile_bits,tile_kernel", [(10, 1), (10, 0)])
def test_ntt_full_tile_kernels(ctx, tile_bits, tile_kernel):
    """ntt_tile_kernel (r04): the specialised full-tile pass kernel (the default; r06: its tile in LDS as limb planes at a swizzled index with skewed
    stage twiddles) and the generic pass kernel instead (ntt_tile_kernel = 0) — forward, inverse with its fused divisor, coset extension with zero
    padding and back, bit-exact"""
    ctx.set_param("ntt_tile_bits", tile_bits)
    ctx.set_param("ntt_tile_kernel", tile_kernel)
    try:
        for log_n in (11, 12, 13, 14, 16):
            a = rand_fr(1 << log_n, 70 + log_n)
            w, winv, div = domain_consts(log_n)
            got = ctx.best_fft(a, w, log_n)
            assert np.array_equal(got, CO.best_fft(a, log_n, w, threads=4)), log_n
            assert np.array_equal(ctx.ifft(got, winv, log_n, div), a), log_n
        for k, ek in ((10, 12), (11, 14), (13, 15)):
            a = rand_fr(1 << k, k)
            we, weinv, ediv = domain_consts(ek)
            z, zinv = fr([O.ZETA]), fr([O.ZETA * O.ZETA % R])
            ext = ctx.coeff_to_extended(a, k, ek, we, z)
            assert np.array_equal(ext, CO.coeff_to_extended(a, k, ek, we, z, threads=4)), (k, ek)
            back = ctx.extended_to_coeff(ext, ek, weinv, ediv, zinv)
            assert np.array_equal(back[: 1 << k], a) and not back[1 << k:].any()
    finally:
        ctx.set_param("ntt_tile_bits", 10)
        ctx.set_param("ntt_tile_kernel", 1)


def test_flex_gate_reference_kats_emulated(ctx):
    """the reference-held known answers of halo2-base/src/gates/tests/flex_gate.rs through the emulated batch kernels (same fixture as the GPU test)"""
    from tests import gate_kats as GK

    class B:
        add, sub, mul, mul_add, invert = (staticmethod(f) for f in (ctx.fr_add, ctx.fr_sub, ctx.fr_mul, ctx.fr_mul_add, ctx.fr_batch_invert))

    assert GK.check_all(B) == 31
