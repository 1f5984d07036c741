"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, must be
bit-exact against the CPU oracle on the same seeded inputs; at BASELINE sizes also via closed forms."""
import numpy as np
import pytest

import halo2_lib_amd as H
from oracle import bn254 as O
from oracle import c_oracle as CO
from tests.util import R, circuit_like_fr, domain_consts, fr, jac_to_affine_ints, rand_fr

pytestmark = pytest.mark.gpu
NT = 16   # oracle threads


@pytest.fixture(scope="module")
def ctx():
    c = H.Context(device=0)   # the hipcc-built libh2hip.so on a real GPU; raises otherwise
    yield c
    c.close()


@pytest.mark.parametrize("log_n", [0, 1, 4, 9, 10, 11, 14, 16, 19, 20, 22])
def test_ntt_bit_exact(ctx, log_n):
    a = rand_fr(1 << log_n, log_n)
    w, winv, div = domain_consts(log_n)
    got = ctx.best_fft(a, w, log_n)
    assert np.array_equal(got, CO.best_fft(a, log_n, w, threads=NT))
    assert np.array_equal(ctx.ifft(got, winv, log_n, div), a)


def test_ntt_22_properties(ctx):
    # config #3 size: linearity + Horner spot checks (size-independent properties)
    log_n = 22
    n = 1 << log_n
    a, b = rand_fr(n, 1), rand_fr(n, 2)
    w, _, _ = domain_consts(log_n)
    fa, fb = ctx.best_fft(a, w, log_n), ctx.best_fft(b, w, log_n)
    fab = ctx.best_fft(CO.fr_add(a, b), w, log_n)
    assert np.array_equal(fab, CO.fr_add(fa, fb))
    wi = O.omega_for(log_n)
    for j in (0, 1, 123457, n - 1):
        assert np.array_equal(fa[j:j + 1], CO.fr_eval_polynomial(a, fr([pow(wi, j, R)])))


@pytest.mark.parametrize("k,ek", [(0, 2), (5, 7), (12, 14), (17, 19), (19, 21)])
def test_coset_extension(ctx, k, ek):
    a = rand_fr(1 << k, k)
    we, weinv, ediv = domain_consts(ek)
    z, zinv = fr([O.ZETA]), fr([O.ZETA * O.ZETA % R])
    ext = ctx.coeff_to_extended(a, k, ek, we, z)
    assert np.array_equal(ext, CO.coeff_to_extended(a, k, ek, we, z, threads=NT))
    back = ctx.extended_to_coeff(ext, ek, weinv, ediv, zinv)
    assert np.array_equal(back[: 1 << k], a) and not back[1 << k:].any()


@pytest.mark.parametrize("n", [1, 2, 33, 1000, 1 << 14, (1 << 16) + 3])
@pytest.mark.parametrize("kind", ["uniform", "circuit"])
def test_msm_bit_exact(ctx, n, kind):
    bases = CO.known_dlog_bases(n, fr([777 + n]), fr([13]))
    s = rand_fr(n, n) if kind == "uniform" else circuit_like_fr(n, n)
    b = ctx.bases_upload(bases)
    want = CO.best_multiexp(s, bases, threads=NT)
    assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), want)
    assert [jac_to_affine_ints(ctx.msm(b, s, H.POINT_JACOBIAN))] == O.limbs_to_points(want)
    b.free()


@pytest.mark.parametrize("kind", ["uniform", "circuit"])
def test_msm_2_20_bit_exact_and_closed_form(ctx, kind):
    # config #2: 2^20 points.  Checked against the C oracle AND the known-dlog closed form
    # sum_i s_i*(k0+i*d) * G  (SURVEY §8c), which does not depend on any MSM implementation.
    n, k0, d = 1 << 20, 987654321, 31
    bases = CO.known_dlog_bases(n, fr([k0]), fr([d]))
    s = rand_fr(n, 7) if kind == "uniform" else circuit_like_fr(n, 7)
    b = ctx.bases_upload(bases)
    got = ctx.msm(b, s, H.POINT_AFFINE)
    assert np.array_equal(got, CO.best_multiexp(s, bases, threads=NT))
    si = O.limbs_to_ints(s, R)
    total = sum(v * (k0 + i * d) for i, v in enumerate(si)) % R
    assert O.limbs_to_points(got) == [O.g1_mul(O.G1_GEN, total)] == [O.g1_mul_complete(O.G1_GEN, total)]   # two independent G1 implementations
    # device-resident scalars give the same answer (the bench path)
    ds = ctx.to_device(s)
    assert [jac_to_affine_ints(ctx.msm_dev(b, ds, n, H.POINT_JACOBIAN))] == O.limbs_to_points(got)
    ctx.free(ds)
    b.free()


def test_msm_reference_edge_cases(ctx):
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
    with pytest.raises(H.H2HipError):
        ctx.msm(b, fr([1, 2, 3, 4]), H.POINT_AFFINE)
    b.free()


def test_msm_many_duplicates_and_single_bucket(ctx):
    # all scalars equal, all bases equal: every entry lands in the same buckets and every add is a doubling
    n = 5000
    P = O.g1_mul(O.G1_GEN, 424242)
    bases = np.repeat(O.points_to_limbs([P]), n, axis=0)
    s = np.repeat(fr([0x1234567]), n, axis=0)
    b = ctx.bases_upload(bases)
    assert O.limbs_to_points(ctx.msm(b, s, H.POINT_AFFINE)) == [O.g1_mul(P, 0x1234567 * n)]
    b.free()


def test_msm_sizes_and_base_kinds(ctx):
    """a ladder of sizes over plain and precomputed bases (one chunk / several chunks per window, a window whose entries all share one key),
    three scalar distributions each, against the oracle"""
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    for n, flags in ((1, 0), (777, 0), (1 << 10, 0), (20000, 0), (1 << 14, BASES_PRECOMPUTE), (70001, BASES_PRECOMPUTE)):
        bases = CO.known_dlog_bases(n, fr([5]), fr([13]))
        b = ctx.bases_upload(bases, flags)
        for s in (rand_fr(n, n), circuit_like_fr(n, n + 1), np.repeat(fr([1]), n, axis=0)):
            assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases, threads=NT)), (n, flags)
        b.free()


def test_fr_batches(ctx):
    a, b, c = rand_fr(100003, 1), rand_fr(100003, 2), rand_fr(100003, 3)
    assert np.array_equal(ctx.fr_mul(a, b), CO.fr_mul(a, b))
    assert np.array_equal(ctx.fr_add(a, b), CO.fr_add(a, b))
    assert np.array_equal(ctx.fr_sub(a, b), CO.fr_sub(a, b))
    assert np.array_equal(ctx.fr_mul_add(a, b, c), CO.fr_add(CO.fr_mul(a, b), c))
    sc = rand_fr(1, 9)
    rep = np.repeat(sc, len(a), 0)
    assert np.array_equal(ctx.fr_axpy(a, sc, b), CO.fr_add(a, CO.fr_mul(rep, b)))
    assert np.array_equal(ctx.fr_scale(a, sc), CO.fr_mul(a, rep))


def test_flex_gate_reference_kats(ctx):
    """the reference's own known answers for GateInstructions' witness values (halo2-base/src/gates/tests/flex_gate.rs:11-150, :217:
    tests/golden/flex_gate_reference_kats.json) through h2hip_fr_{add,sub,mul,mul_add}_batch_dev and h2hip_fr_batch_invert_dev"""
    from tests import gate_kats as GK

    class B:
        add, sub, mul, mul_add, invert = (staticmethod(f) for f in (ctx.fr_add, ctx.fr_sub, ctx.fr_mul, ctx.fr_mul_add, ctx.fr_batch_invert))

    assert GK.check_all(B) == 31


@pytest.mark.parametrize("n", [700, 1 << 16])
def test_msm_precomputed_bases(ctx, n):
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    P = O.g1_mul(O.G1_GEN, 99)
    pts = CO.known_dlog_bases(n - 3, fr([3]), fr([9]))
    bases = np.concatenate([pts, O.points_to_limbs([None, P, P])])
    b = ctx.bases_upload(bases, BASES_PRECOMPUTE)
    for s in (rand_fr(n, 1), circuit_like_fr(n, 2)):
        assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases, threads=NT))
    s = rand_fr(100, 3)
    assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), CO.best_multiexp(s, bases[:100], threads=2))
    b.free()


def test_msm_2_20_precomputed_closed_form(ctx):
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    n, k0, d = 1 << 20, 123456789, 17
    bases = CO.known_dlog_bases(n, fr([k0]), fr([d]))
    b = ctx.bases_upload(bases, BASES_PRECOMPUTE)
    for seed, s in ((1, rand_fr(n, 11)), (2, circuit_like_fr(n, 12))):
        got = ctx.msm(b, s, H.POINT_AFFINE)
        si = O.limbs_to_ints(s, R)
        total = sum(v * (k0 + i * d) for i, v in enumerate(si)) % R
        assert O.limbs_to_points(got) == [O.g1_mul(O.G1_GEN, total)] == [O.g1_mul_complete(O.G1_GEN, total)]
    b.free()


@pytest.mark.parametrize("n", [1, 257, 1 << 19])
def test_batch_invert_and_grand_product(ctx, n):
    a = rand_fr(n, n)
    if n > 2:
        a[1] = 0
        a[n - 1] = 0
    assert np.array_equal(ctx.fr_batch_invert(a), CO.fr_batch_invert(a))
    num, den = rand_fr(n, n + 1), rand_fr(n, n + 2)
    assert np.array_equal(ctx.fr_grand_product(num, den), CO.fr_grand_product(num, den))


@pytest.mark.parametrize("n", [1, 2049, 1 << 19, (1 << 21) + 5])
def test_eval_polynomial_and_kate_division(ctx, n):
    c = rand_fr(n, n)
    x = rand_fr(1, 99)
    assert np.array_equal(ctx.fr_eval_polynomial(c, x), CO.fr_eval_polynomial(c, x))
    if n >= 2:
        assert np.array_equal(ctx.fr_kate_division(c, x), CO.fr_kate_division(c, x))


@pytest.mark.parametrize("t,r_p", [(3, 57), (5, 60)])
def test_poseidon_batch(ctx, t, r_p):
    from oracle.poseidon import Spec

    spec = Spec(t, 8, r_p)
    ctx.poseidon_set_spec(t, 8, r_p, fr([c for row in spec.constants for c in row]), fr([m for row in spec.mds for m in row]))
    n = 300
    states = [list(range(t))] + [O.random_scalars(t, 100 + i) for i in range(n - 1)]
    inputs = [[0] * (t - 1)] + [O.random_scalars(t - 1, 900 + i) for i in range(n - 1)]
    got = ctx.poseidon_permute(np.stack([fr(s) for s in states]), np.stack([fr(i) for i in inputs]))
    assert [O.limbs_to_ints(g, R) for g in got] == [spec.absorb_and_permute(s, i) for s, i in zip(states, inputs)]
    if t == 3:   # reference golden vector, halo2-base/src/poseidon/hasher/tests/state.rs:29-33
        assert O.limbs_to_ints(got[0], R)[0] == 7853200120776062878684798364095072458815029376092732009249414926327459813530


def test_quotient_flex_gate(ctx):
    k, ek = 12, 14
    ne, step = 1 << ek, 1 << (ek - k)
    acc, q, a, y = rand_fr(ne, 1), rand_fr(ne, 2), rand_fr(ne, 3), rand_fr(1, 4)
    got = ctx.quotient_flex_gate(acc, q, a, ek, k, y)
    rot = lambda v, r: np.roll(v, -r * step, axis=0)
    gate = CO.fr_mul(q, CO.fr_sub(CO.fr_add(a, CO.fr_mul(rot(a, 1), rot(a, 2))), rot(a, 3)))
    assert np.array_equal(got, CO.fr_add(CO.fr_mul(acc, np.repeat(y, ne, axis=0)), gate))


def test_msm_batch_pipelined(ctx):
    n = 1 << 16
    bases = CO.known_dlog_bases(n, fr([21]), fr([4]))
    for flags in (0, 1):
        b = ctx.bases_upload(bases, flags)
        cols = [rand_fr(n, 1), circuit_like_fr(n, 2), rand_fr(n, 3), np.zeros((n, 4), dtype=np.uint64), rand_fr(n, 5)]
        dptrs = [ctx.to_device(c) for c in cols]
        got = ctx.msm_batch_dev(b, dptrs, n, H.POINT_AFFINE)
        for j, c in enumerate(cols):
            assert np.array_equal(got[j:j + 1], CO.best_multiexp(c, bases, threads=NT))
        for d in dptrs:
            ctx.free(d)
        b.free()


@pytest.mark.parametrize("log_n,batch", [(19, 4), (19, 8), (20, 4), (20, 8)])
def test_msm_batch_bench_path_closed_form(ctx, log_n, batch):
    """The code path bench.py times, at the sizes it times: h2hip_msm_g1_batch_dev over precomputed bases (lane pipeline, deferred
    joint bucket reduction), `batch` DISTINCT scalar columns, bases built on the GPU exactly as bench.py builds them; every
    result is checked against the known-dlog closed form (no MSM implementation involved) and column 0 against the C oracle."""
    import torch

    import bench as B
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    n, k0, d = 1 << log_n, 0x1234567, 3
    dev = torch.device("cuda", 0)
    pts = B.known_dlog_bases_gpu(ctx, torch, dev, n, k0, d)
    host_pts = pts.cpu().numpy().view(np.uint64).reshape(n, 8)
    idx = [0, 1, 2, 12345, n - 1]
    assert np.array_equal(host_pts[idx], O.points_to_limbs([O.g1_mul(O.G1_GEN, k0 + i * d) for i in idx]))
    b = ctx.bases_from_device(pts.data_ptr(), n, BASES_PRECOMPUTE)
    cols = [rand_fr(n, 500 + j) if j % 3 else circuit_like_fr(n, 500 + j) for j in range(batch)]
    cols[1] = B.synthetic_scalars(n, 2001)
    dcols = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    torch.cuda.synchronize()
    assert ctx.get_param("msm_defer_reduce") == 1 and ctx.get_param("msm_fuse_cols") == 0
    got = ctx.msm_batch_dev(b, [t.data_ptr() for t in dcols], n, H.POINT_JACOBIAN)
    for j in range(batch):
        assert B.jac_to_affine(got[j]) == O.g1_mul_complete(O.G1_GEN, B.closed_form_dlog(cols[j], k0, d)), (log_n, batch, j)
    assert [B.jac_to_affine(got[0])] == O.limbs_to_points(CO.best_multiexp(cols[0], host_pts, threads=NT))
    b.free()


@pytest.mark.parametrize("unsaturated", [1, 0])
def test_quotient_lookup_and_permutation_identities(ctx, unsaturated):
    from tests.test_emu_kernels import _quotient_identity_checks

    ctx.set_param("quotient_29", unsaturated)
    try:
        _quotient_identity_checks(ctx, 7, 9)
        _quotient_identity_checks(ctx, 7, 9, edge_patterns=True)
    finally:
        ctx.set_param("quotient_29", 1)


def test_lookup_permute_expression_pair(ctx):
    from tests.test_emu_kernels import _lookup_permute_checks

    _lookup_permute_checks(ctx, [(5, 2), (3001, 9), ((1 << 17) - 20, 15), ((1 << 19) - 6, 18)])   # the last one sorts with 4096-key tiles
    _lookup_permute_checks(ctx, [(3001, 9), ((1 << 17) - 20, 15), ((1 << 19) - 7, 18)], big=False)                  # counting sort
    _lookup_permute_checks(ctx, [((1 << 19) - 7, 18)], big=False, presort=True)
    _lookup_permute_checks(ctx, [((1 << 16) - 7, 12)], big=True, presort=True)


def test_gpu_matches_committed_golden_fixtures(ctx):
    from tests.golden_checks import check_backend_against_golden

    check_backend_against_golden(ctx)


def test_msm_randomized_shapes(ctx):
    """differential test over random sizes / base kinds / scalar distributions / batch shapes (fused, deferred, per-lane)"""
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    rng = np.random.default_rng(20260925)
    nmax = 40000
    bases_all = CO.known_dlog_bases(nmax, fr([4242]), fr([17]))
    for case in range(10):
        n = int(rng.integers(1, nmax))
        flags = BASES_PRECOMPUTE if rng.integers(0, 2) else 0
        cols = []
        for j in range(int(rng.integers(1, 7))):
            kind = int(rng.integers(0, 4))
            s = rand_fr(n, 100 * case + j) if kind == 0 else circuit_like_fr(n, 100 * case + j)
            if kind == 2:
                s[: n // 2] = s[0]                       # one scalar repeated: long runs in one bucket
            if kind == 3:
                s[rng.integers(0, n, size=max(1, n // 3))] = 0
            cols.append(s)
        b = ctx.bases_upload(bases_all[:n], flags)
        want = [CO.best_multiexp(s, bases_all[:n], threads=8) for s in cols]
        assert np.array_equal(ctx.msm(b, cols[0], H.POINT_AFFINE), want[0]), (case, n, flags)
        assert np.array_equal(ctx.msm_batch(b, cols, H.POINT_AFFINE), np.concatenate(want)), (case, n, flags, "host columns")
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


@pytest.mark.gpu
def test_msm_batch_many_small_columns(ctx):
    """a wide shape's commitment round: 70 columns of 3000 scalars over two base sets — fused groups of up to 16 columns that never span two
    sets, the deferred bucket reduction in chunks of 64 columns — against the oracle, column by column"""
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE

    n = 3000
    bases_a = CO.known_dlog_bases(n, fr([77]), fr([3]))
    bases_b = CO.known_dlog_bases(n, fr([1234567]), fr([11]))
    ba, bb = ctx.bases_upload(bases_a, BASES_PRECOMPUTE), ctx.bases_upload(bases_b, BASES_PRECOMPUTE)
    cols = [rand_fr(n, 500 + j) if j % 3 else circuit_like_fr(n, 500 + j) for j in range(70)]
    which = [ba] * 50 + [bb] * 3 + [ba] * 17          # runs of 50, 3 and 17 columns
    dptrs = [ctx.to_device(c) for c in cols]
    try:
        got = ctx.msm_multi_dev(which, dptrs, n, H.POINT_AFFINE)
        for j in range(70):
            want = CO.best_multiexp(cols[j], bases_a if which[j] is ba else bases_b, threads=8)
            assert np.array_equal(got[j:j + 1], want), j
    finally:
        for d in dptrs:
            ctx.free(d)
        ba.free()
        bb.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(1 << 18) + 5, 1 << 20, 1 << 21])
def test_kate_division_by_coefficient_ranges(ctx, n):
    """h2hip_fr_kate_division_range_dev at the sharded prover's sizes (ranges of 2^16 .. 2^20 coefficients: every tile variant, ranges that end
    on a tile boundary so that the carry opens a tile of its own): the ranges' quotients, with carries assembled from partial evaluations the
    way create_proof does, concatenate to the whole-polynomial quotient"""
    f = rand_fr(n, 51)
    for m, world in ((1, 2), (4, 3), (5, 8)):
        pts, ws = rand_fr(m, 52 + m), rand_fr(m, 60 + m)
        want = ctx.fr_kate_division_multi(f, pts, ws)
        cuts = [n * r // world for r in range(world + 1)]
        if world == 3:
            cuts[1] += 7   # uneven, not tile aligned
        # partial evaluations E_r(p) = sum_j f[lo_r + j] p^j, then carry of rank r = sum_{s > r} E_s(p) * p^(lo_s - hi_r)  (big-int on the host)
        E = [[O.limbs_to_ints(ctx.fr_eval_polynomial(f[cuts[r]:cuts[r + 1]], pts[j:j + 1]), R)[0] for j in range(m)] for r in range(world)]
        pi = O.limbs_to_ints(pts, R)
        got = []
        for r in range(world):
            carries = [sum(E[s][j] * pow(pi[j], cuts[s] - cuts[r + 1], R) for s in range(r + 1, world)) % R for j in range(m)]
            got.append(ctx.fr_kate_division_range(f[cuts[r]:cuts[r + 1]], pts, ws, fr(carries)))
        got = np.concatenate(got)
        assert np.array_equal(got[:-1], want) and not got[-1].any(), (n, m, world)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(1 << 17) + 3, 1 << 18, (1 << 19) - 1, 1 << 20])
def test_kate_division_multi_tile_lengths(ctx, n):
    """the multi-point division picks its tile (1 / 2 / 4 / 8 coefficients per lane) by the polynomial's length: every variant against the
    weighted sum of single-point divisions (pinned to big-int arithmetic by test_prover_steps), 3 and 5 points (one and two passes)"""
    f = rand_fr(n, 31)
    for m in (3, 5):
        pts, ws = rand_fr(m, 32 + m), rand_fr(m, 40 + m)
        want = ctx.fr_linear_combination([ctx.fr_kate_division(f, pts[j:j + 1]) for j in range(m)], ws)
        assert np.array_equal(ctx.fr_kate_division_multi(f, pts, ws), want), (n, m)
        ctx.set_param("kate_29", 0)   # ... and the saturated kernels (the default runs on unsaturated limbs)
        try:
            assert np.array_equal(ctx.fr_kate_division_multi(f, pts, ws), want), (n, m)
        finally:
            ctx.set_param("kate_29", 1)


@pytest.mark.gpu
def test_lookup_permute_batch(ctx):
    from tests.golden_checks import check_lookup_permute_batch

    check_lookup_permute_batch(ctx, u=70000, bits=13, count=9)


@pytest.mark.gpu
@pytest.mark.parametrize("unsaturated", [1, 0])
def test_quotient_batches(ctx, unsaturated):
    from tests.golden_checks import check_quotient_batches

    ctx.set_param("quotient_29", unsaturated)
    try:
        check_quotient_batches(ctx, k=9, gate_cols=130)
    finally:
        ctx.set_param("quotient_29", 1)


@pytest.mark.gpu
def test_ntt_batches(ctx):
    from tests.golden_checks import check_ntt_batches

    check_ntt_batches(ctx, ks=(3, 11, 14), ncols=70)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 37, 2500, 70001])
def test_prover_steps(ctx, n):
    from tests.golden_checks import check_prover_steps

    check_prover_steps(ctx, n)


@pytest.mark.gpu
def test_two_contexts_interleaved_and_second_thread(ctx):
    """Two contexts (own streams, scratch and twiddle caches) used alternately and one of them from a second host thread — the C ABI's
    contract is one thread at a time PER CONTEXT; every entry point makes its context's device current (H2_DEVICE_GUARD).  With more than
    one GPU visible the second context sits on the last device."""
    import threading

    ndev = ctx.lib.h2hip_device_count
    cnt = __import__("ctypes").c_int(0)
    assert ndev(__import__("ctypes").byref(cnt)) == 0 and cnt.value >= 1
    other = H.Context(device=cnt.value - 1)
    try:
        log_n, n = 14, 1 << 12
        a = rand_fr(1 << log_n, 91)
        w, _, _ = domain_consts(log_n)
        bases = CO.known_dlog_bases(n, fr([4242]), fr([11]))
        s = rand_fr(n, 92)
        want_ntt, want_msm = CO.best_fft(a, log_n, w, threads=NT), CO.best_multiexp(s, bases, threads=NT)
        b_main, b_other = ctx.bases_upload(bases), other.bases_upload(bases, 1)
        out = {}

        def worker():
            try:
                for _ in range(6):
                    out["msm"] = other.msm(b_other, s, H.POINT_AFFINE)
                    out["ntt"] = other.best_fft(a, w, log_n)
            except Exception as e:   # surfaced by the assertion below
                out["err"] = e

        t = threading.Thread(target=worker)
        t.start()
        for _ in range(6):   # the main thread keeps the first context busy meanwhile, alternating entry points
            assert np.array_equal(ctx.best_fft(a, w, log_n), want_ntt)
            assert np.array_equal(ctx.msm(b_main, s, H.POINT_AFFINE), want_msm)
        t.join()
        assert "err" not in out, out.get("err")
        assert np.array_equal(out["msm"], want_msm) and np.array_equal(out["ntt"], want_ntt)
        # and strictly interleaved from one thread
        for c, b in ((ctx, b_main), (other, b_other), (ctx, b_main)):
            assert np.array_equal(c.msm(b, s, H.POINT_AFFINE), want_msm)
        b_main.free()
        b_other.free()
    finally:
        other.close()


@pytest.mark.gpu
def test_msm_g2(ctx):
    from tests.test_emu_kernels import _g2_msm_checks

    _g2_msm_checks(ctx, [1, 37, 300, 5000])


@pytest.mark.gpu
def test_full_range_field_inputs_ntt(ctx):
    """rand_fr never leaves [0, 2^252).  The transforms on limb patterns uniform over [0, r) plus the edge patterns
    (r-1, r-2, 2^252 +- 1, R mod r, ...), up to BASELINE configs[2]'s 2^22 and the k = 19 coset extension"""
    from tests import full_range_checks as F

    F.check_ntt(ctx, [4, 10, 11, 14, 19, 20, 22], threads=NT)
    F.check_coset(ctx, [(5, 7), (12, 14), (19, 21)], threads=NT)


@pytest.mark.gpu
def test_full_range_field_inputs_msm(ctx):
    from halo2_lib_amd.h2hip import BASES_PRECOMPUTE
    from tests import full_range_checks as F

    F.check_msm_scalars(ctx, [33, 5000, 1 << 16], threads=NT, flags_list=(0, BASES_PRECOMPUTE))
    F.check_msm_scalars(ctx, [1 << 20], threads=NT, flags_list=(BASES_PRECOMPUTE,))


@pytest.mark.gpu
def test_full_range_field_inputs_pointwise(ctx):
    from tests import full_range_checks as F

    F.check_pointwise(ctx, 100003)
    F.check_inverse_and_products(ctx, [1, 257, 70001, 1 << 19])
    F.check_eval_and_division(ctx, [1, 2049, 1 << 19, (1 << 21) + 5])
