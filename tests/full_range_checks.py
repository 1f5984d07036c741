"""Kernel-level parity on inputs from the WHOLE of [0, r) (tests/util.py:rand_fr only draws limb patterns below
2^252, so patterns in [2^252, r) reached the kernels only as intermediates inside whole proofs).  tests/util.py:full_range_fr draws
uniformly over [0, r) and plants the edge patterns (r-1, r-2, 2^252 +- 1, 2^253 +- 1, R mod r, R^2 mod r, ...).  Shared by the emulated
build (CPU suite, small sizes) and the GPU suite (BASELINE sizes); everything is compared bit for bit with the C oracle."""
import numpy as np

from oracle import bn254 as O
from oracle import c_oracle as CO
from tests.util import R, domain_consts, edge_fr_values, fr, full_range_fr, _raw_limbs


def check_ntt(ctx, log_ns, threads):
    for log_n in log_ns:
        a = full_range_fr(1 << log_n, 100 + log_n)
        w, winv, div = domain_consts(log_n)
        got = ctx.best_fft(a, w, log_n)
        assert np.array_equal(got, CO.best_fft(a, log_n, w, threads=threads)), log_n
        assert np.array_equal(ctx.ifft(got, winv, log_n, div), a), log_n
    # a column that is nothing but edge patterns, repeated
    ev = _raw_limbs(edge_fr_values())
    log_n = 10
    a = np.ascontiguousarray(np.tile(ev, ((1 << log_n) // len(ev) + 1, 1))[: 1 << log_n])
    w, winv, div = domain_consts(log_n)
    got = ctx.best_fft(a, w, log_n)
    assert np.array_equal(got, CO.best_fft(a, log_n, w, threads=threads))
    assert np.array_equal(ctx.ifft(got, winv, log_n, div), a)


def check_coset(ctx, shapes, threads):
    for k, ek in shapes:
        a = full_range_fr(1 << k, 200 + k)
        we, weinv, ediv = domain_consts(ek)
        z, zinv = fr([O.ZETA]), fr([O.ZETA * O.ZETA % R])
        ext = ctx.coeff_to_extended(a, k, ek, we, z)
        assert np.array_equal(ext, CO.coeff_to_extended(a, k, ek, we, z, threads=threads)), (k, ek)
        back = ctx.extended_to_coeff(ext, ek, weinv, ediv, zinv)
        assert np.array_equal(back[: 1 << k], a) and not back[1 << k:].any(), (k, ek)


def check_msm_scalars(ctx, ns, threads, flags_list=(0,)):
    """MSM scalars over the whole range: the top window's digit reaches its maximum only for scalars close to r"""
    import halo2_lib_amd as H

    for n in ns:
        bases = CO.known_dlog_bases(n, fr([31 + n]), fr([7]))
        s = full_range_fr(n, 300 + n)
        want = CO.best_multiexp(s, bases, threads=threads)
        for flags in flags_list:
            b = ctx.bases_upload(bases, flags)
            assert np.array_equal(ctx.msm(b, s, H.POINT_AFFINE), want), (n, flags)
            b.free()
    # every scalar an edge pattern, over a few bases (naive big-int reference)
    vals = edge_fr_values()
    # raw limb pattern v stands for the scalar v * 2^-256 mod r
    rinv = pow(1 << 256, -1, R)
    pts = [O.g1_mul(O.G1_GEN, 5 + 3 * i) for i in range(len(vals))]
    b = ctx.bases_upload(O.points_to_limbs(pts))
    got = ctx.msm(b, _raw_limbs(vals), H.POINT_AFFINE)
    b.free()
    assert O.limbs_to_points(got) == [O.msm_naive([v * rinv % R for v in vals], pts)]


def check_pointwise(ctx, n, threads=4):
    a, b, c = full_range_fr(n, 1), full_range_fr(n, 2), full_range_fr(n, 3)
    # line the edge patterns up against each other: (r-1)*(r-1), (r-1)+(r-1), 0-(r-1), ...
    ev = _raw_limbs(edge_fr_values())
    m = len(ev)
    if n >= m * m:
        a[: m * m] = np.repeat(ev, m, axis=0)
        b[: m * m] = np.tile(ev, (m, 1))
    assert np.array_equal(ctx.fr_mul(a, b), CO.fr_mul(a, b))
    assert np.array_equal(ctx.fr_add(a, b), CO.fr_add(a, b))
    assert np.array_equal(ctx.fr_sub(a, b), CO.fr_sub(a, b))
    assert np.array_equal(ctx.fr_mul_add(a, b, c), CO.fr_add(CO.fr_mul(a, b), c))
    for sc in (full_range_fr(1, 9, edges=False), _raw_limbs([R - 1]), _raw_limbs([(1 << 253) + 1])):
        rep = np.repeat(sc, len(a), 0)
        assert np.array_equal(ctx.fr_axpy(a, sc, b), CO.fr_add(a, CO.fr_mul(rep, b)))
        assert np.array_equal(ctx.fr_scale(a, sc), CO.fr_mul(a, rep))


def check_inverse_and_products(ctx, ns):
    for n in ns:
        a = full_range_fr(n, 400 + n)
        if n > 4:
            a[3] = 0     # zero denominators stay zero (BatchInvert's convention)
        assert np.array_equal(ctx.fr_batch_invert(a), CO.fr_batch_invert(a)), n
        num, den = full_range_fr(n, 401 + n), full_range_fr(n, 402 + n)
        den[~den.any(axis=1)] = _raw_limbs([R - 1])[0]   # the grand product divides: no zero denominators
        assert np.array_equal(ctx.fr_grand_product(num, den), CO.fr_grand_product(num, den)), n
        assert np.array_equal(ctx.assigned_resolve(num, a), CO.fr_mul(num, CO.fr_batch_invert(a))), n
    ev = _raw_limbs(edge_fr_values())
    assert np.array_equal(ctx.fr_batch_invert(ev), CO.fr_batch_invert(ev))


def check_eval_and_division(ctx, ns):
    for n in ns:
        c = full_range_fr(n, 500 + n)
        for x in (full_range_fr(1, 99, edges=False), _raw_limbs([R - 1]), _raw_limbs([(1 << 252) + 1])):
            assert np.array_equal(ctx.fr_eval_polynomial(c, x), CO.fr_eval_polynomial(c, x)), n
            if n >= 2:
                assert np.array_equal(ctx.fr_kate_division(c, x), CO.fr_kate_division(c, x)), n
        if n >= 8:
            pts, ws = full_range_fr(3, 600 + n), full_range_fr(3, 601 + n)
            want = ctx.fr_linear_combination([CO.fr_kate_division(c, pts[j:j + 1]) for j in range(3)], ws)
            assert np.array_equal(ctx.fr_kate_division_multi(c, pts, ws), want), n
