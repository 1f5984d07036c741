"""Checks of a backend (emulated kernels or the GPU library) and of the C oracle against the committed fixtures in
tests/golden/ (see tests/golden/make_golden.py for their provenance)."""
import json
import os

import numpy as np

import halo2_lib_amd as H
from oracle import bn254 as O
from tests.util import R, fr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def ints(hexes):
    return [int(h, 16) for h in hexes]


def check_backend_against_golden(ctx):
    g = load("hotpath_small_cases.json")
    # MSM
    m = g["msm"]
    pts = [(int(x, 16), int(y, 16)) for x, y in m["bases_xy"]]
    want = (int(m["result_xy"][0], 16), int(m["result_xy"][1], 16))
    for flags in (0, 1):
        b = ctx.bases_upload(O.points_to_limbs(pts), flags)
        got = ctx.msm(b, fr(ints(m["scalars"])), H.POINT_AFFINE)
        assert O.limbs_to_points(got)[0] == want
        b.free()
    # NTT family
    t = g["ntt"]
    k, ek, a, w = t["k"], t["extended_k"], fr(ints(t["input"])), int(t["omega"], 16)
    assert O.limbs_to_ints(ctx.best_fft(a, fr([w]), k), R) == ints(t["best_fft"])
    assert O.limbs_to_ints(ctx.ifft(a, fr([O.inv_mod(w, R)]), k, fr([O.inv_mod(1 << k, R)])), R) == ints(t["ifft"])
    ext_w = O.omega_for(ek)
    ext = ctx.coeff_to_extended(a, k, ek, fr([ext_w]), fr([O.ZETA]))
    assert O.limbs_to_ints(ext, R) == ints(t["coeff_to_extended"])
    back = ctx.extended_to_coeff(ext, ek, fr([O.inv_mod(ext_w, R)]), fr([O.inv_mod(1 << ek, R)]), fr([O.ZETA * O.ZETA % R]))
    assert O.limbs_to_ints(back[: 1 << k], R) == ints(t["input"]) and not back[1 << k:].any()
    # pointwise
    p = g["poly"]
    c, x = fr(ints(p["coeffs"])), fr([int(p["x"], 16)])
    assert O.limbs_to_ints(ctx.fr_eval_polynomial(c, x), R) == [int(p["eval"], 16)]
    assert O.limbs_to_ints(ctx.fr_kate_division(c, x), R) == ints(p["kate_division"])
    cv = ints(p["coeffs"])
    assert O.limbs_to_ints(ctx.fr_batch_invert(fr(cv[:8] + [0])), R) == ints(p["batch_invert"])
    assert O.limbs_to_ints(ctx.fr_grand_product(fr(cv[:16]), fr(cv[16:32])), R) == ints(p["grand_product"])
    # lookup permutation
    lk = g["lookup_permute"]
    ap, sp = ctx.lookup_permute(fr(lk["input"]), fr(lk["table"]), len(lk["input"]))
    assert O.limbs_to_ints(ap, R) == lk["permuted_input"] and O.limbs_to_ints(sp, R) == lk["permuted_table"]
    # Poseidon: the reference's own KATs
    kats = load("poseidon_reference_kats.json")
    from oracle.poseidon import Spec

    for key in ("t3", "t5"):
        kat = kats[key]
        spec = Spec(kat["t"], kat["r_f"], kat["r_p"])
        if "mds" in kat:
            assert spec.mds == [[int(v) for v in row] for row in kat["mds"]]
        ctx.poseidon_set_spec(kat["t"], kat["r_f"], kat["r_p"], fr([v for row in spec.constants for v in row]), fr([v for row in spec.mds for v in row]))
        out = ctx.poseidon_permute(fr(kat["state_in"]).reshape(1, kat["t"], 4), fr(kat["inputs"]).reshape(1, -1, 4))
        assert O.limbs_to_ints(out.reshape(-1, 4), R) == [int(v) for v in kat["state_out"]]


def check_c_oracle_against_golden():
    """the C restatement (oracle/h2_oracle.c) against the committed pure-Python outputs"""
    from oracle import c_oracle as CO

    g = load("hotpath_small_cases.json")
    m = g["msm"]
    pts = [(int(x, 16), int(y, 16)) for x, y in m["bases_xy"]]
    got = CO.best_multiexp(fr(ints(m["scalars"])), O.points_to_limbs(pts), threads=3)
    assert O.limbs_to_points(got)[0] == (int(m["result_xy"][0], 16), int(m["result_xy"][1], 16))
    t = g["ntt"]
    k, a, w = t["k"], fr(ints(t["input"])), int(t["omega"], 16)
    assert O.limbs_to_ints(CO.best_fft(a, k, fr([w]), threads=2), R) == ints(t["best_fft"])
    assert O.limbs_to_ints(CO.ifft(a, k, fr([w]), threads=2), R) == ints(t["ifft"])
    ek = t["extended_k"]
    assert O.limbs_to_ints(CO.coeff_to_extended(a, k, ek, fr([O.omega_for(ek)]), fr([O.ZETA]), threads=2), R) == ints(t["coeff_to_extended"])
    p = g["poly"]
    c, x = fr(ints(p["coeffs"])), fr([int(p["x"], 16)])
    assert O.limbs_to_ints(CO.fr_eval_polynomial(c, x), R) == [int(p["eval"], 16)]
    assert O.limbs_to_ints(CO.fr_kate_division(c, x), R) == ints(p["kate_division"])
    cv = ints(p["coeffs"])
    assert O.limbs_to_ints(CO.fr_grand_product(fr(cv[:16]), fr(cv[16:32])), R) == ints(p["grand_product"])


def check_prover_steps(ctx, n, seed=5):
    """the device steps of create_proof that sit between the big kernels, against big-int arithmetic: Assigned::Rational resolution
    (zero denominators included, reference halo2-base/src/gates/flex_gate/mod.rs:677-681), the factors of the permutation and lookup grand
    products, the Horner step over h's pieces, P(X) - r(X), the one-pass linear combinations and the batched evaluation round"""
    from tests.util import rand_fr

    g = np.random.default_rng(seed)
    to_i = lambda a: O.limbs_to_ints(np.asarray(a).reshape(-1, 4), R)
    # Assigned::Rational -> value, 0^-1 := 0; Trivial cells carry den = 1
    num, den = rand_fr(n, seed), rand_fr(n, seed + 1)
    den[:: 7] = 0
    den[1:: 5] = fr([1])[0]
    got = to_i(ctx.assigned_resolve(num, den))
    ni, di = to_i(num), to_i(den)
    assert got == [a * (O.inv_mod(d, R) if d else 0) % R for a, d in zip(ni, di)]
    # permutation set factors
    beta, gamma = [int(v) for v in g.integers(1, 1 << 62, size=2)]
    omega = O.omega_for(max(1, (n - 1).bit_length()))
    cols, sigs = [rand_fr(n, seed + 10 + j) for j in range(3)], [rand_fr(n, seed + 20 + j) for j in range(3)]
    first = 2
    gn, gd = ctx.permutation_product_terms(cols, sigs, first, fr([beta]), fr([gamma]), fr([O.DELTA]), fr([omega]))
    ci, si = [to_i(c) for c in cols], [to_i(c) for c in sigs]
    wn, wd, wpow = [], [], 1
    for i in range(n):
        a = b = 1
        for j in range(3):
            a = a * (ci[j][i] + beta * pow(O.DELTA, first + j, R) % R * wpow + gamma) % R
            b = b * (ci[j][i] + beta * si[j][i] + gamma) % R
        wn.append(a)
        wd.append(b)
        wpow = wpow * omega % R
    assert to_i(gn) == wn and to_i(gd) == wd
    # all sets at once (chunks of 3 and 2 columns, a ragged last set; 70 columns cross the 64-column launch boundary)
    many_c = [rand_fr(n, seed + 300 + j) for j in range(70 if n <= 64 else 7)]
    many_s = [rand_fr(n, seed + 400 + j) for j in range(len(many_c))]
    for chunk in (3, 2):
        nums, dens = ctx.permutation_product_terms_sets(many_c, many_s, chunk, fr([beta]), fr([gamma]), fr([O.DELTA]), fr([omega]))
        for si in range(len(nums)):
            c0, c1 = si * chunk, min((si + 1) * chunk, len(many_c))
            wn1, wd1 = ctx.permutation_product_terms(many_c[c0:c1], many_s[c0:c1], c0, fr([beta]), fr([gamma]), fr([O.DELTA]), fr([omega]))
            assert np.array_equal(nums[si], wn1) and np.array_equal(dens[si], wd1), (chunk, si)
        # a row range of the same columns (the sharded prover's form) = those rows of the whole-column factors
        if n >= 5:
            r0, rn = n // 3, n - n // 3 - 1
            pn, pd = ctx.permutation_product_terms_sets(many_c, many_s, chunk, fr([beta]), fr([gamma]), fr([O.DELTA]), fr([omega]), row0=r0, rows=rn)
            for si in range(len(nums)):
                assert np.array_equal(pn[si], nums[si][r0:r0 + rn]) and np.array_equal(pd[si], dens[si][r0:r0 + rn]), (chunk, si)
    # lookup factors
    a_, s_, ap, sp = (rand_fr(n, seed + 30 + j) for j in range(4))
    gn, gd = ctx.lookup_product_terms(a_, s_, ap, sp, fr([beta]), fr([gamma]))
    ai, si_, api, spi = to_i(a_), to_i(s_), to_i(ap), to_i(sp)
    assert to_i(gn) == [(x + beta) * (y + gamma) % R for x, y in zip(ai, si_)]
    assert to_i(gd) == [(x + beta) * (y + gamma) % R for x, y in zip(api, spi)]
    # s*y + a*x and y - low
    sc, ac = [int(v) for v in g.integers(1, 1 << 62, size=2)]
    assert to_i(ctx.fr_axpby(a_, fr([sc]), fr([ac]), s_)) == [(sc * y + ac * x) % R for y, x in zip(ai, si_)]
    m = min(n, 5)
    low = rand_fr(m, seed + 40)
    want = list(ai)
    for i, v in enumerate(to_i(low)):
        want[i] = (want[i] - v) % R
    assert to_i(ctx.fr_sub_low(a_, low)) == want
    # several grand products at once (zero denominators count as 0): independent (lookups) and chained (permutation sets), 1 / 3 / 34 segments
    for segs in (1, 3, 34):
        seg = max(1, n // segs)
        nums = [rand_fr(seg, seed + 60 + j) for j in range(segs)]
        dens = [rand_fr(seg, seed + 160 + j) for j in range(segs)]
        if seg > 3:
            dens[0][2] = 0
        single = [to_i(ctx.fr_grand_product(a, b)) for a, b in zip(nums, dens)]
        assert single[0][0] == 1 and (seg <= 3 or single[0][-1] == 0)
        got = [to_i(z) for z in ctx.fr_grand_products(nums, dens, chained=False)]
        assert got == single, segs
        got = [to_i(z) for z in ctx.fr_grand_products(nums, dens, chained=True)]
        carry = 1
        for j in range(segs):
            assert got[j] == [v * carry % R for v in single[j]], (segs, j)
            carry = got[j][-1]
    # sum_j c_j P_j in one pass: 1, 5, 6, 15, 16 and 33 terms (five products per reduction, 15 terms per launch), extreme coefficients and values
    pool = [a_, s_, ap, sp] + [rand_fr(n, seed + 50 + j) for j in range(3)]
    pool[4][: max(1, n // 2)] = fr([R - 1])[0]
    for count in (1, 5, 6, 15, 16, 33):
        cs = [int(v) for v in g.integers(0, 1 << 62, size=count)]
        if count >= 5:
            cs[0], cs[1], cs[4] = R - 1, 0, 1
        ps = [pool[j % len(pool)] for j in range(count)]
        pi = [to_i(p_) for p_ in ps]
        want = [sum(c * p_[i] for c, p_ in zip(cs, pi)) % R for i in range(n)]
        assert to_i(ctx.fr_linear_combination(ps, fr(cs))) == want, count
    # batched evaluations: different polynomials / lengths / points, repeated points (the kernel shares their power tables)
    polys = [a_, s_, ap[: max(1, n // 3)], sp[:1], a_]
    pts = [int(v) for v in g.integers(1, 1 << 62, size=3)]
    points = [pts[0], pts[0], pts[1], pts[2], pts[2]]
    got = to_i(ctx.fr_eval_polynomial_batch(polys, fr(points)))

    def horner(c, x):
        acc = 0
        for v in reversed(c):
            acc = (acc * x + v) % R
        return acc

    assert got == [horner(to_i(p), x) for p, x in zip(polys, points)]
    # multi-point division: (f - r) / prod (X - b_j) via the weighted sum of kate divisions, against sequential exact divisions
    if n >= 6:
        f = to_i(a_)
        for m in (1, 2, 3, 4, 5, 8):   # 3: padded to four points; 5, 8: two passes of four with accumulation
            bs = [int(v) for v in g.integers(2, 1 << 62, size=m)]
            ws = []
            for j in range(m):
                d = 1
                for i in range(m):
                    if i != j:
                        d = d * (bs[j] - bs[i]) % R
                ws.append(O.inv_mod(d, R))
            want = None
            for j in range(m):                       # sum_j w_j * kate(f, b_j), big-int
                qj = O.kate_division(f, bs[j])
                want = [w_ * ws[j] % R for w_ in qj] if want is None else [(a0 + w_ * ws[j]) % R for a0, w_ in zip(want, qj)]
            got = to_i(ctx.fr_kate_division_multi(a_, fr(bs), fr(ws)))
            assert got == want
            if m == 3:                               # several polynomials in one call (SHPLONK's sum over its rotation sets): = the sum of the single calls
                polys3 = [a_, s_, ap]
                psets = [fr(bs), fr(bs[:1]), fr(bs[:2])]
                w2 = [O.inv_mod((bs[0] - bs[1]) % R, R), O.inv_mod((bs[1] - bs[0]) % R, R)]
                wsets = [fr(ws), fr([5]), fr(w2)]
                want3 = [0] * (n - 1)
                for pl, ps_, ws_ in zip(polys3, psets, wsets):
                    want3 = [(x + y) % R for x, y in zip(want3, to_i(ctx.fr_kate_division_multi(pl, ps_, ws_)))]
                assert to_i(ctx.fr_kate_division_sets(polys3, psets, wsets)) == want3
            acc0 = to_i(s_)                              # the accumulating form: acc[0..n-1) + the same sum, the last element untouched
            got_acc = to_i(ctx.fr_kate_division_multi_acc(s_, a_, fr(bs), fr(ws)))
            assert got_acc == [(x + y) % R for x, y in zip(acc0, want)] + acc0[len(want):]
            if m > 1:                                # ... and it IS the quotient by the product of the roots: top m-1 coefficients vanish
                assert got[len(got) - (m - 1):] == [0] * (m - 1)
            # the same quotient by COEFFICIENT RANGES (the multi-GPU prover's form): each range divided on its own with the carries of the
            # ranges above it (big-int Horner), concatenated = the whole-polynomial quotient followed by a zero
            cuts = sorted({0, 1, n // 3, n - 2, n})
            pieces = []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                carries = [horner(f[hi:], b) for b in bs]
                pieces += to_i(ctx.fr_kate_division_range(a_[lo:hi], fr(bs), fr(ws), fr(carries)))
            assert pieces == want + [0], (m, cuts)


def check_ntt_batches(ctx, ks=(3, 11), ncols=35):
    """h2hip_ifft_batch_dev / h2hip_coeff_to_extended_batch_dev (32 columns per launch: 35 columns cross a group boundary, k = 11 -> 13 runs
    the multi-pass path with per-column scratch) against the one-column entries, which the other tests pin to the oracle"""
    from tests.util import rand_fr

    for k in ks:
        ek, n = k + 2, 1 << k
        omega, ext_omega = O.omega_for(k), O.omega_for(ek)
        om_inv, div = fr([O.inv_mod(omega, R)]), fr([O.inv_mod(n, R)])
        cols = [rand_fr(n, 900 + 37 * k + j) for j in range(ncols)]
        cols[1][:] = 0
        want_coeff = [ctx.ifft(c, om_inv, k, div) for c in cols]
        want_ext = [ctx.coeff_to_extended(c, k, ek, fr([ext_omega]), fr([O.ZETA])) for c in want_coeff]
        d = [ctx.to_device(c) for c in cols]
        e = [ctx.malloc(32 << ek) for _ in cols]
        try:
            ctx.ifft_batch_dev(d, om_inv, k, div)
            for j in range(ncols):
                assert np.array_equal(ctx.download(d[j], (n, 4)), want_coeff[j]), (k, j)
            ctx.coeff_to_extended_batch_dev(d, k, e, ek, fr([ext_omega]), fr([O.ZETA]))
            for j in range(ncols):
                assert np.array_equal(ctx.download(e[j], (1 << ek, 4)), want_ext[j]), (k, j)
                assert np.array_equal(ctx.download(d[j], (n, 4)), want_coeff[j]), (k, j, "input preserved")
            ctx.ifft_batch_dev([], om_inv, k, div)   # empty batch: nothing to do
        finally:
            for p_ in d + e:
                ctx.free(p_)


def check_quotient_batches(ctx, k=3, gate_cols=67, lookups=34, perm_cols=40, chunk_len=3):
    """the batched quotient entries (64 gate columns / 32 lookups / 12 (set, term) jobs per launch: the counts cross every group boundary,
    the last permutation set is ragged) against the one-identity-per-call entries the other tests pin to the big-int formulae, folded in
    evaluate_h's order"""
    from tests.util import rand_fr

    ek = k + 2
    ne = 1 << ek
    y, beta, gamma = fr([0x1234567]), fr([0x89ABCDEF]), fr([0x13579B])
    col = lambda j: rand_fr(ne, 7000 + j)
    acc0 = col(0)
    l0, l_last, l_blind = col(1), col(2), col(3)
    # gate
    qs, advs = [col(10 + j) for j in range(gate_cols)], [col(200 + j) for j in range(gate_cols)]
    want = acc0
    for q, a in zip(qs, advs):
        want = ctx.quotient_flex_gate(want, q, a, ek, k, y)
    assert np.array_equal(ctx.quotient_flex_gate_batch(acc0, qs, advs, ek, k, y), want)
    assert np.array_equal(ctx.quotient_flex_gate_batch(acc0, [], [], ek, k, y), acc0)
    # lookups
    cols5 = [[col(400 + 40 * t + j) for j in range(lookups)] for t in range(5)]
    want = acc0
    for j in range(lookups):
        want = ctx.quotient_lookup(want, *[cols5[t][j] for t in range(5)], l0, l_last, l_blind, ek, k, beta, gamma, y)
    assert np.array_equal(ctx.quotient_lookups(acc0, *cols5, l0, l_last, l_blind, ek, k, beta, gamma, y), want)
    # permutation argument
    from halo2_lib_amd.h2hip import PERM_CHAIN, PERM_FIRST, PERM_LAST, PERM_PRODUCT

    ext_omega = fr([O.omega_for(ek)])
    delta, zeta = fr([O.DELTA]), fr([O.ZETA])
    for ncols in (perm_cols, 2):
        sets = (ncols + chunk_len - 1) // chunk_len
        zs = [col(800 + j) for j in range(sets)]
        pc, ps = [col(900 + j) for j in range(ncols)], [col(1000 + j) for j in range(ncols)]
        last_rot = -3

        def one(acc, si, terms):
            c0, c1 = si * chunk_len, min((si + 1) * chunk_len, ncols)
            return ctx.quotient_permutation_set(acc, zs[si], zs[si - 1] if si else None, pc[c0:c1], ps[c0:c1], c0, l0, l_last, l_blind, ek, k, terms,
                                                last_rot, beta, gamma, delta, zeta, ext_omega, y)

        want = acc0
        if sets == 1:
            want = one(want, 0, PERM_FIRST | PERM_LAST | PERM_PRODUCT)
        else:
            want = one(want, 0, PERM_FIRST)
            want = one(want, sets - 1, PERM_LAST)
            for si in range(1, sets):
                want = one(want, si, PERM_CHAIN)
            for si in range(sets):
                want = one(want, si, PERM_PRODUCT)
        got = ctx.quotient_permutation_sets(acc0, zs, pc, ps, chunk_len, l0, l_last, l_blind, ek, k, last_rot, beta, gamma, delta, zeta, ext_omega, y)
        assert np.array_equal(got, want), ncols


def check_lookup_permute_batch(ctx, u=700, bits=6, count=5):
    """several input columns against one range table in one call (one host synchronisation): the same columns as one call per input;
    an input value missing from the table in ANY column fails the whole batch"""
    import halo2_lib_amd as H

    g = np.random.default_rng(u + count)
    table = np.concatenate([fr(list(range(1 << bits)) + [0] * (u - (1 << bits))), fr([7] * 5)])   # padded like a halo2-base table column
    inputs = []
    for j in range(count):
        vals = [int(v) for v in g.integers(0, 1 << bits, size=u)]
        if j == 1:
            vals = [0] * u
        inputs.append(np.concatenate([fr(vals), fr([9] * 5)]))
    got = ctx.lookup_permute_batch(inputs, table, u)
    for j in range(count):
        a1, s1 = ctx.lookup_permute(inputs[j], table, u, presort_table=True)
        assert np.array_equal(got[j][0], a1) and np.array_equal(got[j][1], s1), j
    assert ctx.lookup_permute_batch([], table, u) == []
    bad = [c.copy() for c in inputs]
    bad[count - 1][3] = fr([(1 << bits) + 5])[0]
    try:
        ctx.lookup_permute_batch(bad, table, u)
    except H.H2HipError:
        pass
    else:
        raise AssertionError("a value outside the table went unnoticed")
