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
    products, the Horner step over h's pieces, P(X) - r(X), and the batched evaluation round"""
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
        for m in (1, 2, 4):
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
            if m > 1:                                # ... and it IS the quotient by the product of the roots: top m-1 coefficients vanish
                assert got[len(got) - (m - 1):] == [0] * (m - 1)
