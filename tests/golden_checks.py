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
