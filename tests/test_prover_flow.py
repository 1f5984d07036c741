"""End-to-end composition of the hot-path kernels: a small PLONK/KZG prover for halo2-base's circuit shape — one advice
column with the single gate q*(a + b*c - d) (halo2-base/src/gates/flex_gate/mod.rs:80-91), one RangeChip-style lookup
column (halo2-base/src/gates/range/mod.rs:131-150) and a permutation argument over four columns in two sets (advice,
lookup advice, constants, instance: the equality-enabled columns of halo2-base's BaseConfig) — following create_proof's steps with every
data-parallel step going through the C ABI, and a verifier written with the oracle's big-int arithmetic only:

  * quotient identity   N(x) == h(x) * (x^n - 1)   at a random x, N rebuilt from the openings exactly as a halo2
    verifier does (gate term, the permutation argument's terms in evaluate_h's loop order, the lookup argument's five
    identities, folded by y);
  * KZG openings        C - v*G == (s - x) * W     in G1 (the toxic waste s is known to the test, so no pairing).

Config #0 of BASELINE.json (`cargo bench --bench mul`, k = 9) is the GPU case; the emulated build runs k = 5."""
import numpy as np
import pytest

import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from oracle import bn254 as O
from tests.util import R, fr, jac_to_affine_ints

BLINDING_FACTORS = 5


def _ints(a):
    return O.limbs_to_ints(np.asarray(a).reshape(-1, 4), R)


def _prove_and_verify(ctx, k, seed=1):
    rng = np.random.default_rng(seed)
    rnd = lambda m: O.random_scalars(m, int(rng.integers(1, 1 << 30)))
    n = 1 << k
    u = n - (BLINDING_FACTORS + 1)               # usable rows; row u is the l_last row, rows > u are blinding rows
    s_toxic = 0x1D0C0FFEE1234567890ABCDEF
    params = HP.ParamsKZG.setup(ctx, k, s_toxic, precompute=(k >= 8))
    dom = HP.EvaluationDomain(ctx, 5, k)         # lookup identity has degree 5 (with the complex selector) -> ext_k = k + 2
    ek = dom.extended_k
    omega = HP.fr_int(dom.omega)

    # ---- witness: gates (a, b, c, d = a + b*c) at rows 4j..4j+3, selector on row 4j; blinding rows random
    adv = [0] * n
    q = [0] * n
    vals = rnd(3 * (u // 4))
    for j in range(u // 4):
        a, b, c = vals[3 * j: 3 * j + 3]
        adv[4 * j: 4 * j + 4] = [a, b, c, (a + b * c) % R]
        q[4 * j] = 1
    for j in (1, 2):                              # the first three gates share their `a` input (a 3-cycle of copies)
        adv[4 * j] = adv[0]
        adv[4 * j + 3] = (adv[4 * j] + adv[4 * j + 1] * adv[4 * j + 2]) % R
    adv[u:] = rnd(n - u)
    # ---- lookup column: values of a range table 0..2^t padded with zeros (halo2-base's table layout)
    t = k - 1
    table = [i if i < (1 << t) else 0 for i in range(n)]
    lk = [int(v) for v in rng.integers(0, 1 << t, size=n)]
    lk[: u // 3] = [7 % (1 << t)] * (u // 3)      # heavy repetition, like limbs that are mostly small
    lk[u:] = rnd(n - u)
    l0 = [1] + [0] * (n - 1)
    l_last = [0] * n
    l_last[u] = 1
    l_blind = [0] * u + [0] + [1] * (n - u - 1)

    # ---- copy constraints over the columns (adv, lk, cst, inst): gate outputs exposed in a constants column, lookup cells
    #      tied to an instance column, and the 3-cycle above.  sigma maps (column, row) -> (column, row).
    ncopy = min(8, u // 4)
    cst = [adv[4 * j + 3] if j < ncopy else 0 for j in range(n)]
    inst = [lk[r] if r < ncopy else 0 for r in range(n)]
    perm_cols = [adv, lk, cst, inst]
    sigma = {(c, r): (c, r) for c in range(4) for r in range(n)}

    def tie(cells):   # one cycle through the listed cells
        for a_, b_ in zip(cells, cells[1:] + cells[:1]):
            sigma[a_] = b_

    for j in range(ncopy):
        tie([(0, 4 * j + 3), (2, j)])
        tie([(1, j), (3, j)])
    tie([(0, 0), (0, 4), (0, 8)])
    for (c, r), (c2, r2) in sigma.items():
        assert perm_cols[c][r] == perm_cols[c2][r2] and (r < u) == (r2 < u)
    delta = O.DELTA
    sig_vals = [[pow(delta, sigma[(c, r)][0], R) * pow(omega, sigma[(c, r)][1], R) % R for r in range(n)] for c in range(4)]
    id_vals = [[pow(delta, c, R) * pow(omega, r, R) % R for r in range(n)] for c in range(4)]
    perm_sets = [[0, 1, 2], [3]]                  # chunk length = degree - 2 = 3 columns per grand product

    A, Q_, LK, S = fr(adv), fr(q), fr(lk), fr(table)
    PC, SG = [fr(c) for c in perm_cols], [fr(c) for c in sig_vals]
    # ---- step 1: advice commitments
    commit = lambda vals_: jac_to_affine_ints(params.commit_lagrange(vals_, H.POINT_JACOBIAN))
    C_adv, C_lk = commit(A), commit(LK)
    # ---- step 2: lookup permutation (theta-compression of a single expression is the identity)
    ap, sp = HP.permute_expression_pair(ctx, LK, S, u)
    want_ap, want_sp = O.permute_expression_pair(lk[:u], table[:u])
    assert _ints(ap) == want_ap and _ints(sp) == want_sp
    AP = np.concatenate([ap, fr(rnd(n - u))])
    SP = np.concatenate([sp, fr(rnd(n - u))])
    C_ap, C_sp = commit(AP), commit(SP)
    # ---- step 3: lookup grand product z
    beta, gamma = rnd(2)
    Bv, Gv = np.repeat(fr([beta]), u, 0), np.repeat(fr([gamma]), u, 0)
    num = ctx.fr_mul(ctx.fr_add(LK[:u], Bv), ctx.fr_add(S[:u], Gv))
    den = ctx.fr_mul(ctx.fr_add(AP[:u], Bv), ctx.fr_add(SP[:u], Gv))
    z = ctx.fr_grand_product(num, den)            # u + 1 values, z[0] = 1
    assert _ints(z[:1]) == [1] and _ints(z[u:u + 1]) == [1], "lookup product must close"
    Z = np.concatenate([z, fr(rnd(n - u - 1))])
    C_z = commit(Z)
    # ---- permutation grand products, one per column set, chained through the last usable row
    Bu = np.repeat(fr([beta]), u, 0)
    ZP, C_zp, carry = [], [], fr([1])
    for cset in perm_sets:
        num = den = None
        for c in cset:
            t_id = ctx.fr_add(ctx.fr_add(PC[c][:u], ctx.fr_mul(Bu, fr(id_vals[c][:u]))), Gv)
            t_sg = ctx.fr_add(ctx.fr_add(PC[c][:u], ctx.fr_mul(Bu, SG[c][:u])), Gv)
            num = t_id if num is None else ctx.fr_mul(num, t_id)
            den = t_sg if den is None else ctx.fr_mul(den, t_sg)
        zp = ctx.fr_scale(ctx.fr_grand_product(num, den), carry)     # starts where the previous set ended
        carry = zp[u:u + 1]
        ZP.append(np.concatenate([zp, fr(rnd(n - u - 1))]))
        C_zp.append(commit(ZP[-1]))
    assert _ints(carry) == [1], "permutation product must close"
    # ---- step 5: quotient
    y = rnd(1)[0]
    Y = fr([y])
    cols = {"adv": A, "q": Q_, "lk": LK, "s": S, "ap": AP, "sp": SP, "z": Z, "l0": fr(l0), "l_last": fr(l_last), "l_blind": fr(l_blind),
            "cst": PC[2], "inst": PC[3], "zp0": ZP[0], "zp1": ZP[1], "sg0": SG[0], "sg1": SG[1], "sg2": SG[2], "sg3": SG[3]}
    coeff = {name: dom.lagrange_to_coeff(v) for name, v in cols.items()}
    ext = {name: dom.coeff_to_extended(c) for name, c in coeff.items()}
    acc = np.zeros((1 << ek, 4), dtype=np.uint64)
    acc = ctx.quotient_flex_gate(acc, ext["q"], ext["adv"], ek, k, Y)
    # permutation argument in evaluate_h's order: first-set term, last-set term, chaining terms, then every set's product
    pcol = ["adv", "lk", "cst", "inst"]
    last_rot = -(BLINDING_FACTORS + 1)
    ext_w = fr([HP.fr_int(dom.extended_omega)])

    def perm_terms(acc_, j, terms):
        cset = perm_sets[j]
        return ctx.quotient_permutation_set(acc_, ext["zp%d" % j], ext["zp%d" % (j - 1)] if j else None, [ext[pcol[c]] for c in cset],
                                            [ext["sg%d" % c] for c in cset], cset[0], ext["l0"], ext["l_last"], ext["l_blind"], ek, k, terms,
                                            last_rot, fr([beta]), fr([gamma]), fr([delta]), fr([O.ZETA]), ext_w, Y)

    acc = perm_terms(acc, 0, H.h2hip.PERM_FIRST)
    acc = perm_terms(acc, len(perm_sets) - 1, H.h2hip.PERM_LAST)
    for j in range(1, len(perm_sets)):
        acc = perm_terms(acc, j, H.h2hip.PERM_CHAIN)
    for j in range(len(perm_sets)):
        acc = perm_terms(acc, j, H.h2hip.PERM_PRODUCT)
    acc = ctx.quotient_lookup(acc, ext["z"], ext["lk"], ext["s"], ext["ap"], ext["sp"], ext["l0"], ext["l_last"], ext["l_blind"], ek, k,
                              fr([beta]), fr([gamma]), Y)
    acc = dom.divide_by_vanishing_poly(acc)
    h = dom.extended_to_coeff(acc)                # 4n coefficients
    hi = _ints(h)
    assert all(v == 0 for v in hi[4 * n - 3:]), "deg h <= 4(n-1) - n"
    pieces = [h[i * n:(i + 1) * n] for i in range(4)]
    commit_c = lambda c: jac_to_affine_ints(params.commit(c, H.POINT_JACOBIAN))
    C_h = [commit_c(p) for p in pieces]
    # ---- step 6: evaluations at x and its rotations
    x = rnd(1)[0]
    at = lambda name, rot=0: HP.fr_int(HP.eval_polynomial(ctx, coeff[name], fr([x * pow(omega, rot % n, R) % R])))
    ev = {
        "a0": at("adv"), "a1": at("adv", 1), "a2": at("adv", 2), "a3": at("adv", 3), "q": at("q"),
        "lk": at("lk"), "s": at("s"), "ap": at("ap"), "ap_prev": at("ap", -1), "sp": at("sp"),
        "z": at("z"), "z_next": at("z", 1), "l0": at("l0"), "l_last": at("l_last"), "l_blind": at("l_blind"),
        "cst": at("cst"), "inst": at("inst"), "zp0": at("zp0"), "zp0_next": at("zp0", 1), "zp0_last": at("zp0", last_rot),
        "zp1": at("zp1"), "zp1_next": at("zp1", 1), "sg": [at("sg%d" % c) for c in range(4)],
    }
    h_x = 0
    for i, p in enumerate(pieces):
        h_x = (h_x + pow(x, n * i, R) * HP.fr_int(HP.eval_polynomial(ctx, p, fr([x])))) % R

    # ================= verifier (oracle big-int arithmetic only) =================
    active = (1 - ev["l_last"] - ev["l_blind"]) % R
    pv = [ev["a0"], ev["lk"], ev["cst"], ev["inst"]]

    def perm_product(j):
        left, right = ev["zp%d_next" % j], ev["zp%d" % j]
        for c in perm_sets[j]:
            left = left * (pv[c] + beta * ev["sg"][c] + gamma) % R
            right = right * (pv[c] + pow(delta, c, R) * beta % R * x + gamma) % R
        return active * (left - right)

    terms = [
        ev["q"] * (ev["a0"] + ev["a1"] * ev["a2"] - ev["a3"]),
        ev["l0"] * (1 - ev["zp0"]),
        ev["l_last"] * (ev["zp1"] * ev["zp1"] - ev["zp1"]),
        ev["l0"] * (ev["zp1"] - ev["zp0_last"]),
        perm_product(0),
        perm_product(1),
        ev["l0"] * (1 - ev["z"]),
        ev["l_last"] * (ev["z"] * ev["z"] - ev["z"]),
        active * (ev["z_next"] * (ev["ap"] + beta) * (ev["sp"] + gamma) - ev["z"] * (ev["lk"] + beta) * (ev["s"] + gamma)),
        ev["l0"] * (ev["ap"] - ev["sp"]),
        active * (ev["ap"] - ev["sp"]) * (ev["ap"] - ev["ap_prev"]),
    ]
    N = 0
    for term in terms:
        N = (N * y + term) % R
    assert N == h_x * (pow(x, n, R) - 1) % R, "quotient identity fails at x"

    # KZG openings with the known toxic waste: C - v*G == (s - x') * W
    def check_opening(C, name_or_coeffs, point, value):
        cf = coeff[name_or_coeffs] if isinstance(name_or_coeffs, str) else name_or_coeffs
        W = commit_c(HP.kate_division(ctx, cf, fr([point])))
        lhs = O.g1_add(C, O.g1_neg(O.g1_mul(O.G1_GEN, value)))
        assert lhs == O.g1_mul(W, (s_toxic - point) % R)

    check_opening(C_adv, "adv", x, ev["a0"])
    check_opening(C_adv, "adv", x * pow(omega, 3, R) % R, ev["a3"])
    check_opening(C_lk, "lk", x, ev["lk"])
    check_opening(C_ap, "ap", x * pow(omega, n - 1, R) % R, ev["ap_prev"])
    check_opening(C_sp, "sp", x, ev["sp"])
    check_opening(C_z, "z", x * omega % R, ev["z_next"])
    check_opening(C_zp[0], "zp0", x * pow(omega, last_rot % n, R) % R, ev["zp0_last"])
    check_opening(C_zp[1], "zp1", x * omega % R, ev["zp1_next"])
    # h(X) = sum_i X^(n i) h_i(X): its commitment is the same combination of the piece commitments
    C_hx = None
    for i, C in enumerate(C_h):
        C_hx = O.g1_add(C_hx, O.g1_mul(C, pow(x, n * i, R))) if C is not None else C_hx
    comb = [0] * n
    for i in range(4):
        w = pow(x, n * i, R)
        for j, v in enumerate(hi[i * n:(i + 1) * n]):
            comb[j] = (comb[j] + w * v) % R
    check_opening(C_hx, fr(comb), x, h_x)
    # batched opening at x (the multiopen argument's linear combination by powers of v, built with fr_axpy on the GPU):
    # P(X) = sum_j v^j p_j(X) opens to sum_j v^j p_j(x) under the commitment sum_j v^j C_j
    v = rnd(1)[0]
    batch = [("adv", C_adv, ev["a0"]), ("lk", C_lk, ev["lk"]), ("ap", C_ap, ev["ap"]), ("sp", C_sp, ev["sp"]), ("z", C_z, ev["z"])]
    P = np.zeros((n, 4), dtype=np.uint64)
    C_P, val_P, vj = None, 0, 1
    for name, C, val in batch:
        P = ctx.fr_axpy(P, fr([vj]), coeff[name])
        C_P = O.g1_add(C_P, O.g1_mul(C, vj))
        val_P = (val_P + vj * val) % R
        vj = vj * v % R
    assert HP.fr_int(HP.eval_polynomial(ctx, P, fr([x]))) == val_P
    check_opening(C_P, P, x, val_P)

    # soundness smoke: openings of a tampered advice column (one cell flipped) no longer satisfy the identity
    bad = list(adv)
    bad[3] = (bad[3] + 1) % R
    bc = dom.lagrange_to_coeff(fr(bad))
    b = [HP.fr_int(HP.eval_polynomial(ctx, bc, fr([x * pow(omega, r, R) % R]))) for r in range(4)]
    N_bad = 0
    for term in [ev["q"] * (b[0] + b[1] * b[2] - b[3])] + terms[1:]:
        N_bad = (N_bad * y + term) % R
    assert N_bad != h_x * (pow(x, n, R) - 1) % R
    params.free()


def test_prover_flow_emulated():
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        _prove_and_verify(ctx, 5)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [9, 12])
def test_prover_flow_gpu(k):
    ctx = H.Context()
    try:
        _prove_and_verify(ctx, k, seed=k)
    finally:
        ctx.close()


def test_keygen_warm_up_proof_emulated():
    """plonk_warm_keygen = 1 (the GPU build's default): keygen ends with a throw-away proof of the all-zero witness so that the first real
    proof finds the buffer pool, twiddle tables and lanes in place — the key and the proofs made from it are the same with and without it"""
    import halo2_lib_amd as H
    from halo2_lib_amd import halo2_proofs as HP
    from halo2_lib_amd import plonk as PL
    from halo2_lib_amd import testing as T
    from oracle import c_oracle as CO
    from oracle import plonk as P
    from tests.emu_util import emu_context
    from tests.util import PreDrawnRng

    class _B:
        mul = staticmethod(CO.fr_mul)
        add = staticmethod(CO.fr_add)

    ctx = emu_context()
    try:
        shape = (6, 2, 1, 1, 1, 4)
        sh = P.Shape(*shape)
        kzg = HP.ParamsKZG.setup(ctx, 6, 0xBEEF, precompute=False)
        circ = T.build_circuit(sh, 2, _B)
        proofs, reprs = [], []
        for warm in (0, 1):
            ctx.set_param("plonk_warm_keygen", warm)
            pk = PL.keygen(kzg, PL.BaseCircuitParams.new(*shape), circ.fixed, circ.copies)
            reprs.append(pk.transcript_repr)
            proofs.append(PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(2000, 5)))
            assert PL.verify_proof(pk, circ.instances, proofs[-1])
            pk.free()
        assert proofs[0] == proofs[1] and reprs[0] == reprs[1]
        kzg.free()
    finally:
        ctx.set_param("plonk_warm_keygen", 0)
        ctx.close()
