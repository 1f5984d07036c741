"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported by the product
(`halo2-lib_amd/`); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

Pure-Python big-int restatement of the arithmetic behind halo2-lib's proving hot path
(SURVEY.md §8c).  The algorithms live in un-vendored crates — `halo2-axiom 0.5.3`
(git axiom-crypto/halo2.git@5e4f0e52, /root/reference/Cargo.lock:1063-1065) and
`halo2curves-axiom 0.7.3` (crates.io, Cargo.lock:1185-1188) — so this restates their
*published* algorithms and anchors on the reference's own call sites:
  * prove call site           halo2-base/src/utils/testing.rs:40-47
  * Fr/Fq limb layout         halo2-base/src/utils/mod.rs:28-38, 342-377  ([u64;4] LE)
  * G1Affine{x,y}, id=(0,0)   halo2-ecc/benches/msm.rs:62, halo2-ecc/src/ecc/pippenger.rs:217
  * MSM edge cases            halo2-ecc/src/bn254/tests/msm_sum_infinity.rs:16-69
  * Poseidon KATs (pin F_r)   halo2-base/src/poseidon/hasher/tests/state.rs:29-33,55-61
                              halo2-base/src/poseidon/hasher/tests/mod.rs:14-30

PARITY PINNING: the reference holds NO golden vectors for MSM / NTT / commitments
("parity unpinned" for those — SURVEY §8c).  What *is* pinned by reference constants
(Poseidon permutation states + MDS matrix, which exercise F_r add/mul/inv end to end) is
checked in tests/test_oracle.py.  MSM and NTT outputs are mathematically unique (one group
element; one vector for a given omega), so this independent big-int implementation is a valid
bit-exact oracle; it is additionally cross-checked by closed forms (known-dlog bases, O(n^2)
DFT, Horner evaluation).

Memory convention (mirrors halo2curves in-memory types): a field element is
4 x u64 little-endian limbs in MONTGOMERY form (R = 2^256); an affine point is x||y (64 B),
identity = all-zero.
"""
from __future__ import annotations

import math
import numpy as np

# ------------------------------------------------------------------ constants (all re-derived)
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # scalar field r
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # base field q
MONT_R = 1 << 256
TWO_ADICITY = 28
MULT_GEN = 7
ROOT_OF_UNITY = pow(MULT_GEN, (R_MOD - 1) >> TWO_ADICITY, R_MOD)
ZETA = pow(MULT_GEN, 2 * (R_MOD - 1) // 3, R_MOD)  # cube root of unity = coset shift of extended domain
DELTA = pow(MULT_GEN, 1 << TWO_ADICITY, R_MOD)
G1_GEN = (1, 2)
CURVE_B = 3

assert ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
assert ZETA == 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
assert pow(ZETA, 3, R_MOD) == 1 and ZETA != 1


def inv_mod(a: int, p: int) -> int:
    """a^-1 mod p with 0 -> 0 (halo2 `invert().unwrap_or(zero)` convention for batch inversion)."""
    a %= p
    return 0 if a == 0 else pow(a, -1, p)


# ------------------------------------------------------------------ buffer <-> int conversion
def to_mont(v: int, p: int) -> int:
    return (v * MONT_R) % p


def from_mont(v: int, p: int) -> int:
    return (v * inv_mod(MONT_R, p)) % p


_RINV = {R_MOD: inv_mod(MONT_R, R_MOD), Q_MOD: inv_mod(MONT_R, Q_MOD)}


def ints_to_limbs(vals, p: int, mont: bool = True) -> np.ndarray:
    """list[int] (canonical) -> (n,4) uint64 LE limbs, Montgomery form when mont=True."""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        v %= p
        if mont:
            v = (v * MONT_R) % p
        out[i, 0] = v & mask
        out[i, 1] = (v >> 64) & mask
        out[i, 2] = (v >> 128) & mask
        out[i, 3] = (v >> 192) & mask
    return out


def limbs_to_ints(arr: np.ndarray, p: int, mont: bool = True) -> list:
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    rinv = _RINV[p]
    out = []
    for row in arr.tolist():
        v = row[0] | (row[1] << 64) | (row[2] << 128) | (row[3] << 192)
        if mont:
            v = (v * rinv) % p
        out.append(v)
    return out


def points_to_limbs(pts) -> np.ndarray:
    """list of affine points (None = identity) -> (n,8) uint64 (x limbs, y limbs), Montgomery."""
    xs = [0 if P is None else P[0] for P in pts]
    ys = [0 if P is None else P[1] for P in pts]
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    out[:, :4] = ints_to_limbs(xs, Q_MOD)
    out[:, 4:] = ints_to_limbs(ys, Q_MOD)
    return out


def limbs_to_points(arr: np.ndarray) -> list:
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 8)
    xs = limbs_to_ints(arr[:, :4], Q_MOD)
    ys = limbs_to_ints(arr[:, 4:], Q_MOD)
    return [None if (x == 0 and y == 0) else (x, y) for x, y in zip(xs, ys)]


# ------------------------------------------------------------------ BN254 G1 (y^2 = x^3 + 3), affine big-int
def g1_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - CURVE_B) % Q_MOD == 0


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % Q_MOD)


def g1_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % Q_MOD == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, Q_MOD) % Q_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q_MOD) % Q_MOD
    x3 = (lam * lam - x1 - x2) % Q_MOD
    y3 = (lam * (x1 - x3) - y1) % Q_MOD
    return (x3, y3)


# Jacobian (X,Y,Z) big-int ops: used so that python MSMs at 2^12..2^14 finish in seconds.
def _jac_double(P):
    X, Y, Z = P
    if Z == 0:
        return P
    A = X * X % Q_MOD
    B = Y * Y % Q_MOD
    C = B * B % Q_MOD
    D = 2 * ((X + B) * (X + B) - A - C) % Q_MOD
    E = 3 * A % Q_MOD
    F = E * E % Q_MOD
    X3 = (F - 2 * D) % Q_MOD
    Y3 = (E * (D - X3) - 8 * C) % Q_MOD
    Z3 = 2 * Y * Z % Q_MOD
    return (X3, Y3, Z3)


def _jac_add(P, Q):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if Z1 == 0:
        return Q
    if Z2 == 0:
        return P
    Z1Z1 = Z1 * Z1 % Q_MOD
    Z2Z2 = Z2 * Z2 % Q_MOD
    U1 = X1 * Z2Z2 % Q_MOD
    U2 = X2 * Z1Z1 % Q_MOD
    S1 = Y1 * Z2 * Z2Z2 % Q_MOD
    S2 = Y2 * Z1 * Z1Z1 % Q_MOD
    if U1 == U2:
        if S1 == S2:
            return _jac_double(P)
        return (1, 1, 0)
    H = (U2 - U1) % Q_MOD
    I = 4 * H * H % Q_MOD
    J = H * I % Q_MOD
    r = 2 * (S2 - S1) % Q_MOD
    V = U1 * I % Q_MOD
    X3 = (r * r - J - 2 * V) % Q_MOD
    Y3 = (r * (V - X3) - 2 * S1 * J) % Q_MOD
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % Q_MOD
    return (X3, Y3, Z3)


def _to_jac(P):
    return (1, 1, 0) if P is None else (P[0], P[1], 1)


def _from_jac(P):
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, -1, Q_MOD)
    zi2 = zi * zi % Q_MOD
    return (X * zi2 % Q_MOD, Y * zi2 * zi % Q_MOD)


def g1_mul(P, k: int):
    """k*P, k reduced mod r."""
    k %= R_MOD
    acc = (1, 1, 0)
    base = _to_jac(P)
    while k:
        if k & 1:
            acc = _jac_add(acc, base)
        base = _jac_double(base)
        k >>= 1
    return _from_jac(acc)


# ---- a second, structurally different G1 implementation: COMPLETE addition in homogeneous projective coordinates
# (Renes-Costello-Batina 2015, algorithm 7 for y^2 = x^3 + b, a = 0; b3 = 3*b = 9).  One formula covers doubling, the identity
# (0 : 1 : 0) and inverse points, so it shares no case analysis and no coordinate system with g1_add / _jac_add above.  Used by the
# tests as an independent anchor for every closed-form MSM check (the reference holds no MSM vectors).
_B3 = 3 * CURVE_B


def _proj_add_complete(P, Q):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    q = Q_MOD
    t0, t1, t2 = X1 * X2 % q, Y1 * Y2 % q, Z1 * Z2 % q
    t3 = (X1 + Y1) * (X2 + Y2) % q
    t3 = (t3 - t0 - t1) % q
    t4 = (Y1 + Z1) * (Y2 + Z2) % q
    t4 = (t4 - t1 - t2) % q
    Y3 = (X1 + Z1) * (X2 + Z2) % q
    Y3 = (Y3 - t0 - t2) % q
    X3 = 3 * t0 % q
    t2 = _B3 * t2 % q
    Z3 = (t1 + t2) % q
    t1 = (t1 - t2) % q
    Y3 = _B3 * Y3 % q
    Xo = (t3 * t1 - t4 * Y3) % q
    Yo = (Y3 * X3 + t1 * Z3) % q
    Zo = (Z3 * t4 + X3 * t3) % q
    return (Xo, Yo, Zo)


def g1_mul_complete(P, k: int):
    """k*P with the complete formula only (left-to-right double-and-add; no special cases anywhere)"""
    acc = (0, 1, 0)
    base = (0, 1, 0) if P is None else (P[0], P[1], 1)
    for bit in bin(k % R_MOD)[2:] if k % R_MOD else "":
        acc = _proj_add_complete(acc, acc)
        if bit == "1":
            acc = _proj_add_complete(acc, base)
    X, Y, Z = acc
    if Z == 0:
        return None
    zi = inv_mod(Z, Q_MOD)
    return (X * zi % Q_MOD, Y * zi % Q_MOD)


def msm_complete(scalars, points):
    """sum_i scalars[i]*points[i] through the complete projective formulas"""
    acc = (0, 1, 0)
    for k, P in zip(scalars, points):
        term = g1_mul_complete(P, k)
        acc = _proj_add_complete(acc, (0, 1, 0) if term is None else (term[0], term[1], 1))
    X, Y, Z = acc
    if Z == 0:
        return None
    zi = inv_mod(Z, Q_MOD)
    return (X * zi % Q_MOD, Y * zi % Q_MOD)


def msm_naive(scalars, points):
    """sum_i s_i * P_i by independent scalar multiplications (small n only)."""
    acc = (1, 1, 0)
    for s, P in zip(scalars, points):
        acc = _jac_add(acc, _to_jac(g1_mul(P, s)))
    return _from_jac(acc)


def multiexp_serial(scalars, points):
    """
    Restatement of halo2_proofs::arithmetic::multiexp_serial [UPSTREAM, halo2-axiom 0.5.3; reached from
    the reference only through create_proof, halo2-base/src/utils/testing.rs:40-47]:
    window c = 1 (n<4) / 3 (n<32) / ceil(ln n); segments = 256/c + 1; unsigned digits read from the
    canonical little-endian repr; (1<<c)-1 buckets; running-sum ("summation by parts") fold; c doublings
    between segments.
    """
    n = len(points)
    if n < 4:
        c = 1
    elif n < 32:
        c = 3
    else:
        c = int(math.ceil(math.log(n)))
    segments = 256 // c + 1
    scalars = [s % R_MOD for s in scalars]
    acc = (1, 1, 0)
    jpts = [_to_jac(P) for P in points]
    for seg in reversed(range(segments)):
        for _ in range(c):
            acc = _jac_double(acc)
        buckets = [None] * ((1 << c) - 1)
        for s, P in zip(scalars, jpts):
            d = (s >> (seg * c)) & ((1 << c) - 1)
            if d:
                b = buckets[d - 1]
                buckets[d - 1] = P if b is None else _jac_add(b, P)
        running = (1, 1, 0)
        for b in reversed(buckets):
            if b is not None:
                running = _jac_add(running, b)
            acc = _jac_add(acc, running)
    return _from_jac(acc)


def known_dlog_bases(n: int, k0: int, d: int):
    """P_i = (k0 + i*d)*G by repeated affine addition of D = d*G (SURVEY §8c large-n oracle):
    then MSM(s, P) == (sum_i s_i*(k0+i*d) mod r) * G, a single scalar multiplication."""
    P = g1_mul(G1_GEN, k0)
    D = g1_mul(G1_GEN, d)
    out = []
    for _ in range(n):
        out.append(P)
        P = g1_add(P, D)
    return out


# ------------------------------------------------------------------ F_r NTT family
def omega_for(k: int) -> int:
    """EvaluationDomain omega for 2^k rows: ROOT_OF_UNITY^(2^(S-k))."""
    assert 0 <= k <= TWO_ADICITY
    return pow(ROOT_OF_UNITY, 1 << (TWO_ADICITY - k), R_MOD)


def dft_quadratic(a, omega):
    """O(n^2) definition: X[k] = sum_j a[j] * omega^(jk)."""
    n = len(a)
    out = []
    for k in range(n):
        wk = pow(omega, k, R_MOD)
        acc, w = 0, 1
        for j in range(n):
            acc = (acc + a[j] * w) % R_MOD
            w = w * wk % R_MOD
        out.append(acc)
    return out


def _bitrev(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def best_fft(a, omega, log_n):
    """
    Restatement of halo2_proofs::arithmetic::best_fft [UPSTREAM]: bit-reversal permutation, then log_n
    layers of radix-2 decimation-in-time butterflies with twiddles omega^(i * n/(2*half)); natural order in
    and out.  Returns a new list.
    """
    n = 1 << log_n
    assert len(a) == n
    a = list(a)
    for k in range(n):
        rk = _bitrev(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [1] * max(n // 2, 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % R_MOD
    chunk, tchunk = 2, n // 2
    for _ in range(log_n):
        half = chunk // 2
        for start in range(0, n, chunk):
            for i in range(half):
                t = a[start + half + i] * tw[i * tchunk] % R_MOD
                u = a[start + i]
                a[start + i] = (u + t) % R_MOD
                a[start + half + i] = (u - t) % R_MOD
        chunk *= 2
        tchunk //= 2
    return a


def ifft(a, omega, log_n):
    """EvaluationDomain::ifft: best_fft with omega^-1 then multiply by n^-1."""
    out = best_fft(a, inv_mod(omega, R_MOD), log_n)
    ninv = inv_mod(1 << log_n, R_MOD)
    return [x * ninv % R_MOD for x in out]


def coeff_to_extended(a, k, ext_k):
    """EvaluationDomain::coeff_to_extended: a[i] *= zeta^(i mod 3); zero-pad to 2^ext_k; fft (A.2)."""
    assert len(a) == 1 << k
    z = [1, ZETA, ZETA * ZETA % R_MOD]
    b = [x * z[i % 3] % R_MOD for i, x in enumerate(a)] + [0] * ((1 << ext_k) - (1 << k))
    return best_fft(b, omega_for(ext_k), ext_k)


def extended_to_coeff(a, ext_k):
    """EvaluationDomain::extended_to_coeff (before truncation): ifft on the extended domain, then
    a[i] *= [1, zeta^2, zeta][i mod 3] (A.2)."""
    b = ifft(a, omega_for(ext_k), ext_k)
    z = [1, ZETA * ZETA % R_MOD, ZETA]
    return [x * z[i % 3] % R_MOD for i, x in enumerate(b)]


def eval_polynomial(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


def batch_invert(vals):
    """Montgomery's trick with 0 -> 0 (halo2 BatchInvert semantics)."""
    return [inv_mod(v, R_MOD) for v in vals]


def kate_division(coeffs, b):
    """halo2 `kate_division`: quotient of (f(X) - f(b)) / (X - b); returns len(coeffs)-1 coefficients."""
    n = len(coeffs)
    q = [0] * (n - 1)
    tmp = 0
    for i in range(n - 1, 0, -1):
        tmp = (coeffs[i] + tmp * b) % R_MOD
        q[i - 1] = tmp
    return q


def grand_product(num, den):
    """z[0]=1, z[i+1] = z[i]*num[i]/den[i]  (permutation/lookup running product)."""
    z = [1]
    for a, b in zip(num, den):
        z.append(z[-1] * a % R_MOD * inv_mod(b, R_MOD) % R_MOD)
    return z


# ------------------------------------------------------------------ deterministic test data
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & ((1 << 64) - 1)

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
        return z ^ (z >> 31)

    def field(self, p: int) -> int:
        v = 0
        for i in range(8):
            v |= self.next() << (64 * i)
        return v % p


def random_scalars(n: int, seed: int):
    g = SplitMix64(seed)
    return [g.field(R_MOD) for _ in range(n)]


def circuit_like_scalars(n: int, seed: int):
    """advice-column-shaped data (SURVEY §7 hard parts): ~50% zero, ~25% one, rest < 2^88."""
    g = SplitMix64(seed)
    out = []
    for _ in range(n):
        t = g.next() & 3
        if t < 2:
            out.append(0)
        elif t == 2:
            out.append(1)
        else:
            out.append((g.next() | (g.next() << 64)) & ((1 << 88) - 1))
    return out


# ------------------------------------------------------------------ quotient identities, pointwise
def quotient_lookup_terms(acc, z, a, s, ap, sp, l0, l_last, l_blind, step, beta, gamma, y):
    """acc[i] folded with the lookup argument's five identities in upstream's order; all lists are extended-domain
    evaluations as canonical integers."""
    ne, out = len(acc), []
    for i in range(ne):
        active = (1 - (l_last[i] + l_blind[i])) % R_MOD
        v = acc[i]
        v = (v * y + l0[i] * (1 - z[i])) % R_MOD
        v = (v * y + l_last[i] * (z[i] * z[i] - z[i])) % R_MOD
        left = z[(i + step) % ne] * (ap[i] + beta) % R_MOD * (sp[i] + gamma) % R_MOD
        right = z[i] * (a[i] + beta) % R_MOD * (s[i] + gamma) % R_MOD
        v = (v * y + active * (left - right)) % R_MOD
        v = (v * y + l0[i] * (ap[i] - sp[i])) % R_MOD
        v = (v * y + active * (ap[i] - sp[i]) % R_MOD * (ap[i] - ap[(i - step) % ne])) % R_MOD
        out.append(v)
    return out


def quotient_permutation_set_terms(acc, z, z_prev, cols, sigmas, first_col_index, l0, l_last, l_blind, step, terms,
                                   last_rotation, beta, gamma, delta, zeta, ext_omega, y):
    """terms: mask 1 = first-set term, 2 = last-set term, 4 = chaining term, 8 = product term (include/h2hip.h H2HIP_PERM_*)"""
    ne, out = len(acc), []
    for i in range(ne):
        active = (1 - (l_last[i] + l_blind[i])) % R_MOD
        v = acc[i]
        if terms & 1:
            v = (v * y + l0[i] * (1 - z[i])) % R_MOD
        if terms & 2:
            v = (v * y + l_last[i] * (z[i] * z[i] - z[i])) % R_MOD
        if terms & 4:
            v = (v * y + l0[i] * (z[i] - z_prev[(i + last_rotation * step) % ne])) % R_MOD
        if terms & 8:
            x = zeta * pow(ext_omega, i, R_MOD) % R_MOD
            left, right = z[(i + step) % ne], z[i]
            for j, (p, sg) in enumerate(zip(cols, sigmas)):
                left = left * ((p[i] + beta * sg[i] + gamma) % R_MOD) % R_MOD
                right = right * ((p[i] + pow(delta, first_col_index + j, R_MOD) * beta % R_MOD * x + gamma) % R_MOD) % R_MOD
            v = (v * y + active * (left - right)) % R_MOD
        out.append(v)
    return out


def permute_expression_pair(a, s):
    """Restatement of halo2's lookup `permute_expression_pair` over the usable rows [UPSTREAM plonk/lookup/prover.rs,
    SURVEY.md A.5]: sorted input; table element placed wherever a new input value starts; leftover table elements
    (BTreeMap order = ascending) assigned to the repeated rows popped from the END of the list.  Raises if an input
    value is not in the table."""
    from collections import Counter

    ap = sorted(v % R_MOD for v in a)
    left = Counter(v % R_MOD for v in s)
    sp = [0] * len(ap)
    repeated = []
    for row, v in enumerate(ap):
        if row == 0 or v != ap[row - 1]:
            sp[row] = v
            if left[v] == 0:
                raise ValueError("input value missing from the table")
            left[v] -= 1
        else:
            repeated.append(row)
    for v in sorted(left):
        for _ in range(left[v]):
            sp[repeated.pop()] = v
    assert not repeated
    return ap, sp
