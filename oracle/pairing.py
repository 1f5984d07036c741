"""
ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  BN254 G2 and a pairing, pure-Python big-int.

The KZG verifier of the reference's `check_proof` (halo2-base/src/utils/testing.rs:64-88:
`verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<_>, _, _, SingleStrategy<_>>`) ends in one pairing
check  e(L, s*g2) == e(R, g2)  [UPSTREAM halo2-axiom poly/kzg/msm.rs DualMSM::check].  The pairing library
itself (halo2curves-axiom 0.7.3 bn256, optimal ate) is un-vendored; this file restates the *published*
construction and uses the plain **ate pairing** a(Q, P) = f_{t-1,Q}(P)^((q^12-1)/r) — a different Miller loop
than upstream's optimal ate, but any non-degenerate bilinear map on G1 x G2 decides the same relation
(e(L, sQ) == e(R, Q)  <=>  s*L == R), which is all the verifier needs.  Checked by bilinearity tests
(tests/test_plonk_oracle.py).

Tower: Fq2 = Fq[u]/(u^2+1);  Fq6 = Fq2[v]/(v^3 - xi), xi = 9+u;  Fq12 = Fq6[w]/(w^2 - v).
G2: the sextic D-twist  y^2 = x^3 + 3/xi  over Fq2; untwist (x, y) -> (x*w^2, y*w^3).
"""
from __future__ import annotations

from .bn254 import Q_MOD as P, R_MOD

BN_X = 4965661367192848881                      # curve parameter: q = 36x^4+36x^3+24x^2+6x+1, t = 6x^2+1
ATE_LOOP = 6 * BN_X * BN_X                      # t - 1
assert 36 * BN_X**4 + 36 * BN_X**3 + 24 * BN_X**2 + 6 * BN_X + 1 == P
assert 36 * BN_X**4 + 36 * BN_X**3 + 18 * BN_X**2 + 6 * BN_X + 1 == R_MOD
assert (ATE_LOOP - P) % R_MOD == 0              # t - 1 = q (mod r): the Frobenius eigenvalue on G2

# ------------------------------------------------------------------ Fq2 (tuples (c0, c1) = c0 + c1*u)
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b):
    t0, t1 = a[0] * b[0], a[1] * b[1]
    return ((t0 - t1) % P, ((a[0] + a[1]) * (b[0] + b[1]) - t0 - t1) % P)
def f2_sqr(a): return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)
def f2_scal(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)
def f2_mul_xi(a): return ((9 * a[0] - a[1]) % P, (a[0] + 9 * a[1]) % P)   # * (9 + u)
F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)
TWIST_B = f2_mul((3, 0), f2_inv(XI))

# ------------------------------------------------------------------ Fq6 (triples of Fq2: a0 + a1 v + a2 v^2)
def f6_add(a, b): return (f2_add(a[0], b[0]), f2_add(a[1], b[1]), f2_add(a[2], b[2]))
def f6_sub(a, b): return (f2_sub(a[0], b[0]), f2_sub(a[1], b[1]), f2_sub(a[2], b[2]))
def f6_neg(a): return (f2_neg(a[0]), f2_neg(a[1]), f2_neg(a[2]))
def f6_mul(a, b):
    t0, t1, t2 = f2_mul(a[0], b[0]), f2_mul(a[1], b[1]), f2_mul(a[2], b[2])
    c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a[1], a[2]), f2_add(b[1], b[2])), t1), t2)))
    c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a[0], a[1]), f2_add(b[0], b[1])), t0), t1), f2_mul_xi(t2))
    c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a[0], a[2]), f2_add(b[0], b[2])), t0), t2), t1)
    return (c0, c1, c2)
def f6_mul_v(a): return (f2_mul_xi(a[2]), a[0], a[1])      # * v
def f6_inv(a):
    c0 = f2_sub(f2_sqr(a[0]), f2_mul_xi(f2_mul(a[1], a[2])))
    c1 = f2_sub(f2_mul_xi(f2_sqr(a[2])), f2_mul(a[0], a[1]))
    c2 = f2_sub(f2_sqr(a[1]), f2_mul(a[0], a[2]))
    t = f2_inv(f2_add(f2_mul(a[0], c0), f2_mul_xi(f2_add(f2_mul(a[2], c1), f2_mul(a[1], c2)))))
    return (f2_mul(c0, t), f2_mul(c1, t), f2_mul(c2, t))
F6_ZERO, F6_ONE = (F2_ZERO, F2_ZERO, F2_ZERO), (F2_ONE, F2_ZERO, F2_ZERO)

# ------------------------------------------------------------------ Fq12 (pairs of Fq6: c0 + c1 w)
def f12_mul(a, b):
    t0, t1 = f6_mul(a[0], b[0]), f6_mul(a[1], b[1])
    return (f6_add(t0, f6_mul_v(t1)), f6_sub(f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), t0), t1))
def f12_sqr(a): return f12_mul(a, a)
def f12_inv(a):
    t = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mul_v(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], t), f6_neg(f6_mul(a[1], t)))
def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r
F12_ONE = (F6_ONE, F6_ZERO)

# ------------------------------------------------------------------ G2 (affine over Fq2, None = identity)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def g2_is_on_curve(Q):
    if Q is None:
        return True
    x, y = Q
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), TWIST_B)) == F2_ZERO


def g2_neg(Q): return None if Q is None else (Q[0], f2_neg(Q[1]))


def g2_add(A, B):
    if A is None: return B
    if B is None: return A
    if A[0] == B[0]:
        if f2_add(A[1], B[1]) == F2_ZERO:
            return None
        lam = f2_mul(f2_scal(f2_sqr(A[0]), 3), f2_inv(f2_scal(A[1], 2)))
    else:
        lam = f2_mul(f2_sub(B[1], A[1]), f2_inv(f2_sub(B[0], A[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), A[0]), B[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(A[0], x3)), A[1]))


def g2_mul(Q, k):
    k %= R_MOD
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, Q)
    return acc


assert g2_is_on_curve(G2_GEN)


# ------------------------------------------------------------------ ate pairing
def _line(T, lam, Pt):
    """line through T (twist coordinates) with twist-slope lam, evaluated at the G1 point Pt after untwisting:
    l = yP - lam*xP*w + (lam*xT - yT)*w^3   (w^3 = v*w)"""
    xP, yP = Pt
    c0 = ((yP % P, 0), F2_ZERO, F2_ZERO)
    c1 = (f2_neg(f2_scal(lam, xP)), f2_sub(f2_mul(lam, T[0]), T[1]), F2_ZERO)
    return (c0, c1)


def miller_loop(Pt, Q):
    """f_{t-1,Q}(P); P in G1 (affine ints), Q in G2 (affine Fq2).  Identity inputs give 1."""
    if Pt is None or Q is None:
        return F12_ONE
    f, T = F12_ONE, Q
    for bit in bin(ATE_LOOP)[3:]:
        lam = f2_mul(f2_scal(f2_sqr(T[0]), 3), f2_inv(f2_scal(T[1], 2)))
        f = f12_mul(f12_sqr(f), _line(T, lam, Pt))
        T = g2_add(T, T)
        if bit == "1":
            if T[0] == Q[0]:      # T = -Q can only happen at the very end of a loop over a multiple of the order
                T = g2_add(T, Q)
                continue
            lam = f2_mul(f2_sub(Q[1], T[1]), f2_inv(f2_sub(Q[0], T[0])))
            f = f12_mul(f, _line(T, lam, Pt))
            T = g2_add(T, Q)
    return f


FINAL_EXP = (P**12 - 1) // R_MOD


def final_exponentiation(f):
    # easy part first: f^(q^6 - 1) = conj(f) / f keeps the big exponentiation on a unitary element (same result)
    g = f12_mul((f[0], f6_neg(f[1])), f12_inv(f))
    return f12_pow(g, FINAL_EXP // (P**6 - 1))


def pairing(Pt, Q):
    return final_exponentiation(miller_loop(Pt, Q))


def pairing_product_is_one(pairs) -> bool:
    """prod_i e(P_i, Q_i) == 1 with a single final exponentiation [UPSTREAM multi_miller_loop + final_exponentiation]"""
    f = F12_ONE
    for Pt, Q in pairs:
        f = f12_mul(f, miller_loop(Pt, Q))
    return final_exponentiation(f) == F12_ONE
