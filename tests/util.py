"""Shared helpers for parity tests (host side only)."""
import numpy as np

from oracle import bn254 as O

R, Q = O.R_MOD, O.Q_MOD


def fr(vals):
    return O.ints_to_limbs(vals, R)


def rand_fr(n, seed):
    """n uniformly random *valid* field elements as raw Montgomery limbs (value < 2^252 < r), generated with
    numpy so that 2^20..2^22-element inputs take milliseconds."""
    a = np.random.default_rng(seed).integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + \
        np.random.default_rng(seed + 1).integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


def circuit_like_fr(n, seed):
    """advice-column-shaped scalars: ~50% zero, ~25% one, rest < 2^88, Montgomery limbs."""
    g = np.random.default_rng(seed)
    kind = g.integers(0, 4, size=n)
    one = fr([1])[0]
    small = fr([int(x) for x in g.integers(0, 2**62, size=64)] + [(1 << 88) - 1])   # pool of <2^88 values
    out = np.zeros((n, 4), dtype=np.uint64)
    out[kind == 2] = one
    idx = np.where(kind == 3)[0]
    out[idx] = small[g.integers(0, len(small), size=len(idx))]
    return out


def jac_to_affine_ints(j):
    X, Y, Z = [O.limbs_to_ints(j.reshape(-1, 12)[:, 4 * i:4 * i + 4], Q)[0] for i in range(3)]
    if Z == 0:
        return None
    zi = pow(Z, -1, Q)
    return (X * zi * zi % Q, Y * zi * zi * zi % Q)


def domain_consts(k):
    """(omega, omega_inv, 2^-k) as (1,4) limb arrays."""
    w = O.omega_for(k)
    return fr([w]), fr([O.inv_mod(w, R)]), fr([O.inv_mod(1 << k, R)])


class PreDrawnRng:
    """The prover's `Fr::random(rng)` stream as a pre-drawn array, served identically to the oracle prover (fill / next_fr, oracle/plonk.py)
    and to the HIP prover (fill_into, halo2-lib_amd/plonk.py): both sides consume the same values in the same order."""

    def __init__(self, count, seed):
        self.values = rand_fr(count, seed)
        self.pos = 0

    def fill(self, m):
        if self.pos + m > len(self.values):
            raise RuntimeError("PreDrawnRng exhausted")
        out = self.values[self.pos:self.pos + m]
        self.pos += m
        return out

    def next_fr(self):
        return O.limbs_to_ints(self.fill(1), R)[0]

    def fill_into(self, dst, m):
        import ctypes

        a = self.fill(m)
        ctypes.memmove(dst, a.ctypes.data, 32 * m)
