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


def edge_fr_values():
    """the limb patterns rand_fr never produces: values in [2^252, r), the top of the range, the Montgomery
    constants — as RAW limb patterns (what the kernels see; as Montgomery residues they stand for value * 2^-256 mod r)"""
    return [0, 1, 2, R - 1, R - 2, R - 3, (1 << 252) - 1, 1 << 252, (1 << 252) + 1, (1 << 253) - 1, 1 << 253, (1 << 253) + 1,
            (1 << 256) % R, (1 << 512) % R, (R - (1 << 256) % R) % R, (R + 1) // 2, (R - 1) // 2, R - (1 << 64), R - (1 << 128), R - (1 << 192),
            (1 << 64) - 1, (1 << 128) - 1, (1 << 192) - 1, ((1 << 253) | ((1 << 192) - 1)) % R, 0x30644e72e131a029 << 192]


def full_range_fr(n, seed, edges=True):
    """n raw limb patterns UNIFORM over the whole of [0, r) (rejection sampling on 254-bit draws, numpy-vectorised), with the edge patterns
    of edge_fr_values() written over pseudo-random positions (and positions 0 / n-1) when `edges`: two thirds of the draws lie in
    [2^252, r), the range rand_fr leaves out."""
    g = np.random.default_rng([seed, 0xF011])
    out = np.empty((0, 4), dtype=np.uint64)
    rl = np.array([(R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    while len(out) < n:
        m = max(16, int((n - len(out)) * 1.4) + 8)
        a = g.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64, endpoint=False)
        a[:, 3] &= np.uint64((1 << 62) - 1)
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (a[:, i] < rl[i])
            eq &= a[:, i] == rl[i]
        out = np.concatenate([out, a[lt]])
    out = np.ascontiguousarray(out[:n])
    if edges and n:
        ev = _raw_limbs(edge_fr_values())
        pos = g.integers(0, n, size=len(ev))
        pos[0], pos[1] = 0, n - 1
        out[pos] = ev[: len(pos)]
    return out


def _raw_limbs(vals):
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)


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
