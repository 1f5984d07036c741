"""
Host-side mirror of the `halo2_proofs` names on the proving hot path, over libh2hip's C ABI.

The reference reaches these through `halo2_base::halo2_proofs` (halo2-base/src/lib.rs:25-28); the names, argument
meaning and error behaviour below follow upstream halo2-axiom 0.5.3 [UPSTREAM, SURVEY.md A.2/A.6] so that parity
tests read like the reference's own:

    arithmetic::{best_multiexp, best_fft, eval_polynomial, kate_division}
    poly::EvaluationDomain::{new, lagrange_to_coeff, coeff_to_extended, extended_to_coeff, ...}
    poly::kzg::commitment::ParamsKZG::{setup, commit, commit_lagrange, read, write}
      (call sites in the reference: halo2-base/src/utils/mod.rs:401-443, halo2-base/benches/mul.rs:39,
       halo2-base/src/utils/testing.rs:32-50)

All heavy arithmetic runs on the GPU; Python only prepares a handful of domain constants (integers mod r).
Field elements cross this layer as numpy uint64 arrays of Montgomery limbs, (n, 4).
"""
from __future__ import annotations

import struct
from typing import Optional

import numpy as np

from .h2hip import BASES_PLAIN, BASES_PRECOMPUTE, POINT_AFFINE, POINT_JACOBIAN, Bases, Context

# BN254 scalar field constants
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
TWO_ADICITY = 28
ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23


def fr_limbs(v: int) -> np.ndarray:
    """canonical integer -> (1,4) Montgomery limbs"""
    v = ((v % R_MOD) << 256) % R_MOD
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64)


def fr_int(limbs: np.ndarray) -> int:
    row = np.asarray(limbs, dtype=np.uint64).reshape(4).tolist()
    v = row[0] | (row[1] << 64) | (row[2] << 128) | (row[3] << 192)
    return v * pow(1 << 256, -1, R_MOD) % R_MOD


# ------------------------------------------------------------------ arithmetic::*
def best_multiexp(ctx: Context, coeffs: np.ndarray, bases: Bases, point_format: int = POINT_JACOBIAN) -> np.ndarray:
    """arithmetic::best_multiexp(coeffs, bases) -> C::Curve.  `bases` is a resident base set (h2hip_bases)."""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    if len(coeffs) != len(bases):
        raise AssertionError("assertion failed: coeffs.len() == bases.len()")   # upstream assert_eq!
    return ctx.msm(bases, coeffs, point_format)


def best_fft(ctx: Context, a: np.ndarray, omega: np.ndarray, log_n: int) -> np.ndarray:
    """arithmetic::best_fft(&mut a, omega, log_n): natural order in/out, returns the transformed copy."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    if len(a) != 1 << log_n:
        raise AssertionError("assertion failed: a.len() == 1 << log_n")
    return ctx.best_fft(a, omega, log_n)


def eval_polynomial(ctx: Context, poly: np.ndarray, point: np.ndarray) -> np.ndarray:
    return ctx.fr_eval_polynomial(poly, point)


def kate_division(ctx: Context, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(a(X) - a(b)) / (X - b)"""
    return ctx.fr_kate_division(a, b)


def permute_expression_pair(ctx: Context, input_expression: np.ndarray, table_expression: np.ndarray, usable_rows: int):
    """plonk::lookup::prover::permute_expression_pair over the usable rows (the caller appends the blinding rows):
    returns (permuted_input, permuted_table); raises like upstream's ConstraintSystemFailure if an input is not in the table."""
    return ctx.lookup_permute(input_expression, table_expression, usable_rows)


# ------------------------------------------------------------------ poly::EvaluationDomain
class EvaluationDomain:
    """EvaluationDomain::new(j, k): j = constraint-system degree, n = 2^k rows, extended domain 2^extended_k with
    extended_k = k + ceil(log2(j-1)); coset generator g_coset = ZETA."""

    def __init__(self, ctx: Context, j: int, k: int):
        self.ctx, self.k = ctx, k
        self.quotient_poly_degree = j - 1
        n = 1 << k
        extended_k = k
        while (1 << extended_k) < n * self.quotient_poly_degree:
            extended_k += 1
        if extended_k > TWO_ADICITY:
            raise ValueError("extended_k exceeds the field's 2-adicity")
        self.extended_k = extended_k
        self.n = n
        omega = pow(ROOT_OF_UNITY, 1 << (TWO_ADICITY - k), R_MOD)
        ext_omega = pow(ROOT_OF_UNITY, 1 << (TWO_ADICITY - extended_k), R_MOD)
        inv = lambda v: pow(v, -1, R_MOD)
        self.omega, self.omega_inv = fr_limbs(omega), fr_limbs(inv(omega))
        self.extended_omega, self.extended_omega_inv = fr_limbs(ext_omega), fr_limbs(inv(ext_omega))
        self.g_coset, self.g_coset_inv = fr_limbs(ZETA), fr_limbs(ZETA * ZETA % R_MOD)
        self.ifft_divisor = fr_limbs(inv(n))
        self.extended_ifft_divisor = fr_limbs(inv(1 << extended_k))

    def extended_len(self) -> int:
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        assert len(a) == self.n
        return self.ctx.ifft(a, self.omega_inv, self.k, self.ifft_divisor)

    def coeff_to_lagrange(self, a: np.ndarray) -> np.ndarray:   # not in upstream's prover, used by tests
        return self.ctx.best_fft(a, self.omega, self.k)

    def coeff_to_extended(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        assert len(a) == self.n
        return self.ctx.coeff_to_extended(a, self.k, self.extended_k, self.extended_omega, self.g_coset)

    def extended_to_coeff(self, a: np.ndarray) -> np.ndarray:
        """returns the n*(j-1) coefficients (upstream truncates the Vec the same way)"""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        assert len(a) == self.extended_len()
        out = self.ctx.extended_to_coeff(a, self.extended_k, self.extended_omega_inv, self.extended_ifft_divisor, self.g_coset_inv)
        return out[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a: np.ndarray) -> np.ndarray:
        """a[i] /= t(zeta * extended_omega^i), t(X) = X^n - 1, on the extended (coset) domain"""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        assert len(a) == self.extended_len()
        return self.ctx.divide_by_vanishing_poly(a, self.extended_k, self.k, self.extended_omega, self.g_coset)

    def rotate_omega(self, value: np.ndarray, rotation: int) -> np.ndarray:
        w = pow(fr_int(self.omega), rotation % self.n, R_MOD)
        return fr_limbs(fr_int(value) * w % R_MOD)


# ------------------------------------------------------------------ poly::kzg::commitment::ParamsKZG
class ParamsKZG:
    """The prover half of ParamsKZG<Bn256>: k, g (monomial SRS) and g_lagrange, both resident in HBM.
    G2 elements are verifier-side and are carried opaquely (raw bytes) when reading/writing files."""

    def __init__(self, ctx: Context, k: int, g: Bases, g_lagrange: Bases, g2_raw: bytes = b""):
        self.ctx, self.k, self.n = ctx, k, 1 << k
        self.g, self.g_lagrange, self.g2_raw = g, g_lagrange, g2_raw

    @classmethod
    def setup(cls, ctx: Context, k: int, s: int, precompute: bool = True) -> "ParamsKZG":
        """ParamsKZG::setup(k, rng) with the toxic waste `s` passed explicitly (the RNG stays on the host side of
        the FFI).  Both bases are generated on the GPU."""
        g, gl = ctx.params_kzg_setup(k, fr_limbs(s), BASES_PRECOMPUTE if precompute else BASES_PLAIN)
        return cls(ctx, k, g, gl)

    @classmethod
    def from_parts(cls, ctx: Context, k: int, g: Bases, g_lagrange: Bases = None, g2_raw: bytes = b"", precompute: bool = True) -> "ParamsKZG":
        """ParamsKZG::from_parts(k, g, g_lagrange: Option<..>, g2, s_g2): a missing Lagrange basis is derived on the GPU
        (upstream's g_to_lagrange: group inverse FFT of g)."""
        if g_lagrange is None:
            g_lagrange = ctx.g1_to_lagrange(g, k, BASES_PRECOMPUTE if precompute else BASES_PLAIN)
        return cls(ctx, k, g, g_lagrange, g2_raw)

    def commit(self, coeffs: np.ndarray, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        """Params::commit(poly): best_multiexp(coeffs, g[..len]) (KZG ignores the blind)"""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        if len(coeffs) > self.n:
            raise AssertionError("assertion failed: bases.len() >= size")
        return self.ctx.msm(self.g, coeffs, point_format)

    def commit_lagrange(self, values: np.ndarray, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        values = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        if len(values) > self.n:
            raise AssertionError("assertion failed: bases.len() >= size")
        return self.ctx.msm(self.g_lagrange, values, point_format)

    def commit_many(self, polys, lagrange: bool = False, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        """the commitments of one prover round (equal-length host polynomials) as ONE batch call: uploads overlapped with
        compute, sort / accumulation pipelined over lanes, one joint bucket reduction"""
        cols = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys]
        if any(len(c) > self.n for c in cols):
            raise AssertionError("assertion failed: bases.len() >= size")
        return self.ctx.msm_batch(self.g_lagrange if lagrange else self.g, cols, point_format)

    # SerdeFormat::RawBytes layout [UPSTREAM-RECALL, unverified for 0.5.3 — SURVEY.md A.8]:
    # u32 k (LE) | g[0..n) 64 B each | g_lagrange[0..n) 64 B each | g2 128 B | s_g2 128 B
    def write(self, path: str):
        with open(path, "wb") as f:
            f.write(struct.pack("<I", self.k))
            f.write(self.ctx.bases_download(self.g).tobytes())
            f.write(self.ctx.bases_download(self.g_lagrange).tobytes())
            f.write(self.g2_raw)

    @classmethod
    def read(cls, ctx: Context, path: str, precompute: bool = True) -> "ParamsKZG":
        with open(path, "rb") as f:
            (k,) = struct.unpack("<I", f.read(4))
            if k > 26:
                raise ValueError("k too large")
            n = 1 << k
            raw = f.read(2 * n * 64)
            if len(raw) != 2 * n * 64:
                raise ValueError("truncated SRS file")
            g2 = f.read()
        pts = np.frombuffer(raw, dtype=np.uint64).reshape(2 * n, 8)
        flags = BASES_PRECOMPUTE if precompute else BASES_PLAIN
        return cls(ctx, k, ctx.bases_upload(pts[:n], flags), ctx.bases_upload(pts[n:], flags), g2)

    def free(self):
        self.g.free()
        self.g_lagrange.free()
