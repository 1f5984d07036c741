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

import ctypes
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
        return cls(ctx, k, g, gl, _g2_raw(G2_GENERATOR) + _g2_raw(_g2_mul(G2_GENERATOR, s)))

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

    # ---- files: ParamsKZG::write / read [UPSTREAM poly/kzg/commitment.rs, SURVEY.md A.8] as the reference uses them at
    # halo2-base/src/utils/mod.rs:401-435 (read_params / read_or_create_srs).  Layout: u32 k (little-endian), g[0..n), g_lagrange[0..n), g2,
    # s_g2; the element encoding is the SerdeFormat:
    #   "raw"        RawBytes (what write()/read() use upstream): 64-byte G1 (x, y as 4 x u64 Montgomery limbs), 128-byte G2
    #   "processed"  Processed: 32-byte compressed G1, 64-byte compressed G2
    # Which of the two a given file holds follows from its size; which one upstream 0.5.3 writes by default could not be checked against
    # its sources here, so both are read (and both can be written).  G2 elements are verifier-side and are carried as opaque bytes.
    def write(self, path: str, fmt: str = "raw"):
        g, gl = self.ctx.bases_download(self.g), self.ctx.bases_download(self.g_lagrange)
        with open(path, "wb") as f:
            f.write(struct.pack("<I", self.k))
            if fmt == "raw":
                f.write(g.tobytes())
                f.write(gl.tobytes())
            elif fmt == "processed":
                f.write(_compress_g1(g))
                f.write(_compress_g1(gl))
            else:
                raise ValueError("fmt must be 'raw' or 'processed'")
            f.write(self.g2_raw)

    @classmethod
    def read(cls, ctx: Context, path: str, precompute: bool = True) -> "ParamsKZG":
        """every point is validated on the device (canonical coordinates, on the curve) like upstream's checked formats; a corrupt or
        foreign file raises ValueError instead of yielding wrong commitments"""
        import os

        size = os.path.getsize(path)
        with open(path, "rb") as f:
            (k,) = struct.unpack("<I", f.read(4))
            if k > 26:
                raise ValueError("k too large")
            n = 1 << k
            if size >= 4 + 2 * n * 64:
                fmt, per = "raw", 64
            elif size >= 4 + 2 * n * 32:
                fmt, per = "processed", 32
            else:
                raise ValueError("truncated SRS file")
            raw = f.read(2 * n * per)
            g2 = f.read()
        flags = BASES_PRECOMPUTE if precompute else BASES_PLAIN
        d_in = ctx.to_device(np.frombuffer(raw, dtype=np.uint8))
        d_pts = d_in
        try:
            if fmt == "processed":
                d_pts = ctx.malloc(2 * n * 64)
                try:
                    ctx._chk(ctx.lib.h2hip_g1_decompress_batch_dev(ctx.handle, d_in, 2 * n, d_pts, 6, 7))
                except Exception as e:
                    raise ValueError("invalid SRS file: %s" % e)
            bad = ctypes.c_size_t(0)
            ctx._chk(ctx.lib.h2hip_g1_validate_dev(ctx.handle, d_pts, 2 * n, ctypes.byref(bad)))
            if bad.value:
                raise ValueError("invalid SRS file: %d of %d points are not canonical curve points" % (bad.value, 2 * n))
            return cls(ctx, k, ctx.bases_from_device(d_pts, n, flags), ctx.bases_from_device(d_pts + 64 * n, n, flags), g2)
        finally:
            ctx.free(d_in)
            if d_pts != d_in:
                ctx.free(d_pts)

    def free(self):
        self.g.free()
        self.g_lagrange.free()


# ------------------------------------------------------------------ verifier-side G2 elements of the SRS (host arithmetic only)
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
# bn256 G2 generator, x = x0 + x1*u, y = y0 + y1*u over F_q[u]/(u^2 + 1)
G2_GENERATOR = ((0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED, 0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
                (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA, 0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B))


def _f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def _f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q_MOD)
    return (a[0] * d % Q_MOD, -a[1] * d % Q_MOD)


def _g2_add(A, B):
    if A is None:
        return B
    if B is None:
        return A
    (x1, y1), (x2, y2) = A, B
    sub = lambda u, v: ((u[0] - v[0]) % Q_MOD, (u[1] - v[1]) % Q_MOD)
    if x1 == x2:
        if sub(y1, y2) != (0, 0) or y1 == (0, 0):
            return None
        x1sq = _f2_mul(x1, x1)
        lam = _f2_mul(((3 * x1sq[0]) % Q_MOD, (3 * x1sq[1]) % Q_MOD), _f2_inv(((2 * y1[0]) % Q_MOD, (2 * y1[1]) % Q_MOD)))
    else:
        lam = _f2_mul(sub(y2, y1), _f2_inv(sub(x2, x1)))
    x3 = sub(sub(_f2_mul(lam, lam), x1), x2)
    return (x3, sub(_f2_mul(lam, sub(x1, x3)), y1))


def _g2_mul(P, k: int):
    acc = None
    for bit in bin(k % R_MOD)[2:] if k % R_MOD else "":
        acc = _g2_add(acc, acc)
        if bit == "1":
            acc = _g2_add(acc, P)
    return acc


def _g2_raw(P) -> bytes:
    """G2Affine in SerdeFormat::RawBytes: x.c0, x.c1, y.c0, y.c1 as 4 x u64 Montgomery limbs (128 bytes)"""
    if P is None:
        return bytes(128)
    out = b""
    for c in (P[0][0], P[0][1], P[1][0], P[1][1]):
        out += ((c << 256) % Q_MOD).to_bytes(32, "little")
    return out


def _compress_g1(points: np.ndarray) -> bytes:
    """(n, 8) Montgomery affine points -> n x 32 bytes: x little-endian canonical, bit 6 of the top byte = y mod 2, bit 7 = identity"""
    rinv = pow(1 << 256, -1, Q_MOD)
    out = bytearray()
    for row in np.asarray(points, dtype=np.uint64).reshape(-1, 8).tolist():
        x = sum(v << (64 * i) for i, v in enumerate(row[:4]))
        y = sum(v << (64 * i) for i, v in enumerate(row[4:]))
        if x == 0 and y == 0:
            b = bytearray(32)
            b[31] |= 0x80
        else:
            b = bytearray((x * rinv % Q_MOD).to_bytes(32, "little"))
            b[31] |= ((y * rinv % Q_MOD) & 1) << 6
        out += b
    return bytes(out)


# ------------------------------------------------------------------ gen_srs (halo2-base/src/utils/mod.rs:401-443)
def _chacha20_block_zero_key(counter: int) -> bytes:
    """keystream block `counter` of ChaCha20 with an all-zero key and nonce (rand_chacha's ChaCha20Rng::from_seed([0; 32]), stream 0)"""
    def rotl(v, c):
        return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))

    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + [0] * 8 + [counter & 0xFFFFFFFF, counter >> 32, 0, 0]
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return b"".join(((x[i] + init[i]) & 0xFFFFFFFF).to_bytes(4, "little") for i in range(16))


def default_srs_secret() -> int:
    """the `s` of `ParamsKZG::<Bn256>::setup(k, ChaCha20Rng::from_seed(Default::default()))` (gen_srs, utils/mod.rs:439-443): Fr::random draws
    eight u64 = the first 64 keystream bytes and reduces them as a 512-bit little-endian integer [UPSTREAM-RECALL for Fr::random]"""
    return int.from_bytes(_chacha20_block_zero_key(0), "little") % R_MOD


def gen_srs(ctx: Context, k: int, params_dir: Optional[str] = None, precompute: bool = True) -> ParamsKZG:
    """halo2_base::utils::fs::gen_srs: read `{PARAMS_DIR or ./params}/kzg_bn254_{k}.srs` if it exists, else run the (insecure, test-only)
    setup with the reference's fixed seed and write the file"""
    import os

    d = params_dir or os.environ.get("PARAMS_DIR", "./params")
    path = os.path.join(d, "kzg_bn254_%d.srs" % k)
    if os.path.exists(path):
        return ParamsKZG.read(ctx, path, precompute)
    os.makedirs(d, exist_ok=True)
    params = ParamsKZG.setup(ctx, k, default_srs_secret(), precompute)
    params.write(path)
    return params
