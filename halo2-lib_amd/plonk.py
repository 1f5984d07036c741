"""
Host-side mirror of `halo2_proofs::plonk::{keygen_vk, keygen_pk, create_proof}` for halo2-base circuits, over libh2hip's
prover entry points (include/h2hip.h, "a1").  The reference reaches these names at

    halo2-base/src/utils/testing.rs:224-227   keygen_vk / keygen_pk
    halo2-base/src/utils/testing.rs:32-50     create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<_>, Challenge255<_>, _,
                                              Blake2bWrite<_, _, _>, _>(params, pk, &[circuit], &[instances], rng, &mut transcript)

The circuit here is what the reference's `BaseCircuitBuilder` hands to the prover after synthesis: its BaseCircuitParams
(halo2-base/src/gates/circuit/mod.rs:25-45), the fixed columns, the copy constraints and the advice columns produced by
`assign_witnesses` (halo2-base/src/gates/flex_gate/threads/single_phase.rs:273-312).  The whole proof is computed on the GPU by
libh2hip; this module only marshals pointers.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import hashlib
from typing import Optional, Sequence

import numpy as np

from .h2hip import Context, _fe, _ptr
from .halo2_proofs import ParamsKZG, R_MOD, fr_limbs

_vp = C.c_void_p
PLONK_STAGES = 12


class BaseCircuitParams(C.Structure):
    """BaseCircuitParams, first phase (gates/circuit/mod.rs:25-45); lookup_bits < 0 means None"""
    _fields_ = [("k", C.c_uint32), ("num_advice", C.c_uint32), ("num_lookup_advice", C.c_uint32), ("num_fixed", C.c_uint32),
                ("num_instance", C.c_uint32), ("lookup_bits", C.c_int32)]

    @classmethod
    def new(cls, k, num_advice, num_lookup_advice, num_fixed, num_instance=0, lookup_bits: Optional[int] = None):
        return cls(k, num_advice, num_lookup_advice, num_fixed, num_instance, -1 if lookup_bits is None else lookup_bits)


class ConstraintSystemShape(C.Structure):
    """what BaseConfig::configure derives from the params (h2hip_plonk_shape)"""
    _fields_ = [("num_advice_total", C.c_uint32), ("num_fixed_total", C.c_uint32), ("table_col", C.c_int32), ("first_constant_col", C.c_int32),
                ("q_lookup_col", C.c_int32), ("first_q_enable_col", C.c_int32), ("num_lookups", C.c_uint32), ("num_perm_columns", C.c_uint32),
                ("num_perm_sets", C.c_uint32), ("degree", C.c_uint32), ("extended_k", C.c_uint32), ("blinding_factors", C.c_uint32),
                ("usable_rows", C.c_uint32), ("quotient_pieces", C.c_uint32), ("num_commitments", C.c_uint32), ("num_evals", C.c_uint32)]


def shape_of(ctx: Context, params: BaseCircuitParams) -> ConstraintSystemShape:
    out = ConstraintSystemShape()
    ctx._chk(ctx.lib.h2hip_plonk_shape_of(C.byref(params), C.byref(out)))
    return out


def transcript_repr(params: BaseCircuitParams, fixed_commitments: np.ndarray, permutation_commitments: np.ndarray) -> int:
    """Stand-in for VerifyingKey::transcript_repr.  Upstream hashes the Rust Debug rendering of the pinned verifying key with
    Blake2b-512("Halo2-Verify-Key"); that rendering belongs to the Rust side of the FFI (the shim passes the value in).  Without Rust
    the same construction is applied to an equivalent description of the key."""
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    s = ("halo2-lib_amd BaseConfig k=%d advice=%d lookup_advice=%d fixed=%d instance=%d lookup_bits=%s" % (
        params.k, params.num_advice, params.num_lookup_advice, params.num_fixed, params.num_instance,
        None if params.lookup_bits < 0 else params.lookup_bits)).encode()
    h.update(len(s).to_bytes(8, "little") + s)
    q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    rinv = pow(1 << 256, -1, q)
    for pts in (fixed_commitments, permutation_commitments):
        for row in np.asarray(pts, dtype=np.uint64).reshape(-1, 8).tolist():
            x = sum(v << (64 * i) for i, v in enumerate(row[:4])) * rinv % q
            y = sum(v << (64 * i) for i, v in enumerate(row[4:])) * rinv % q
            h.update(x.to_bytes(32, "little") + y.to_bytes(32, "little"))     # the identity (0, 0) hashes as 64 zero bytes
    return int.from_bytes(h.digest(), "little") % R_MOD


_RNG_FN = C.CFUNCTYPE(None, _vp, _vp, C.c_size_t)


class _ArrayRngState(C.Structure):   # h2hip_array_rng
    _fields_ = [("values", _vp), ("count", C.c_size_t), ("pos", C.c_size_t), ("exhausted", C.c_int)]


class ArrayRng:
    """`Fr::random(rng)` stream served from a pre-drawn (m, 4) Montgomery array (the RNG itself stays with the caller: upstream draws
    from the `StdRng` the reference seeds at halo2-base/src/utils/testing.rs:38)."""

    def __init__(self, values: np.ndarray):
        self.values = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        self.pos = 0

    def fill_into(self, dst: int, n: int):
        if self.pos + n > len(self.values):
            raise RuntimeError("ArrayRng exhausted")
        C.memmove(dst, self.values.ctypes.data + 32 * self.pos, 32 * n)
        self.pos += n


class _ChaChaRngState(C.Structure):   # h2hip_chacha_rng
    _fields_ = [("seed", C.c_uint8 * 32), ("rounds", C.c_int32), ("pos", C.c_uint64)]


class ChaChaRng:
    """The `Fr::random(&mut rng)` stream of a seeded rand_chacha generator — what the reference hands to create_proof
    (halo2-base/src/utils/testing.rs:38: `StdRng::seed_from_u64(0)`, rand 0.8's StdRng = ChaCha12).  libh2hip's own generator
    (csrc/rng.hip): create_proof produces its large draws ON THE DEVICE from (seed, position); `device=False` forces every draw through the
    one-thread host generator behind a Python callback (same values, same bytes: the A/B the bench line and the tests use).
    seed: 32 bytes, or an int for `seed_from_u64`."""

    def __init__(self, lib, seed=0, rounds: int = 12, device: bool = True):
        self.lib, self.device = lib, device
        if isinstance(seed, int):
            buf = (C.c_uint8 * 32)()
            lib.h2hip_rng_seed_from_u64(C.c_uint64(seed), buf)
            seed = bytes(buf)
        if len(seed) != 32:
            raise ValueError("ChaChaRng: the seed has 32 bytes")
        self.state = _ChaChaRngState()
        lib.h2hip_chacha_rng_init(C.byref(self.state), (C.c_uint8 * 32)(*seed), rounds)

    @property
    def pos(self) -> int:
        return int(self.state.pos)

    def fill_into(self, dst: int, n: int):
        self.lib.h2hip_chacha_rng_fill(C.byref(self.state), _vp(dst), n)

    def fill(self, n: int) -> np.ndarray:
        out = np.empty((n, 4), dtype=np.uint64)
        self.fill_into(out.ctypes.data, n)
        return out


class CallbackRng:
    """adapts any object with fill(n) -> (n, 4) uint64 Montgomery array"""

    def __init__(self, src):
        self.src = src

    def fill_into(self, dst: int, n: int):
        a = np.ascontiguousarray(self.src.fill(n), dtype=np.uint64).reshape(n, 4)
        C.memmove(dst, a.ctypes.data, 32 * n)


class ProvingKey:
    """ProvingKey<G1Affine> resident on the GPU: fixed / permutation polynomials in Lagrange, coefficient and extended form, l_0 /
    l_last / l_blind, and the verifying key's commitments."""

    def __init__(self, ctx: Context, handle, params: BaseCircuitParams, shape: ConstraintSystemShape, kzg: ParamsKZG):
        self.ctx, self.handle, self.params, self.shape, self.kzg = ctx, handle, params, shape, kzg
        fc = np.zeros((shape.num_fixed_total, 8), dtype=np.uint64)
        pc = np.zeros((max(shape.num_perm_columns, 1), 8), dtype=np.uint64)
        ctx._chk(ctx.lib.h2hip_plonk_pk_commitments(handle, _ptr(fc), _ptr(pc)))
        self.fixed_commitments, self.permutation_commitments = fc, pc[: shape.num_perm_columns]
        self.set_transcript_repr(transcript_repr(params, self.fixed_commitments, self.permutation_commitments))

    def set_transcript_repr(self, value: int):
        self.transcript_repr = value % R_MOD
        self.ctx._chk(self.ctx.lib.h2hip_plonk_pk_set_transcript_repr(self.handle, _ptr(fr_limbs(self.transcript_repr))))

    def proof_size(self) -> int:
        return 32 * (self.shape.num_commitments + self.shape.num_evals)

    def free(self):
        if self.handle:
            self.ctx.lib.h2hip_plonk_pk_free(self.ctx.handle, self.handle)
            self.handle = None


def perm_column_index(params: BaseCircuitParams, shape: ConstraintSystemShape, kind: str, index: int) -> int:
    """position of a column among the equality-enabled columns: constants, gate advice, lookup advice, instance"""
    if kind == "fixed":
        return index - shape.first_constant_col
    if kind == "advice":
        return params.num_fixed + index
    if kind == "instance":
        return params.num_fixed + shape.num_advice_total + index
    raise ValueError(kind)


def keygen(kzg: ParamsKZG, params: BaseCircuitParams, fixed: Sequence[np.ndarray], copies) -> ProvingKey:
    """keygen_vk + keygen_pk.  fixed: num_fixed_total (n,4) Lagrange columns; copies: (m,4) uint32 (column, row, column, row) over the
    permutation columns, or a list of (((kind, column), row), ((kind, column), row)) in the order the circuit emitted them."""
    ctx = kzg.ctx
    shape = shape_of(ctx, params)
    n = 1 << params.k
    cols = [_fe(c) for c in fixed]
    if len(cols) != shape.num_fixed_total or any(len(c) != n for c in cols):
        raise ValueError("keygen: need %d fixed columns of 2^k elements" % shape.num_fixed_total)
    if not isinstance(copies, np.ndarray):
        copies = np.array([[perm_column_index(params, shape, l[0][0], l[0][1]), l[1], perm_column_index(params, shape, r[0][0], r[0][1]), r[1]]
                           for l, r in copies], dtype=np.uint32).reshape(-1, 4)
    copies = np.ascontiguousarray(copies, dtype=np.uint32).reshape(-1, 4)
    arr = (_vp * len(cols))(*[_vp(c.ctypes.data) for c in cols])
    out = _vp()
    ctx._chk(ctx.lib.h2hip_plonk_keygen(ctx.handle, C.byref(params), kzg.g.handle, kzg.g_lagrange.handle, arr, _vp(copies.ctypes.data), len(copies),
                                        C.byref(out)))
    return ProvingKey(ctx, out, params, shape, kzg)


def create_proof(pk: ProvingKey, advice: Sequence, instances: Sequence[np.ndarray], rng, timings: Optional[dict] = None,
                 advice_on_device: bool = False) -> bytes:
    """create_proof for one circuit: advice columns (host (n,4) arrays, or device pointers with advice_on_device), instance columns
    ((m,4) arrays), rng = ArrayRng / ChaChaRng / CallbackRng.  Returns the proof bytes (Blake2bWrite::finalize)."""
    ctx, sh = pk.ctx, pk.shape
    n = 1 << pk.params.k
    if len(advice) != sh.num_advice_total:
        raise ValueError("create_proof: need %d advice columns" % sh.num_advice_total)
    keep = None
    if advice_on_device:
        adv = (_vp * len(advice))(*[_vp(int(p)) for p in advice])
    else:
        keep = [_fe(c) for c in advice]
        if any(len(c) != n for c in keep):
            raise ValueError("create_proof: advice columns must have 2^k elements")
        adv = (_vp * len(keep))(*[_vp(c.ctypes.data) for c in keep])
    inst = [_fe(c) for c in instances]
    if len(inst) != pk.params.num_instance:
        raise ValueError("create_proof: need %d instance columns" % pk.params.num_instance)
    ip = (_vp * max(len(inst), 1))(*[_vp(c.ctypes.data) for c in inst])
    il = (C.c_size_t * max(len(inst), 1))(*[len(c) for c in inst])
    err = []

    def _fill(_user, out, count):
        try:
            rng.fill_into(out, count)
        except BaseException as e:   # never unwind through the C frames
            err.append(e)
            C.memset(out, 0, 32 * count)

    proof = np.zeros(pk.proof_size(), dtype=np.uint8)
    plen = C.c_size_t(0)
    stage = (C.c_double * PLONK_STAGES)() if timings is not None else None
    if isinstance(rng, ArrayRng):   # libh2hip's own array RNG: no Python frame per draw (a wide shape draws several hundred blinding tails)
        st = _ArrayRngState(rng.values.ctypes.data + 32 * rng.pos, len(rng.values) - rng.pos, 0, 0)
        rc = ctx.lib.h2hip_plonk_create_proof(ctx.handle, pk.handle, adv, 1 if advice_on_device else 0, ip, il,
                                              C.cast(ctx.lib.h2hip_array_rng_fill, _vp), C.cast(C.pointer(st), _vp), _ptr(proof), proof.nbytes,
                                              C.byref(plen), stage)
        rng.pos += st.pos
        if st.exhausted:
            raise RuntimeError("ArrayRng exhausted")
    elif isinstance(rng, ChaChaRng) and rng.device:   # libh2hip's seeded generator: the prover's device path for the large draws
        rc = ctx.lib.h2hip_plonk_create_proof(ctx.handle, pk.handle, adv, 1 if advice_on_device else 0, ip, il,
                                              C.cast(ctx.lib.h2hip_chacha_rng_fill, _vp), C.cast(C.pointer(rng.state), _vp), _ptr(proof), proof.nbytes,
                                              C.byref(plen), stage)
    else:
        cb = _RNG_FN(_fill)
        rc = ctx.lib.h2hip_plonk_create_proof(ctx.handle, pk.handle, adv, 1 if advice_on_device else 0, ip, il, C.cast(cb, _vp), None, _ptr(proof),
                                              proof.nbytes, C.byref(plen), stage)
    if err:
        raise err[0]
    ctx._chk(rc)
    if timings is not None:
        for i in range(PLONK_STAGES):
            name = ctx.lib.h2hip_plonk_stage_name(i).decode()
            timings[name] = timings.get(name, 0.0) + stage[i]
    del keep
    return proof[: plen.value].tobytes()


def verify_proof(pk: ProvingKey, instances: Sequence[np.ndarray], proof: bytes) -> bool:
    """verify_proof with the key's verifying half (VerifierSHPLONK, SingleStrategy) — check_proof of halo2-base/src/utils/testing.rs:64-88.
    Host code inside libh2hip; needs the G2 half of the SRS (ParamsKZG.g2_raw)."""
    ctx, kzg = pk.ctx, pk.kzg
    if len(kzg.g2_raw) != 256:
        raise ValueError("verify_proof: the ParamsKZG carries no G2 elements (g2_raw)")
    inst = [_fe(c) for c in instances]
    if len(inst) != pk.params.num_instance:
        raise ValueError("verify_proof: need %d instance columns" % pk.params.num_instance)
    ip = (_vp * max(len(inst), 1))(*[_vp(c.ctypes.data) for c in inst])
    il = (C.c_size_t * max(len(inst), 1))(*[len(c) for c in inst])
    g0 = ctx.bases_download(kzg.g)[:1].copy() if not hasattr(kzg, "_g0") else kzg._g0
    kzg._g0 = g0
    g2 = np.frombuffer(kzg.g2_raw, dtype=np.uint8).copy()
    buf = np.frombuffer(bytes(proof), dtype=np.uint8).copy()
    ok = C.c_int(0)
    pc = pk.permutation_commitments if len(pk.permutation_commitments) else np.zeros((1, 8), dtype=np.uint64)
    ctx._chk(ctx.lib.h2hip_plonk_verify_proof(C.byref(pk.params), _ptr(np.ascontiguousarray(pk.fixed_commitments)), _ptr(np.ascontiguousarray(pc)),
                                              _ptr(fr_limbs(pk.transcript_repr)), _ptr(g0), _vp(g2.ctypes.data), _vp(g2.ctypes.data + 128), ip, il,
                                              _vp(buf.ctypes.data), len(buf), C.byref(ok)))
    return bool(ok.value)
