"""
ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).

Restatement of the transcript the reference proves with (halo2-base/src/utils/testing.rs:38-47:
`Blake2bWrite<Vec<u8>, G1Affine, Challenge255<_>>`, read back with `Blake2bRead` at :72-80).  The implementation
lives in un-vendored halo2-axiom 0.5.3 (transcript/blake2b.rs) — [UPSTREAM-RECALL], SURVEY.md A.7:

  * state: Blake2b-512 with personalization "Halo2-Transcript", never reset
  * common_point(P):   absorb 0x01 || x.to_repr() || y.to_repr()        (32-byte little-endian canonical each)
  * common_scalar(s):  absorb 0x02 || s.to_repr()
  * squeeze_challenge: absorb 0x00, finalize a CLONE of the state -> 64 bytes -> Fr::from_uniform_bytes
                       (512-bit little-endian integer mod r)                                  [Challenge255]
  * write_point = common_point + append the 32-byte compressed point to the proof; write_scalar likewise.

Compressed G1 (halo2curves `new_curve_impl!`, bn256 has two spare bits in the top byte): x little-endian with
sign(y) = y.to_repr()[0] & 1 in bit 6 of byte 31 and the identity flagged by bit 7 (all other bits zero).
The flag positions are UNVERIFIED for halo2curves-axiom 0.7.3 (older releases kept the sign in bit 7), so they are
parameters here (`SIGN_BIT`, `INF_BIT`); nothing on the device side depends on them.
"""
from __future__ import annotations

import hashlib

from .bn254 import Q_MOD, R_MOD

PREFIX_CHALLENGE, PREFIX_POINT, PREFIX_SCALAR = b"\x00", b"\x01", b"\x02"
SIGN_BIT, INF_BIT = 6, 7


def fr_repr(v: int) -> bytes:
    return (v % R_MOD).to_bytes(32, "little")


def fq_repr(v: int) -> bytes:
    return (v % Q_MOD).to_bytes(32, "little")


def fr_from_uniform_bytes(b: bytes) -> int:
    assert len(b) == 64
    return int.from_bytes(b, "little") % R_MOD


def g1_compress(P) -> bytes:
    if P is None:
        out = bytearray(32)
        out[31] |= 1 << INF_BIT
        return bytes(out)
    x, y = P
    out = bytearray(fq_repr(x))
    out[31] |= (y & 1) << SIGN_BIT
    return bytes(out)


def g1_decompress(b: bytes):
    assert len(b) == 32
    top = b[31]
    if top >> INF_BIT & 1:
        assert all(v == 0 for v in b[:31]) and top == 1 << INF_BIT, "non-canonical identity encoding"
        return None
    sign = top >> SIGN_BIT & 1
    x = int.from_bytes(b[:31] + bytes([top & ~((1 << SIGN_BIT) | (1 << INF_BIT)) & 0xFF]), "little")
    assert x < Q_MOD, "x not canonical"
    y2 = (x * x * x + 3) % Q_MOD
    y = pow(y2, (Q_MOD + 1) // 4, Q_MOD)          # q = 3 mod 4
    assert y * y % Q_MOD == y2, "point not on the curve"
    if y & 1 != sign:
        y = Q_MOD - y
    return (x, y)


class _Blake2bTranscript:
    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")

    def common_point(self, P):
        if P is None:
            raise ValueError("cannot write points at infinity to the transcript")   # upstream: io::Error
        self.state.update(PREFIX_POINT + fq_repr(P[0]) + fq_repr(P[1]))

    def common_scalar(self, s: int):
        self.state.update(PREFIX_SCALAR + fr_repr(s))

    def squeeze_challenge(self) -> int:
        self.state.update(PREFIX_CHALLENGE)
        return fr_from_uniform_bytes(self.state.copy().digest())


class Blake2bWrite(_Blake2bTranscript):
    """Blake2bWrite::init(vec![]) ... finalize() -> proof bytes"""

    def __init__(self):
        super().__init__()
        self.proof = bytearray()

    def write_point(self, P):
        self.common_point(P)
        self.proof += g1_compress(P)

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += fr_repr(s)

    def finalize(self) -> bytes:
        return bytes(self.proof)


class Blake2bRead(_Blake2bTranscript):
    def __init__(self, proof: bytes):
        super().__init__()
        self.proof, self.pos = bytes(proof), 0

    def _take(self) -> bytes:
        if self.pos + 32 > len(self.proof):
            raise ValueError("proof too short")
        b = self.proof[self.pos:self.pos + 32]
        self.pos += 32
        return b

    def read_point(self):
        P = g1_decompress(self._take())
        self.common_point(P)
        return P

    def read_scalar(self) -> int:
        b = self._take()
        s = int.from_bytes(b, "little")
        if s >= R_MOD:
            raise ValueError("invalid field element encoding in proof")
        self.common_scalar(s)
        return s

    def exhausted(self) -> bool:
        return self.pos == len(self.proof)
