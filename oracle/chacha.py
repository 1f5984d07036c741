"""
ORACLE — TEST INFRASTRUCTURE ONLY.  The `Fr::random(&mut rng)` stream of a seeded rand_chacha generator, restated independently of
halo2-lib_amd/csrc/rng.hip (numpy over whole arrays of blocks + Python big integers for the reduction).

What the reference uses (and what this restates):
  * halo2-base/src/utils/testing.rs:38   `StdRng::seed_from_u64(0)` handed to create_proof  (rand 0.8: StdRng = ChaCha12Rng)
  * halo2-base/src/utils/mod.rs:441      `ChaCha20Rng::from_seed(Default::default())` for the SRS secret

The ChaCha block function is RFC 8439's (section 2.3; tests/test_rng_chacha.py checks the RFC's vectors).  The STREAM LAYOUT is recalled from
rand_chacha 0.3 / rand_core 0.6 / ff 0.13 / halo2curves-axiom 0.7.3, none of which is vendored [UPSTREAM-RECALL, INTEGRATION.md section 8]:
  - state words 12, 13 = 64-bit block counter starting at 0; words 14, 15 = 64-bit stream id 0; key = the 32 seed bytes as 8 LE words
  - next_u64 = two consecutive 32-bit words, low word first; Fr::random = Fr::from_u512 of eight next_u64 = ONE 64-byte block
  - from_u512 = the 512-bit little-endian integer reduced mod r
  - seed_from_u64 = rand_core's PCG32 expansion (MUL 6364136223846793005, INC 11634580027462260723, output xorshift-rotate, LE bytes)
"""
from __future__ import annotations

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _rotl(v, c):
    return (v << np.uint32(c)) | (v >> np.uint32(32 - c))


def chacha_blocks(seed: bytes, counters, rounds: int = 20, stream: int = 0) -> np.ndarray:
    """keystream blocks for an array of 64-bit counters -> (len, 16) uint32 words"""
    assert len(seed) == 32 and rounds % 2 == 0
    ctr = np.asarray(counters, dtype=np.uint64).reshape(-1)
    m = len(ctr)
    key = np.frombuffer(seed, dtype="<u4")
    init = np.empty((16, m), dtype=np.uint32)
    for i, c in enumerate((0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)):
        init[i] = c
    for i in range(8):
        init[4 + i] = key[i]
    init[12] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    init[13] = (ctr >> np.uint64(32)).astype(np.uint32)
    init[14] = stream & 0xFFFFFFFF
    init[15] = (stream >> 32) & 0xFFFFFFFF
    x = init.copy()

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] += x[d]; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] += x[b]; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] += x[d]; x[b] = _rotl(x[b] ^ x[c], 7)

    with np.errstate(over="ignore"):
        for _ in range(rounds // 2):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        x += init
    return np.ascontiguousarray(x.T)


def seed_from_u64(state: int) -> bytes:
    """rand_core::SeedableRng::seed_from_u64 for a 32-byte seed"""
    MUL, INC, M64 = 6364136223846793005, 11634580027462260723, (1 << 64) - 1
    out = b""
    for _ in range(8):
        state = (state * MUL + INC) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) % 32))) & 0xFFFFFFFF
        out += x.to_bytes(4, "little")
    return out


def fr_random_ints(seed: bytes, first: int, count: int, rounds: int = 12) -> list:
    """elements first .. first + count - 1 of the `Fr::random` stream as canonical integers"""
    w = chacha_blocks(seed, np.arange(first, first + count, dtype=np.uint64), rounds)
    raw = w.astype("<u4").tobytes()
    return [int.from_bytes(raw[64 * i:64 * i + 64], "little") % R_MOD for i in range(count)]


class ChaChaFrRng:
    """oracle-side RNG object with the interface oracle/plonk.py's create_proof expects (next_fr / fill): the seeded stream, in order"""

    def __init__(self, seed=0, rounds: int = 12):
        self.seed = seed_from_u64(seed) if isinstance(seed, int) else bytes(seed)
        self.rounds, self.pos = rounds, 0

    def next_fr(self) -> int:
        v = fr_random_ints(self.seed, self.pos, 1, self.rounds)[0]
        self.pos += 1
        return v

    def fill(self, m: int) -> np.ndarray:
        from oracle import bn254 as O

        vals = fr_random_ints(self.seed, self.pos, m, self.rounds)
        self.pos += m
        return O.ints_to_limbs(vals, R_MOD)
