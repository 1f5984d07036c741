"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Poseidon (x^5, BN254 F_r) restated from its published specification:
round constants and the Cauchy MDS matrix from the Grain LFSR exactly as `poseidon-primitives 0.2.0`
(reference halo2-base/Cargo.toml:32; used by halo2-base/src/poseidon/hasher/spec.rs:88-175) derives them, and
the textbook (un-optimised) permutation, which halo2-base's optimised `PoseidonState::permutation`
(halo2-base/src/poseidon/hasher/state.rs:35-83) is algebraically equal to.

PINNED against the reference's own golden vectors (tests/test_oracle.py):
  * MDS matrix               halo2-base/src/poseidon/hasher/tests/mod.rs:14-30
  * poseidonperm_x5_254_3    halo2-base/src/poseidon/hasher/tests/state.rs:29-33
  * poseidonperm_x5_254_5    halo2-base/src/poseidon/hasher/tests/state.rs:55-61
These KATs exercise F_r addition, multiplication and inversion end to end, which is what anchors the
field arithmetic shared by every other oracle function.
"""
from __future__ import annotations

from .bn254 import R_MOD

P = R_MOD


class Grain:
    def __init__(self, t: int, r_f: int, r_p: int, n_bits: int = 254):
        bits = []

        def push(v, width):
            bits.extend((v >> (width - 1 - i)) & 1 for i in range(width))

        push(1, 2)        # prime field
        push(0, 4)        # s-box x^alpha
        push(n_bits, 12)
        push(t, 12)
        push(r_f, 10)
        push(r_p, 10)
        bits.extend([1] * 30)
        assert len(bits) == 80
        self.state = bits
        self.n_bits = n_bits
        for _ in range(160):
            self._update()

    def _update(self) -> int:
        s = self.state
        new = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        self.state = s[1:] + [new]
        return new

    def _bit(self) -> int:
        while True:
            a = self._update()
            b = self._update()
            if a:
                return b

    def _raw(self) -> int:
        v = 0
        for _ in range(self.n_bits):
            v = (v << 1) | self._bit()
        return v

    def next_field_element(self) -> int:
        while True:
            v = self._raw()
            if v < P:
                return v

    def next_field_element_without_rejection(self) -> int:
        return self._raw() % P


class Spec:
    """round constants [(r_f + r_p)][t] and the t x t MDS matrix."""

    def __init__(self, t: int, r_f: int, r_p: int):
        self.t, self.r_f, self.r_p = t, r_f, r_p
        g = Grain(t, r_f, r_p)
        self.constants = [[g.next_field_element() for _ in range(t)] for _ in range(r_f + r_p)]
        xs = [g.next_field_element_without_rejection() for _ in range(t)]
        ys = [g.next_field_element_without_rejection() for _ in range(t)]
        self.mds = [[pow((xs[i] + ys[j]) % P, -1, P) for j in range(t)] for i in range(t)]

    def permute(self, state):
        """textbook Poseidon permutation (ARK -> S-box -> MDS per round)."""
        t, half = self.t, self.r_f // 2
        s = [v % P for v in state]
        for r in range(self.r_f + self.r_p):
            s = [(a + c) % P for a, c in zip(s, self.constants[r])]
            if r < half or r >= half + self.r_p:
                s = [pow(a, 5, P) for a in s]
            else:
                s[0] = pow(s[0], 5, P)
            s = [sum(self.mds[i][j] * s[j] for j in range(t)) % P for i in range(t)]
        return s

    def absorb_and_permute(self, state, inputs):
        """PoseidonState::permutation with a fixed-length `inputs` (len <= RATE = t-1): add inputs to
        s[1..], add the padding 1 to the first lane without input (if any), permute
        (halo2-base/src/poseidon/hasher/state.rs:124-160)."""
        s = list(state)
        assert len(inputs) < self.t
        for i, v in enumerate(inputs):
            s[1 + i] = (s[1 + i] + v) % P
        if len(inputs) + 1 < self.t:
            s[1 + len(inputs)] = (s[1 + len(inputs)] + 1) % P
        return self.permute(s)
