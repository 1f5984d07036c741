"""
Synthetic halo2-base circuits for the real-prover tests and the create_proof benchmark — the role
`halo2_base::utils::testing` plays in the reference (halo2-base/src/utils/testing.rs:198-247: build a circuit for a
BaseCircuitParams shape, then keygen, gen_proof, check_proof).

The reference fills its advice columns by running halo2-ecc gadgets (e.g. halo2-ecc/src/secp256k1/tests/ecdsa.rs:45-65)
and laying the cells out with `assign_witnesses` (halo2-base/src/gates/flex_gate/threads/single_phase.rs:273-312).  Those
gadgets are the workload, not the hot path, and are not rebuilt; this module produces columns of the same *shape and
statistics* — every row group [a, b, c, d] satisfies the single gate a + b*c = d, values are the 0/1 bits, < 2^lookup_bits
limbs and full-width field elements a non-native-arithmetic circuit is made of, range-checked cells hit the lookup
table, and copy constraints tie cells to constants, to the instance column and to each other — with all field arithmetic
done through a caller-supplied batch backend (`mul`, `add` over (m,4) Montgomery-limb arrays: the GPU context's
fr_mul/fr_add in product use, the CPU oracle in oracle-side tests).
"""
from __future__ import annotations

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
_R2 = pow(1 << 256, 2, R_MOD)


def _raw(vals: np.ndarray) -> np.ndarray:
    """small non-negative integers (uint64 array) -> (m,4) raw limb arrays (NOT yet Montgomery)"""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    out[:, 0] = vals
    return out


def _limbs_of(v: int) -> np.ndarray:
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64)


class SyntheticCircuit:
    """advice: list of (n,4) arrays (gate columns then lookup-advice columns); fixed: list of (n,4) arrays in the shape's fixed-column
    order (table, constants, selector columns); copies: list of (((kind, column), row), ((kind, column), row)); instances: list of int lists."""

    def __init__(self, advice, fixed, copies, instances):
        self.advice, self.fixed, self.copies, self.instances = advice, fixed, copies, instances


def build_circuit(shape, seed: int, backend, num_instance_values: int = 3, cell_mix=None, range_checked_bits=None) -> SyntheticCircuit:
    """`shape`: an object with the attributes of the BaseConfig constraint system (k, n, usable_rows, num_advice, gate_advice,
    lookup_advice, lookups, table_col, constant_cols, q_lookup_col, q_enable_cols, num_fixed_total, num_instance, lookup_bits).

    Cell statistics.  Every gate's a, b, c cells are drawn independently from four kinds — 0, 1, small (< 2^lookup_bits, or < 2^16 without a
    table), full-width (uniform below 2^252) — with probabilities `cell_mix` = (p0, p1, p_small, p_full), default 1/4 each; then the `a` cell of
    every third gate is overwritten by a range-checked value < 2^lookup_bits, every fifth gate's `d` feeds the next gate's `a`, and
    d = a + b*c (full-width whenever b*c or a is).  Resulting column at the default mix (measured by `cell_statistics`, k = 14 ... 19): 17-18 % zeros,
    18 % ones, 33 % other values below 2^lookup_bits, 2-3 % below 2^88, 29 % wider (full-width).  `range_checked_bits`: width of the range-checked
    values (default lookup_bits).  A real halo2-ecc column (secp256k1/tests/ecdsa.rs:104-146 through the
    CRT limbs of fields/fp.rs) holds mostly 0/1 bits, lookup-sized limbs and < 2^88 / 2^90 limbs and FEWER full-width cells: `bench.py` brackets the
    dependence of the proof time on this distribution with cell_mix = (1/2, 1/2, 0, 0) + range_checked_bits = 1 ("all bits": 0 / 1 / 2 only) and
    (0, 0, 0, 1) ("all uniform": 92 % full-width, the rest the range-checked cells)."""
    n, u = shape.n, shape.usable_rows
    g = np.random.default_rng(seed)
    r2 = np.repeat(_limbs_of(_R2), 1, axis=0)
    mont = lambda raw: backend.mul(raw, np.repeat(r2, len(raw), axis=0)) if len(raw) else raw     # x -> x*R (Montgomery form)
    one = _limbs_of((1 << 256) % R_MOD)[0]
    lb = shape.lookup_bits or 0
    m = u // 4                                  # gates per column
    advice, copies = [], []
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(shape.num_fixed_total)]
    if shape.table_col is not None:             # range/mod.rs:154-170: 0..2^lookup_bits, default 0 elsewhere
        assert (1 << lb) <= u
        fixed[shape.table_col][: 1 << lb] = mont(_raw(np.arange(1 << lb, dtype=np.uint64)))
    single = shape.q_lookup_col is not None
    small_vals = {}
    for ci, col in enumerate(shape.gate_advice):
        def draw(count, kinds):
            """circuit-like cells: kind 0 -> 0, 1 -> 1, 2 -> < 2^lookup_bits (or < 2^16), 3 -> full-width"""
            out = np.zeros((count, 4), dtype=np.uint64)
            out[kinds == 1] = one
            idx = np.where(kinds == 2)[0]
            out[idx] = mont(_raw(g.integers(0, 1 << (lb or 16), size=len(idx), dtype=np.uint64)))
            idx = np.where(kinds == 3)[0]
            full = g.integers(0, 2**63, size=(len(idx), 4), dtype=np.uint64) * np.uint64(2) + g.integers(0, 2, size=(len(idx), 4), dtype=np.uint64)
            full[:, 3] &= np.uint64((1 << 60) - 1)
            out[idx] = full
            return out
        # (the default mix keeps the generator's original stream: the committed golden digests were made from it)
        kinds = (lambda: g.integers(0, 4, size=m)) if cell_mix is None else (lambda: g.choice(4, size=m, p=np.asarray(cell_mix, dtype=np.float64)))
        A = draw(m, kinds())
        B = draw(m, kinds())
        Cc = draw(m, kinds())
        j = np.arange(m)
        look = (j % 3 == 0) if lb else np.zeros(m, dtype=bool)          # gates whose `a` cell is range-checked
        if lb:
            raw_small = g.integers(0, 1 << (lb if range_checked_bits is None else min(lb, range_checked_bits)), size=int(look.sum()), dtype=np.uint64)
            raw_small[: min(len(raw_small), 8)] = (1 << lb) - 1            # the table's largest value appears too
            A[look] = mont(_raw(raw_small))
            small_vals[col] = (np.where(look)[0], A[look])
        D = backend.add(A, backend.mul(B, Cc))
        chain = np.where((j % 5 == 0) & ((j + 1) % 3 != 0) & (j + 1 < m))[0]   # a_{j+1} = d_j (an inner-product style chain)
        A[chain + 1] = D[chain]
        D = backend.add(A, backend.mul(B, Cc))
        colv = np.zeros((n, 4), dtype=np.uint64)
        colv[0:4 * m:4], colv[1:4 * m:4], colv[2:4 * m:4], colv[3:4 * m:4] = A, B, Cc, D
        advice.append(colv)
        fixed[shape.q_enable_cols[ci]][0:4 * m:4] = one
        for t in chain[: 64 if m > 256 else len(chain)]:
            copies.append(((("advice", col), 4 * int(t) + 3), (("advice", col), 4 * int(t) + 4)))
        if single:
            fixed[shape.q_lookup_col][4 * np.where(look)[0]] = one
    # lookup-advice columns (multi-column shapes): copies of the range-checked cells of gate column 0, zero elsewhere
    for li, la in enumerate(shape.lookup_advice):
        colv = np.zeros((n, 4), dtype=np.uint64)
        rows, vals = small_vals[shape.gate_advice[li % len(shape.gate_advice)]]
        cnt = min(len(rows), u)
        colv[:cnt] = vals[:cnt]
        advice.append(colv)
        src = shape.gate_advice[li % len(shape.gate_advice)]
        for t in range(min(cnt, 64)):
            copies.append(((("advice", la), t), (("advice", src), 4 * int(rows[t]))))
    # constants: constant j of column c is exposed as the `b` cell of gate j in gate column 0 (copy_constraints.rs: constants assigned
    # to fixed cells and constrained equal to the advice cells that use them)
    ncst = min(16, m)
    for c in shape.constant_cols:
        fixed[c][:ncst] = advice[0][1:4 * ncst:4]
        for t in range(ncst):
            copies.append(((("fixed", c), t), (("advice", shape.gate_advice[0]), 4 * t + 1)))
    # instances: the `c` cells of the first gates are public
    instances = []
    for i in range(shape.num_instance):
        cnt = min(num_instance_values, m)
        vals = advice[0][2:4 * cnt:4]
        instances.append(vals.copy())
        for t in range(cnt):
            copies.append(((("instance", i), t), (("advice", shape.gate_advice[0]), 4 * t + 2)))
    return SyntheticCircuit(advice, fixed, copies, instances)


def cell_statistics(column: np.ndarray, usable_rows: int, lookup_bits: int) -> dict:
    """fractions of a column's assigned cells (raw Montgomery limbs, (n,4) u64) that are 0, 1, another value below 2^lookup_bits, another value
    below 2^88 (a CRT limb), anything wider — what the counting sort of the commitment MSM and its zero-digit skipping see"""
    a = np.ascontiguousarray(column[:usable_rows], dtype=np.uint64)
    rinv = pow(1 << 256, -1, R_MOD)
    one = _limbs_of((1 << 256) % R_MOD)[0]
    zero = ~a.any(axis=1)
    is_one = (a == one).all(axis=1)
    rest = np.where(~zero & ~is_one)[0]
    # canonical values of a sample of the remaining cells (big-int conversion: 4096 cells are plenty for two-digit fractions)
    pick = rest if len(rest) <= 4096 else rest[np.random.default_rng(1).choice(len(rest), 4096, replace=False)]
    vals = [((int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192) * rinv) % R_MOD for r in a[pick]]
    small = sum(1 for v in vals if v < (1 << max(lookup_bits, 1)))
    limb = sum(1 for v in vals if (1 << max(lookup_bits, 1)) <= v < (1 << 88))
    tot, nrest, ns = float(len(a)), float(len(rest)), float(max(len(vals), 1))
    return {"cells": int(tot), "zero": float(zero.sum()) / tot, "one": float(is_one.sum()) / tot, "below_2^lookup_bits": nrest / tot * small / ns,
            "below_2^88": nrest / tot * limb / ns, "wider": nrest / tot * (len(vals) - small - limb) / ns}
