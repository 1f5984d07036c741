"""
ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  CPU restatement of `create_proof` / `verify_proof` for
halo2-base circuits, i.e. of what the reference executes at

    halo2-base/src/utils/testing.rs:32-50   gen_proof_with_instances -> create_proof::<KZGCommitmentScheme<Bn256>,
                                            ProverSHPLONK<_>, Challenge255<_>, _, Blake2bWrite<..>, _>
    halo2-base/src/utils/testing.rs:64-88   check_proof_with_instances -> verify_proof::<.., VerifierSHPLONK<_>, ..,
                                            SingleStrategy<_>>

The prover/verifier themselves live in the un-vendored halo2-axiom 0.5.3 (Cargo.lock:1063-1065); everything marked
[UPSTREAM-RECALL] below is restated from memory of that code base (PSE halo2 lineage) and could not be diffed against
it here.  What the reference *does* hold — the constraint system the prover runs on — is followed exactly:

    BaseConfig::configure          halo2-base/src/gates/circuit/mod.rs:70-96
    RangeConfig::configure/lookup  halo2-base/src/gates/range/mod.rs:71-150   (table column first; q_lookup complex selector when
                                   there is a single advice column, else dedicated lookup-advice columns without selector)
    FlexGateConfig / BasicGate     halo2-base/src/gates/flex_gate/mod.rs:61-91,120-146  (constants + enable_equality, one selector
                                   per advice column, gate q*(a + b*c - out) at rotations 0..3)

PARITY: the reference has no golden proofs ("no golden proof bytes anywhere"), so proof bytes are *parity
unpinned* against upstream; they are pinned (i) between this restatement and the HIP prover (byte equality), and (ii) by the
independent verifier below, which accepts only if every commitment, evaluation and opening is consistent (real pairing check).

Heavy vector steps run in the C restatement (oracle/h2_oracle.c) on numpy Montgomery-limb arrays; this file holds the
protocol: ordering, transcript, challenges, SHPLONK.  Vectors: (n,4) uint64; scalars: Python ints (canonical).
"""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import bn254 as O
from . import c_oracle as CO
from . import pairing as PR
from .transcript import Blake2bRead, Blake2bWrite, fr_from_uniform_bytes

R = O.R_MOD


def fr1(v: int) -> np.ndarray:
    return O.ints_to_limbs([v % R], R)


def to_int(a) -> int:
    return O.limbs_to_ints(np.asarray(a).reshape(1, 4), R)[0]


# ====================================================================================== constraint system shape
@dataclass
class Shape:
    """The ConstraintSystem BaseConfig::configure builds (phase 0 only), as index lists.

    Column indices follow the creation order in the reference's configure functions; query lists follow the order of the
    first query_* call (enable_equality queries the column at Rotation::cur(), halo2 `query_any_index`) [UPSTREAM-RECALL for
    the bookkeeping inside ConstraintSystem; the call order is the reference's]."""
    k: int
    num_advice: int                 # gate advice columns (num_advice_per_phase[0])
    num_lookup_advice: int          # num_lookup_advice_per_phase[0]
    num_fixed: int                  # constants columns
    num_instance: int = 0
    lookup_bits: Optional[int] = None

    def __post_init__(self):
        self.n = 1 << self.k
        self.with_range = self.lookup_bits is not None and self.num_lookup_advice != 0
        single = self.with_range and self.num_advice == 1          # range/mod.rs:93-95: lookups on the gate column itself
        nf = 0
        self.table_col = None
        if self.with_range:
            self.table_col = nf                                     # meta.lookup_table_column() comes first (range/mod.rs:82)
            nf += 1
        self.constant_cols = list(range(nf, nf + self.num_fixed))   # flex_gate/mod.rs:123-129
        nf += self.num_fixed
        self.gate_advice = list(range(self.num_advice))
        self.lookup_advice = [] if (single or not self.with_range) else list(range(self.num_advice, self.num_advice + self.num_lookup_advice))
        self.num_advice_total = self.num_advice + len(self.lookup_advice)
        # selectors -> fixed columns (compress_selectors [UPSTREAM-RECALL]: complex / degree-0 selectors get their column first, then
        # the simple ones; the per-column gate selectors are enabled on common rows and therefore never share a column)
        self.q_lookup_col = None
        if single:
            self.q_lookup_col = nf
            nf += 1
        self.q_enable_cols = list(range(nf, nf + self.num_advice))
        nf += self.num_advice
        self.num_fixed_total = nf
        # gates: (selector fixed column, advice column); lookups: (selector fixed column | None, advice column, table fixed column)
        self.gates = [(self.q_enable_cols[i], i) for i in range(self.num_advice)]
        self.lookups = []
        if single:
            self.lookups.append((self.q_lookup_col, 0, self.table_col))
        for la in self.lookup_advice:
            self.lookups.append((None, la, self.table_col))
        # permutation columns in enable_equality order: constants, gate advice, lookup advice, instance
        self.perm_columns = [("fixed", c) for c in self.constant_cols] + [("advice", a) for a in self.gate_advice] + \
                            [("advice", a) for a in self.lookup_advice] + [("instance", i) for i in range(self.num_instance)]
        # queries
        self.advice_queries = []
        for a in self.gate_advice:
            self.advice_queries += [(a, 0), (a, 1), (a, 2), (a, 3)]
        self.advice_queries += [(a, 0) for a in self.lookup_advice]
        self.fixed_queries = [(c, 0) for c in self.constant_cols]
        if self.with_range:
            self.fixed_queries.append((self.table_col, 0))
        if single:
            self.fixed_queries.append((self.q_lookup_col, 0))
        self.fixed_queries += [(c, 0) for c in self.q_enable_cols]
        self.instance_queries = [(i, 0) for i in range(self.num_instance)]
        # degrees: gate 3; lookup max(4, 2 + deg(input) + deg(table)); permutation 3
        deg = 3
        for q, _, _ in self.lookups:
            deg = max(deg, 4, 2 + (2 if q is not None else 1) + 1)
        self.degree = deg
        max_queries = max([sum(1 for (c, _) in self.advice_queries if c == a) for a in range(self.num_advice_total)] + [1])
        self.blinding_factors = max(3, max_queries) + 2
        self.usable_rows = self.n - (self.blinding_factors + 1)
        self.chunk_len = self.degree - 2
        self.quotient_poly_degree = self.degree - 1
        ek = self.k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        self.num_perm_sets = (len(self.perm_columns) + self.chunk_len - 1) // self.chunk_len if self.perm_columns else 0

    def pinned(self) -> str:
        """our own stand-in for upstream's `format!("{:?}", vk.pinned())` (Rust Debug output; not reproducible here)"""
        return "halo2-lib_amd BaseConfig k=%d advice=%d lookup_advice=%d fixed=%d instance=%d lookup_bits=%s" % (
            self.k, self.num_advice, self.num_lookup_advice, self.num_fixed, self.num_instance, self.lookup_bits)


class Domain:
    """EvaluationDomain::new(degree, k)  """

    def __init__(self, shape: Shape):
        self.k, self.ek, self.n = shape.k, shape.extended_k, shape.n
        self.omega = O.omega_for(self.k)
        self.omega_inv = O.inv_mod(self.omega, R)
        self.ext_omega = O.omega_for(self.ek)
        self.step = 1 << (self.ek - self.k)

    def rotate_omega(self, x: int, rot: int) -> int:
        return x * pow(self.omega, rot % self.n, R) % R

    def lagrange_to_coeff(self, a, threads=1):
        return CO.ifft(a, self.k, fr1(self.omega), threads)

    def coeff_to_extended(self, a, threads=1):
        return CO.coeff_to_extended(a, self.k, self.ek, fr1(self.ext_omega), fr1(O.ZETA), threads)

    def extended_to_coeff(self, a, threads=1):
        return CO.extended_to_coeff(a, self.ek, fr1(self.ext_omega), fr1(O.ZETA), threads)


# ====================================================================================== SRS (ParamsKZG)
@dataclass
class Params:
    """ParamsKZG<Bn256> as numpy point arrays (g, g_lagrange: (n,8) Montgomery affine) + verifier elements as Python points"""
    k: int
    g: np.ndarray
    g_lagrange: np.ndarray
    g2: tuple = None
    s_g2: tuple = None

    @classmethod
    def setup(cls, k: int, s: int, g=None, g_lagrange=None, threads: int = 4):
        """ParamsKZG::setup with the toxic waste given explicitly: g[i] = s^i G, g_lagrange[i] = L_i(s) G  (SURVEY.md A.8).
        The base arrays may be supplied (e.g. downloaded from the GPU setup, itself tested against this definition)."""
        n = 1 << k
        G = O.points_to_limbs([O.G1_GEN])
        if g is None:
            g = CO.g1_fixed_base_batch(G, CO.fr_geom(fr1(1), fr1(s), n, threads), threads)
        if g_lagrange is None:
            # L_i(s) = (s^n - 1) * omega^i / (n * (s - omega^i)); s in the domain: L_i(s) = [omega^i == s]
            w = O.omega_for(k)
            wi = CO.fr_geom(fr1(1), fr1(w), n, threads)
            den = CO.fr_sub(np.repeat(fr1(s), n, axis=0), wi)
            if not den.any(axis=1).all():
                li = np.where(den.any(axis=1)[:, None], np.zeros((n, 4), dtype=np.uint64), np.repeat(fr1(1), n, axis=0))
            else:
                num = (pow(s, n, R) - 1) * O.inv_mod(n, R) % R
                li = CO.fr_mul_mt(CO.fr_mul_mt(wi, CO.fr_batch_invert_mt(den, threads), threads), np.repeat(fr1(num), n, axis=0), threads)
            g_lagrange = CO.g1_fixed_base_batch(G, li, threads)
        return cls(k, np.ascontiguousarray(g), np.ascontiguousarray(g_lagrange), PR.G2_GEN, PR.g2_mul(PR.G2_GEN, s))

    def commit(self, coeffs, threads=1):
        c = np.ascontiguousarray(coeffs).reshape(-1, 4)
        return O.limbs_to_points(CO.best_multiexp(c, self.g[: len(c)], threads))[0]

    def commit_lagrange(self, values, threads=1):
        c = np.ascontiguousarray(values).reshape(-1, 4)
        return O.limbs_to_points(CO.best_multiexp(c, self.g_lagrange[: len(c)], threads))[0]


# ====================================================================================== keygen
class PermutationAssembly:
    """permutation::keygen::Assembly [UPSTREAM-RECALL]: cycles merged by `copy`, sigma_i(omega^j) = delta^i' * omega^j' for mapping[i][j] = (i', j')"""

    def __init__(self, shape: Shape):
        self.shape = shape
        m, n = len(shape.perm_columns), shape.n
        self.col_index = {c: i for i, c in enumerate(shape.perm_columns)}
        self.mapping = [[(i, j) for j in range(n)] for i in range(m)]
        self.aux = [[(i, j) for j in range(n)] for i in range(m)]
        self.sizes = [[1] * n for _ in range(m)]

    def copy(self, left, right):
        """left/right: ((kind, column), row)"""
        (lc, lr), (rc, rr) = (self.col_index[left[0]], left[1]), (self.col_index[right[0]], right[1])
        assert lr < self.shape.usable_rows and rr < self.shape.usable_rows, "NotEnoughRowsAvailable"
        lcy, rcy = self.aux[lc][lr], self.aux[rc][rr]
        if lcy == rcy:
            return
        if self.sizes[lcy[0]][lcy[1]] < self.sizes[rcy[0]][rcy[1]]:
            lcy, rcy = rcy, lcy
        self.sizes[lcy[0]][lcy[1]] += self.sizes[rcy[0]][rcy[1]]
        i = rcy
        while True:
            self.aux[i[0]][i[1]] = lcy
            i = self.mapping[i[0]][i[1]]
            if i == rcy:
                break
        self.mapping[lc][lr], self.mapping[rc][rr] = self.mapping[rc][rr], self.mapping[lc][lr]

    def sigma_values(self) -> List[np.ndarray]:
        """Lagrange values of the permutation polynomials, one (n,4) array per permutation column"""
        sh = self.shape
        n, w = sh.n, O.omega_for(sh.k)
        wpow = [1] * n
        for j in range(1, n):
            wpow[j] = wpow[j - 1] * w % R
        dpow = [pow(O.DELTA, i, R) for i in range(len(sh.perm_columns))]
        return [O.ints_to_limbs([dpow[pi] * wpow[pj] % R for (pi, pj) in col], R) for col in self.mapping]


@dataclass
class VerifyingKey:
    shape: Shape
    fixed_commitments: list
    permutation_commitments: list
    transcript_repr: int


@dataclass
class ProvingKey:
    vk: VerifyingKey
    fixed_values: list          # Lagrange (n,4)
    fixed_polys: list           # coefficient form
    fixed_cosets: list          # extended domain
    sigma_values: list
    sigma_polys: list
    sigma_cosets: list
    l0: np.ndarray              # extended-domain evaluations of l_0, l_last, l_active_row
    l_last: np.ndarray
    l_active: np.ndarray


def transcript_repr_for(shape: Shape, fixed_commitments, permutation_commitments) -> int:
    """VerifyingKey::transcript_repr: Blake2b-512("Halo2-Verify-Key") over a description of the key -> from_uniform_bytes.  Upstream
    hashes the Rust Debug rendering of the pinned key, which cannot be reproduced without the sources; the value is an INPUT of
    create_proof here (it stays on the host side of the FFI), and this stand-in hashes an equivalent description."""
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    s = shape.pinned().encode()
    h.update(len(s).to_bytes(8, "little") + s)
    for P in list(fixed_commitments) + list(permutation_commitments):
        h.update(b"\x00" * 64 if P is None else P[0].to_bytes(32, "little") + P[1].to_bytes(32, "little"))
    return fr_from_uniform_bytes(h.digest())


class _LazyCosets:
    """list-like: item i = coeff_to_extended(polys[i]), computed at every access and never cached (H2_ORACLE_LOWMEM=1)"""

    def __init__(self, dom, polys, threads):
        self.dom, self.polys, self.threads = dom, polys, threads

    def __len__(self):
        return len(self.polys)

    def __getitem__(self, i):
        return self.dom.coeff_to_extended(self.polys[i], self.threads)

    def __iter__(self):
        return (self[i] for i in range(len(self.polys)))


def keygen(params: Params, shape: Shape, fixed_values: List[np.ndarray], assembly: PermutationAssembly, threads=1) -> ProvingKey:
    """keygen_vk + keygen_pk for the shape: fixed_values = Lagrange values of ALL fixed columns in Shape order (table, constants,
    selector columns), each (n,4)."""
    assert len(fixed_values) == shape.num_fixed_total
    dom = Domain(shape)
    n, bf = shape.n, shape.blinding_factors
    fixed_values = [np.ascontiguousarray(v, dtype=np.uint64).reshape(n, 4) for v in fixed_values]
    fixed_polys = [dom.lagrange_to_coeff(v, threads) for v in fixed_values]
    sig_vals = assembly.sigma_values()
    sig_polys = [dom.lagrange_to_coeff(v, threads) for v in sig_vals]
    if os.environ.get("H2_ORACLE_LOWMEM") == "1":
        # r06 (tests/golden/make_proof_goldens.py for bench_msm.config:13, k = 23 with 6 + 1 advice columns): the proving key's extended-domain forms —
        # 1 GiB per column at k = 23, 17 columns — are recomputed from the coefficient forms whenever create_proof indexes them instead of being held
        # for the key's lifetime (same values, hence the same proof bytes: checked against a committed digest in tests/test_oracle.py)
        fixed_cosets = _LazyCosets(dom, fixed_polys, threads)
        sig_cosets = _LazyCosets(dom, sig_polys, threads)
    else:
        fixed_cosets = [dom.coeff_to_extended(p, threads) for p in fixed_polys]
        sig_cosets = [dom.coeff_to_extended(p, threads) for p in sig_polys]
    fixed_comm = [params.commit_lagrange(v, threads) for v in fixed_values]
    perm_comm = [params.commit_lagrange(v, threads) for v in sig_vals]
    one, zero = fr1(1)[0], fr1(0)[0]
    l0 = np.tile(zero, (n, 1))
    l0[0] = one
    l_last = np.tile(zero, (n, 1))
    l_last[n - bf - 1] = one
    l_blind = np.tile(zero, (n, 1))
    l_blind[n - bf:] = one
    ext = lambda v: dom.coeff_to_extended(dom.lagrange_to_coeff(v, threads), threads)
    l0e, lle, lbe = ext(l0), ext(l_last), ext(l_blind)
    ones = np.tile(one, (1 << shape.extended_k, 1))
    l_active = CO.fr_sub(CO.fr_sub(ones, lle), lbe)
    vk = VerifyingKey(shape, fixed_comm, perm_comm, transcript_repr_for(shape, fixed_comm, perm_comm))
    return ProvingKey(vk, fixed_values, fixed_polys, fixed_cosets, sig_vals, sig_polys, sig_cosets, l0e, lle, l_active)


# ====================================================================================== SHPLONK helpers
def lagrange_interpolate(points: List[int], evals: List[int]) -> List[int]:
    """coefficients (low to high) of the polynomial of degree < len(points) through (points[i], evals[i])"""
    m = len(points)
    if m == 1:
        return [evals[0] % R]
    out = [0] * m
    for j in range(m):
        num, den = [1], 1
        for i in range(m):
            if i == j:
                continue
            # num *= (X - x_i)
            nxt = [0] * (len(num) + 1)
            for t, c in enumerate(num):
                nxt[t] = (nxt[t] - c * points[i]) % R
                nxt[t + 1] = (nxt[t + 1] + c) % R
            num = nxt
            den = den * (points[j] - points[i]) % R
        scale = evals[j] * O.inv_mod(den, R) % R
        for t, c in enumerate(num):
            out[t] = (out[t] + c * scale) % R
    return out


def eval_small(coeffs: List[int], x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def evaluate_vanishing_polynomial(roots: List[int], z: int) -> int:
    acc = 1
    for r_ in roots:
        acc = acc * (z - r_) % R
    return acc


def construct_intermediate_sets(queries):
    """poly/kzg/multiopen/shplonk.rs::construct_intermediate_sets [UPSTREAM-RECALL].  queries: list of (commitment key, point, eval) in
    query order.  Returns (rotation_sets, super_point_set): rotation_sets = [(points sorted ascending, [(key, [evals at those points])])]
    in order of first appearance; BTreeSet<Fr> order = numeric order of the canonical value."""
    super_points = sorted({p for _, p, _ in queries})
    get_eval = {}
    for key, p, e in queries:
        get_eval.setdefault((key, p), e)
    commitment_rotation = []                      # [(key, set(points))] in first-appearance order
    index = {}
    for key, p, _ in queries:
        if key in index:
            commitment_rotation[index[key]][1].add(p)
        else:
            index[key] = len(commitment_rotation)
            commitment_rotation.append((key, {p}))
    rot_sets = []                                 # [(frozenset(points), [keys])]
    for key, pts in commitment_rotation:
        fs = frozenset(pts)
        for entry in rot_sets:
            if entry[0] == fs:
                entry[1].append(key)
                break
        else:
            rot_sets.append((fs, [key]))
    out = []
    for fs, keys in rot_sets:
        pts = sorted(fs)
        out.append((pts, [(key, [get_eval[(key, p)] for p in pts]) for key in keys]))
    return out, super_points


# ====================================================================================== create_proof
class CountingRng:
    """Source of the prover's blinding scalars (`Fr::random(rng)` upstream; the RNG itself stays on the host side of the FFI).
    Deterministic SplitMix64 stream -> 512 bits -> mod r, so that the oracle and the HIP prover can be fed identical values."""

    def __init__(self, seed: int):
        self.sm = O.SplitMix64(seed)
        self.count = 0

    def next_fr(self) -> int:
        self.count += 1
        v = 0
        for i in range(8):
            v |= self.sm.next() << (64 * i)
        return v % R

    def fill(self, m: int) -> np.ndarray:
        return O.ints_to_limbs([self.next_fr() for _ in range(m)], R)


def create_proof(params: Params, pk: ProvingKey, advice: List[np.ndarray], instances: List[List[int]], rng, threads: int = 1, timings: dict = None) -> bytes:
    """plonk::create_proof for ONE circuit of the BaseConfig shape [UPSTREAM-RECALL, SURVEY.md §3.2].

    advice: one (n,4) array per advice column (gate columns, then lookup-advice columns), the witness as `assign_witnesses`
    lays it out (halo2-base/src/gates/flex_gate/threads/single_phase.rs:273-312); rows >= usable_rows are overwritten with blinding
    values.  instances: the public inputs per instance column (short lists).  Returns the proof bytes."""
    import time as _time

    sh = pk.vk.shape
    dom = Domain(sh)
    n, bf, u, T = sh.n, sh.blinding_factors, sh.usable_rows, threads
    tr = Blake2bWrite()
    t_last = [_time.perf_counter()]

    def lap(name):
        if os.environ.get("H2_ORACLE_TRACE") == "1":   # (the golden generator's large shapes: which step holds how much memory)
            import resource
            print("  oracle create_proof: %-28s %6.1f s, peak rss %.1f GB" % (name, _time.perf_counter() - t_last[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576), flush=True)
            if timings is None:
                t_last[0] = _time.perf_counter()
        if timings is not None:
            now = _time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - t_last[0]
            t_last[0] = now

    tr.common_scalar(pk.vk.transcript_repr)                    # vk.hash_into(transcript)
    # ---- instances (KZG: QUERY_INSTANCE = false -> values are hashed, not committed)
    assert len(instances) == sh.num_instance
    inst_values = []
    for vals in instances:
        assert len(vals) <= u, "InstanceTooLarge"
        for v in vals:
            tr.common_scalar(v)
        col = np.tile(fr1(0)[0], (n, 1))
        if len(vals):
            col[: len(vals)] = O.ints_to_limbs(list(vals), R)
        inst_values.append(col)
    inst_polys = [dom.lagrange_to_coeff(v, T) for v in inst_values]
    # ---- advice: blinding rows, commitments
    assert len(advice) == sh.num_advice_total
    adv_values = []
    for col in advice:
        col = np.array(col, dtype=np.uint64).reshape(n, 4)
        col[u:] = rng.fill(n - u)                               # rows unusable_rows_start.. <- Fr::random
        adv_values.append(col)
    for _ in adv_values:
        rng.next_fr()                                           # Blind(Fr::random) per column (KZG ignores the blind)
    lap("witness_blinding")
    for col in adv_values:
        tr.write_point(params.commit_lagrange(col, T))
    lap("commit_advice")
    theta = tr.squeeze_challenge()
    # ---- lookups: compress (single expression -> identity), permute, commit
    fixed_v, lookups = pk.fixed_values, []
    for (qcol, acol, tcol) in sh.lookups:
        inp = adv_values[acol] if qcol is None else CO.fr_mul_mt(fixed_v[qcol], adv_values[acol], T)
        tab = fixed_v[tcol]
        ap, sp = CO.permute_expression_pair(inp, tab, u)
        ap = np.concatenate([ap, rng.fill(bf + 1)])
        sp = np.concatenate([sp, rng.fill(bf + 1)])
        lap("lookup_permute")
        rng.next_fr()
        rng.next_fr()                                           # the two commit_values blinds
        tr.write_point(params.commit_lagrange(ap, T))
        tr.write_point(params.commit_lagrange(sp, T))
        lap("commit_lookup_permuted")
        lookups.append({"input": inp, "table": tab, "ap": ap, "sp": sp, "q": qcol, "a": acol, "t": tcol})
    beta = tr.squeeze_challenge()
    gamma = tr.squeeze_challenge()
    # ---- permutation grand products
    col_values = {"advice": adv_values, "fixed": fixed_v, "instance": inst_values}
    perm_z, last_z = [], 1
    deltaomega_start = 1
    for s0 in range(0, len(sh.perm_columns), sh.chunk_len):
        cols = sh.perm_columns[s0:s0 + sh.chunk_len]
        modified = None
        for j, (kind, idx) in enumerate(cols):
            t = CO.fr_lincomb(pk.sigma_values[s0 + j], fr1(beta), col_values[kind][idx], None, fr1(gamma), T)   # beta*sigma + gamma + value
            modified = t if modified is None else CO.fr_mul_mt(modified, t, T)
        modified = CO.fr_batch_invert_mt(modified, T)
        for (kind, idx) in cols:
            dw = CO.fr_geom(fr1(deltaomega_start * beta % R), fr1(dom.omega), n, T)                              # delta^j * omega^i * beta
            t = CO.fr_lincomb(dw, None, col_values[kind][idx], None, fr1(gamma), T)
            modified = CO.fr_mul_mt(modified, t, T)
            deltaomega_start = deltaomega_start * O.DELTA % R
        z = CO.fr_running_product(fr1(last_z), modified[: n - 1])                                                 # z[0] = last_z, n values
        z[n - bf:] = rng.fill(bf)
        last_z = to_int(z[n - bf - 1])
        rng.next_fr()                                           # blind
        lap("permutation_product")
        tr.write_point(params.commit_lagrange(z, T))
        lap("commit_permutation")
        perm_z.append(z)
    # ---- lookup grand products
    for lk in lookups:
        den = CO.fr_mul_mt(CO.fr_lincomb(lk["ap"], None, None, None, fr1(beta), T), CO.fr_lincomb(lk["sp"], None, None, None, fr1(gamma), T), T)
        den = CO.fr_batch_invert_mt(den, T)
        num = CO.fr_mul_mt(CO.fr_lincomb(lk["input"], None, None, None, fr1(beta), T), CO.fr_lincomb(lk["table"], None, None, None, fr1(gamma), T), T)
        prod = CO.fr_mul_mt(den, num, T)
        z = CO.fr_running_product(fr1(1), prod[: n - bf - 1])                                                     # n - bf values, z[0] = 1
        z = np.concatenate([z, rng.fill(bf)])
        rng.next_fr()                                           # blind
        lap("lookup_product")
        tr.write_point(params.commit_lagrange(z, T))
        lap("commit_lookup_product")
        lk["z"] = z
    # ---- vanishing argument: random polynomial
    random_poly = rng.fill(n)
    rng.next_fr()                                               # random_blind
    lap("witness_blinding")
    tr.write_point(params.commit(random_poly, T))
    lap("commit_random_poly")
    y = tr.squeeze_challenge()
    # ---- to coefficient form
    adv_polys = [dom.lagrange_to_coeff(v, T) for v in adv_values]
    perm_polys = [dom.lagrange_to_coeff(z, T) for z in perm_z]
    for lk in lookups:
        for name in ("ap", "sp", "z"):
            lk[name + "_poly"] = dom.lagrange_to_coeff(lk[name], T)
    lap("lagrange_to_coeff")
    # ---- evaluate_h on the extended domain
    ne = 1 << sh.extended_k
    adv_cosets = [dom.coeff_to_extended(p, T) for p in adv_polys]
    inst_cosets = [dom.coeff_to_extended(p, T) for p in inst_polys]
    lap("coeff_to_extended")
    Y = fr1(y)
    acc = np.tile(fr1(0)[0], (ne, 1))
    for (qcol, acol) in sh.gates:
        CO.quotient_gate(acc, pk.fixed_cosets[qcol], adv_cosets[acol], Y, dom.step, T)
    lap("quotient_gates")
    if perm_polys:
        perm_cosets = [dom.coeff_to_extended(p, T) for p in perm_polys]
        lap("coeff_to_extended")
        cos = {"advice": adv_cosets, "fixed": pk.fixed_cosets, "instance": inst_cosets}
        CO.quotient_permutation(acc, perm_cosets, [cos[kind][idx] for kind, idx in sh.perm_columns], pk.sigma_cosets, sh.chunk_len, pk.l0, pk.l_last,
                                pk.l_active, dom.step, -(bf + 1), fr1(beta), fr1(gamma), Y, fr1(O.DELTA), fr1(O.ZETA), fr1(dom.ext_omega), T)
        del perm_cosets
        lap("quotient_permutation")
    for lk in lookups:
        zc, apc, spc = (dom.coeff_to_extended(lk[name + "_poly"], T) for name in ("z", "ap", "sp"))
        lap("coeff_to_extended")
        inp = adv_cosets[lk["a"]] if lk["q"] is None else CO.fr_mul_mt(pk.fixed_cosets[lk["q"]], adv_cosets[lk["a"]], T)
        CO.quotient_lookup(acc, zc, inp, pk.fixed_cosets[lk["t"]], apc, spc, pk.l0, pk.l_last, pk.l_active, dom.step, fr1(beta), fr1(gamma), Y, T)
        del zc, apc, spc
        lap("quotient_lookup")
    del adv_cosets, inst_cosets
    # ---- vanishing.construct: h = numerator / (X^n - 1), split, commit
    CO.divide_by_vanishing(acc, sh.extended_k, sh.k, fr1(dom.ext_omega), fr1(O.ZETA), T)
    h = dom.extended_to_coeff(acc, T)[: n * sh.quotient_poly_degree]
    del acc
    lap("quotient_to_coeff")
    h_pieces = [h[i * n:(i + 1) * n] for i in range(sh.quotient_poly_degree)]
    for _ in h_pieces:
        rng.next_fr()                                           # h_blinds
    for piece in h_pieces:
        tr.write_point(params.commit(piece, T))
    lap("commit_h_pieces")
    x = tr.squeeze_challenge()
    xn = pow(x, n, R)
    # ---- evaluations
    ev = lambda poly, point: to_int(CO.fr_eval_polynomial(poly, fr1(point)))
    queries = []                                                # (key, point, eval) in upstream's query order
    polys = {}
    for (col, rot) in sh.advice_queries:
        point = dom.rotate_omega(x, rot)
        e = ev(adv_polys[col], point)
        tr.write_scalar(e)
        polys[("advice", col)] = adv_polys[col]
        queries.append((("advice", col), point, e))
    fixed_q = []
    for (col, rot) in sh.fixed_queries:
        point = dom.rotate_omega(x, rot)
        e = ev(pk.fixed_polys[col], point)
        tr.write_scalar(e)
        polys[("fixed", col)] = pk.fixed_polys[col]
        fixed_q.append((("fixed", col), point, e))
    # vanishing.evaluate: h(X) = sum_i xn^i h_i(X); random_eval
    h_poly = np.array(h_pieces[-1])
    for piece in reversed(h_pieces[:-1]):
        h_poly = CO.fr_axpy(piece, fr1(xn), h_poly, T)          # piece + xn * acc
    random_eval = ev(random_poly, x)
    tr.write_scalar(random_eval)
    polys[("h",)] = h_poly
    polys[("random",)] = random_poly
    # permutation: common sigma evals, then the sets
    sigma_q = []
    for j, p in enumerate(pk.sigma_polys):
        e = ev(p, x)
        tr.write_scalar(e)
        polys[("sigma", j)] = p
        sigma_q.append((("sigma", j), x, e))
    x_next, x_last, x_inv = dom.rotate_omega(x, 1), dom.rotate_omega(x, -(bf + 1)), dom.rotate_omega(x, -1)
    perm_q_a, perm_q_b = [], []
    for si, p in enumerate(perm_polys):
        polys[("perm_z", si)] = p
        e0, e1 = ev(p, x), ev(p, x_next)
        tr.write_scalar(e0)
        tr.write_scalar(e1)
        perm_q_a += [(("perm_z", si), x, e0), (("perm_z", si), x_next, e1)]
        if si != len(perm_polys) - 1:
            e2 = ev(p, x_last)
            tr.write_scalar(e2)
            perm_q_b.append((("perm_z", si), x_last, e2))
    perm_q = perm_q_a + list(reversed(perm_q_b))               # open(): sets at x, x_next; then sets.rev().skip(1) at x_last
    lookup_q = []
    for li, lk in enumerate(lookups):
        polys[("lk_z", li)], polys[("lk_a", li)], polys[("lk_s", li)] = lk["z_poly"], lk["ap_poly"], lk["sp_poly"]
        pe, pne = ev(lk["z_poly"], x), ev(lk["z_poly"], x_next)
        ae, aie, se = ev(lk["ap_poly"], x), ev(lk["ap_poly"], x_inv), ev(lk["sp_poly"], x)
        for e in (pe, pne, ae, aie, se):
            tr.write_scalar(e)
        lookup_q += [(("lk_z", li), x, pe), (("lk_a", li), x, ae), (("lk_s", li), x, se), (("lk_a", li), x_inv, aie), (("lk_z", li), x_next, pne)]
    lap("evaluations")
    queries = queries + perm_q + lookup_q + fixed_q + sigma_q + [(("h",), x, None), (("random",), x, random_eval)]
    # the prover never writes h(x); ProverQuery carries no eval: fill evals from the polynomials for the interpolation below
    queries = [(k_, p_, ev(polys[k_], p_) if e_ is None else e_) for (k_, p_, e_) in queries]
    shplonk_prove(params, tr, polys, queries, n, T)
    lap("multiopen_shplonk")
    return tr.finalize()


def _poly_sub_low(poly: np.ndarray, low: List[int]) -> np.ndarray:
    out = np.array(poly)
    out[: len(low)] = CO.fr_sub(out[: len(low)], O.ints_to_limbs(low, R))
    return out


def shplonk_prove(params: Params, tr, polys: dict, queries, n: int, T: int):
    """ProverSHPLONK::create_proof [UPSTREAM-RECALL poly/kzg/multiopen/shplonk/prover.rs]"""
    y = tr.squeeze_challenge()
    rotation_sets, super_points = construct_intermediate_sets(queries)
    v = tr.squeeze_challenge()
    ext_sets = []
    for points, commitments in rotation_sets:
        ext_sets.append((points, [(key, evals, lagrange_interpolate(points, evals)) for key, evals in commitments]))

    def div_by_vanishing(poly, roots):
        for r_ in roots:
            poly = CO.fr_kate_division(poly, fr1(r_))
        return poly

    # h(X) = sum_i v^i * ( sum_j y^j (P_ij(X) - R_ij(X)) ) / Z_i(X)
    h_x, vpow = None, 1
    for points, commitments in ext_sets:
        n_x, ypow = None, 1
        for key, evals, low in commitments:
            num = _poly_sub_low(polys[key], low)
            n_x = CO.fr_scale(num, fr1(ypow), T) if n_x is None else CO.fr_axpy(n_x, fr1(ypow), num, T)
            ypow = ypow * y % R
        q = div_by_vanishing(n_x, points)
        q = np.concatenate([q, np.zeros((n - len(q), 4), dtype=np.uint64)])
        h_x = CO.fr_scale(q, fr1(vpow), T) if h_x is None else CO.fr_axpy(h_x, fr1(vpow), q, T)
        vpow = vpow * v % R
    tr.write_point(params.commit(h_x, T))
    u = tr.squeeze_challenge()
    # linearisation: L(X) = sum_i v^i z_i sum_j y^j (P_ij(X) - R_ij(u)) - Z_T(u) h(X)
    l_x, vpow, z_diffs = None, 1, []
    for points, commitments in ext_sets:
        diffs = [p for p in super_points if p not in points]
        z_i = evaluate_vanishing_polynomial(diffs, u)
        z_diffs.append(z_i)
        inner, ypow = None, 1
        for key, evals, low in commitments:
            num = _poly_sub_low(polys[key], [eval_small(low, u)])
            inner = CO.fr_scale(num, fr1(ypow), T) if inner is None else CO.fr_axpy(inner, fr1(ypow), num, T)
            ypow = ypow * y % R
        l_x = CO.fr_scale(inner, fr1(vpow * z_i % R), T) if l_x is None else CO.fr_axpy(l_x, fr1(vpow * z_i % R), inner, T)
        vpow = vpow * v % R
    zt_eval = evaluate_vanishing_polynomial(super_points, u)
    l_x = CO.fr_axpy(l_x, fr1(-zt_eval % R), h_x, T)
    assert to_int(CO.fr_eval_polynomial(l_x, fr1(u))) == 0, "linearisation polynomial must vanish at u"
    h2 = div_by_vanishing(l_x, [u])
    h2 = CO.fr_scale(h2, fr1(O.inv_mod(z_diffs[0], R)), T)
    tr.write_point(params.commit(h2, T))


# ====================================================================================== verify_proof
class VerifyError(Exception):
    pass


def _l_i_range(dom: Domain, x: int, xn: int, lo: int, hi: int) -> List[int]:
    """EvaluationDomain::l_i_range: [l_i(x) for i in lo..=hi], l_i(x) = (x^n - 1) * omega^i / (n * (x - omega^i))"""
    n = dom.n
    out = []
    for rot in range(lo, hi + 1):
        wi = pow(dom.omega, rot % n, R)
        out.append((xn - 1) * wi % R * O.inv_mod(n * (x - wi) % R, R) % R)
    return out


def verify_proof(params: Params, vk: VerifyingKey, instances: List[List[int]], proof: bytes) -> bool:
    """plonk::verify_proof + VerifierSHPLONK + SingleStrategy [UPSTREAM-RECALL]: recomputes the quotient identity from the openings and checks the
    batched KZG opening with a real pairing.  Independent of the prover code above except for Shape and construct_intermediate_sets."""
    sh = vk.shape
    dom = Domain(sh)
    n, bf = sh.n, sh.blinding_factors
    tr = Blake2bRead(proof)
    try:
        tr.common_scalar(vk.transcript_repr)
        assert len(instances) == sh.num_instance
        for vals in instances:
            if len(vals) > sh.usable_rows:
                raise VerifyError("InstanceTooLarge")
            for v_ in vals:
                tr.common_scalar(v_)
        advice_comm = [tr.read_point() for _ in range(sh.num_advice_total)]
        theta = tr.squeeze_challenge()
        lk_perm_comm = [(tr.read_point(), tr.read_point()) for _ in sh.lookups]
        beta = tr.squeeze_challenge()
        gamma = tr.squeeze_challenge()
        perm_comm = [tr.read_point() for _ in range(sh.num_perm_sets)]
        lk_z_comm = [tr.read_point() for _ in sh.lookups]
        random_comm = tr.read_point()
        y = tr.squeeze_challenge()
        h_comm = [tr.read_point() for _ in range(sh.quotient_poly_degree)]
        x = tr.squeeze_challenge()
        advice_evals = [tr.read_scalar() for _ in sh.advice_queries]
        fixed_evals = [tr.read_scalar() for _ in sh.fixed_queries]
        random_eval = tr.read_scalar()
        sigma_evals = [tr.read_scalar() for _ in sh.perm_columns]
        perm_evals = []
        for si in range(sh.num_perm_sets):
            e0, e1 = tr.read_scalar(), tr.read_scalar()
            e2 = tr.read_scalar() if si != sh.num_perm_sets - 1 else None
            perm_evals.append((e0, e1, e2))
        lk_evals = [tuple(tr.read_scalar() for _ in range(5)) for _ in sh.lookups]
    except (ValueError, AssertionError) as e:
        raise VerifyError("malformed proof: %s" % e)
    xn = pow(x, n, R)
    # instance evaluations are computed by the verifier (QUERY_INSTANCE = false)
    max_inst = max([len(v_) for v_ in instances] + [0])
    l_i_s = _l_i_range(dom, x, xn, 0, max_inst - 1) if max_inst else []
    instance_evals = [sum(v_ * l for v_, l in zip(instances[col], l_i_s)) % R for (col, rot) in sh.instance_queries]
    l_evals = _l_i_range(dom, x, xn, -(bf + 1), 0)
    l_last, l_blind, l_0 = l_evals[0], sum(l_evals[1:1 + bf]) % R, l_evals[1 + bf]
    a_eval = {q: e for q, e in zip(sh.advice_queries, advice_evals)}
    f_eval = {q: e for q, e in zip(sh.fixed_queries, fixed_evals)}
    i_eval = {q: e for q, e in zip(sh.instance_queries, instance_evals)}
    col_eval = lambda kind, idx: {"advice": a_eval, "fixed": f_eval, "instance": i_eval}[kind][(idx, 0)]
    active = (1 - l_last - l_blind) % R
    exprs = []
    for (qcol, acol) in sh.gates:
        exprs.append(f_eval[(qcol, 0)] * (a_eval[(acol, 0)] + a_eval[(acol, 1)] * a_eval[(acol, 2)] - a_eval[(acol, 3)]) % R)
    if sh.num_perm_sets:
        exprs.append(l_0 * (1 - perm_evals[0][0]) % R)
        zl = perm_evals[-1][0]
        exprs.append(l_last * (zl * zl - zl) % R)
        for si in range(1, sh.num_perm_sets):
            exprs.append(l_0 * (perm_evals[si][0] - perm_evals[si - 1][2]) % R)
        for si in range(sh.num_perm_sets):
            cols = sh.perm_columns[si * sh.chunk_len:(si + 1) * sh.chunk_len]
            left, right = perm_evals[si][1], perm_evals[si][0]
            cur_delta = beta * x % R * pow(O.DELTA, si * sh.chunk_len, R) % R
            for j, (kind, idx) in enumerate(cols):
                left = left * (col_eval(kind, idx) + beta * sigma_evals[si * sh.chunk_len + j] + gamma) % R
            for (kind, idx) in cols:
                right = right * (col_eval(kind, idx) + cur_delta + gamma) % R
                cur_delta = cur_delta * O.DELTA % R
            exprs.append(active * (left - right) % R)
    for (qcol, acol, tcol), (pe, pne, ae, aie, se) in zip(sh.lookups, lk_evals):
        inp = a_eval[(acol, 0)] if qcol is None else f_eval[(qcol, 0)] * a_eval[(acol, 0)] % R
        tab = f_eval[(tcol, 0)]
        exprs.append(l_0 * (1 - pe) % R)
        exprs.append(l_last * (pe * pe - pe) % R)
        exprs.append(active * (pne * (ae + beta) % R * (se + gamma) - pe * (inp + beta) % R * (tab + gamma)) % R)
        exprs.append(l_0 * (ae - se) % R)
        exprs.append(active * (ae - se) % R * (ae - aie) % R)
    expected_h = 0
    for e in exprs:
        expected_h = (expected_h * y + e) % R
    expected_h = expected_h * O.inv_mod(xn - 1, R) % R
    # h commitment = sum_i xn^i H_i
    h_commitment = None
    for H in reversed(h_comm):
        h_commitment = O.g1_add(O.g1_mul(h_commitment, xn) if h_commitment is not None else None, H)
    # queries in the prover's order
    comm = {}
    queries = []
    for (col, rot), e in zip(sh.advice_queries, advice_evals):
        comm[("advice", col)] = advice_comm[col]
        queries.append((("advice", col), dom.rotate_omega(x, rot), e))
    x_next, x_last, x_inv = dom.rotate_omega(x, 1), dom.rotate_omega(x, -(bf + 1)), dom.rotate_omega(x, -1)
    tail = []
    for si, (e0, e1, e2) in enumerate(perm_evals):
        comm[("perm_z", si)] = perm_comm[si]
        queries += [(("perm_z", si), x, e0), (("perm_z", si), x_next, e1)]
        if e2 is not None:
            tail.append((("perm_z", si), x_last, e2))
    queries += list(reversed(tail))
    for li, (pe, pne, ae, aie, se) in enumerate(lk_evals):
        comm[("lk_z", li)], comm[("lk_a", li)], comm[("lk_s", li)] = lk_z_comm[li], lk_perm_comm[li][0], lk_perm_comm[li][1]
        queries += [(("lk_z", li), x, pe), (("lk_a", li), x, ae), (("lk_s", li), x, se), (("lk_a", li), x_inv, aie), (("lk_z", li), x_next, pne)]
    for (col, rot), e in zip(sh.fixed_queries, fixed_evals):
        comm[("fixed", col)] = vk.fixed_commitments[col]
        queries.append((("fixed", col), dom.rotate_omega(x, rot), e))
    for j, e in enumerate(sigma_evals):
        comm[("sigma", j)] = vk.permutation_commitments[j]
        queries.append((("sigma", j), x, e))
    comm[("h",)], comm[("random",)] = h_commitment, random_comm
    queries += [(("h",), x, expected_h), (("random",), x, random_eval)]
    # ---- VerifierSHPLONK
    rotation_sets, super_points = construct_intermediate_sets(queries)
    y2 = tr.squeeze_challenge()
    v = tr.squeeze_challenge()
    try:
        h1 = tr.read_point()
        u = tr.squeeze_challenge()
        h2 = tr.read_point()
    except (ValueError, AssertionError) as e:
        raise VerifyError("malformed proof: %s" % e)
    if not tr.exhausted():
        raise VerifyError("trailing bytes in proof")
    outer, r_outer, z_0, z_0_diff_inv, vpow = None, 0, 0, 0, 1
    for i, (points, commitments) in enumerate(rotation_sets):
        diffs = [p for p in super_points if p not in points]
        z_diff_i = evaluate_vanishing_polynomial(diffs, u)
        if i == 0:
            z_0 = evaluate_vanishing_polynomial(points, u)
            z_0_diff_inv = O.inv_mod(z_diff_i, R)
            z_diff_i = 1
        else:
            z_diff_i = z_diff_i * z_0_diff_inv % R
        inner, r_inner, ypow = None, 0, 1
        for key, evals in commitments:
            r_eval = ypow * eval_small(lagrange_interpolate(points, evals), u) % R
            inner = O.g1_add(inner, O.g1_mul(comm[key], ypow))
            r_inner = (r_inner + r_eval) % R
            ypow = ypow * y2 % R
        outer = O.g1_add(outer, O.g1_mul(inner, vpow * z_diff_i % R))
        r_outer = (r_outer + vpow * r_inner % R * z_diff_i) % R
        vpow = vpow * v % R
    g0 = O.limbs_to_points(params.g[:1])[0]
    outer = O.g1_add(outer, O.g1_mul(g0, -r_outer % R))
    outer = O.g1_add(outer, O.g1_mul(h1, -z_0 % R))
    outer = O.g1_add(outer, O.g1_mul(h2, u))
    # DualMSM::check: e(left, s_g2) * e(-right, g2) == 1  with left = h2, right = outer
    return PR.pairing_product_is_one([(h2, params.s_g2), (O.g1_neg(outer), params.g2)])
