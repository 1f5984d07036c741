#!/usr/bin/env python3
"""Regenerates the [UPSTREAM-RECALL] table of INTEGRATION.md (between the `<!-- recall-table:begin/end -->` markers): every statement about
halo2-axiom 0.5.3 / halo2curves-axiom 0.7.3 that this repository restates from memory (the pinned sources are not in the container), with
the product line and the oracle line to change if upstream turns out to differ.  Each location is found by a unique anchor string, so the
table follows the code; tests/test_static_names.py::test_recall_table_is_current fails when the committed table is stale.

usage: python tools/recall_table.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, V, L, S = "halo2-lib_amd/csrc/plonk.hip", "halo2-lib_amd/csrc/verifier.hip", "halo2-lib_amd/csrc/lookup.hip", "halo2-lib_amd/csrc/srs.hip"
OP, OT, OB = "oracle/plonk.py", "oracle/transcript.py", "oracle/bn254.py"
HP, VR, PL = "halo2-lib_amd/halo2_proofs.py", "halo2-lib_amd/virtual_region.py", "halo2-lib_amd/plonk.py"
RG, OC = "halo2-lib_amd/csrc/rng.hip", "oracle/chacha.py"

# (item, what is assumed about upstream, [(file, anchor)] product, [(file, anchor)] oracle, how to flip)
ITEMS = [
    ("y-fold order of h(X)", "evaluate_h folds: every gate column's `q*(a+b*c-d)`, then the permutation argument (l_0(1-z_0), l_last(z_last^2-z_last), the chain terms, the product terms), then per lookup its five identities — `acc = acc*y + term`",
     [(P, "auto quotient_pass = [&]")], [(OP, "# ---- evaluate_h on the extended domain")],
     "reorder the three calls inside `quotient_pass` (and the job order inside `h2hip_quotient_permutation_sets_dev`, fr_ops.hip `items.push_back`) and the oracle's block alike; verifier: `verifier.hip` expression list"),
    ("order of the evaluations in the proof", "advice queries, fixed queries, random poly, sigma polys, permutation sets (z(x), z(wx), z(w^last x) except the last set), lookups (z(x), z(wx), a'(x), a'(w^-1 x), s'(x)); h(x) is NOT written",
     [(P, "std::vector<Query> evq;")], [(OP, "def create_proof(params: Params, pk: ProvingKey")],
     "permute the `want(...)` calls; the multiopen's query order is the separate `ask(...)` list just below"),
    ("`permute_expression_pair` tie-breaks", "A' = sorted input; the first row of each run takes its value from the table, leftover table values fill the repeated rows from the LAST repeated row backwards in ascending order (BTreeMap iteration, `repeated_input_rows.pop()`)",
     [(L, "upstream: BTreeMap iteration ascending, repeated_input_rows.pop()")], [(OB, "def permute_expression_pair(a, s):")],
     "`lk_assign_kernel`'s index arithmetic (lookup.hip) and the oracle function"),
    ("blinding rows", "`blinding_factors = max(3, max distinct rotations of an advice column) + 2 = 6`, usable rows = n - 7, the last 7 rows of every advice / permuted column and the last 6 of every grand product are `Fr::random`",
     [(P, "blinding_factors = std::max<uint32_t>(3, 4) + 2;")], [(OP, "self.blinding_factors = max(3, max_queries) + 2")],
     "the constant in `Shape::init` (both sides derive every row count from it)"),
    ("RNG draw order", "per advice column its 7 tail rows, then one (unused) blind per column; per lookup a' tail, s' tail, 2 blinds; per permutation set 6 tail rows + blind; per lookup z 6 tail rows + blind; n scalars of the vanishing argument's random polynomial + blind; one blind per h piece; SHPLONK none",
     [(P, "const Fr *tail = draw(n - u);")], [(OP, "class CountingRng:")],
     "the RNG is a callback (`h2hip_rng_fill_fn`): move the `draw(...)` calls of `create_proof_impl`; no kernel changes"),
    ("SHPLONK rotation sets", "`construct_intermediate_sets`: commitments grouped by their set of opening points in first-appearance order, `super_point_set` in first-appearance order; challenges y, v, then u after the first commitment",
     [(P, "static void construct_intermediate_sets(")], [(OP, "def construct_intermediate_sets(queries):")],
     "both functions (the verifier in verifier.hip shares the product's)"),
    ("compressed G1 flag bits", "32-byte little-endian x with sign(y) in bit 6 and the identity flag in bit 7 of byte 31",
     [(P, "static const unsigned SIGN_BIT = 6, INF_BIT = 7;"), (V, "static const unsigned SIGN_BIT = 6, INF_BIT = 7;")], [(OT, "SIGN_BIT, INF_BIT = 6, 7")],
     "the two constants (three places); SRS files in `Processed` encoding pass the positions as arguments (`h2hip_g1_decompress_batch_dev`)"),
    ("transcript framing", "Blake2b-512 personalised `Halo2-Transcript`; prefix bytes 0x01 point (x, y canonical LE), 0x02 scalar, 0x00 before a squeeze of a CLONE; challenge = 64-byte digest reduced mod r",
     [(P, "struct Transcript {   // Blake2bWrite")], [(OT, "def squeeze_challenge(self) -> int:")],
     "`Transcript` (plonk.hip) / `TranscriptRead` (verifier.hip) and oracle/transcript.py; the hash itself is pinned by RFC 7693 (tests/test_external_vectors.py)"),
    ("`vk.transcript_repr`", "an INPUT (`h2hip_plonk_pk_set_transcript_repr`): upstream hashes the Debug rendering of the pinned verifying key, which only Rust can produce; Python uses a stand-in of the same construction",
     [(P, "int h2hip_plonk_pk_set_transcript_repr("), (PL, "def transcript_repr(params: BaseCircuitParams")], [(OP, "def transcript_repr_for(shape: Shape")],
     "nothing in the library: the Rust shim passes `vk.transcript_repr()`"),
    ("selector compression", "`q_lookup` (complex selector) and every gate column's `q_enable` keep a fixed column of their own, created after the table and constants columns in that order",
     [(P, "if (single) q_lookup_col = nf++;"), (P, "static int check_selectors_stay_apart(")], [(OP, "self.q_lookup_col = nf")],
     "`Shape::init` column numbering; circuits whose selectors upstream would merge are rejected by keygen"),
    ("permutation `Assembly::copy`", "cycles merged smaller-into-larger through `mapping` / `aux` / `sizes`; sigma_i(w^j) = delta^i' w^j' for mapping[i][j] = (i', j'); columns in enable_equality order (constants, gate advice, lookup advice, instance)",
     [(P, "void copy(uint32_t lc, uint32_t lr, uint32_t rc, uint32_t rr) {")], [(OP, "def copy(self, left, right):")],
     "`Assembly::copy` and `PermutationAssembly.copy`"),
    ("degree and extended domain", "cs.degree() = max(3, lookup: 4 or 5) -> quotient_poly_degree = degree - 1 pieces, extended_k = k + ceil(log2(degree - 1)), coset generator `ZETA = 7^(2(r-1)/3)` with period-3 scaling",
     [(P, "degree = 3;   // gate and permutation argument"), (P, "static const uint64_t ZETA[4]")], [(OP, "self.degree = deg"), (OB, "ZETA = pow(MULT_GEN, 2 * (R_MOD - 1) // 3, R_MOD)")],
     "`Shape::init`; the constants are derived values (checked in SURVEY.md Appendix B), only their ROLE is recalled"),
    ("`Fr::random`", "`Fr::from_u512` of 64 bytes of the RNG's keystream, little-endian; `gen_srs` seeds ChaCha20 with 32 zero bytes",
     [(HP, "the `s` of `ParamsKZG::<Bn256>::setup(k, ChaCha20Rng::from_seed(Default::default()))`")], [],
     "`halo2_proofs.py` only (the prover's own randomness always comes through the callback)"),
    ("seeded RNG stream (`StdRng::seed_from_u64(0)`, `ChaCha20Rng::from_seed`)", "rand 0.8 `StdRng` = ChaCha12; keystream block b = ChaCha(key = the 32 seed bytes, 64-bit counter b in state words 12 / 13, stream id 0 in 14 / 15); `next_u64` = two consecutive words, low first; `Fr::random` = `from_u512` of eight `next_u64` = ONE 64-byte block per element; `seed_from_u64` = rand_core's PCG32 expansion (MUL 6364136223846793005, INC 11634580027462260723).  The block function itself is pinned to RFC 8439 (tests/test_rng_chacha.py)",
     [(RG, "H2_HD void chacha_block(const ChaChaKey &key, uint64_t counter, uint64_t stream, int rounds, uint32_t (&out)[16]) {"), (RG, "void h2hip_rng_seed_from_u64(uint64_t state, uint8_t *seed_out) {"), (P, "rng_ahead_pos = cr->pos + A * (uint64_t)(n - u)")],
     [(OC, "def chacha_blocks(seed: bytes, counters, rounds: int = 20, stream: int = 0) -> np.ndarray:"), (OC, "def seed_from_u64(state: int) -> bytes:")],
     "`chacha_block`'s state layout / `fr_from_block` (rng.hip) and oracle/chacha.py; a prover that keeps its own generator simply passes its own `h2hip_rng_fill_fn` and none of this is used"),
    ("SRS file layout", "`u32 k` LE, then g[0..n), g_lagrange[0..n), g2, s_g2 in `SerdeFormat::RawBytes` (Montgomery limbs) or `Processed` (compressed) encoding",
     [(HP, "def read(cls, ctx: Context, path: str, precompute: bool = True)")], [],
     "`ParamsKZG.read` / `.write` (host code); the device side validates whatever points it is given"),
    ("`constrain_instance` / `F: Ord`", "an instance copy is recorded as (advice cell, instance cell); constants are sorted by numeric value of the canonical representation",
     [(VR, 'region.constrain_equal(self.copy_manager.assigned_advices[inst.cell], (("instance", col), i))'), (VR, "copy_manager.constant_equalities.sort(key=lambda t: (t[0], t[1]))")], [],
     "`virtual_region.py` (and `host/halo2_proofs.hpp`); affects the sigma polynomials (verifying key), not validity"),
]


def locate(path, anchor):
    lines = open(os.path.join(ROOT, path)).read().split("\n")
    hits = [i + 1 for i, l in enumerate(lines) if anchor in l]
    if len(hits) != 1:
        raise SystemExit("anchor %r matches %d lines of %s" % (anchor, len(hits), path))
    return hits[0]


def render():
    out = ["| # | recalled statement | product: change here | oracle: change here | what to flip |", "|---|---|---|---|---|"]
    for n, (item, what, prod, orac, flip) in enumerate(ITEMS, 1):
        loc = lambda pairs: "<br>".join("`%s:%d` `%s`" % (p, locate(p, a), a.replace("|", "\\|")[:60]) for p, a in pairs) or "—"
        out.append("| %d | **%s** — %s | %s | %s | %s |" % (n, item, what.replace("|", "\\|"), loc(prod), loc(orac), flip.replace("|", "\\|")))
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(path).read()
    m = re.search(r"(<!-- recall-table:begin -->\n)(.*?)(\n<!-- recall-table:end -->)", doc, re.S)
    if not m:
        raise SystemExit("INTEGRATION.md lacks the recall-table markers")
    table = render()
    if "--check" in sys.argv:
        if m.group(2) != table:
            raise SystemExit("INTEGRATION.md: the [UPSTREAM-RECALL] table is stale — run python tools/recall_table.py")
        print("recall table is current (%d items)" % len(ITEMS))
        return
    open(path, "w").write(doc[:m.start(2)] + table + doc[m.end(2):])
    print("wrote %d items" % len(ITEMS))


if __name__ == "__main__":
    main()
