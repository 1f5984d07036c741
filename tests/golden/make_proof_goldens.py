"""Generator of tests/golden/reference_shapes_proof_digests.json (TEST INFRASTRUCTURE; CPU only, no GPU library is loaded).

For every BaseCircuitParams shape of the reference's two benchmark sweeps (halo2-ecc/configs/secp256k1/bench_ecdsa.config and
halo2-ecc/configs/bn254/bench_pairing.config: 18 shapes, k = 11 ... 22) the ORACLE prover (oracle/plonk.py over oracle/h2_oracle.c)
makes its own SRS from a fixed toxic scalar (Params.setup: g[i] = s^i G and g_lagrange[i] = L_i(s) G point by point), builds the synthetic
halo2-base circuit of that shape (halo2_lib_amd/testing.py with the oracle's field arithmetic), runs keygen and create_proof with a
pre-drawn RNG stream, verifies the proof with the oracle verifier and records

    sha256(proof), len(proof), vk.transcript_repr, sha256(g), sha256(g_lagrange)

tests/test_reference_shapes_golden.py (-m gpu) then proves the same shapes with libh2hip on its own GPU-made SRS and compares digests: the
byte-for-byte comparison with the oracle prover at EVERY reference shape (VERDICT r03 "next" 1b) without minutes of CPU prover time on the
GPU box.  k = 22 takes several minutes here; the file is written shape by shape and finished shapes are skipped on a re-run.

usage: python tests/golden/make_proof_goldens.py [threads] [name-filter, e.g. ecdsa or pairing-22]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from halo2_lib_amd import testing as T   # noqa: E402  (pure numpy; importing the package does not load libh2hip)
from oracle import bn254 as O            # noqa: E402
from oracle import c_oracle as CO        # noqa: E402
from oracle import plonk as P            # noqa: E402
from tests.util import PreDrawnRng       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "reference_shapes_proof_digests.json")
TOXIC_S = 0x1D0C0FFEE1234567890ABCDEF0F1E2D3C4B5A69788796A5B4C3D2E1F
CIRCUIT_SEED = 20260926
RNG_SEED = 777
# (k, num_advice, num_lookup_advice, num_fixed, lookup_bits) — the config files' lines, in file order
PAIRING = [(14, 211, 27, 1, 13), (15, 105, 14, 1, 14), (16, 50, 6, 1, 15), (17, 25, 3, 1, 16), (18, 13, 2, 1, 17), (19, 6, 1, 1, 18),
           (20, 3, 1, 1, 19), (21, 2, 1, 1, 20), (22, 1, 1, 1, 21)]
ECDSA = [(19, 1, 1, 1, 18), (18, 2, 1, 1, 17), (17, 4, 1, 1, 16), (16, 8, 2, 1, 15), (15, 17, 3, 1, 14), (14, 34, 6, 1, 13),
         (13, 68, 12, 1, 12), (12, 139, 24, 2, 11), (11, 291, 53, 4, 10)]


# r05: the reference's other three benchmark files (README.md:297-305), every line in file order — halo2-ecc/configs/bn254/bench_msm.config:1-13,
# bench_fixed_msm.config:1-12 (num_fixed up to 7), bench_ec_add.config:1-5.  A line with num_lookup_advice = 0 configures NO lookup table at all
# (halo2-base/src/gates/circuit/mod.rs:76-78), whatever its lookup_bits says.
MSM = [(16, 170, 23, 1, 15), (17, 84, 11, 1, 16), (18, 42, 6, 1, 17), (19, 20, 3, 1, 18), (20, 11, 2, 1, 19), (21, 6, 1, 1, 20), (22, 3, 1, 1, 21),
       (23, 2, 1, 1, 22), (24, 1, 0, 1, 22), (19, 6, 1, 1, 18), (20, 6, 1, 1, 19), (21, 21, 3, 1, 20), (23, 6, 1, 1, 22)]
FIXED_MSM = [(17, 83, 9, 7, 16), (18, 42, 5, 4, 17), (19, 20, 2, 2, 18), (20, 10, 1, 1, 19), (21, 5, 1, 1, 20), (22, 3, 1, 1, 21), (23, 2, 1, 1, 22),
             (24, 1, 0, 1, 22), (19, 6, 1, 1, 18), (20, 6, 1, 1, 19), (21, 21, 3, 3, 20), (23, 6, 1, 1, 22)]
EC_ADD = [(15, 10, 2, 1, 14), (16, 5, 1, 1, 15), (17, 3, 1, 1, 16), (18, 2, 1, 1, 17), (19, 1, 0, 1, 18)]


def shapes():
    """the 18 shapes of the two sweeps the README times line by line (bench_ecdsa.config, bench_pairing.config)"""
    for name, lst in (("ecdsa", ECDSA), ("pairing", PAIRING)):
        for k, na, nl, nf, lb in lst:
            yield "%s-%d" % (name, k), (k, na, nl, nf, 0, lb)


def more_shapes():
    """(name, shape, alias_of) for every line of the other three benchmark files: name = <file>-L<line>; alias_of = the entry that already holds the
    same BaseCircuitParams (several lines of these files repeat a shape of another file or line), else None"""
    seen = {p: n for n, p in shapes()}
    for fname, lst in (("msm", MSM), ("fixed_msm", FIXED_MSM), ("ec_add", EC_ADD)):
        for line, (k, na, nl, nf, lb) in enumerate(lst, 1):
            name, p = "%s-L%d" % (fname, line), (k, na, nl, nf, 0, lb)
            yield name, p, seen.get(p)
            seen.setdefault(p, name)


def all_shapes():
    yield from shapes()
    for name, p, alias in more_shapes():
        if alias is None:
            yield name, p


class OracleBackend:
    mul = staticmethod(CO.fr_mul)
    add = staticmethod(CO.fr_add)


def rng_budget(sh):
    """Fr::random draws of one proof, with slack"""
    n, bf = sh.n, sh.blinding_factors
    return (sh.num_advice_total * (bf + 2) + len(sh.lookups) * (2 * (bf + 1) + 2 + bf + 1) + sh.num_perm_sets * (bf + 1) + n + 1 +
            sh.quotient_poly_degree + 16)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes() if not isinstance(a, (bytes, bytearray)) else a).hexdigest()


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    doc = {"what": "oracle-prover proof digests for the BaseCircuitParams shapes of the reference's five benchmark files; generator: tests/golden/make_proof_goldens.py",
           "toxic_s": hex(TOXIC_S), "circuit_seed": CIRCUIT_SEED, "rng_seed": RNG_SEED, "shapes": {}}
    if os.path.exists(OUT):
        old = json.load(open(OUT))
        if (old.get("toxic_s"), old.get("circuit_seed"), old.get("rng_seed")) == (doc["toxic_s"], CIRCUIT_SEED, RNG_SEED):
            doc["shapes"] = old["shapes"]
    g_full = None   # g[i] = s^i G does not depend on k: computed once at the largest k and sliced
    kcap = int(os.environ.get("GOLDEN_MAX_K", "24"))
    todo = [(name, p) for name, p in all_shapes() if flt in name and name not in doc["shapes"] and p[0] <= kcap]
    if not todo:
        print("nothing to do")
        return
    kmax = max(p[0] for _, p in todo)
    t0 = time.time()
    g_full = P.Params.setup(kmax, TOXIC_S, g_lagrange=np.zeros((1 << kmax, 8), dtype=np.uint64), threads=threads).g
    print("g for k=%d: %.1f s" % (kmax, time.time() - t0), flush=True)
    for name, (k, na, nl, nf, ni, lb) in sorted(todo, key=lambda t: t[1][0]):
        t0 = time.time()
        sh = P.Shape(k, na, nl, nf, ni, lb)
        params = P.Params.setup(k, TOXIC_S, g=g_full[: 1 << k], threads=threads)
        circ = T.build_circuit(sh, CIRCUIT_SEED + k, OracleBackend)
        asm = P.PermutationAssembly(sh)
        for l, r in circ.copies:
            asm.copy(l, r)
        if os.environ.get("H2_ORACLE_TRACE") == "1":
            import resource
            print("  %s: SRS + circuit + permutation assembly %.1f s, peak rss %.1f GB" % (name, time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576), flush=True)
        pk = P.keygen(params, sh, circ.fixed, asm, threads)
        if os.environ.get("H2_ORACLE_TRACE") == "1":
            print("  %s: keygen done %.1f s, peak rss %.1f GB" % (name, time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576), flush=True)
        inst = [O.limbs_to_ints(v, O.R_MOD) for v in circ.instances]
        proof = P.create_proof(params, pk, circ.advice, inst, PreDrawnRng(rng_budget(sh), RNG_SEED + k), threads)
        assert P.verify_proof(params, pk.vk, inst, proof), name
        doc["shapes"][name] = {"k": k, "num_advice": na, "num_lookup_advice": nl, "num_fixed": nf, "num_instance": ni, "lookup_bits": lb,
                               "proof_sha256": sha(proof), "proof_len": len(proof), "transcript_repr": hex(pk.vk.transcript_repr),
                               "g_sha256": sha(params.g), "g_lagrange_sha256": sha(params.g_lagrange),
                               "advice_sha256": sha(np.concatenate(circ.advice)), "verified_by_oracle_verifier": True}
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
        print("%s: %d bytes, %.1f s" % (name, len(proof), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
