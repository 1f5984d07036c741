"""GPU: h2hip_plonk_create_proof over every BaseCircuitParams shape of the reference's two benchmark sweeps
(halo2-ecc/configs/bn254/bench_pairing.config and halo2-ecc/configs/secp256k1/bench_ecdsa.config: degree, num_advice, num_lookup_advice,
num_fixed, lookup_bits), synthetic halo2-base circuits of those shapes (tools/prove_time.py).  Prints a markdown table.
usage: python tools/config_sweep.py [pairing|ecdsa|all] [reps]"""
import os
import re
import subprocess
import sys

PAIRING = [(14, 211, 27, 1, 13), (15, 105, 14, 1, 14), (16, 50, 6, 1, 15), (17, 25, 3, 1, 16), (18, 13, 2, 1, 17), (19, 6, 1, 1, 18),
           (20, 3, 1, 1, 19), (21, 2, 1, 1, 20), (22, 1, 1, 1, 21)]
ECDSA = [(19, 1, 1, 1, 18), (18, 2, 1, 1, 17), (17, 4, 1, 1, 16), (16, 8, 2, 1, 15), (15, 17, 3, 1, 14), (14, 34, 6, 1, 13),
         (13, 68, 12, 1, 12), (12, 139, 24, 2, 11), (11, 291, 53, 4, 10)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = sys.argv[2] if len(sys.argv) > 2 else "4"
here = os.path.dirname(os.path.abspath(__file__))
stages = None
rows = []
for name, shapes in (("pairing", PAIRING), ("ecdsa", ECDSA)):
    if which not in (name, "all"):
        continue
    for k, na, nl, nf, lb in shapes:
        out = subprocess.run([sys.executable, os.path.join(here, "prove_time.py"), str(k), str(na), str(nl), str(nf), "0", str(lb), reps],
                             capture_output=True, text=True).stdout
        times = [float(m) for m in re.findall(r"create_proof rep \d+: ([0-9.]+) ms", out)]
        size = re.findall(r"\((\d+) bytes\)", out)
        st = re.findall(r"^  (\S+)\s+([0-9.]+) ms$", out, flags=re.M)
        if not times:
            rows.append((name, k, na, nl, nf, lb, None, None, []))
            print(out[-400:], file=sys.stderr)
            continue
        stages = stages or [s for s, _ in st if s != "sum"]
        rows.append((name, k, na, nl, nf, lb, min(times[1:] or times), size[-1], [float(v) for s, v in st if s != "sum"]))
        print("%s k=%d advice=%d lookup=%d: %.2f ms" % (name, k, na, nl, rows[-1][6]), file=sys.stderr, flush=True)
short = {"advice_upload_blinding": "upload", "lookup_permute": "permute", "commit_advice_lookup_permuted": "commit 1", "grand_products": "products",
         "ntt_round1_columns_and_commit_products_random": "NTT + commit 2", "lagrange_to_coeff": "to coeff", "coeff_to_extended": "to ext",
         "quotient_terms": "quotient", "quotient_to_coeff": "h coeff", "commit_h_pieces": "commit h", "evaluations": "evals",
         "multiopen_shplonk": "SHPLONK"}
print("| sweep | k | advice | lookup advice | fixed | lookup bits | proof bytes | create_proof ms | " + " | ".join(short.get(s, s) for s in stages or []) + " |")
print("|---" * (8 + len(stages or [])) + "|")
for name, k, na, nl, nf, lb, ms, size, st in rows:
    if ms is None:
        print("| %s | %d | %d | %d | %d | %d | failed | | |" % (name, k, na, nl, nf, lb))
    else:
        print("| %s | %d | %d | %d | %d | %d | %s | **%.2f** | " % (name, k, na, nl, nf, lb, size, ms) + " | ".join("%.2f" % v for v in st) + " |")
