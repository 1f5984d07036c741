#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.

* poseidon_reference_kats.json — the ONLY golden vectors the reference holds for this path, transcribed from
  /root/reference/halo2-base/src/poseidon/hasher/tests/state.rs:29-33,55-61 (t=3 and t=5 permutation states) and
  tests/mod.rs:14-30 (MDS matrix of the t=3 spec).  They pin the oracle (tests/test_oracle.py).
* hotpath_small_cases.json — small inputs/outputs of every hot-path function produced by the pure-Python big-int
  oracle (oracle/bn254.py), so that the C restatement, the emulated kernels and the GPU library are all compared
  against committed bytes as well as against the live oracle.  The reference has no vectors for these functions
  ("parity unpinned"), so these are fixtures of the oracle, not of the reference.

usage (from the repo root):  python tests/golden/make_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bn254 as O  # noqa: E402

R = O.R_MOD


def hexes(vals):
    return [hex(v) for v in vals]


def main():
    out = {}
    # --- MSM: known-dlog bases, uniform + circuit-like + edge-case scalars
    n = 48
    pts = O.known_dlog_bases(n, 5, 3)
    sc = O.random_scalars(n - 6, 11) + [0, 1, R - 1, 0, 1, 2]
    res = O.msm_naive(sc, pts)
    out["msm"] = {"bases_xy": [[hex(p[0]), hex(p[1])] for p in pts], "scalars": hexes(sc), "result_xy": [hex(res[0]), hex(res[1])]}
    # --- NTT / iNTT / coset extension at k = 5 (extended_k = 7)
    k, ek = 5, 7
    a = O.random_scalars(1 << k, 21)
    w = O.omega_for(k)
    out["ntt"] = {"k": k, "omega": hex(w), "input": hexes(a), "best_fft": hexes(O.best_fft(list(a), w, k)),
                  "ifft": hexes(O.ifft(list(a), w, k)),
                  "coeff_to_extended": hexes(O.coeff_to_extended(list(a), k, ek)), "extended_k": ek}
    # --- pointwise
    x = O.random_scalars(1, 31)[0]
    out["poly"] = {"coeffs": hexes(a), "x": hex(x), "eval": hex(O.eval_polynomial(a, x)), "kate_division": hexes(O.kate_division(a, x)),
                   "batch_invert": hexes(O.batch_invert(a[:8] + [0])),
                   "grand_product": hexes(O.grand_product(a[:16], a[16:32]))}
    # --- lookup permutation
    u = 29
    table = [i % 8 for i in range(u)]
    inp = [(i * i) % 8 for i in range(u)]
    ap, sp = O.permute_expression_pair(inp, table)
    out["lookup_permute"] = {"input": inp, "table": table, "permuted_input": ap, "permuted_table": sp}
    json.dump(out, open(os.path.join(HERE, "hotpath_small_cases.json"), "w"), indent=1)

    kats = {
        "source": "halo2-base/src/poseidon/hasher/tests/state.rs:29-33,55-61 ; halo2-base/src/poseidon/hasher/tests/mod.rs:14-30",
        "t3": {"t": 3, "r_f": 8, "r_p": 57, "state_in": [0, 1, 2], "inputs": [0, 0],
               "mds": [["7511745149465107256748700652201246547602992235352608707588321460060273774987",
                        "10370080108974718697676803824769673834027675643658433702224577712625900127200",
                        "19705173408229649878903981084052839426532978878058043055305024233888854471533"],
                       ["18732019378264290557468133440468564866454307626475683536618613112504878618481",
                        "20870176810702568768751421378473869562658540583882454726129544628203806653987",
                        "7266061498423634438633389053804536045105766754026813321943009179476902321146"],
                       ["9131299761947733513298312097611845208338517739621853568979632113419485819303",
                        "10595341252162738537912664445405114076324478519622938027420701542910180337937",
                        "11597556804922396090267472882856054602429588299176362916247939723151043581408"]],
               "state_out": ["7853200120776062878684798364095072458815029376092732009249414926327459813530",
                             "7142104613055408817911962100316808866448378443474503659992478482890339429929",
                             "6549537674122432311777789598043107870002137484850126429160507761192163713804"]},
        "t5": {"t": 5, "r_f": 8, "r_p": 60, "state_in": [0, 1, 2, 3, 4], "inputs": [0, 0, 0, 0],
               "state_out": ["18821383157269793795438455681495246036402687001665670618754263018637548127333",
                             "7817711165059374331357136443537800893307845083525445872661165200086166013245",
                             "16733335996448830230979566039396561240864200624113062088822991822580465420551",
                             "6644334865470350789317807668685953492649391266180911382577082600917830417726",
                             "3372108894677221197912083238087960099443657816445944159266857514496320565191"]},
    }
    json.dump(kats, open(os.path.join(HERE, "poseidon_reference_kats.json"), "w"), indent=1)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
