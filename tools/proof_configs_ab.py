"""GPU: same-process A/B of whole create_proof calls over several PARAMETER SETS (each a comma-separated list of name=value context parameters read
at launch time), alternated `rounds` times: min / median ms per proof and the proof's sha256 (all sets must give the same bytes).
usage: python tools/proof_configs_ab.py k num_advice num_lookup_advice num_fixed lookup_bits rounds "name=v,name=v" "name=v" ...   ("-" = defaults)"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import _ShapeView
from halo2_lib_amd import halo2_proofs as HP
from halo2_lib_amd import plonk as PL
from halo2_lib_amd import testing as T

k, na, nl, nf, lb, rounds = [int(v) for v in sys.argv[1:7]]
sets = [[] if s == "-" else [(kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",")] for s in sys.argv[7:]]
ctx = H.Context()
kzg = HP.ParamsKZG.setup(ctx, k, 0x1D0C0FFEE1234567890ABCDEF, precompute=True)
bp = PL.BaseCircuitParams.new(k, na, nl, nf, 0, lb)
sh = PL.shape_of(ctx, bp)


class Backend:
    mul = staticmethod(ctx.fr_mul)
    add = staticmethod(ctx.fr_add)


circ = T.build_circuit(_ShapeView(bp, sh), 5, Backend)
pk = PL.keygen(kzg, bp, circ.fixed, circ.copies)
adv = [ctx.to_device(np.ascontiguousarray(c)) for c in circ.advice]
prove = lambda: PL.create_proof(pk, adv, circ.instances, PL.ChaChaRng(ctx.lib, 0, 12), advice_on_device=True)
defaults = {}
for s in sets:
    for name, _ in s:
        defaults.setdefault(name, ctx.get_param(name))
digests = set()
for rnd in range(rounds):
    for s in sets:
        for name, v in defaults.items():
            ctx.set_param(name, v)
        for name, v in s:
            ctx.set_param(name, v)
        prove()
        times = []
        for _ in range(8):
            t = time.perf_counter()
            proof = prove()
            times.append((time.perf_counter() - t) * 1e3)
        digests.add(hashlib.sha256(bytes(proof)).hexdigest()[:16])
        print("k=%d %-60s min %.2f median %.2f ms" % (k, ",".join("%s=%d" % kv for kv in s) or "(defaults)", min(times), sorted(times)[len(times) // 2]), flush=True)
print("proof digests:", digests, "OK" if len(digests) == 1 else "MISMATCH")
