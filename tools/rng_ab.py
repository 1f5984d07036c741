"""GPU: create_proof (k = 19 ECDSA shape, advice resident) with the device ChaCha stream vs a pre-drawn array, alternating in one process: totals and stage laps"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP, plonk as PL, testing as T

args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
k = int(args_[0]) if args_ else 19
ctx = H.Context()
kzg = HP.ParamsKZG.setup(ctx, k, 0x1D0C0FFEE1234567890ABCDEF, precompute=True)
bp = PL.BaseCircuitParams.new(k, 1, 1, 1, 0, k - 1)
sh = PL.shape_of(ctx, bp)


class Backend:
    mul = staticmethod(ctx.fr_mul)
    add = staticmethod(ctx.fr_add)


class SV:
    pass


sv = SV()
sv.k, sv.n, sv.usable_rows, sv.num_advice, sv.lookup_bits = k, 1 << k, sh.usable_rows, 1, k - 1
sv.gate_advice, sv.lookup_advice = [0], list(range(1, sh.num_advice_total))
sv.table_col = sh.table_col if sh.table_col >= 0 else None
sv.constant_cols = [sh.first_constant_col]
sv.q_lookup_col = sh.q_lookup_col if sh.q_lookup_col >= 0 else None
sv.q_enable_cols = [sh.first_q_enable_col]
sv.num_fixed_total, sv.num_instance = sh.num_fixed_total, 0
circ = T.build_circuit(sv, 5, Backend)
pk = PL.keygen(kzg, bp, circ.fixed, circ.copies)
adv_dev = [ctx.to_device(c) for c in circ.advice]
n = 1 << k
vals = PL.ChaChaRng(ctx.lib, 0, 12).fill(n + 4096)
for mode in ("device", "array", "device", "array"):
    tot, stages = [], []
    for rep in range(8):
        rng = PL.ChaChaRng(ctx.lib, 0, 12) if mode == "device" else PL.ArrayRng(vals)
        tm = {} if rep >= 6 else None
        t = time.time()
        proof = PL.create_proof(pk, adv_dev, circ.instances, rng, tm, advice_on_device=True)
        tot.append((time.time() - t) * 1e3)
        if tm:
            stages.append(tm)
    print("%s: min %.2f median %.2f ms" % (mode, min(tot[1:6]), sorted(tot[1:6])[2]), flush=True)
    print("   " + " ".join("%s=%.2f" % (kk[:14], v) for kk, v in stages[-1].items()))

for opt in sys.argv[1:]:   # --ab=name:a,b: a context parameter alternated with the device generator (what the bench's timed loop uses)
    if opt.startswith("--ab="):
        import hashlib
        name, _, pair = opt[len("--ab="):].partition(":")
        lo, hi = [int(v) for v in (pair or "0,1").split(",")]
        for value in (lo, hi, lo, hi):
            ctx.set_param(name, value)
            tot = []
            for rep in range(8):
                t = time.time()
                proof = PL.create_proof(pk, adv_dev, circ.instances, PL.ChaChaRng(ctx.lib, 0, 12), None, advice_on_device=True)
                tot.append((time.time() - t) * 1e3)
            print("%s=%d: create_proof min %.2f median %.2f ms  sha256 %s" % (name, value, min(tot[1:]), sorted(tot[1:])[3], hashlib.sha256(bytes(proof)).hexdigest()[:16]),
                  flush=True)
