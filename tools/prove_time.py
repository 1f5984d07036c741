"""GPU: time h2hip_plonk_create_proof for a BaseCircuitParams shape (default: the k=19 ECDSA configuration) with per-stage laps.
usage: python tools/prove_time.py [k] [num_advice] [num_lookup_advice] [num_fixed] [num_instance] [lookup_bits] [reps]
       [--register] [--verify] [--param=name=value ...] [--ab=name[:a,b]]   (--ab: same-run A/B of a context parameter, values a / b (default 0 / 1) alternated)"""
import sys
import time

import numpy as np

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from halo2_lib_amd import plonk as PL
from halo2_lib_amd import testing as T

a = [int(v) for v in sys.argv[1:] if not v.startswith('--')]
k, na, nl, nf, ni, lb, reps = (a + [19, 1, 1, 1, 0, 18, 5][len(a):])[:7]
ctx = H.Context()
for opt in sys.argv[1:]:
    if opt.startswith("--param="):
        name, value = opt[len("--param="):].split("=")
        ctx.set_param(name, int(value))


class Backend:
    mul = staticmethod(ctx.fr_mul)
    add = staticmethod(ctx.fr_add)


t = time.time()
kzg = HP.ParamsKZG.setup(ctx, k, 0x1D0C0FFEE1234567890ABCDEF, precompute=True)
bp = PL.BaseCircuitParams.new(k, na, nl, nf, ni, lb)
sh = PL.shape_of(ctx, bp)
print("srs %.2fs" % (time.time() - t), flush=True)


class ShapeView:   # what testing.build_circuit reads
    pass


sv = ShapeView()
sv.k, sv.n, sv.usable_rows, sv.num_advice, sv.lookup_bits = k, 1 << k, sh.usable_rows, na, lb
sv.gate_advice = list(range(na))
sv.lookup_advice = list(range(na, sh.num_advice_total))
sv.table_col = sh.table_col if sh.table_col >= 0 else None
sv.constant_cols = list(range(sh.first_constant_col, sh.first_constant_col + nf)) if nf else []
sv.q_lookup_col = sh.q_lookup_col if sh.q_lookup_col >= 0 else None
sv.q_enable_cols = list(range(sh.first_q_enable_col, sh.first_q_enable_col + na))
sv.num_fixed_total, sv.num_instance = sh.num_fixed_total, ni
t = time.time()
circ = T.build_circuit(sv, 5, Backend)
print("circuit %.2fs" % (time.time() - t), flush=True)
t = time.time()
pk = PL.keygen(kzg, bp, circ.fixed, circ.copies)
print("keygen %.2fs" % (time.time() - t), flush=True)
n = 1 << k
g = np.random.default_rng(1)
vals = g.integers(0, 2**63, size=(n + 4096 + 64 * (na + 4 * nl + 64), 4), dtype=np.uint64)   # blinding rows of every column + the random polynomial
vals[:, 3] &= np.uint64((1 << 60) - 1)
if "--register" in sys.argv:   # page-lock the advice columns (a prover keeps them across proofs)
    for c in circ.advice:
        ctx.host_register(c)
for rep in range(reps):
    tm = {}
    if rep == reps - 1:
        ctx.bench_modmul(1, 1)   # a kernel the prover never launches: tools/rocprof_proof.py takes what follows the last one as the proof
    t = time.time()
    proof = PL.create_proof(pk, circ.advice, circ.instances, PL.ArrayRng(vals), tm if rep == reps - 1 else None)
    dt = time.time() - t
    print("create_proof rep %d: %.2f ms (%d bytes)" % (rep, dt * 1e3, len(proof)), flush=True)
for opt in sys.argv[1:]:
    if opt.startswith("--ab="):
        import hashlib
        name, _, pair = opt[len("--ab="):].partition(":")
        lo, hi = [int(v) for v in (pair or "0,1").split(",")]
        for value in (lo, hi, lo, hi):
            ctx.set_param(name, value)
            times = []
            for rep in range(6):
                t = time.time()
                proof = PL.create_proof(pk, circ.advice, circ.instances, PL.ArrayRng(vals), None)
                times.append((time.time() - t) * 1e3)
            print("%s=%d: create_proof min %.2f median %.2f ms  sha256 %s" % (name, value, min(times[1:]), sorted(times[1:])[2],
                                                                             hashlib.sha256(bytes(proof)).hexdigest()[:16]), flush=True)
if "--verify" in sys.argv:
    t = time.time()
    print("h2hip_plonk_verify_proof:", PL.verify_proof(pk, circ.instances, proof), "%.1f ms" % ((time.time() - t) * 1e3), flush=True)
for name, ms in tm.items():
    print("  %-26s %8.3f ms" % (name, ms))
print("  %-26s %8.3f ms" % ("sum", sum(tm.values())))
