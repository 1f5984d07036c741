"""Race / nondeterminism detector: N create_proof calls on the same inputs and RNG stream must give byte-identical proofs (k = 19 ECDSA
configuration by default), interleaved with proofs of a second and a third key (a mid-size and a wide shape) sharing the context, and every 50th proof is verified."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP, plonk as PL, testing as T
from bench import _ShapeView, synthetic_scalars

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = H.Context()


class Backend:
    mul = staticmethod(ctx.fr_mul)
    add = staticmethod(ctx.fr_add)


def make(k, na, nl, nf, ni, lb, seed):
    kzg = HP.ParamsKZG.setup(ctx, k, 0xABCDEF + k, precompute=True)
    bp = PL.BaseCircuitParams.new(k, na, nl, nf, ni, lb)
    sh = PL.shape_of(ctx, bp)
    circ = T.build_circuit(_ShapeView(bp, sh), seed, Backend)
    return kzg, PL.keygen(kzg, bp, circ.fixed, circ.copies), circ


kzg1, pk1, c1 = make(19, 1, 1, 1, 0, 18, 1)
kzg2, pk2, c2 = make(16, 3, 1, 1, 1, 15, 2)
kzg3, pk3, c3 = make(13, 68, 12, 1, 0, 12, 3)   # a wide shape: every batched path (fused MSMs, column-batched NTTs, batched lookups / products / quotient)
d1, d2, d3 = synthetic_scalars((1 << 19) + 4096, 9), synthetic_scalars((1 << 16) + 4096, 10), synthetic_scalars((1 << 13) + 65536, 11)
ref1 = PL.create_proof(pk1, c1.advice, c1.instances, PL.ArrayRng(d1))
ref2 = PL.create_proof(pk2, c2.advice, c2.instances, PL.ArrayRng(d2))
ref3 = PL.create_proof(pk3, c3.advice, c3.instances, PL.ArrayRng(d3))
assert PL.verify_proof(pk1, c1.instances, ref1) and PL.verify_proof(pk2, c2.instances, ref2) and PL.verify_proof(pk3, c3.instances, ref3)
t = time.time()
bad = 0
for i in range(N):
    p1 = PL.create_proof(pk1, c1.advice, c1.instances, PL.ArrayRng(d1))
    p2 = PL.create_proof(pk2, c2.advice, c2.instances, PL.ArrayRng(d2)) if i % 3 == 0 else ref2
    p3 = PL.create_proof(pk3, c3.advice, c3.instances, PL.ArrayRng(d3)) if i % 2 == 0 else ref3
    if p1 != ref1 or p2 != ref2 or p3 != ref3:
        bad += 1
        print("MISMATCH at iteration", i, p1 != ref1, p2 != ref2, p3 != ref3, flush=True)
    if i % 50 == 49:
        assert PL.verify_proof(pk1, c1.instances, p1)
print("soak: %d iterations, %d mismatches, %.1f s" % (N, bad, time.time() - t))
sys.exit(1 if bad else 0)
