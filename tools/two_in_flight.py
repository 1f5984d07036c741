"""GPU: create_proof THROUGHPUT with several proofs in flight — T host threads, each with its own libh2hip context, SRS tables and proving key of the same
circuit, each calling h2hip_plonk_create_proof in a loop (ctypes releases the GIL inside the call).  One proof alone leaves the chip under-used for about
half of its 13 ms (the Fiat-Shamir chain: every round's sorts, bucket reductions and pointwise kernels wait for the previous challenge); another proof's
accumulations fill those stretches.  Prints ms per proof (wall / proofs) for 1 .. T threads and checks that every proof has the same bytes.
usage: python tools/two_in_flight.py k num_advice num_lookup_advice num_fixed lookup_bits proofs_per_thread max_threads"""
import hashlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_lib_amd as H
from bench import _ShapeView
from halo2_lib_amd import halo2_proofs as HP
from halo2_lib_amd import plonk as PL
from halo2_lib_amd import testing as T

k, na, nl, nf, lb, per, tmax = [int(v) for v in sys.argv[1:8]]


class Prover:
    def __init__(self):
        self.ctx = ctx = H.Context()
        self.kzg = HP.ParamsKZG.setup(ctx, k, 0x1D0C0FFEE1234567890ABCDEF, precompute=True)
        bp = PL.BaseCircuitParams.new(k, na, nl, nf, 0, lb)
        sh = PL.shape_of(ctx, bp)

        class Backend:
            mul = staticmethod(ctx.fr_mul)
            add = staticmethod(ctx.fr_add)

        self.circ = T.build_circuit(_ShapeView(bp, sh), 5, Backend)
        self.pk = PL.keygen(self.kzg, bp, self.circ.fixed, self.circ.copies)
        self.adv = [ctx.to_device(np.ascontiguousarray(c)) for c in self.circ.advice]
        self.digests = set()

    def prove(self):
        p = PL.create_proof(self.pk, self.adv, self.circ.instances, PL.ChaChaRng(self.ctx.lib, 0, 12), advice_on_device=True)
        self.digests.add(hashlib.sha256(bytes(p)).hexdigest()[:16])


provers = [Prover() for _ in range(tmax)]
for p in provers:
    p.prove()
    p.prove()


def run(nthreads):
    barrier = threading.Barrier(nthreads + 1)

    def work(p):
        barrier.wait()
        for _ in range(per):
            p.prove()

    ths = [threading.Thread(target=work, args=(provers[i],)) for i in range(nthreads)]
    for t in ths:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    return (time.perf_counter() - t0) * 1e3


for rnd in range(3):
    for nt in range(1, tmax + 1):
        ms = run(nt)
        print("k=%d  %d proof(s) in flight: %.2f ms per proof (%d proofs in %.1f ms)" % (k, nt, ms / (nt * per), nt * per, ms), flush=True)
digests = set().union(*[p.digests for p in provers])
print("proof digests:", digests, "OK" if len(digests) == 1 else "MISMATCH")
