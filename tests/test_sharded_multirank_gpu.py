"""The sharded prover with MORE THAN ONE rank on the real HIP kernels (VERDICT r04 "next" 1a / 1c): 2 and 3 processes share GPU 0 — the 1-GPU
box cannot host two RCCL ranks, so the exchange is libh2hip's callback transport over a gloo process group, exactly what
`bench.py --gpus N --dist-backend gloo --share-device` runs — and prove the two BASELINE proof configurations

    ecdsa-19    halo2-ecc/configs/secp256k1/bench_ecdsa.config:1   (configs[3])
    pairing-21  halo2-ecc/configs/bn254/bench_pairing.config:8     (configs[4]: the shape north_star shards over 8 GPUs)

through h2hip_plonk_create_proof (the prover call of halo2-base/src/utils/testing.rs:32-50) with

  * EVERY stage sharded: commitments by point range, h(X)'s numerator and extended_to_coeff by coset, grand products by row range, evaluations and
    SHPLONK by coefficient range, and lagrange_to_coeff by column (`shard_ntt_columns=True`: the code path an 8-GPU run takes);
  * commitments (+ evaluations + SHPLONK) only;
  * a ragged point-range tiling (uneven slices),
  * the column-dealt lagrange_to_coeff switched by the MEASURED decision (multi_gpu.decide_shard_ntt_columns: one transform against one all-gather),

and EVERY rank must emit sha256(proof) == the digest the ORACLE prover alone produced for that shape
(tests/golden/reference_shapes_proof_digests.json, generator tests/golden/make_proof_goldens.py).  Each rank also reports the exchange schedule
libh2hip ran and that its communicator saw the expected world."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_shapes_proof_digests.json")


def _sha(a) -> str:
    return hashlib.sha256(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes()).hexdigest()


def _worker(rank, world, port, names, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    try:
        import ctypes as C

        import halo2_lib_amd as H
        from halo2_lib_amd import halo2_proofs as HP
        from halo2_lib_amd import plonk as PL
        from halo2_lib_amd import testing as T
        from halo2_lib_amd.multi_gpu import Comm, shard_proving_key, shard_range
        from oracle import plonk as P   # Shape only: the column bookkeeping of the synthetic circuit builder
        from tests.golden import make_proof_goldens as M
        from tests.util import PreDrawnRng

        doc = json.load(open(GOLDEN))["shapes"]
        rccl = os.environ.get("H2_MULTIRANK_RCCL") == "1"   # tools/rccl_cpx_try.sh: a partitioned GPU shows several logical devices, one rank on each
        ctx = H.Context(device=rank if rccl else 0)   # default: every rank on GPU 0

        class Backend:
            mul, add = staticmethod(ctx.fr_mul), staticmethod(ctx.fr_add)

        comm = Comm(ctx, rccl=rccl)   # default: callback transport over the gloo group; RCCL: the 128-byte id travels over the gloo group
        cw, cr, crccl = C.c_int(), C.c_int(), C.c_int()
        ctx._chk(ctx.lib.h2hip_comm_info(comm.handle, C.byref(cw), C.byref(cr), C.byref(crccl)))
        out["comm"] = (cw.value, cr.value, crccl.value)
        for name in names:
            e = doc[name]
            k, na, nl, nf, ni, lb = (e[f] for f in ("k", "num_advice", "num_lookup_advice", "num_fixed", "num_instance", "lookup_bits"))
            n = 1 << k
            sh = P.Shape(k, na, nl, nf, ni, lb)
            kzg = HP.ParamsKZG.setup(ctx, k, M.TOXIC_S, precompute=False)   # the full SRS only for keygen; a rank's slice gets the window tables
            circ = T.build_circuit(sh, M.CIRCUIT_SEED + k, Backend)
            pk = PL.keygen(kzg, PL.BaseCircuitParams.new(k, na, nl, nf, ni, lb), circ.fixed, circ.copies)
            res = {"vk": hex(pk.transcript_repr) == e["transcript_repr"]}
            g, gl = ctx.bases_download(kzg.g), ctx.bases_download(kzg.g_lagrange)
            rng = lambda: PreDrawnRng(M.rng_budget(sh), M.RNG_SEED + k)
            cuts = [0, 5] + [shard_range(n, r, world)[1] for r in range(1, world - 1)] + [n]   # rank 0 holds 5 rows
            cuts[-2] = n - 3 if world > 2 else cuts[-2]                                          # world 3: the last rank holds 3 blinding rows only
            for label, kw in (("all_stages", dict(shard_ntt_columns=True)),
                              ("commitments_only", dict(shard_quotient=False, shard_products=False, shard_ntt_columns=False)),
                              ("ragged", dict(shard_ntt_columns=True, point_range=(cuts[rank], cuts[rank + 1]))),
                              ("measured_decision", dict(shard_ntt_columns=None))):   # lagrange_to_coeff by column or not: timed on this machine, all ranks agree
                sk = shard_proving_key(pk, g, gl, precompute=True, comm=comm, **kw)
                proof = PL.create_proof(pk, circ.advice, circ.instances, rng())
                res[label] = _sha(proof) == e["proof_sha256"]
                if label == "measured_decision":
                    res["ntt_decision"] = (sk.shard_ntt_columns, sk.ntt_decision)
                if label == "all_stages":
                    cnt, sizes = C.c_size_t(0), (C.c_size_t * 32)()
                    ctx._chk(ctx.lib.h2hip_plonk_pk_last_exchanges(pk.handle, sizes, 32, C.byref(cnt)))
                    res["exchanges"] = [int(sizes[i]) for i in range(cnt.value)]
                    res["verified"] = bool(PL.verify_proof(pk, circ.instances, proof))
                sk.free()
            res["unsharded_again"] = _sha(PL.create_proof(pk, circ.advice, circ.instances, rng())) == e["proof_sha256"]
            out[name] = res
            pk.free()
            kzg.free()
        comm.destroy()
        ctx.close()
        q.put((rank, out))
    except BaseException as ex:   # the parent must hear about it instead of waiting for the queue
        q.put((rank, {"error": repr(ex)}))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_prover_multirank_shared_gpu_golden_digests(world):
    import torch.multiprocessing as mp

    names = ["ecdsa-19", "pairing-21"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, names, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=900) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in range(world):
        o = res[r]
        assert "error" not in o, (r, o)
        rccl = int(os.environ.get("H2_MULTIRANK_RCCL") == "1")
        assert o["comm"] == (world, r, rccl), (r, o)   # libh2hip's communicator: `world` ranks, this rank, callback transport (RCCL only under tools/rccl_cpx_try.sh)
        for name in names:
            e = o[name]
            assert e["vk"] and e["all_stages"] and e["commitments_only"] and e["ragged"] and e["measured_decision"] and e["unsharded_again"] and e["verified"], (r, name, e)
            assert e["ntt_decision"][0] == res[0][name]["ntt_decision"][0], (r, name, e["ntt_decision"])   # every rank took the same decision
            # every stage sharded incl. the column-dealt lagrange_to_coeff: 13 host exchanges, three of them status-only go-aheads
            assert len(e["exchanges"]) == 13 and e["exchanges"][0] == 72, (r, name, e["exchanges"])
    assert res[0]["ecdsa-19"]["exchanges"] == res[world - 1]["ecdsa-19"]["exchanges"]
    print("world %d: measured lagrange_to_coeff-by-column decisions: %r" % (world, {n: res[0][n]["ntt_decision"] for n in names}))
