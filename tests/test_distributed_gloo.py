"""CPU-only, world_size 2 over gloo: the multi-GPU point-range sharding (halo2-lib_amd/multi_gpu.py) with the
emulated kernels standing in for the GPU.  Every rank must return the full MSM."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import halo2_lib_amd as H
        from halo2_lib_amd.multi_gpu import columns_for_rank, shard_range, sharded_msm, sharded_msm_batch
        from oracle import c_oracle as CO
        from tests.emu_util import emu_context
        from tests.util import circuit_like_fr, fr, jac_to_affine_ints, rand_fr
        from oracle import bn254 as O

        ctx = emu_context()
        bases = CO.known_dlog_bases(n, fr([99]), fr([5]))
        scal = np.concatenate([rand_fr(n // 2, 1), circuit_like_fr(n - n // 2, 2)])
        lo, hi = shard_range(n, rank, world)
        b = ctx.bases_upload(bases[lo:hi])
        ds = ctx.to_device(scal[lo:hi])
        got = sharded_msm(ctx, b, ds, hi - lo)
        want = CO.best_multiexp(scal, bases, threads=2)
        ok = [jac_to_affine_ints(got)] == O.limbs_to_points(want)
        got_aff = sharded_msm(ctx, b, ds, hi - lo, point_format=H.POINT_AFFINE)
        ok = ok and np.array_equal(got_aff, want)
        scal2 = rand_fr(n, 77)
        ds2 = ctx.to_device(scal2[lo:hi])
        gotb = sharded_msm_batch(ctx, b, [ds, ds2, ds], hi - lo)
        want2 = CO.best_multiexp(scal2, bases, threads=2)
        ok = ok and [jac_to_affine_ints(gotb[j]) for j in range(3)] == O.limbs_to_points(np.concatenate([want, want2, want]))
        # the same through libh2hip's own communicator (callback transport over gloo here, RCCL on a multi-GPU node) and its host-side sum
        from halo2_lib_amd.multi_gpu import Comm

        comm = Comm(ctx)
        got_c = sharded_msm(ctx, b, ds, hi - lo, comm=comm)
        ok = ok and [jac_to_affine_ints(got_c)] == O.limbs_to_points(want)
        ok = ok and np.array_equal(sharded_msm(ctx, b, ds, hi - lo, point_format=H.POINT_AFFINE, comm=comm), want)
        gotc = sharded_msm_batch(ctx, b, [ds, ds2, ds], hi - lo, comm=comm)
        ok = ok and [jac_to_affine_ints(gotc[j]) for j in range(3)] == O.limbs_to_points(np.concatenate([want, want2, want]))
        comm.destroy()
        ctx.free(ds2)
        cols = columns_for_rank(7, rank, world)
        q.put((rank, ok, cols))
        ctx.free(ds)
        b.free()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_sharded_msm_world2():
    from tests.emu_util import emu_context

    emu_context().close()   # build the emulated library once, before forking workers
    world, n = 2, 777
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert sorted(res[0][2] + res[1][2]) == list(range(7))


def test_shard_range_covers_everything():
    from halo2_lib_amd.multi_gpu import shard_range

    for n in (0, 1, 7, 8, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def _worker_prover(rank, world, port, q):
    """world-2 create_proof with point-range-sharded commitments + column-sharded NTTs on the emulated kernels"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from halo2_lib_amd import halo2_proofs as HP
        from halo2_lib_amd import plonk as PL
        from halo2_lib_amd import testing as T
        from halo2_lib_amd.multi_gpu import shard_proving_key, sharded_ntt_columns
        from oracle import c_oracle as CO
        from oracle import plonk as P
        from tests.emu_util import emu_context
        from tests.test_plonk_prover import _OracleBackend, _rng_budget
        from tests.util import PreDrawnRng, domain_consts, rand_fr

        ctx = emu_context()
        k, shape = 6, (6, 2, 1, 1, 1, 4)
        sh = P.Shape(*shape)
        kzg = HP.ParamsKZG.setup(ctx, k, 0xABCDEF0123, precompute=False)
        circ = T.build_circuit(sh, 4, _OracleBackend)
        pk = PL.keygen(kzg, PL.BaseCircuitParams.new(*shape), circ.fixed, circ.copies)
        budget = _rng_budget(sh)
        single = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9))
        sk = shard_proving_key(pk, ctx.bases_download(kzg.g), ctx.bases_download(kzg.g_lagrange), precompute=False)
        sharded = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9))
        sk.free()
        after = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9))
        ok_proof = sharded == single == after
        # column-sharded NTTs: 5 columns over 2 ranks, values checked against the oracle on every rank
        log_n = 7
        cols = [rand_fr(1 << log_n, 100 + j) for j in range(5)]
        w, winv, div = domain_consts(log_n)
        got = sharded_ntt_columns(ctx, cols, lambda c, p: c.ifft_dev(p, winv, log_n, div), gather=True)
        ok_ntt = all(np.array_equal(got[j], CO.ifft(cols[j], log_n, w)) for j in range(5))
        owned = sharded_ntt_columns(ctx, cols, lambda c, p: c.best_fft_dev(p, w, log_n))
        ok_owned = sorted(owned) == list(range(rank, 5, world)) and all(
            np.array_equal(ctx.download(d, (1 << log_n, 4)), CO.best_fft(cols[j], log_n, w)) for j, d in owned.items())
        for d in owned.values():
            ctx.free(d)
        # r06: the all-to-all primitive (callback transport here; grouped ncclSend / ncclRecv under RCCL): recv[p] = rank p's send[me]
        from halo2_lib_amd.multi_gpu import Comm

        comm = Comm(ctx)
        blk = 5
        send = np.array([[1000 * rank + 10 * p + i for i in range(blk)] for p in range(world)], dtype=np.uint64)
        d_send, d_recv = ctx.to_device(send), ctx.malloc(8 * blk * world)
        comm.alltoall_dev(d_send, 8 * blk, d_recv)
        ctx.sync()
        got_a2a = ctx.download(d_recv, (world, blk))
        ok_a2a = np.array_equal(got_a2a, np.array([[1000 * p + 10 * rank + i for i in range(blk)] for p in range(world)], dtype=np.uint64))
        ctx.free(d_send)
        ctx.free(d_recv)
        comm.destroy()
        q.put((rank, ok_proof, ok_ntt, ok_owned and ok_a2a, sharded.hex()[:32]))
        pk.free()
        kzg.free()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_sharded_create_proof_and_ntt_columns_world2():
    from tests.emu_util import emu_context

    emu_context().close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_prover, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] and r[2] and r[3] for r in res), res
    assert res[0][4] == res[1][4]   # both ranks emit the same proof


def _worker_scenarios(rank, world, port, q):
    """sharded create_proof beyond the plain case: uneven coset / point-range splits (world 3: four cosets dealt 2 + 1 + 1), the single-column
    q_lookup shape (degree 5: the product coset q_lookup * a), MSM-only sharding, and the failure protocol — a rank whose witness misses
    the lookup table, or whose RNG stream differs, makes EVERY rank return an error instead of leaving the others in a collective"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import halo2_lib_amd as H
        from halo2_lib_amd import halo2_proofs as HP
        from halo2_lib_amd import plonk as PL
        from halo2_lib_amd import testing as T
        from halo2_lib_amd.multi_gpu import shard_proving_key
        from oracle import bn254 as O
        from oracle import plonk as P
        from tests.emu_util import emu_context
        from tests.test_plonk_prover import _OracleBackend, _rng_budget
        from tests.util import PreDrawnRng

        ctx = emu_context()
        out = {}
        full = world == 2   # world 3 (uneven splits) runs the multi-column shape and the failure case only: the CPU suite's time budget
        for name, shape in (("multi", (6, 2, 1, 1, 1, 4)), ("single", (6, 1, 1, 1, 0, 4))) if full else (("multi", (6, 2, 1, 1, 1, 4)),):
            sh = P.Shape(*shape)
            kzg = HP.ParamsKZG.setup(ctx, shape[0], 0xABCDEF0123, precompute=False)
            circ = T.build_circuit(sh, 4, _OracleBackend)
            pk = PL.keygen(kzg, PL.BaseCircuitParams.new(*shape), circ.fixed, circ.copies)
            budget = _rng_budget(sh)
            single = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9))
            g, gl = ctx.bases_download(kzg.g), ctx.bases_download(kzg.g_lagrange)
            sk = shard_proving_key(pk, g, gl, precompute=False, shard_ntt_columns=True)   # every stage that can be sharded
            out[name] = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
            # r06: with lagrange_to_coeff dealt by column the grand products' rows travel to the columns' owners by an ALL-TO-ALL
            # (h2hip_comm_alltoall_dev; default) — and by r05's all-gather to every rank with plonk_route_rows = 0: same bytes either way
            ctx.set_param("plonk_route_rows", 0)
            out[name + "_allgather_rows"] = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
            ctx.set_param("plonk_route_rows", 1)
            if name == "multi":
                # the exchange schedule libh2hip reports for that proof (bench.py --gpus N prints it): 13 host exchanges with every stage sharded,
                # the status-only go-aheads at their places, the SHPLONK carry slots at a fixed 64 x 32 bytes
                import ctypes as C_
                cnt, sizes = C_.c_size_t(0), (C_.c_size_t * 32)()
                ctx._chk(ctx.lib.h2hip_plonk_pk_last_exchanges(pk.handle, sizes, 32, C_.byref(cnt)))
                sched = [int(sizes[i]) for i in range(cnt.value)]
                out["sched"] = (len(sched) == 13 and sched[0] == 72 and sched[3] == 0 and sched[5] == 0 and sched[6] == 0 and sched[9] == 64 * 32 and sched[11] == 32
                                and sched[10] == 96 and sched[12] == 96)
            if name == "multi":
                # a witness outside the lookup table on ONE rank: that rank reports the cause, the others H2HIP_ERR_PEER; nobody hangs
                adv = [np.array(c) for c in circ.advice]
                if rank == world - 1:
                    adv[sh.lookup_advice[0]][0] = O.ints_to_limbs([(1 << 4) + 3], O.R_MOD)[0]
                try:
                    PL.create_proof(pk, adv, circ.instances, PreDrawnRng(budget, 9))
                    out["fail"] = "no error"
                except H.H2HipError as e:
                    out["fail"] = e.code
                # the key still proves afterwards
                out["after_fail"] = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
                # ranks on different RNG streams: caught at the first exchange on every rank
                if not full:
                    out["rng"], out["after_rng"] = -1, True
                try:
                    if not full:
                        raise H.H2HipError(-1, "skipped")
                    PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9 + (rank == 0)))
                    out["rng"] = "no error"
                except H.H2HipError as e:
                    out["rng"] = e.code
                out["after_rng"] = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
            sk.free()
            if name == "multi" and not full:
                # a very uneven tiling: the last rank's rows are blinding rows only (no factor of any grand product), the first holds 5 rows
                n_ = 1 << shape[0]
                cuts = [0, 5, n_ - 3, n_]
                sk = shard_proving_key(pk, g, gl, precompute=False, point_range=(cuts[rank], cuts[rank + 1]))
                out["ragged"] = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
                sk.free()
            if name == "multi" and not full:
                out["msm_only"] = out["single"] = out["single_unsharded_again"] = True
            if name == "multi" and full:   # commitments only (h(X) replicated)
                sk = shard_proving_key(pk, g, gl, precompute=False, shard_quotient=False, shard_products=False)
                out["msm_only"] = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
                sk.free()
            out[name + "_unsharded_again"] = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
            pk.free()
            kzg.free()
        q.put((rank, out))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_create_proof_scenarios(world):
    from tests.emu_util import emu_context

    emu_context().close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker_scenarios, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert o["multi"] and o["single"] and o["msm_only"] and o["multi_unsharded_again"] and o["single_unsharded_again"], (r, o)
        assert o["after_fail"] and o["after_rng"] and o.get("ragged", True) and o["sched"], (r, o)
        assert o["multi_allgather_rows"] and o.get("single_allgather_rows", True), (r, o)   # r06: rows routed by all-to-all (default) == rows all-gathered
        assert o["fail"] == (-1 if r == world - 1 else -5), (r, o)     # H2HIP_ERR_INVALID where the witness is wrong, H2HIP_ERR_PEER elsewhere
        assert o["rng"] == -1, (r, o)                                   # every rank sees the mismatch in the hello exchange
