"""The sharded create_proof code path (h2hip_plonk_pk_set_sharding: point-range commitments through h2hip_comm exchanges, h(X)'s numerator by
cosets of the extended domain, the coset all-gather and interleave) run by ONE rank that owns every coset (H2HIP_SHARD_FORCE):

  * CPU suite: callback transport on the emulated kernels, proof bytes equal to the unsharded proof's;
  * GPU suite: the RCCL transport — libh2hip dlopen's librccl, creates a one-rank communicator from its own unique id and runs ncclAllGather
    on the context's stream (the 1-GPU box cannot host two ranks: RCCL refuses two ranks on one device) — plus the coset kernels on the
    real GPU at the k = 12 .. 16 sizes, and the coset helpers against their definition.

The multi-rank behaviour (world 2 and 3: uneven splits, failure protocol) is in tests/test_distributed_gloo.py."""
import ctypes as C

import numpy as np
import pytest

import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from halo2_lib_amd import plonk as PL
from halo2_lib_amd import testing as T
from oracle import bn254 as O
from oracle import c_oracle as CO
from oracle import plonk as P
from tests.util import PreDrawnRng, R, rand_fr

SHARD_QUOTIENT, SHARD_FORCE, SHARD_PRODUCTS, SHARD_NTT_COLUMNS = 1, 2, 4, 8


class _OracleBackend:
    mul = staticmethod(CO.fr_mul)
    add = staticmethod(CO.fr_add)


def _callback_comm(ctx):
    def _allgather(_user, local, nbytes, out):   # world 1: the gathered buffer is the local one
        C.memmove(out, local, nbytes)
        return 0

    cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)(_allgather)
    h = C.c_void_p()
    ctx._chk(ctx.lib.h2hip_comm_init_callback(1, 0, C.cast(cb, C.c_void_p), None, C.byref(h)))
    return h, cb


def _rccl_comm(ctx):
    uid = (C.c_uint8 * 128)()
    ctx._chk(ctx.lib.h2hip_comm_rccl_unique_id(uid))
    h = C.c_void_p()
    ctx._chk(ctx.lib.h2hip_comm_init_rccl(ctx.handle, uid, 1, 0, C.byref(h)))
    return h, None


def _prove_both_ways(ctx, shape, make_comm, precompute, seed=4):
    from tests.test_plonk_prover import _rng_budget

    sh = P.Shape(*shape)
    kzg = HP.ParamsKZG.setup(ctx, shape[0], 0xABCDEF0123 + seed, precompute=precompute)
    circ = T.build_circuit(sh, seed, _OracleBackend)
    pk = PL.keygen(kzg, PL.BaseCircuitParams.new(*shape), circ.fixed, circ.copies)
    budget = _rng_budget(sh)
    single = PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9))
    comm, keep = make_comm(ctx)
    n = 1 << shape[0]
    for flags in (SHARD_QUOTIENT | SHARD_PRODUCTS | SHARD_NTT_COLUMNS | SHARD_FORCE, SHARD_FORCE):   # every sharded stage / commitments, evaluations and SHPLONK only
        ctx._chk(ctx.lib.h2hip_plonk_pk_set_sharding(pk.handle, comm, kzg.g.handle, kzg.g_lagrange.handle, 0, n, flags))
        assert PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single, flags
    ctx._chk(ctx.lib.h2hip_plonk_pk_set_sharding(pk.handle, None, None, None, 0, 0, 0))
    assert PL.create_proof(pk, circ.advice, circ.instances, PreDrawnRng(budget, 9)) == single
    assert PL.verify_proof(pk, circ.instances, single)
    ctx.lib.h2hip_comm_destroy(comm)
    pk.free()
    kzg.free()
    del keep


def _coset_helpers(ctx):
    """f(X) -> f(sX), coset gather / interleave against their definitions (numpy / the C oracle's field arithmetic)"""
    n, log_c = 64, 2
    s = 0x1234567 % R
    cols = [rand_fr(n, 1), rand_fr(n, 2), rand_fr(n, 3)]
    d_in = [ctx.to_device(c) for c in cols]
    d_out = [ctx.malloc(32 * n) for _ in cols]
    outs = (C.c_void_p * 3)(*d_out)
    ins = (C.c_void_p * 3)(*d_in)
    sl = O.ints_to_limbs([s], R)
    ctx._chk(ctx.lib.h2hip_fr_coset_scale_batch_dev(ctx.handle, outs, ins, 3, n, sl.ctypes.data))
    pw = O.ints_to_limbs([pow(s, t, R) for t in range(n)], R)
    for c, d in zip(cols, d_out):
        assert np.array_equal(ctx.download(d, (n, 4)), CO.fr_mul(c, pw))
    full = rand_fr(n << log_c, 9)
    d_full = ctx.to_device(full)
    d_g = ctx.malloc(32 * n * 3)
    cosets = (C.c_uint32 * 3)(3, 0, 2)
    ctx._chk(ctx.lib.h2hip_fr_coset_gather_dev(ctx.handle, d_g, d_full, cosets, 3, log_c, n))
    got = ctx.download(d_g, (3 * n, 4))
    for m, c in enumerate((3, 0, 2)):
        assert np.array_equal(got[m * n:(m + 1) * n], full[c::4])
    d_all = ctx.to_device(np.concatenate([full[c::4] for c in (2, 0, 3, 1)]))
    slots = (C.c_uint32 * 4)(1, 3, 0, 2)      # coset c sits at position slots[c]
    d_back = ctx.malloc(32 * (n << log_c))
    ctx._chk(ctx.lib.h2hip_fr_coset_interleave_dev(ctx.handle, d_back, d_all, slots, log_c, n))
    assert np.array_equal(ctx.download(d_back, (n << log_c, 4)), full)
    for d in d_in + d_out + [d_full, d_g, d_all, d_back]:
        ctx.free(d)
    # extended_to_coeff by cosets: the size-n inverse transform of every coset (iNTT, then s_c^-t) + h2hip_fr_coset_combine_dev = the whole-domain
    # extended_to_coeff of the same evaluations, for 2, 4 and 8 cosets
    k = 6
    for log_c in (1, 2, 3):
        ek, nc = k + log_c, 1 << log_c
        omega, omega_e, zeta = O.omega_for(k), O.omega_for(ek), O.ZETA
        evals = rand_fr(n << log_c, 20 + log_c)
        want = ctx.extended_to_coeff(evals, ek, O.ints_to_limbs([O.inv_mod(omega_e, R)], R), O.ints_to_limbs([O.inv_mod(1 << ek, R)], R),
                                     O.ints_to_limbs([O.inv_mod(zeta, R)], R))
        order = list(range(nc))[::-1]                                  # coset c sits at position slots[c] of the gathered buffer
        parts = [None] * nc
        for c in range(nc):
            s_c = zeta * pow(omega_e, c, R) % R
            p_c = ctx.ifft(evals[c::nc], O.ints_to_limbs([O.inv_mod(omega, R)], R), k, O.ints_to_limbs([O.inv_mod(n, R)], R))
            parts[order[c]] = CO.fr_mul(p_c, O.ints_to_limbs([pow(O.inv_mod(s_c, R), t, R) for t in range(n)], R))
        d_p = ctx.to_device(np.concatenate(parts))
        d_h = ctx.malloc(32 * (n << log_c))
        slots = (C.c_uint32 * nc)(*order)
        rho_inv = O.ints_to_limbs([O.inv_mod(pow(omega_e, n, R), R)], R)
        zeta_n_inv = O.ints_to_limbs([O.inv_mod(pow(zeta, n, R), R)], R)
        ctx._chk(ctx.lib.h2hip_fr_coset_combine_dev(ctx.handle, d_h, d_p, slots, log_c, n, rho_inv.ctypes.data, zeta_n_inv.ctypes.data))
        assert np.array_equal(ctx.download(d_h, (n << log_c, 4)), want), log_c
        ctx.free(d_p)
        ctx.free(d_h)


@pytest.mark.parametrize("shape", [(6, 2, 1, 1, 1, 4), (6, 1, 1, 1, 0, 4)])
def test_sharded_path_single_rank_emulated(shape):
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        _prove_both_ways(ctx, shape, _callback_comm, precompute=False)
        if shape[1] == 2:
            _coset_helpers(ctx)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(12, 3, 1, 1, 1, 11), (14, 1, 1, 1, 0, 13), (16, 2, 1, 1, 0, 15)])
def test_sharded_path_single_rank_rccl_gpu(shape):
    ctx = H.Context()
    try:
        _prove_both_ways(ctx, shape, _rccl_comm, precompute=True)
        _prove_both_ways(ctx, shape, _callback_comm, precompute=True)
        _coset_helpers(ctx)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_rccl_transport_allgather_gpu():
    """h2hip_comm over RCCL (one rank): device and host all-gathers and the all-to-all (grouped ncclSend / ncclRecv) return the rank's own payload;
    the library reports itself as RCCL"""
    ctx = H.Context()
    try:
        comm, _ = _rccl_comm(ctx)
        world, rank, is_rccl = C.c_int(), C.c_int(), C.c_int()
        ctx._chk(ctx.lib.h2hip_comm_info(comm, C.byref(world), C.byref(rank), C.byref(is_rccl)))
        assert (world.value, rank.value, is_rccl.value) == (1, 0, 1)
        a = rand_fr(1 << 16, 5)
        d_a, d_b = ctx.to_device(a), ctx.malloc(a.nbytes)
        ctx._chk(ctx.lib.h2hip_comm_allgather_dev(comm, ctx.handle, d_a, a.nbytes, d_b))
        assert np.array_equal(ctx.download(d_b, a.shape), a)
        # r06: the all-to-all — one group of ncclSend / ncclRecv per peer (here: to itself) on the context's stream
        d_c = ctx.malloc(a.nbytes)
        ctx._chk(ctx.lib.h2hip_comm_alltoall_dev(comm, ctx.handle, d_a, a.nbytes, d_c))
        assert np.array_equal(ctx.download(d_c, a.shape), a)
        ctx.free(d_c)
        send = bytes(range(200))
        recv = C.create_string_buffer(200)
        ctx._chk(ctx.lib.h2hip_comm_allgather_host(comm, ctx.handle, send, 200, recv))
        assert recv.raw == send
        ctx.free(d_a)
        ctx.free(d_b)
        ctx.lib.h2hip_comm_destroy(comm)
    finally:
        ctx.close()
