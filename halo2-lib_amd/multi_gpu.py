"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed ("nccl" = RCCL
over xGMI on ROCm; "gloo" in CPU tests).

* MSM — point-range sharding: rank g keeps bases[g*n/N .. (g+1)*n/N) resident and receives the matching
  scalar slice; it computes a full partial MSM locally (no data-path collective).  The only exchange is an
  all-gather of the N partial results (96 B Jacobian each — RCCL has no elliptic-curve reduction op), after
  which every rank sums the N points (h2hip_g1_sum_partials_host: all columns of a round in one host call; the exchange itself is
  libh2hip's `Comm`).
* NTT — independent columns are dealt round-robin to ranks (`columns_for_rank`); `sharded_ntt_columns` runs each rank's transforms
  on its own GPU and leaves the results there (where the matching commitments / quotient terms are computed), or, on request,
  all-gathers them so that every rank holds every transformed column.
* create_proof — `shard_proving_key` (h2hip_plonk_pk_set_sharding): all ranks run the same create_proof call; commitments are point-range
  sharded (one all-gather of 96-byte partials per round), h(X)'s numerator is evaluated by cosets of the extended domain (one
  device-to-device all-gather); the exchange is libh2hip's own (`Comm`: RCCL via dlopen, or a torch.distributed callback).
"""
from __future__ import annotations

import numpy as np

from .h2hip import POINT_JACOBIAN, Bases, Context


def shard_range(n: int, rank: int, world: int):
    """[lo, hi) of the point range owned by `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def columns_for_rank(num_columns: int, rank: int, world: int):
    return list(range(rank, num_columns, world))


def sharded_msm(ctx: Context, bases: Bases, scalars_dptr: int, n_local: int, group=None, device=None, point_format: int = POINT_JACOBIAN, comm=None) -> np.ndarray:
    """Partial MSM over this rank's slice, all-gather of the partials, local sum.  Every rank returns the full result.  With `comm` (a
    `Comm`: libh2hip's RCCL communicator or callback transport) the exchange and the sum run inside the library; without it, over
    torch.distributed tensors (kept for hosts that already hold a torch process group and no h2hip_comm)."""
    import torch
    import torch.distributed as dist

    part = ctx.msm_dev(bases, scalars_dptr, n_local, POINT_JACOBIAN)
    if comm is not None:
        return _gather_and_sum(ctx, comm, part, point_format)[0:1]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if point_format == POINT_JACOBIAN:
            return part
        t = torch.from_numpy(part.view(np.int64).copy()).reshape(1, 1, 12)
        t = t.to(device) if device is not None else t
        return _sum_gathered(ctx, t, 1, 1, point_format, device)
    world = dist.get_world_size(group)
    t = torch.from_numpy(part.view(np.int64).copy()).reshape(1, 12)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    allp = torch.cat(gathered, dim=0).contiguous()
    return _sum_gathered(ctx, allp, world, 1, point_format, device)[0:1]


def _sum_gathered(ctx: Context, allp, world: int, count: int, point_format: int, device) -> np.ndarray:
    """allp: (count, world, 12) int64 tensor of Jacobian partials -> (count, 8|12) sums computed on the GPU.
    A tensor gathered on the CPU (gloo) is staged into device memory first: the library only takes device pointers."""
    import torch

    width = 12 if point_format == POINT_JACOBIAN else 8
    out = np.zeros((count, width), dtype=np.uint64)
    staged = None
    if allp.device.type == "cpu":
        host = allp.numpy().view(np.uint64).reshape(-1)
        staged = ctx.to_device(host)
        base = staged
    else:
        torch.cuda.current_stream(device).synchronize()
        base = allp.data_ptr()
    try:
        stride = world * 12 * 8
        for j in range(count):
            out[j] = ctx.g1_sum_jacobian_dev(base + j * stride, world, point_format)[0]
    finally:
        if staged is not None:
            ctx.free(staged)
    return out


def _gather_and_sum(ctx: Context, comm, parts: np.ndarray, point_format: int) -> np.ndarray:
    """parts: (count, 12) Jacobian partials of this rank -> (count, 12 | 8) sums over all ranks: ONE all-gather through libh2hip's
    communicator (RCCL over xGMI, or the host callback) and ONE host-side summation call for all columns"""
    import ctypes as C

    count = len(parts)
    parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(count, 12)
    allp = np.zeros((comm.world, count, 12), dtype=np.uint64)
    ctx._chk(ctx.lib.h2hip_comm_allgather_host(comm.handle, ctx.handle, parts.ctypes.data, parts.nbytes, allp.ctypes.data))
    out = np.zeros((count, 12 if point_format == POINT_JACOBIAN else 8), dtype=np.uint64)
    ctx._chk(ctx.lib.h2hip_g1_sum_partials_host(allp.ctypes.data, comm.world, count, point_format, out.ctypes.data))
    return out


def sharded_msm_batch(ctx: Context, bases: Bases, scalar_dptrs, n_local: int, group=None, device=None, comm=None) -> np.ndarray:
    """`len(scalar_dptrs)` independent MSMs (e.g. all columns of a phase) over this rank's slice, pipelined on two
    streams; ONE all-gather carries all the partials (count x 96 B per rank); every rank returns all full results
    as a (count, 12) Jacobian array."""
    import torch
    import torch.distributed as dist

    count = len(scalar_dptrs)
    parts = ctx.msm_batch_dev(bases, scalar_dptrs, n_local, POINT_JACOBIAN)
    if comm is not None:   # libh2hip's own exchange: no torch tensors, no per-column device calls
        return parts if comm.world == 1 else _gather_and_sum(ctx, comm, parts, POINT_JACOBIAN)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return parts
    world = dist.get_world_size(group)
    t = torch.from_numpy(parts.view(np.int64).copy()).reshape(1, count, 12)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    allp = torch.cat(gathered, dim=0).permute(1, 0, 2).contiguous()   # (count, world, 12)
    return _sum_gathered(ctx, allp, world, count, POINT_JACOBIAN, device)


def sharded_ntt_columns(ctx: Context, columns, transform, group=None, gather: bool = False, device=None):
    """Column-sharded NTTs: `columns` is the full list of independent polynomials (host (n,4) arrays, every rank passes
    the same list); rank r transforms columns r, r+N, ... with `transform(ctx, dptr)` — any of the in-place `_dev` NTT entry points,
    e.g. `lambda c, p: c.ifft_dev(p, omega_inv, k, divisor)` — on its own GPU.  Returns {column index: device pointer} of the columns this
    rank owns (resident: the caller frees them with ctx.free), or with gather=True a list of ALL transformed columns as host arrays
    (one all-gather per column slot; ranks without a column in the last slot contribute a dummy)."""
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    mine = columns_for_rank(len(columns), rank, world)
    owned = {}
    for j in mine:
        col = np.ascontiguousarray(columns[j], dtype=np.uint64).reshape(-1, 4)
        d = ctx.to_device(col)
        transform(ctx, d)
        owned[j] = (d, col.shape)
    if not gather:
        ctx.sync()
        return {j: d for j, (d, _) in owned.items()}
    out = [None] * len(columns)
    slots = (len(columns) + world - 1) // world
    for slot in range(slots):
        j = slot * world + rank
        shape = np.ascontiguousarray(columns[min(j, len(columns) - 1)]).reshape(-1, 4).shape
        local = ctx.download(owned[j][0], owned[j][1]) if j in owned else np.zeros(shape, dtype=np.uint64)
        if not distributed:
            out[j] = local
            continue
        t = torch.from_numpy(local.view(np.int64).copy())
        if device is not None:
            t = t.to(device)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t, group=group)
        for r in range(world):
            jj = slot * world + r
            if jj < len(columns):
                out[jj] = gathered[r].cpu().numpy().view(np.uint64).reshape(-1, 4)
    for d, _ in owned.values():
        ctx.free(d)
    return out


class Comm:
    """h2hip_comm: the exchange step of the sharded prover inside libh2hip.  `rccl=True`: an RCCL communicator of the library's own
    (librccl is dlopen'ed by libh2hip; the 128-byte id travels from rank 0 over torch.distributed's object broadcast) — device-to-device
    all-gathers over xGMI on the context's stream.  Otherwise a callback transport over torch.distributed's all_gather on host tensors
    (gloo: the CPU tests; also usable with any backend)."""

    def __init__(self, ctx: Context, group=None, rccl: bool = False, device=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        self.ctx, self.handle, self._cb = ctx, C.c_void_p(), None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.world, self.rank = world, rank
        if rccl:
            uid = (C.c_uint8 * 128)()
            if rank == 0:
                ctx._chk(ctx.lib.h2hip_comm_rccl_unique_id(uid))
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group, device=device)
            ctx._chk(ctx.lib.h2hip_comm_init_rccl(ctx.handle, box[0], world, rank, C.byref(self.handle)))
            return

        def _allgather(_user, local, nbytes, out):
            try:
                buf = (C.c_uint8 * nbytes).from_address(local)
                t = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
                if device is not None:
                    t = t.to(device)
                gathered = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(gathered, t, group=group)
                flat = torch.cat(gathered).cpu().numpy()
                C.memmove(out, flat.ctypes.data, nbytes * world)
                return 0
            except BaseException:   # never unwind through the C frames
                return 1

        self._cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)(_allgather)
        ctx._chk(ctx.lib.h2hip_comm_init_callback(world, rank, C.cast(self._cb, C.c_void_p), None, C.byref(self.handle)))

    def allgather_dev(self, send_dptr: int, nbytes: int, recv_dptr: int):
        self.ctx._chk(self.ctx.lib.h2hip_comm_allgather_dev(self.handle, self.ctx.handle, send_dptr, nbytes, recv_dptr))

    def alltoall_dev(self, send_dptr: int, nbytes_per_peer: int, recv_dptr: int):
        """recv[p] = rank p's send[me] (blocks of nbytes_per_peer): grouped ncclSend / ncclRecv, or the callback transport"""
        self.ctx._chk(self.ctx.lib.h2hip_comm_alltoall_dev(self.handle, self.ctx.handle, send_dptr, nbytes_per_peer, recv_dptr))

    def destroy(self):
        if self.handle:
            self.ctx.lib.h2hip_comm_destroy(self.handle)
            self.handle = None


SHARD_QUOTIENT, SHARD_FORCE, SHARD_PRODUCTS, SHARD_NTT_COLUMNS = 1, 2, 4, 8


def decide_shard_ntt_columns(ctx: Context, comm: "Comm", k: int, num_columns: int, reps: int = 3):
    """Should lagrange_to_coeff be dealt by column (H2HIP_SHARD_NTT_COLUMNS)?  MEASURED on the machine the proof will run on (r05; r04 switched it
    on from 8 ranks): one 2^k inverse transform on this GPU against one all-gather of a 2^k-element column per rank through the communicator the
    proof will use (RCCL over xGMI, or the callback).  Dealing `num_columns` columns over N ranks costs ceil(C/N) transforms plus an all-gather of
    ceil(C/N) columns per rank; not dealing costs C transforms.  Every rank measures; the ranks exchange their figures and all take the SLOWEST
    rank's, so that all of them decide the same way (the exchange schedule depends on the decision).  Returns (decision, figures)."""
    import ctypes as C
    import time

    n, world = 1 << k, comm.world
    if world == 1 or num_columns == 0:
        return False, {"world": world}
    from .halo2_proofs import EvaluationDomain

    dom = EvaluationDomain(ctx, 4, k)
    d_col, d_all = ctx.to_device(np.zeros((n, 4), dtype=np.uint64)), ctx.malloc(32 * n * world)   # (zeros: valid field elements; the time does not depend on the values)
    try:
        ctx.ifft_dev(d_col, dom.omega_inv, k, dom.ifft_divisor)   # warm: twiddle tables
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.ifft_dev(d_col, dom.omega_inv, k, dom.ifft_divisor)
        ctx.sync()
        t_intt = (time.perf_counter() - t0) / reps
        comm.allgather_dev(d_col, 32 * n, d_all)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            comm.allgather_dev(d_col, 32 * n, d_all)
        ctx.sync()
        t_gather = (time.perf_counter() - t0) / reps
    finally:
        ctx.free(d_col)
        ctx.free(d_all)
    mine = np.array([t_intt, t_gather], dtype=np.float64)
    allv = np.zeros((world, 2), dtype=np.float64)
    ctx._chk(ctx.lib.h2hip_comm_allgather_host(comm.handle, ctx.handle, mine.ctypes.data, mine.nbytes, allv.ctypes.data))
    t_intt, t_gather = float(allv[:, 0].max()), float(allv[:, 1].max())
    per_rank = -(-num_columns // world)
    dealt, local = per_rank * (t_intt + t_gather), num_columns * t_intt
    return dealt < local, {"world": world, "k": k, "columns": num_columns, "intt_ms": t_intt * 1e3, "allgather_one_column_per_rank_ms": t_gather * 1e3,
                           "dealt_ms": dealt * 1e3, "local_ms": local * 1e3}


class ShardedKey:
    """keeps the shard base sets and the communicator of a sharded proving key alive"""

    def __init__(self, pk, g_shard, g_lagrange_shard, comm, own_comm=True):
        self.pk, self.g_shard, self.g_lagrange_shard, self.comm, self.own_comm = pk, g_shard, g_lagrange_shard, comm, own_comm

    def free(self):
        self.pk.ctx._chk(self.pk.ctx.lib.h2hip_plonk_pk_set_sharding(self.pk.handle, None, None, None, 0, 0, 0))
        self.g_shard.free()
        self.g_lagrange_shard.free()
        if self.own_comm:
            self.comm.destroy()


def shard_proving_key(pk, g_points: np.ndarray, g_lagrange_points: np.ndarray, group=None, device=None, precompute: bool = True,
                      rccl: bool = None, shard_quotient: bool = True, comm=None, shard_products: bool = True, point_range=None, shard_ntt_columns: bool = None) -> ShardedKey:
    """Shards a proving key's create_proof over the process group (h2hip_plonk_pk_set_sharding): this rank uploads ONLY its point range of
    the SRS (g_points / g_lagrange_points: the full (n, 8) affine arrays or anything sliceable that yields them) as base sets with their own
    window tables; h(X)'s numerator is evaluated by cosets of the extended domain (shard_quotient), the grand products by row range
    (shard_products); evaluations and SHPLONK's polynomial work always run on the rank's coefficient range.  point_range: this rank's (lo, hi) instead
    of the even split.  shard_ntt_columns: lagrange_to_coeff dealt by column with an all-gather of the coefficient forms (None: decided by
    `decide_shard_ntt_columns` — one transform timed against one column-per-rank all-gather through this communicator, slowest rank's figures).  The exchange runs inside libh2hip:
    over its own RCCL communicator when the process group's backend is nccl (rccl=None: decided from the backend), else over a
    torch.distributed callback (gloo on the CPU)."""
    import torch.distributed as dist

    from .h2hip import BASES_PLAIN, BASES_PRECOMPUTE

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = 1 << pk.params.k
    lo, hi = point_range if point_range is not None else shard_range(n, rank, world)   # (the ranks' ranges must tile [0, n): checked by the proof's first exchange)
    ctx = pk.ctx
    own_comm = comm is None
    if own_comm:
        if rccl is None:
            rccl = dist.get_backend(group) == "nccl"
        comm = Comm(ctx, group=group, rccl=rccl, device=device)
    ntt_decision = None
    if shard_ntt_columns is None:   # measured here, on this machine's links (r04: "from 8 ranks")
        sh = pk.shape
        shard_ntt_columns, ntt_decision = decide_shard_ntt_columns(ctx, comm, pk.params.k, sh.num_advice_total + 2 * sh.num_lookups)
    flags = BASES_PRECOMPUTE if precompute else BASES_PLAIN
    gs = ctx.bases_upload(np.ascontiguousarray(g_points[lo:hi]), flags)
    gls = ctx.bases_upload(np.ascontiguousarray(g_lagrange_points[lo:hi]), flags)
    ctx._chk(ctx.lib.h2hip_plonk_pk_set_sharding(pk.handle, comm.handle, gs.handle, gls.handle, lo, hi - lo,
                                                 (SHARD_QUOTIENT if shard_quotient else 0) | (SHARD_PRODUCTS if shard_products else 0) |
                                                 (SHARD_NTT_COLUMNS if shard_ntt_columns else 0)))
    sk = ShardedKey(pk, gs, gls, comm, own_comm)
    sk.shard_ntt_columns, sk.ntt_decision = bool(shard_ntt_columns), ntt_decision
    return sk
