// h2hip_comm — the exchange step of the multi-GPU prover (one process per GPU, one tiny exchange per commitment round, one
// all-gather of the quotient's cosets) inside libh2hip, so that a Rust / C++ host needs no collective library of its own:
//
//   * RCCL transport: librccl.so is dlopen'ed on first use (libh2hip does not link it: single-GPU users never load it), the communicator
//     is created from a 128-byte unique id that the host distributes over whatever channel it already has (h2hip_comm_rccl_unique_id on rank
//     0), and ncclAllGather runs on the context's stream over device buffers — xGMI peer-to-peer, no host staging;
//   * callback transport: the host supplies an all-gather over host memory (torch.distributed / gloo in the CPU tests, MPI, ...); device
//     payloads are staged through pinned host memory.  This is the fallback the world-2 tests run on the emulated kernels.
//
// The reference has no counterpart (its prover is single-process, rayon threads): the north star adds the 8-GPU split.
#include <dlfcn.h>

#include <mutex>

#include "internal.h"

namespace h2 {
// the handful of RCCL entry points used, with the types of rccl.h (ncclUniqueId is a 128-byte struct passed by value; ncclUint8 = 1)
struct NcclId {
    char internal[128];
};
typedef int (*nccl_get_unique_id_fn)(NcclId *);
typedef int (*nccl_comm_init_rank_fn)(void **, int, NcclId, int);
typedef int (*nccl_all_gather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_send_fn)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_recv_fn)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_group_fn)(void);
typedef int (*nccl_comm_destroy_fn)(void *);
typedef const char *(*nccl_get_error_string_fn)(int);
struct RcclApi {
    void *lib = nullptr;
    nccl_get_unique_id_fn get_unique_id = nullptr;
    nccl_comm_init_rank_fn comm_init_rank = nullptr;
    nccl_all_gather_fn all_gather = nullptr;
    nccl_send_fn send = nullptr;   // r06: the all-to-all (rows -> column owners) is a group of point-to-point transfers, one per peer link
    nccl_recv_fn recv = nullptr;
    nccl_group_fn group_start = nullptr, group_end = nullptr;
    nccl_comm_destroy_fn comm_destroy = nullptr;
    nccl_get_error_string_fn get_error_string = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;   // contexts of different threads may create their communicators at the same time
static int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return H2HIP_OK;
    const char *names[] = {getenv("H2HIP_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *nm : names)
        if (nm && *nm && (lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) {
        set_error("h2hip_comm: librccl.so could not be loaded (%s); set H2HIP_RCCL_LIBRARY or use the callback transport", dlerror());
        return H2HIP_ERR_INVALID;
    }
    RcclApi a;
    a.lib = lib;
    a.get_unique_id = (nccl_get_unique_id_fn)dlsym(lib, "ncclGetUniqueId");
    a.comm_init_rank = (nccl_comm_init_rank_fn)dlsym(lib, "ncclCommInitRank");
    a.all_gather = (nccl_all_gather_fn)dlsym(lib, "ncclAllGather");
    a.send = (nccl_send_fn)dlsym(lib, "ncclSend");
    a.recv = (nccl_recv_fn)dlsym(lib, "ncclRecv");
    a.group_start = (nccl_group_fn)dlsym(lib, "ncclGroupStart");
    a.group_end = (nccl_group_fn)dlsym(lib, "ncclGroupEnd");
    a.comm_destroy = (nccl_comm_destroy_fn)dlsym(lib, "ncclCommDestroy");
    a.get_error_string = (nccl_get_error_string_fn)dlsym(lib, "ncclGetErrorString");
    if (!a.get_unique_id || !a.comm_init_rank || !a.all_gather || !a.comm_destroy || !a.send || !a.recv || !a.group_start || !a.group_end) {
        set_error("h2hip_comm: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd / ncclCommDestroy");
        dlclose(lib);
        return H2HIP_ERR_INVALID;
    }
    g_rccl = a;
    return H2HIP_OK;
}
static int nccl_check(int rc, const char *what) {
    if (rc == 0) return H2HIP_OK;
    set_error("h2hip_comm: %s failed: %s (%d)", what, g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "?", rc);
    return H2HIP_ERR_HIP;
}
}  // namespace h2

struct h2hip_comm {
    int world = 1, rank = 0;
    // callback transport
    h2hip_allgather_fn cb = nullptr;
    void *cb_user = nullptr;
    // RCCL transport
    void *nccl = nullptr;
    int device = 0;
    // staging (callback transport: device payloads through pinned host memory; RCCL transport: host payloads through device memory)
    void *host_stage = nullptr, *dev_stage = nullptr;
    size_t host_cap = 0, dev_cap = 0;
};

using namespace h2;

static int comm_host_stage(h2hip_comm *c, size_t bytes) {
    if (c->host_cap >= bytes) return H2HIP_OK;
    if (c->host_stage) hipHostFree(c->host_stage);
    c->host_stage = nullptr;
    c->host_cap = 0;
    H2_HIPCHK(hipHostMalloc(&c->host_stage, bytes, 0));
    c->host_cap = bytes;
    return H2HIP_OK;
}
static int comm_dev_stage(h2hip_comm *c, size_t bytes) {
    if (c->dev_cap >= bytes) return H2HIP_OK;
    if (c->dev_stage) hipFree(c->dev_stage);
    c->dev_stage = nullptr;
    c->dev_cap = 0;
    H2_HIPCHK(hipMalloc(&c->dev_stage, bytes));
    c->dev_cap = bytes;
    return H2HIP_OK;
}

// every fallible preparation of a later h2hip_comm_allgather_dev(bytes per rank) — the callback transport's pinned staging area — done NOW, so
// that a caller can put it before the go-ahead exchange that precedes the collective (plonk.hip: the coset all-gather)
int h2::comm_reserve_allgather_dev(h2hip_comm *c, size_t bytes) {
    H2_REQUIRE(c, "NULL argument");
    if (c->nccl || !bytes) return H2HIP_OK;
    return comm_host_stage(c, bytes + bytes * (size_t)c->world);
}
// the same for a later h2hip_comm_alltoall_dev(bytes per peer)
int h2::comm_reserve_alltoall_dev(h2hip_comm *c, size_t bytes) {
    H2_REQUIRE(c, "NULL argument");
    if (c->nccl || !bytes) return H2HIP_OK;
    const size_t mine = bytes * (size_t)c->world;
    return comm_host_stage(c, mine + mine * (size_t)c->world);
}

extern "C" {

int h2hip_comm_rccl_unique_id(void *out128) {
    H2_REQUIRE(out128, "NULL argument");
    H2_CHK(load_rccl());
    NcclId id;
    H2_CHK(nccl_check(g_rccl.get_unique_id(&id), "ncclGetUniqueId"));
    memcpy(out128, id.internal, 128);
    return H2HIP_OK;
}

int h2hip_comm_init_rccl(h2hip_ctx *ctx, const void *unique_id128, int world, int rank, h2hip_comm **out) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && unique_id128 && out && world >= 1 && rank >= 0 && rank < world, "bad argument");
    H2_CHK(load_rccl());
    NcclId id;
    memcpy(id.internal, unique_id128, 128);
    void *nc = nullptr;
    H2_CHK(nccl_check(g_rccl.comm_init_rank(&nc, world, id, rank), "ncclCommInitRank"));   // collective: every rank of the group calls it
    h2hip_comm *c = new h2hip_comm();
    c->world = world;
    c->rank = rank;
    c->nccl = nc;
    c->device = ctx->device;
    *out = c;
    return H2HIP_OK;
}

int h2hip_comm_init_callback(int world, int rank, h2hip_allgather_fn allgather, void *user, h2hip_comm **out) {
    H2_REQUIRE(out && allgather && world >= 1 && rank >= 0 && rank < world, "bad argument");
    h2hip_comm *c = new h2hip_comm();
    c->world = world;
    c->rank = rank;
    c->cb = allgather;
    c->cb_user = user;
    *out = c;
    return H2HIP_OK;
}

int h2hip_comm_info(const h2hip_comm *comm, int *world, int *rank, int *is_rccl) {
    H2_REQUIRE(comm, "NULL argument");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    if (is_rccl) *is_rccl = comm->nccl ? 1 : 0;
    return H2HIP_OK;
}

void h2hip_comm_destroy(h2hip_comm *comm) {
    if (!comm) return;
    if (comm->nccl && g_rccl.comm_destroy) g_rccl.comm_destroy(comm->nccl);
    if (comm->host_stage) hipHostFree(comm->host_stage);
    if (comm->dev_stage) hipFree(comm->dev_stage);
    delete comm;
}

// recv_dev[r * bytes ...] = rank r's send_dev[0 .. bytes); ordered on the context's stream (send_dev is read and recv_dev written after the
// work already queued there).  RCCL: ncclAllGather on that stream, the call returns as soon as it is queued.  Callback: the stream is
// drained, the payload staged through pinned host memory.
int h2hip_comm_allgather_dev(h2hip_comm *comm, h2hip_ctx *ctx, const void *send_dev, size_t bytes, void *recv_dev) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(comm && ctx && (bytes == 0 || (send_dev && recv_dev)), "NULL argument");
    if (!bytes) return H2HIP_OK;
    if (comm->nccl) return nccl_check(g_rccl.all_gather(send_dev, recv_dev, bytes, /* ncclUint8 */ 1, comm->nccl, ctx->stream), "ncclAllGather");
    const size_t total = bytes * (size_t)comm->world;
    H2_CHK(comm_host_stage(comm, bytes + total));
    char *hs = (char *)comm->host_stage, *hr = hs + bytes;
    H2_HIPCHK(hipMemcpyAsync(hs, send_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    if (comm->cb(comm->cb_user, hs, bytes, hr) != 0) {
        set_error("h2hip_comm: the all-gather callback failed");
        return H2HIP_ERR_INVALID;
    }
    H2_HIPCHK(hipMemcpyAsync(recv_dev, hr, total, hipMemcpyHostToDevice, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by the next exchange
    return H2HIP_OK;
}

// recv_dev[p * bytes ...] = what rank p put at send_dev[me * bytes ...]: every rank hands every peer a DIFFERENT block (the rows of the columns
// that peer owns).  RCCL: one group of ncclSend / ncclRecv per peer on the context's stream — xGMI is point to point, every block takes its own
// link, and a rank receives 1 / world of what the matching all-gather would bring it.  Callback transport (tests): through the all-gather
// callback over the whole send buffers (world times the traffic: correctness only), staged through pinned host memory.
int h2hip_comm_alltoall_dev(h2hip_comm *comm, h2hip_ctx *ctx, const void *send_dev, size_t bytes, void *recv_dev) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(comm && ctx && (bytes == 0 || (send_dev && recv_dev)), "NULL argument");
    if (!bytes) return H2HIP_OK;
    const size_t W = (size_t)comm->world;
    if (comm->nccl) {
        H2_CHK(nccl_check(g_rccl.group_start(), "ncclGroupStart"));
        int rc = 0;
        for (size_t p = 0; p < W && rc == 0; ++p) {
            rc = g_rccl.send((const char *)send_dev + p * bytes, bytes, /* ncclUint8 */ 1, (int)p, comm->nccl, ctx->stream);
            if (rc == 0) rc = g_rccl.recv((char *)recv_dev + p * bytes, bytes, 1, (int)p, comm->nccl, ctx->stream);
        }
        const int rc_end = g_rccl.group_end();   // always closed, whatever a send / recv said
        H2_CHK(nccl_check(rc, "ncclSend / ncclRecv"));
        return nccl_check(rc_end, "ncclGroupEnd");
    }
    const size_t mine = bytes * W, total = mine * W;
    H2_CHK(comm_host_stage(comm, mine + total));
    char *hs = (char *)comm->host_stage, *hr = hs + mine;
    H2_HIPCHK(hipMemcpyAsync(hs, send_dev, mine, hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    if (comm->cb(comm->cb_user, hs, mine, hr) != 0) {
        set_error("h2hip_comm: the all-gather callback failed");
        return H2HIP_ERR_INVALID;
    }
    for (size_t p = 0; p < W; ++p)   // rank p's block for this rank
        H2_HIPCHK(hipMemcpyAsync((char *)recv_dev + p * bytes, hr + p * mine + (size_t)comm->rank * bytes, bytes, hipMemcpyHostToDevice, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by the next exchange
    return H2HIP_OK;
}

// the same for a (small) host payload — the 96-byte commitment partials of a round.  RCCL: staged through device memory on the context's stream.
int h2hip_comm_allgather_host(h2hip_comm *comm, h2hip_ctx *ctx, const void *send_host, size_t bytes, void *recv_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(comm && (bytes == 0 || (send_host && recv_host)), "NULL argument");
    if (!bytes) return H2HIP_OK;
    if (!comm->nccl) {
        if (comm->cb(comm->cb_user, send_host, bytes, recv_host) != 0) {
            set_error("h2hip_comm: the all-gather callback failed");
            return H2HIP_ERR_INVALID;
        }
        return H2HIP_OK;
    }
    H2_REQUIRE(ctx, "the RCCL transport needs a context (stream)");
    const size_t total = bytes * (size_t)comm->world;
    H2_CHK(comm_dev_stage(comm, bytes + total));
    char *ds = (char *)comm->dev_stage, *dr = ds + bytes;
    H2_HIPCHK(hipMemcpyAsync(ds, send_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    H2_CHK(nccl_check(g_rccl.all_gather(ds, dr, bytes, 1, comm->nccl, ctx->stream), "ncclAllGather"));
    H2_HIPCHK(hipMemcpyAsync(recv_host, dr, total, hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    return H2HIP_OK;
}

}  // extern "C"
