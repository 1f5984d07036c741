// C ABI of libh2hip (declared in include/h2hip.h): context, memory, timers, and the host-buffer / device-
// pointer entry points that route to the kernels in ntt.hip, msm.hip and fr_ops.hip.
#include <stdarg.h>

#include <algorithm>

#include "internal.h"
#include <sched.h>

namespace h2 {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ws_reserve(h2hip_ctx *ctx, int slot, size_t bytes, void **out) {
    DevBuf &b = ctx->ws[slot];
    if (bytes == 0) bytes = 256;
    if (b.cap < bytes) {
        if (b.p) {
            H2_HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->clean_stream) H2_HIPCHK(hipStreamSynchronize(ctx->clean_stream));   // a pending zero-fill of this buffer ...
            if (ctx->lane[0]) H2_HIPCHK(hipStreamSynchronize(ctx->lane[0]->stream));      // ... (the batch's runs on its first lane's stream)
            H2_HIPCHK(hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        size_t cap = (bytes + 0xFFFFF) & ~(size_t)0xFFFFF;   // 1 MiB granules
        hipError_t e = hipMalloc(&b.p, cap);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) for workspace slot %d failed: %s", cap, slot, hipGetErrorString(e));
            b.p = nullptr;
            return H2HIP_ERR_NOMEM;
        }
        b.cap = cap;
    }
    *out = b.p;
    return H2HIP_OK;
}

// ---------------------------------------------------------------------------------------------- host round trips through a host-mapped flag (r05)
constexpr size_t POLL_BYTES = 16384;
__global__ __launch_bounds__(256) void publish_kernel(const uint32_t *__restrict__ src, uint32_t *dst_host, uint32_t nwords, unsigned long long *flag_host,
                                                      unsigned long long seq) {
    for (uint32_t i = threadIdx.x; i < nwords; i += 256) dst_host[i] = src[i];
#ifdef H2_HIPEMU
    __syncthreads();
    if (threadIdx.x == 0) *flag_host = seq;
#else
    __threadfence_system();   // every lane's payload stores are visible to the host before ...
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... the flag is
#endif
}
static int poll_init(h2hip_ctx *ctx) {
    if (ctx->poll_host) return H2HIP_OK;
    void *h = nullptr;
    H2_HIPCHK(hipHostMalloc(&h, 64 + POLL_BYTES, hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: the flag store is visible while the stream goes on
    memset(h, 0, 64 + POLL_BYTES);
    ctx->poll_host = (char *)h;
#ifdef H2_HIPEMU
    ctx->poll_dev = ctx->poll_host;
#else
    void *d = nullptr;
    H2_HIPCHK(hipHostGetDevicePointer(&d, h, 0));
    ctx->poll_dev = (char *)d;
#endif
    return H2HIP_OK;
}
int sync_results(h2hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!ctx->host_poll || bytes > POLL_BYTES || (bytes & 3) || ((uintptr_t)src_dev & 3)) {
        if (bytes) H2_HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));
        return H2HIP_OK;
    }
    H2_CHK(poll_init(ctx));
    const unsigned long long seq = ++ctx->poll_seq;
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t *)src_dev, (uint32_t *)(ctx->poll_dev + 64), (uint32_t)(bytes / 4),
                       (unsigned long long *)ctx->poll_dev, seq);
    H2_HIPCHK(hipGetLastError());
    const volatile unsigned long long *flag = (const volatile unsigned long long *)ctx->poll_host;
    for (uint64_t spins = 1; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
        if ((spins & 0x3FFFF) == 0) {   // every ~10 ms: a faulted stream would never raise the flag
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) break;   // (the stream has drained: the kernel's stores are complete)
            if (e != hipErrorNotReady) H2_HIPCHK(e);
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if (spins > 20000 && (spins & 0x3F) == 0) sched_yield();   // a wait of more than ~1 ms: leave the core to the other contexts' host threads
    }
    if (bytes) memcpy(dst_host, ctx->poll_host + 64, bytes);
    return H2HIP_OK;
}
int sync_stream(h2hip_ctx *ctx) { return sync_results(ctx, nullptr, ctx->poll_dev ? ctx->poll_dev + 64 : nullptr, 0); }

static hipEvent_t get_event(h2hip_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void prof_begin(h2hip_ctx *ctx, const char *name) {
    if (!ctx->profiling) return;
    if (!ctx->prof_filter.empty() && strncmp(name, ctx->prof_filter.c_str(), ctx->prof_filter.size()) != 0) return;
    hipEvent_t a = get_event(ctx), b = get_event(ctx);
    if (!a || !b) return;
    hipEventRecord(a, ctx->stream);
    ctx->pending.push_back({name, a, b, false});
}
// a bracket whose two events the LAUNCH itself records (hipExtLaunchKernelGGL's start / stop events: no marker packets of their own in the
// stream — the bench's timed region brackets every accumulation launch, and separate event records cost the proof ~0.3 ms).  false: not profiled
bool prof_launch_events(h2hip_ctx *ctx, const char *name, hipEvent_t *start, hipEvent_t *stop) {
    *start = *stop = nullptr;
    if (!ctx->profiling) return false;
    if (!ctx->prof_filter.empty() && strncmp(name, ctx->prof_filter.c_str(), ctx->prof_filter.size()) != 0) return false;
    hipEvent_t a = get_event(ctx), b = get_event(ctx);
    if (!a || !b) return false;
    ctx->pending.push_back({name, a, b, true});
    *start = a;
    *stop = b;
    return true;
}
// closes the innermost open bracket (brackets nest: the lookup permutation's bracket contains the scan's)
void prof_end(h2hip_ctx *ctx) {
    if (!ctx->profiling || !ctx->prof_filter.empty()) {
        // with a filter the brackets kept are leaf brackets of one name: close the open one of that name, if any
        if (ctx->profiling && !ctx->pending.empty() && !ctx->pending.back().ended) {
            hipEventRecord(ctx->pending.back().end, ctx->stream);
            ctx->pending.back().ended = true;
        }
        return;
    }
    for (size_t i = ctx->pending.size(); i-- > 0;)
        if (!ctx->pending[i].ended) {
            hipEventRecord(ctx->pending[i].end, ctx->stream);
            ctx->pending[i].ended = true;
            return;
        }
}
static void prof_collect(h2hip_ctx *ctx) {
    if (ctx->pending.empty()) return;
    hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->pending) {
        float ms = 0;
        if (p.ended && hipEventElapsedTime(&ms, p.begin, p.end) == hipSuccess) {
            KernelStat &s = ctx->stats[p.name];
            s.total_ms += ms;
            s.launches += 1;
            float t0 = 0;
            if (ctx->prof_ref && hipEventElapsedTime(&t0, ctx->prof_ref, p.begin) == hipSuccess) s.spans.push_back({t0, t0 + ms});
        }
        ctx->event_pool.push_back(p.begin);
        ctx->event_pool.push_back(p.end);
    }
    (void)hipGetLastError();   // a failed elapsed-time query must not surface as the next launch's error
    ctx->pending.clear();
}

static void prof_collect_all(h2hip_ctx *ctx);
// a child context's (MSM lane, the prover's side stream) kernel timers into the parent's table
void prof_fold_child(h2hip_ctx *parent, h2hip_ctx *child) {
    prof_collect(child);
    for (auto &kv : child->stats) {
        parent->stats[kv.first].total_ms += kv.second.total_ms;
        parent->stats[kv.first].launches += kv.second.launches;
        parent->stats[kv.first].spans.insert(parent->stats[kv.first].spans.end(), kv.second.spans.begin(), kv.second.spans.end());
    }
    child->stats.clear();
}
// this context's timers and its lanes': the lanes' are folded when the table is READ, not after every batch (the elapsed-time queries of a
// proof's bracketed launches are host time inside the bench's timed region otherwise)
static void prof_collect_all(h2hip_ctx *ctx) {
    prof_collect(ctx);
    for (h2hip_ctx *l : ctx->lane)
        if (l) prof_fold_child(ctx, l);
}

__global__ void point_finish_kernel(const XYZZ *__restrict__ in, G1Jac *__restrict__ jac, G1Affine *__restrict__ aff) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        XYZZ p = in[0];
        if (jac) jac[0] = xyzz_to_jacobian(p);
        if (aff) aff[0] = xyzz_to_affine(p);
    }
}

// workgroup b: in[b] -> slot first_slot + b of the result array
__global__ void point_finish_slot_kernel(const XYZZ *__restrict__ in, G1Jac *__restrict__ jac, G1Affine *__restrict__ aff, uint32_t first_slot) {
    if (threadIdx.x == 0) {
        XYZZ p = in[blockIdx.x];
        if (jac) jac[first_slot + blockIdx.x] = xyzz_to_jacobian(p);
        if (aff) aff[first_slot + blockIdx.x] = xyzz_to_affine(p);
    }
}

// sum of n Jacobian points (multi-GPU partial results): one workgroup, strided accumulate + LDS tree
__global__ __launch_bounds__(64) void jac_sum_kernel(const G1Jac *__restrict__ pts, uint32_t n, XYZZ *__restrict__ out) {
    __shared__ XYZZ sh[64];
    uint32_t tid = threadIdx.x;
    XYZZ acc = XYZZ::identity();
    for (uint32_t i = tid; i < n; i += 64) {
        G1Jac p = pts[i];
        if (p.z.is_zero()) continue;
        XYZZ q;
        q.x = p.x;
        q.y = p.y;
        q.zz = fe_sqr(p.z);
        q.zzz = fe_mul(q.zz, p.z);
        xyzz_add(acc, q);
    }
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        if (tid < d) {
            XYZZ a = sh[tid];
            xyzz_add(a, sh[tid + d]);
            sh[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) out[0] = sh[0];
}

}  // namespace h2

using namespace h2;

extern "C" {

const char *h2hip_last_error(void) { return g_err; }
int h2hip_version(void) { return 100; }

int h2hip_device_count(int *count) {
    H2_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return H2HIP_ERR_NO_DEVICE;
    }
    *count = n;
    return H2HIP_OK;
}

int h2hip_init(int device, void *hip_stream, h2hip_ctx **out) {
    H2_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available (libh2hip has no CPU fallback)");
        return H2HIP_ERR_NO_DEVICE;
    }
    H2_REQUIRE(device >= 0 && device < n, "device index out of range");
    H2_HIPCHK(hipSetDevice(device));
    h2hip_ctx *ctx = new h2hip_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete ctx;
            return H2HIP_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return H2HIP_OK;
}

}  // extern "C"
namespace h2 {
// Small host tables (job descriptors) -> device without a synchronisation: the bytes are copied into a pinned ring first, so the caller's
// buffer may go away at once and the asynchronous copy has a stable source.  The ring only wraps after a stream synchronisation.
constexpr size_t JOB_RING_BYTES = (size_t)1 << 20;
int upload_jobs(h2hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!bytes) return H2HIP_OK;
    if (bytes > JOB_RING_BYTES / 4) {   // large tables: plain (staged, host-synchronous) copy
        H2_HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));
        return H2HIP_OK;
    }
    if (!ctx->job_ring) H2_HIPCHK(hipHostMalloc((void **)&ctx->job_ring, JOB_RING_BYTES, 0));
    const size_t need = (bytes + 255) / 256 * 256;
    if (ctx->job_ring_off + need > JOB_RING_BYTES) {
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));   // everything staged so far has been consumed
        ctx->job_ring_off = 0;
    }
    char *slot = ctx->job_ring + ctx->job_ring_off;
    memcpy(slot, src_host, bytes);
    ctx->job_ring_off += need;
    H2_HIPCHK(hipMemcpyAsync(dst_dev, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return H2HIP_OK;
}
}  // namespace h2
extern "C" {

void h2hip_destroy(h2hip_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (int l = 0; l < 4; ++l)
        if (ctx->lane[l]) hipStreamSynchronize(ctx->lane[l]->stream);   // (a zero-fill of this context's bucket array may be queued on a lane's stream)
    if (ctx->clean_stream) {
        hipStreamSynchronize(ctx->clean_stream);
        hipStreamDestroy(ctx->clean_stream);
        hipEventDestroy(ctx->clean_ev);
        hipEventDestroy(ctx->clean_ev1);
        hipEventDestroy(ctx->used_ev);
    }
    if (ctx->tail_ev) hipEventDestroy(ctx->tail_ev);
    if (ctx->sorted_ev) hipEventDestroy(ctx->sorted_ev);
    if (ctx->job_ring) hipHostFree(ctx->job_ring);
    if (ctx->poll_host) hipHostFree(ctx->poll_host);
    for (auto &b : ctx->ws)
        if (b.p) hipFree(b.p);
    for (auto &t : ctx->twiddles) {
        hipFree(t.t1);
        hipFree(t.t2);
        for (int k = 0; k < 4; ++k) {
            if (t.direct[k]) hipFree(t.direct[k]);
        }
    }
    for (auto &p : ctx->pending) {
        hipEventDestroy(p.begin);
        hipEventDestroy(p.end);
    }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    if (ctx->prof_ref && ctx->own_prof_ref) hipEventDestroy(ctx->prof_ref);
    for (int l = 0; l < 4; ++l)
        if (ctx->lane[l]) {
            h2hip_destroy(ctx->lane[l]);
            hipEventDestroy(ctx->lane_ev[l]);
        }
    if (ctx->fork_ev) hipEventDestroy(ctx->fork_ev);
    if (ctx->fork_ev2) hipEventDestroy(ctx->fork_ev2);
    if (ctx->fork_ev3) hipEventDestroy(ctx->fork_ev3);
    for (auto e : ctx->timer_ev)
        if (e) hipEventDestroy(e);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int h2hip_sync(h2hip_ctx *ctx) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx, "ctx is NULL");
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    return H2HIP_OK;
}

static int *param_slot(h2hip_ctx *ctx, const char *name) {
    if (!strcmp(name, "msm_window_bits")) return &ctx->msm_window_bits;
    if (!strcmp(name, "msm_chunk")) return &ctx->msm_chunk;
    if (!strcmp(name, "msm_seg")) return &ctx->msm_seg;
    if (!strcmp(name, "ntt_tile_bits")) return &ctx->ntt_tile_bits;
    if (!strcmp(name, "ntt_debug_skip")) return &ctx->ntt_debug_skip;
    if (!strcmp(name, "ntt_tile_kernel")) return &ctx->ntt_tile_kernel;
    if (!strcmp(name, "plonk_warm_keygen")) return &ctx->plonk_warm_keygen;
    if (!strcmp(name, "plonk_tail_overlap")) return &ctx->plonk_tail_overlap;
    if (!strcmp(name, "plonk_side_on_lanes")) return &ctx->plonk_side_on_lanes;
    if (!strcmp(name, "kate_coeffs_per_lane")) return &ctx->kate_coeffs_per_lane;
    if (!strcmp(name, "quotient_29")) return &ctx->quotient_29;
    if (!strcmp(name, "kate_29")) return &ctx->kate_29;
    if (!strcmp(name, "host_poll")) return &ctx->host_poll;
    if (!strcmp(name, "plonk_merge_products")) return &ctx->plonk_merge_products;
    if (!strcmp(name, "plonk_shard_side")) return &ctx->plonk_shard_side;
    if (!strcmp(name, "plonk_route_rows")) return &ctx->plonk_route_rows;
    if (!strcmp(name, "plonk_lazy_upload")) return &ctx->plonk_lazy_upload;
    if (!strcmp(name, "plonk_early_intt")) return &ctx->plonk_early_intt;
    if (!strcmp(name, "plonk_gate_before_join")) return &ctx->plonk_gate_before_join;
    if (!strcmp(name, "msm_stagger_sorts")) return &ctx->msm_stagger_sorts;
    if (!strcmp(name, "msm_table_split")) return &ctx->msm_table_split;
    if (!strcmp(name, "clean_on_lane")) return &ctx->clean_on_lane;
    if (!strcmp(name, "plonk_permute_in_commit")) return &ctx->plonk_permute_in_commit;
    if (!strcmp(name, "ntt_full_table")) return &ctx->ntt_full_table;
    if (!strcmp(name, "ntt_min_col_bits")) return &ctx->ntt_min_col_bits;
    if (!strcmp(name, "msm_lanes")) return &ctx->msm_lanes;
    if (!strcmp(name, "msm_quad_tails")) return &ctx->msm_quad_tails;
    if (!strcmp(name, "msm_scatter_split")) return &ctx->msm_scatter_split;
    if (!strcmp(name, "msm_hist_packed")) return &ctx->msm_hist_packed;
    if (!strcmp(name, "msm_chunk_lone")) return &ctx->msm_chunk_lone;
    if (!strcmp(name, "msm_sort_groups")) return &ctx->msm_sort_groups;
    if (!strcmp(name, "msm_hist_split")) return &ctx->msm_hist_split;
    if (!strcmp(name, "msm_scatter_full_lds")) return &ctx->msm_scatter_full_lds;
    if (!strcmp(name, "msm_sort_threads")) return &ctx->msm_sort_threads;
    if (!strcmp(name, "msm_fuse_cols")) return &ctx->msm_fuse_cols;
    if (!strcmp(name, "msm_defer_reduce")) return &ctx->msm_defer_reduce;
    if (!strcmp(name, "msm_quad_seg_max")) return &ctx->msm_quad_seg_max;
    if (!strcmp(name, "lookup_big_tile_bits")) return &ctx->lookup_big_tile_bits;
    if (!strcmp(name, "fr_invert_run")) return &ctx->fr_invert_run;
    return nullptr;
}
int h2hip_set_param(h2hip_ctx *ctx, const char *name, int value) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && name, "NULL argument");
    int *p = param_slot(ctx, name);
    H2_REQUIRE(p, "unknown parameter name");
    if (p == &ctx->msm_window_bits) H2_REQUIRE(value == 0 || (value >= 4 && value <= 16), "msm_window_bits must be 0 (auto) or 4..16 (a window's histogram lives in LDS; at most 64 windows)");
    if (p == &ctx->msm_chunk) H2_REQUIRE(value == 0 || (value >= 2 && value <= 4096), "msm_chunk must be 0 (auto) or 2..4096");
    if (p == &ctx->msm_seg) H2_REQUIRE(value >= 1 && value <= 1024 && (value & (value - 1)) == 0, "msm_seg must be a power of two <= 1024");
    if (p == &ctx->msm_fuse_cols) H2_REQUIRE(value >= 0 && value <= (int)MSM_MAX_COLS, "msm_fuse_cols must be 0 (auto) or 1..32");
    if (p == &ctx->fr_invert_run) H2_REQUIRE(value >= 0 && value <= 1024, "fr_invert_run must be 0 (auto) or 1..1024");
    if (p == &ctx->lookup_big_tile_bits) H2_REQUIRE(value >= 12 && value <= 28, "lookup_big_tile_bits must be 12..28");
    if (p == &ctx->msm_sort_threads) H2_REQUIRE(value == 256 || value == 512 || value == 1024, "msm_sort_threads must be 256, 512 or 1024");
    if (p == &ctx->msm_scatter_split) H2_REQUIRE(value >= 0 && value <= 64 && (value & (value - 1)) == 0, "msm_scatter_split must be 0 or a power of two <= 64");
    if (p == &ctx->msm_lanes) H2_REQUIRE(value >= 0 && value <= 4, "msm_lanes must be 0 (auto) or 1..4");
    if (p == &ctx->ntt_min_col_bits) H2_REQUIRE(value >= 0 && value <= 5, "ntt_min_col_bits must be 0..5");
    if (p == &ctx->ntt_tile_bits) H2_REQUIRE(value >= 4 && value <= 10, "ntt_tile_bits must be 4..10");
    *p = value;
    return H2HIP_OK;
}
int h2hip_get_param(h2hip_ctx *ctx, const char *name, int *value) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && name && value, "NULL argument");
    int *p = param_slot(ctx, name);
    H2_REQUIRE(p, "unknown parameter name");
    *value = *p;
    return H2HIP_OK;
}

int h2hip_malloc(h2hip_ctx *ctx, size_t bytes, void **dptr) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && dptr, "NULL argument");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 256);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return H2HIP_ERR_NOMEM;
    }
    return H2HIP_OK;
}
int h2hip_free(h2hip_ctx *ctx, void *dptr) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx, "ctx is NULL");
    if (!dptr) return H2HIP_OK;
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    H2_HIPCHK(hipFree(dptr));
    return H2HIP_OK;
}
// Page-locks a caller-owned host buffer (e.g. the Vec<Fr> columns a prover re-uses across proofs): uploads from it then run on the DMA
// engines asynchronously instead of through the runtime's pageable staging copies.
int h2hip_host_register(h2hip_ctx *ctx, void *host_ptr, size_t bytes) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && host_ptr && bytes, "NULL argument");
    H2_HIPCHK(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return H2HIP_OK;
}
int h2hip_host_unregister(h2hip_ctx *ctx, void *host_ptr) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && host_ptr, "NULL argument");
    H2_HIPCHK(hipHostUnregister(host_ptr));
    return H2HIP_OK;
}
int h2hip_upload(h2hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (bytes == 0 || (dst_dev && src_host)), "NULL argument");
    if (!bytes) return H2HIP_OK;
    H2_HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    return H2HIP_OK;
}
int h2hip_download(h2hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (bytes == 0 || (dst_host && src_dev)), "NULL argument");
    if (!bytes) return H2HIP_OK;
    H2_HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    return H2HIP_OK;
}

// ------------------------------------------------------------------ profiling
int h2hip_profile_enable(h2hip_ctx *ctx, int on) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx, "ctx is NULL");
    prof_collect_all(ctx);
    ctx->profiling = on != 0;
    return H2HIP_OK;
}
int h2hip_profile_filter(h2hip_ctx *ctx, const char *prefix) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx, "ctx is NULL");
    prof_collect_all(ctx);
    ctx->prof_filter = prefix ? prefix : "";
    return H2HIP_OK;
}
int h2hip_profile_reset(h2hip_ctx *ctx) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx, "ctx is NULL");
    prof_collect_all(ctx);
    ctx->stats.clear();
    if (!ctx->prof_ref) {
        H2_HIPCHK(hipEventCreate(&ctx->prof_ref));
        ctx->own_prof_ref = true;
    }
    if (ctx->own_prof_ref) {
        H2_HIPCHK(hipEventRecord(ctx->prof_ref, ctx->stream));
        H2_HIPCHK(hipEventSynchronize(ctx->prof_ref));
    }
    return H2HIP_OK;
}
// time during which at least one launch of the matching kernels was executing (union of the launch spans): with
// pipelined MSMs several launches of one kernel overlap, and busy / launches is what one launch effectively costs
int h2hip_profile_get_busy(h2hip_ctx *ctx, const char *prefix, double *busy_ms) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && prefix && busy_ms, "NULL argument");
    prof_collect_all(ctx);
    std::vector<std::pair<float, float>> all;
    size_t len = strlen(prefix);
    for (auto &kv : ctx->stats)
        if (kv.first.compare(0, len, prefix) == 0) all.insert(all.end(), kv.second.spans.begin(), kv.second.spans.end());
    std::sort(all.begin(), all.end());
    double busy = 0;
    float cur0 = 0, cur1 = -1;
    for (auto &sp : all) {
        if (cur1 < cur0 || sp.first > cur1) {
            if (cur1 >= cur0) busy += cur1 - cur0;
            cur0 = sp.first;
            cur1 = sp.second;
        } else if (sp.second > cur1) {
            cur1 = sp.second;
        }
    }
    if (cur1 >= cur0) busy += cur1 - cur0;
    *busy_ms = busy;
    return H2HIP_OK;
}
int h2hip_profile_get(h2hip_ctx *ctx, const char *prefix, double *total_ms, uint64_t *launches) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && prefix, "NULL argument");
    prof_collect_all(ctx);
    double ms = 0;
    uint64_t cnt = 0;
    size_t len = strlen(prefix);
    for (auto &kv : ctx->stats)
        if (kv.first.compare(0, len, prefix) == 0) {
            ms += kv.second.total_ms;
            cnt += kv.second.launches;
        }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    return H2HIP_OK;
}
// every kernel name with launches since the last reset, as "name total_ms launches busy_ms\n" lines (NUL-terminated; truncated at cap)
int h2hip_profile_dump(h2hip_ctx *ctx, char *out, size_t cap, size_t *needed) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (out || cap == 0), "NULL argument");
    prof_collect_all(ctx);
    std::string text;
    for (auto &kv : ctx->stats) {
        std::vector<std::pair<float, float>> all(kv.second.spans.begin(), kv.second.spans.end());
        std::sort(all.begin(), all.end());
        double busy = 0;
        float cur0 = 0, cur1 = -1;
        for (auto &sp : all) {
            if (cur1 < cur0 || sp.first > cur1) {
                if (cur1 >= cur0) busy += cur1 - cur0;
                cur0 = sp.first;
                cur1 = sp.second;
            } else if (sp.second > cur1) {
                cur1 = sp.second;
            }
        }
        if (cur1 >= cur0) busy += cur1 - cur0;
        char line[256];
        snprintf(line, sizeof(line), "%s %.6f %llu %.6f\n", kv.first.c_str(), kv.second.total_ms, (unsigned long long)kv.second.launches, busy);
        text += line;
    }
    if (needed) *needed = text.size() + 1;
    if (cap) {
        size_t m = text.size() < cap - 1 ? text.size() : cap - 1;
        memcpy(out, text.data(), m);
        out[m] = 0;
    }
    return H2HIP_OK;
}
int h2hip_timer_start(h2hip_ctx *ctx) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx, "ctx is NULL");
    if (!ctx->timer_ev[0]) {   // per context: an event belongs to the device it was created on
        H2_HIPCHK(hipEventCreate(&ctx->timer_ev[0]));
        H2_HIPCHK(hipEventCreate(&ctx->timer_ev[1]));
    }
    H2_HIPCHK(hipEventRecord(ctx->timer_ev[0], ctx->stream));
    return H2HIP_OK;
}
int h2hip_timer_stop(h2hip_ctx *ctx, double *elapsed_ms) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && elapsed_ms && ctx->timer_ev[0], "timer not started");
    H2_HIPCHK(hipEventRecord(ctx->timer_ev[1], ctx->stream));
    H2_HIPCHK(hipEventSynchronize(ctx->timer_ev[1]));
    float ms = 0;
    H2_HIPCHK(hipEventElapsedTime(&ms, ctx->timer_ev[0], ctx->timer_ev[1]));
    *elapsed_ms = ms;
    return H2HIP_OK;
}

// ------------------------------------------------------------------ MSM
static int bases_create(h2hip_ctx *ctx, const void *src, bool src_on_device, size_t n, uint32_t flags, h2hip_bases **out) {
    H2_REQUIRE(ctx && out && (n == 0 || src), "NULL argument");
    H2_REQUIRE((flags & ~H2HIP_BASES_PRECOMPUTE) == 0, "unknown flags");
    h2hip_bases *b = new h2hip_bases();
    b->n = n;
    hipError_t e = hipMalloc((void **)&b->pts, sizeof(G1Affine) * (n ? n : 1));
    if (e != hipSuccess) {
        set_error("hipMalloc for %zu bases failed: %s", n, hipGetErrorString(e));
        delete b;
        return H2HIP_ERR_NOMEM;
    }
    if (n) {
        e = hipMemcpyAsync(b->pts, src, sizeof(G1Affine) * n, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("copying bases failed: %s", hipGetErrorString(e));
            hipFree(b->pts);
            delete b;
            return H2HIP_ERR_HIP;
        }
    }
    {
        int rc = msm_prepare_bases(ctx, b, (flags & H2HIP_BASES_PRECOMPUTE) != 0);
        if (rc != H2HIP_OK) {
            hipFree(b->pts);
            delete b;
            return rc;
        }
    }
    *out = b;
    return H2HIP_OK;
}
int h2hip_bases_upload(h2hip_ctx *ctx, const void *g1_affine_host, size_t n, uint32_t flags, h2hip_bases **out) {
    H2_DEVICE_GUARD(ctx);
    return bases_create(ctx, g1_affine_host, false, n, flags, out);
}
int h2hip_bases_from_device(h2hip_ctx *ctx, const void *g1_affine_dev, size_t n, uint32_t flags, h2hip_bases **out) {
    H2_DEVICE_GUARD(ctx);
    return bases_create(ctx, g1_affine_dev, true, n, flags, out);
}
void h2hip_bases_free(h2hip_ctx *ctx, h2hip_bases *bases) {
    H2_DEVICE_GUARD(ctx);
    if (!bases) return;
    if (ctx) hipStreamSynchronize(ctx->stream);
    if (bases->pts) hipFree(bases->pts);
    if (bases->pts29) hipFree(bases->pts29);
    delete bases;
}
size_t h2hip_bases_len(const h2hip_bases *bases) { return bases ? bases->n : 0; }

static int finish_point(h2hip_ctx *ctx, char *outbuf, int point_format, void *out_host);
static int msm_batch_impl(h2hip_ctx *ctx, const h2hip_bases *bases, const h2hip_bases *const *bases_per_col, const void *const *scalars_in,
                          bool scalars_on_host, size_t n, size_t count, int point_format, void *out_host);
int h2hip_msm_g1_dev(h2hip_ctx *ctx, const h2hip_bases *bases, const void *scalars_dev, size_t n, int point_format, void *out_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && bases && out_host && (n == 0 || scalars_dev), "NULL argument");
    H2_REQUIRE(point_format == H2HIP_POINT_JACOBIAN || point_format == H2HIP_POINT_AFFINE, "unknown point_format");
    char *outbuf = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OUT, 1024, (void **)&outbuf));
    H2_CHK(msm_run(ctx, bases, (const Fr *)scalars_dev, n, (XYZZ *)outbuf));
    return finish_point(ctx, outbuf, point_format, out_host);
}
// Several independent MSMs over the same bases (e.g. the h(X) pieces, or all advice columns of a phase): MSM j runs
// on lane j mod 2 — a child context with its own stream and scratch — so the latency-bound tail of one MSM (merge,
// bucket reduction) overlaps the multiplier-bound accumulation of the next.
// once groups are queued on the lanes, an error return must not leave them running on buffers the caller's next call reuses
static void join_lanes(h2hip_ctx *ctx, int nl) {
    for (int l = 0; l < nl; ++l)
        if (ctx->lane[l]) hipStreamSynchronize(ctx->lane[l]->stream);
}
#define H2_LANES(expr)                                                                           \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            h2::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            join_lanes(ctx, NL);                                                                 \
            return H2HIP_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)
#define H2_LANES_RC(expr)         \
    do {                          \
        int r__ = (expr);         \
        if (r__ != H2HIP_OK) {    \
            join_lanes(ctx, NL);  \
            return r__;           \
        }                         \
    } while (0)
// bases_per_col (optional): a base set per column — columns over different sets (e.g. a Lagrange-basis and a monomial-basis commitment of
// the same prover round) share the lanes and, when their window tables match, the deferred bucket reduction
static int msm_batch_impl(h2hip_ctx *ctx, const h2hip_bases *bases, const h2hip_bases *const *bases_per_col, const void *const *scalars_in,
                          bool scalars_on_host, size_t n, size_t count, int point_format, void *out_host) {
    if (bases_per_col && count) bases = bases_per_col[0];
    H2_REQUIRE(ctx && bases && (count == 0 || (scalars_in && out_host)), "NULL argument");
    bool mixed = false;
    if (bases_per_col)
        for (size_t j = 0; j < count; ++j) {
            H2_REQUIRE(bases_per_col[j] && n <= bases_per_col[j]->n, "NULL base set / more scalars than bases");
            if (bases_per_col[j] != bases) mixed = true;
            H2_REQUIRE((bases_per_col[j]->tables > 1) == (bases->tables > 1) && bases_per_col[j]->window_bits == bases->window_bits,
                       "the base sets of one batch must share their table layout (plain, or precomputed with the same window)");
        }
    auto bases_of = [&](size_t j) -> const h2hip_bases * { return bases_per_col ? bases_per_col[j] : bases; };
    H2_REQUIRE(point_format == H2HIP_POINT_JACOBIAN || point_format == H2HIP_POINT_AFFINE, "unknown point_format");
    if (!count) return H2HIP_OK;
    H2_REQUIRE(n <= bases->n, "more scalars than bases");
    for (size_t j = 0; j < count; ++j) H2_REQUIRE(n == 0 || scalars_in[j], "NULL scalar column");   // everything checked before the first launch
    const bool affine = point_format == H2HIP_POINT_AFFINE;
    const size_t psz = affine ? sizeof(G1Affine) : sizeof(G1Jac);
    // lanes: the kernels of one MSM are issue-bound or latency-bound, so lanes that overlap whole MSMs mostly contend (measured,
    // tools/batch_ab.py, batches of 4 with the deferred reduction: 2^20 1.71 / 1.75 / 1.80 / 1.84 ms per MSM on 1 / 2 / 3 / 4 lanes, 2^19
    // 0.99 / 0.95 / 0.98 / 1.00: r02's kernels) — auto picks 2 lanes from 2^20 points (r04, measured in proofs), 3 below
    int NL = ctx->msm_lanes;
    if (NL <= 0) NL = n >= ((size_t)1 << 20) ? 2 : 3;   // (2^18 / 2^19 were on 2 lanes until the window model moved them to c = 15: 3 lanes now win by 2 %, k = 18 / 19 proofs;
                                                        //  r04: 2^21 on 2 lanes 60.3 ms per k = 21 proof against 62.2 on one and 61.0 on three — the next column's sort
                                                        //  runs beside the accumulation: profiles/archive/r04_msm_lanes_large.log)
    if (NL > 4) NL = 4;
    // (a third context exists even where only two lanes carry columns: the prover's side transforms run on the LAST lane's context, and with two
    // lanes a context of their own ended up behind the grand products on a shared hardware queue — k = 21: the products waited 3.5 ms for
    // the transforms they were meant to run beside, profiles/archive/r04_timeline_k21.md)
    for (int l = 0; l < (NL < 3 ? 3 : NL); ++l) {
        if (!ctx->lane[l]) {
            h2hip_ctx *c = nullptr;
            H2_CHK(h2hip_init(ctx->device, nullptr, &c));
            c->msm_window_bits = ctx->msm_window_bits;
            c->is_lane = true;
            ctx->lane[l] = c;
            H2_HIPCHK(hipEventCreate(&ctx->lane_ev[l]));
        }
        h2hip_ctx *c = ctx->lane[l];
        inherit_knobs(c, ctx);
    }
    if (!ctx->fork_ev) H2_HIPCHK(hipEventCreate(&ctx->fork_ev));
    char *results = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_BATCH, psz * count, (void **)&results));
    H2_HIPCHK(hipEventRecord(ctx->fork_ev, ctx->stream));   // inputs produced on the caller's stream are ready after this
    for (int l = 0; l < NL; ++l) H2_HIPCHK(hipStreamWaitEvent(ctx->lane[l]->stream, ctx->fork_ev, 0));
    // Precomputed bases, two shapes (measured, tools/fuse_sweep*.sh):
    //  * up to 2^17 points: columns are FUSED into groups that go through the whole pipeline as one multi-column MSM;
    //  * larger: every column runs its own sort / accumulation / merge on a lane (pipelined), and the latency-bound
    //    bucket reduction is DEFERRED: it runs once, for all columns together, after the lanes have joined.
    const bool precomp = bases->tables > 1;
    size_t fuse = precomp ? (size_t)ctx->msm_fuse_cols : 1;
    if (precomp && ctx->msm_fuse_cols == 0) {   // auto: about 2^19 scalars per fused MSM, at most 16 columns (2^17: 4, 2^16: 8, <= 2^15: 16); larger sizes run one by one
        fuse = 1;
        if (n <= ((size_t)1 << 17))
            while (fuse < 16 && fuse * 2 * (n ? n : 1) <= ((size_t)1 << 19)) fuse *= 2;
    }
    if (fuse < 1) fuse = 1;
    if (fuse > MSM_MAX_COLS) fuse = MSM_MAX_COLS;
    // groups of columns that go through the pipeline as one fused MSM: a fused MSM reads one table, so a group never spans two base sets
    // (runs of columns over the same set are split into balanced groups of at most `fuse`)
    std::vector<std::pair<size_t, size_t>> groups;   // (first column, size)
    for (size_t r0 = 0; r0 < count;) {
        size_t r1 = r0 + 1;
        while (r1 < count && bases_of(r1) == bases_of(r0)) ++r1;
        const size_t run = r1 - r0, ng = (run + fuse - 1) / fuse;
        for (size_t g = 0, j = r0; g < ng; ++g) {
            const size_t gs = (r1 - j + (ng - g) - 1) / (ng - g);
            groups.push_back({j, gs});
            j += gs;
        }
        r0 = r1;
    }
    (void)mixed;
    // deferred bucket reduction: every column (or fused group of columns) stops after its merge and leaves its buckets in one array; the
    // latency-bound reduction then runs once per 64 columns for the whole batch instead of once per MSM / group
    size_t keys_per_col = 0;
    if (precomp) {
        const uint32_t cw = bases->window_bits;
        const uint32_t wcol = (255 + cw - 1) / cw;
        keys_per_col = (size_t)wcol << (cw - 1);   // one bucket set per window
    }
    const bool deferred = precomp && ctx->msm_defer_reduce && count >= 2 && n > 0 && (fuse == 1 ? count <= 64 : true) &&
                          sizeof(XYZZ29) * keys_per_col * count <= ((size_t)2 << 30);
    XYZZ29 *all_buckets = nullptr;
    if (deferred) H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_BATCH_BUCKETS, sizeof(XYZZ29) * keys_per_col * count, (void **)&all_buckets));
    // the shared bucket array was zero-filled behind the previous batch's reduction (side stream): the lanes wait for that instead of filling
    // (a fill of the SAME buffer may still be pending on lane 0's stream even when it covered fewer bytes than this batch needs: the lanes wait
    // for it either way, or it could wipe partial sums of lanes 1 / 2 that zero their own regions and start accumulating — ADVICE r04)
    const bool fill_pending = deferred && ctx->clean_ev1 && ctx->clean_ptr[1] == all_buckets;
    const bool buckets_zeroed = deferred && buckets_prezeroed(ctx, 1, all_buckets, sizeof(XYZZ29) * keys_per_col * count);
    if (buckets_zeroed || fill_pending)
        for (int l = 0; l < NL; ++l) H2_HIPCHK(hipStreamWaitEvent(ctx->lane[l]->stream, ctx->clean_ev1, 0));
    // host columns: one staging area for all of them; column j is copied on its lane's stream right before its kernels are
    // queued, so the (host-blocking, pageable) copy of column j+1 overlaps the GPU work of column j
    std::vector<const void *> staged(count, nullptr);
    const void *const *scalars_dev = scalars_in;
    if (scalars_on_host && n) {
        char *stage = nullptr;
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_STAGE, sizeof(Fr) * n * count, (void **)&stage));
        for (size_t j = 0; j < count; ++j) staged[j] = stage + sizeof(Fr) * n * j;
        scalars_dev = staged.data();
    }
    // (two other schedules were built, measured slower and removed in r04: every accumulation on one stream with all sorts / merges on a
    // second, higher-priority one — 2^19 1.00 vs 0.95 ms per MSM, tools/batch_ab.py in r02 — and a column's windows dealt to two lanes —
    // the k = 19 proof 15.7-15.8 vs 14.7 ms, profiles/archive/r03_msm_split_windows_ab.log)
    // late columns (msm_mid_hook, internal.h): the hook runs once, before the first group that holds a column >= msm_mid_after
    std::function<int()> mid_hook;
    mid_hook.swap(ctx->msm_mid_hook);
    const size_t mid_after = ctx->msm_mid_after;
    auto run_mid = [&]() -> int {
        if (!mid_hook) return H2HIP_OK;
        std::function<int()> f;
        f.swap(mid_hook);
        H2_CHK(f());
        if (!ctx->fork_ev2) H2_HIPCHK(hipEventCreateWithFlags(&ctx->fork_ev2, hipEventDisableTiming));
        H2_HIPCHK(hipEventRecord(ctx->fork_ev2, ctx->stream));   // what the hook queued on the caller's stream produces the remaining columns
        for (int l = 0; l < NL; ++l) H2_HIPCHK(hipStreamWaitEvent(ctx->lane[l]->stream, ctx->fork_ev2, 0));
        return H2HIP_OK;
    };
    std::function<int(size_t)> col_hook;   // columns that arrive one by one (msm_col_hook, internal.h)
    col_hook.swap(ctx->msm_col_hook);
    const size_t ngroups = groups.size();
    // (r05, last: the lanes' streams created with the lowest / the highest HIP priority — either way 7 - 10 % slower at k = 17 / 19, profiles/r05_lane_priority_ab.log; removed)
    // (r05 built and measured a third schedule — the sorts of a round of columns queued on ALL lanes before any of their accumulations, so that no
    // sort starts beside an accumulation that holds every CU: the k = 19 proof 13.8-14.0 vs 13.7-13.9 ms, k = 21 53.4-54.0 vs 52.8-53.2, k = 15 / 18
    // equal — profiles/r05_msm_sort_first_ab.log; removed)
    for (size_t g = 0; g < ngroups; ++g) {
        const size_t j0 = groups[g].first, gsize = groups[g].second;
        h2hip_ctx *c = ctx->lane[g % NL];
        if (col_hook)   // the group's columns are produced now, on the caller's stream (before the mid hook: what it queues may read them)
            for (size_t j = j0; j < j0 + gsize; ++j) H2_LANES_RC(col_hook(j));
        if (mid_hook && j0 + gsize > mid_after) H2_LANES_RC(run_mid());
        if (col_hook) {   // the lane waits for them
            if (!ctx->fork_ev3) H2_LANES(hipEventCreateWithFlags(&ctx->fork_ev3, hipEventDisableTiming));
            H2_LANES(hipEventRecord(ctx->fork_ev3, ctx->stream));
            H2_LANES(hipStreamWaitEvent(c->stream, ctx->fork_ev3, 0));
        }
        const h2hip_bases *gb = bases_of(j0);
        if ((ctx->msm_stagger_sorts > 0 || (ctx->msm_stagger_sorts < 0 && NL == 2)) && precomp && fuse == 1 && g < (size_t)NL) {   // the first round of columns: lane g sorts behind lane g - 1's sort
            if (g > 0 && ctx->lane[g - 1]->sorted_ev) H2_LANES(hipStreamWaitEvent(c->stream, ctx->lane[g - 1]->sorted_ev, 0));
            if (g + 1 < (size_t)NL && g + 1 < ngroups) {
                if (!c->sorted_ev) H2_LANES(hipEventCreateWithFlags(&c->sorted_ev, hipEventDisableTiming));
                c->sorted_arm = true;
            }
        }
        for (size_t j = j0; j < j0 + gsize; ++j) {
            if (scalars_on_host && n) H2_LANES(hipMemcpyAsync((void *)staged[j], scalars_in[j], sizeof(Fr) * n, hipMemcpyHostToDevice, c->stream));
        }
        char *outbuf = nullptr;
        H2_LANES_RC(ws_reserve(c, h2hip_ctx::WS_OUT, sizeof(XYZZ) * MSM_MAX_COLS, (void **)&outbuf));
        H2_LANES_RC(msm_run_cols(c, gb, (const Fr *const *)(scalars_dev + j0), (uint32_t)gsize, n, (XYZZ *)outbuf,
                                 deferred ? all_buckets + keys_per_col * j0 : nullptr, buckets_zeroed));
        if (!deferred) {   // the group's results, one lane each, into their slots of the batch's result array
            prof_begin(c, "point_finish_kernel");
            hipLaunchKernelGGL(point_finish_slot_kernel, dim3((uint32_t)gsize), dim3(64), 0, c->stream, (const XYZZ *)outbuf,
                               affine ? (G1Jac *)nullptr : (G1Jac *)results, affine ? (G1Affine *)results : (G1Affine *)nullptr, (uint32_t)j0);
            prof_end(c);
        }
        H2_LANES(hipGetLastError());
    }
    if (mid_hook) H2_LANES_RC(run_mid());   // (no column behind msm_mid_after: the hook still runs, before the join)
    for (int l = 0; l < NL; ++l) {
        H2_HIPCHK(hipEventRecord(ctx->lane_ev[l], ctx->lane[l]->stream));
        H2_HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->lane_ev[l], 0));
    }
    // the accumulations are queued and joined: what follows on this stream is the reduction's tail.  The hook runs HERE, before the tail is
    // queued: the device is still hundreds of microseconds of accumulation behind the host at this point, so the hook's few launches do not
    // delay the reduction — and a wait on the event must be issued before more work follows it on this stream (r04 timeline,
    // profiles/archive/r04_timeline_k19.md: issued after the tail and the result copy had been queued, the side transforms started 20 us after that
    // copy FINISHED — the runtime resolved the cross-stream wait against what the stream held at the time of the wait, not of the record)
    int hook_rc = H2HIP_OK;
    if (ctx->msm_tail_hook) {
        std::function<int(hipEvent_t)> tail_hook;
        tail_hook.swap(ctx->msm_tail_hook);
        if (!ctx->tail_ev) H2_HIPCHK(hipEventCreateWithFlags(&ctx->tail_ev, hipEventDisableTiming));
        H2_HIPCHK(hipEventRecord(ctx->tail_ev, ctx->stream));
        hook_rc = tail_hook(ctx->tail_ev);
    }
    if (deferred) {   // one bucket reduction per 64 columns, on the caller's stream
        XYZZ *sums = nullptr;
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OUT, sizeof(XYZZ) * count, (void **)&sums));
        for (size_t c0 = 0; c0 < count; c0 += 64) {
            const uint32_t cc = (uint32_t)(count - c0 < 64 ? count - c0 : 64);
            H2_CHK(msm_reduce_cols(ctx, bases, bases->window_bits, all_buckets + keys_per_col * c0, cc, sums + c0));
        }
        prof_begin(ctx, "point_finish_kernel");
        hipLaunchKernelGGL(point_finish_slot_kernel, dim3((uint32_t)count), dim3(64), 0, ctx->stream, (const XYZZ *)sums,
                           affine ? (G1Jac *)nullptr : (G1Jac *)results, affine ? (G1Affine *)results : (G1Affine *)nullptr, 0u);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
    }
    // zero-fill of the shared bucket array for the next batch, off its critical path — queued AFTER the hook's work and on the first lane's stream
    // (the hook works on the last lane's): a fill that waits for the reduction must not sit in front of that work in a shared hardware queue
    if (deferred) H2_CHK(buckets_clean_after_use(ctx, 1, all_buckets, sizeof(XYZZ29) * keys_per_col * count, ctx->clean_on_lane ? ctx->lane[0]->stream : nullptr));
    H2_CHK(sync_results(ctx, out_host, results, psz * count));   // the commitments come back (r05: through the host-mapped flag, no runtime wait)
    H2_CHK(hook_rc);
    return H2HIP_OK;   // (the lanes' kernel timers are folded into this context's table when it is read: prof_collect_all)
}

int h2hip_msm_g1_batch_dev(h2hip_ctx *ctx, const h2hip_bases *bases, const void *const *scalars_dev, size_t n, size_t count, int point_format,
                           void *out_host) {
    H2_DEVICE_GUARD(ctx);
    return msm_batch_impl(ctx, bases, nullptr, scalars_dev, false, n, count, point_format, out_host);
}
int h2hip_msm_g1_multi_dev(h2hip_ctx *ctx, const h2hip_bases *const *bases_per_column, const void *const *scalars_dev, size_t n, size_t count,
                           int point_format, void *out_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (count == 0 || bases_per_column), "NULL argument");
    if (!count) return H2HIP_OK;
    return msm_batch_impl(ctx, nullptr, bases_per_column, scalars_dev, false, n, count, point_format, out_host);
}
int h2hip_msm_g1_batch(h2hip_ctx *ctx, const h2hip_bases *bases, const void *const *scalars_host, size_t n, size_t count, int point_format,
                       void *out_host) {
    H2_DEVICE_GUARD(ctx);
    return msm_batch_impl(ctx, bases, nullptr, scalars_host, true, n, count, point_format, out_host);
}

int h2hip_msm_g1(h2hip_ctx *ctx, const h2hip_bases *bases, const void *scalars_host, size_t n, int point_format, void *out_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && bases && out_host && (n == 0 || scalars_host), "NULL argument");
    Fr *stage = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_STAGE, sizeof(Fr) * n, (void **)&stage));
    if (n) H2_HIPCHK(hipMemcpyAsync(stage, scalars_host, sizeof(Fr) * n, hipMemcpyHostToDevice, ctx->stream));
    return h2hip_msm_g1_dev(ctx, bases, stage, n, point_format, out_host);
}

static int finish_point(h2hip_ctx *ctx, char *outbuf, int point_format, void *out_host) {
    XYZZ *acc = (XYZZ *)outbuf;
    G1Jac *jac = (G1Jac *)(outbuf + 256);
    G1Affine *aff = (G1Affine *)(outbuf + 512);
    const bool affine = point_format == H2HIP_POINT_AFFINE;
    prof_begin(ctx, "point_finish_kernel");
    hipLaunchKernelGGL(point_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, (const XYZZ *)acc, affine ? (G1Jac *)nullptr : jac,
                       affine ? aff : (G1Affine *)nullptr);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return sync_results(ctx, out_host, affine ? (void *)aff : (void *)jac, affine ? sizeof(G1Affine) : sizeof(G1Jac));
}

int h2hip_g1_sum_jacobian_dev(h2hip_ctx *ctx, const void *points_dev, size_t n, int point_format, void *out_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && out_host && (n == 0 || points_dev), "NULL argument");
    H2_REQUIRE(point_format == H2HIP_POINT_JACOBIAN || point_format == H2HIP_POINT_AFFINE, "unknown point_format");
    H2_REQUIRE(n < (1u << 24), "too many points");
    char *outbuf = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OUT, 1024, (void **)&outbuf));
    prof_begin(ctx, "jac_sum_kernel");
    hipLaunchKernelGGL(jac_sum_kernel, dim3(1), dim3(64), 0, ctx->stream, (const G1Jac *)points_dev, (uint32_t)n, (XYZZ *)outbuf);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return finish_point(ctx, outbuf, point_format, out_host);
}

// The host half of a point-range sharded commitment round: out[j] = sum over ranks r of gathered[r * count + j] (Jacobian partials as
// h2hip_comm_allgather_host leaves them, rank-major), for all `count` columns in one call — a few dozen point additions on the host,
// no device round trip.  point_format of the OUTPUT: Jacobian (z = 1 or 0) or affine.
int h2hip_g1_sum_partials_host(const void *gathered_jacobian, size_t world, size_t count, int point_format, void *out) {
    H2_REQUIRE((count == 0 || (gathered_jacobian && out)) && world >= 1, "bad argument");
    H2_REQUIRE(point_format == H2HIP_POINT_JACOBIAN || point_format == H2HIP_POINT_AFFINE, "unknown point_format");
    const G1Jac *in = (const G1Jac *)gathered_jacobian;
    for (size_t j = 0; j < count; ++j) {
        XYZZ acc = XYZZ::identity();
        for (size_t r = 0; r < world; ++r) {
            G1Jac p;
            memcpy(&p, &in[r * count + j], sizeof(G1Jac));
            if (p.z.is_zero()) continue;
            XYZZ q;
            q.x = p.x;
            q.y = p.y;
            q.zz = fe_sqr(p.z);
            q.zzz = fe_mul(q.zz, p.z);
            xyzz_add(acc, q);
        }
        const G1Affine a = xyzz_to_affine(acc);
        if (point_format == H2HIP_POINT_AFFINE) {
            memcpy((char *)out + sizeof(G1Affine) * j, &a, sizeof(G1Affine));
        } else {
            G1Jac o;
            o.x = a.x;
            o.y = a.y;
            o.z = a.is_identity() ? Fq::zero() : Fq::one();
            memcpy((char *)out + sizeof(G1Jac) * j, &o, sizeof(G1Jac));
        }
    }
    return H2HIP_OK;
}

// ------------------------------------------------------------------ NTT family
static Fr load_fr(const void *p) {
    Fr r;
    memcpy(&r, p, sizeof(Fr));
    return r;
}
static int stage_in(h2hip_ctx *ctx, const void *host, size_t elems, Fr **dev) {
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_STAGE, sizeof(Fr) * elems, (void **)dev));
    H2_HIPCHK(hipMemcpyAsync(*dev, host, sizeof(Fr) * elems, hipMemcpyHostToDevice, ctx->stream));
    return H2HIP_OK;
}
static int stage_out(h2hip_ctx *ctx, void *host, const Fr *dev, size_t elems) {
    H2_HIPCHK(hipMemcpyAsync(host, dev, sizeof(Fr) * elems, hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    return H2HIP_OK;
}

int h2hip_best_fft_dev(h2hip_ctx *ctx, void *a_dev, const void *omega, uint32_t log_n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a_dev && omega, "NULL argument");
    return ntt_run(ctx, (Fr *)a_dev, log_n, load_fr(omega), nullptr, 0, nullptr, nullptr);
}
int h2hip_best_fft(h2hip_ctx *ctx, void *a_host, const void *omega, uint32_t log_n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a_host && omega && log_n <= 28, "bad argument");
    Fr *d = nullptr;
    H2_CHK(stage_in(ctx, a_host, (size_t)1 << log_n, &d));
    H2_CHK(h2hip_best_fft_dev(ctx, d, omega, log_n));
    return stage_out(ctx, a_host, d, (size_t)1 << log_n);
}
int h2hip_ifft_dev(h2hip_ctx *ctx, void *a_dev, const void *omega_inv, uint32_t log_n, const void *divisor) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a_dev && omega_inv && divisor, "NULL argument");
    Fr d = load_fr(divisor);
    Fr out3[3] = {d, d, d};
    return ntt_run(ctx, (Fr *)a_dev, log_n, load_fr(omega_inv), nullptr, 0, nullptr, out3);
}
int h2hip_ifft(h2hip_ctx *ctx, void *a_host, const void *omega_inv, uint32_t log_n, const void *divisor) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a_host && omega_inv && divisor && log_n <= 28, "bad argument");
    Fr *d = nullptr;
    H2_CHK(stage_in(ctx, a_host, (size_t)1 << log_n, &d));
    H2_CHK(h2hip_ifft_dev(ctx, d, omega_inv, log_n, divisor));
    return stage_out(ctx, a_host, d, (size_t)1 << log_n);
}
int h2hip_coeff_to_extended_dev(h2hip_ctx *ctx, const void *coeffs_dev, uint32_t k, void *out_dev, uint32_t ext_k, const void *ext_omega,
                                const void *zeta) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && coeffs_dev && out_dev && ext_omega && zeta, "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    H2_REQUIRE(coeffs_dev != out_dev || k == ext_k, "coeffs and out must not alias");
    Fr z = load_fr(zeta);
    Fr in3[3] = {Fr::one(), z, fe_mul(z, z)};
    return ntt_run(ctx, (Fr *)out_dev, ext_k, load_fr(ext_omega), (const Fr *)coeffs_dev, (uint64_t)1 << k, in3, nullptr);
}
// the same two transforms over `count` columns at once (host arrays of device pointers): 32 columns per launch
int h2hip_ifft_batch_dev(h2hip_ctx *ctx, void *const *cols_dev, size_t count, const void *omega_inv, uint32_t log_n, const void *divisor) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && omega_inv && divisor && (count == 0 || cols_dev), "NULL argument");
    Fr d = load_fr(divisor);
    Fr out3[3] = {d, d, d};
    return ntt_run_batch(ctx, (Fr *const *)cols_dev, nullptr, count, log_n, load_fr(omega_inv), 0, nullptr, out3);
}
int h2hip_coeff_to_extended_batch_dev(h2hip_ctx *ctx, const void *const *coeffs_dev, uint32_t k, void *const *outs_dev, uint32_t ext_k, size_t count,
                                      const void *ext_omega, const void *zeta) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && ext_omega && zeta && (count == 0 || (coeffs_dev && outs_dev)), "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    for (size_t j = 0; j < count; ++j) H2_REQUIRE(coeffs_dev[j] && outs_dev[j] && (coeffs_dev[j] != outs_dev[j] || k == ext_k), "NULL column, or coeffs and out alias");
    Fr z = load_fr(zeta);
    Fr in3[3] = {Fr::one(), z, fe_mul(z, z)};
    return ntt_run_batch(ctx, (Fr *const *)outs_dev, (const Fr *const *)coeffs_dev, count, ext_k, load_fr(ext_omega), (uint64_t)1 << k, in3, nullptr);
}
int h2hip_coeff_to_extended(h2hip_ctx *ctx, const void *coeffs_host, uint32_t k, void *out_host, uint32_t ext_k, const void *ext_omega,
                            const void *zeta) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && coeffs_host && out_host && ext_omega && zeta, "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    Fr *d = nullptr;
    const size_t n = (size_t)1 << k, ne = (size_t)1 << ext_k;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_STAGE, sizeof(Fr) * (n + ne), (void **)&d));
    H2_HIPCHK(hipMemcpyAsync(d, coeffs_host, sizeof(Fr) * n, hipMemcpyHostToDevice, ctx->stream));
    H2_CHK(h2hip_coeff_to_extended_dev(ctx, d, k, d + n, ext_k, ext_omega, zeta));
    return stage_out(ctx, out_host, d + n, ne);
}
int h2hip_extended_to_coeff_dev(h2hip_ctx *ctx, void *a_dev, uint32_t ext_k, const void *ext_omega_inv, const void *ext_divisor,
                                const void *zeta_inv) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a_dev && ext_omega_inv && ext_divisor && zeta_inv, "NULL argument");
    Fr d = load_fr(ext_divisor), zi = load_fr(zeta_inv);
    Fr out3[3] = {d, fe_mul(d, zi), fe_mul(d, fe_mul(zi, zi))};
    return ntt_run(ctx, (Fr *)a_dev, ext_k, load_fr(ext_omega_inv), nullptr, 0, nullptr, out3);
}
int h2hip_extended_to_coeff(h2hip_ctx *ctx, void *a_host, uint32_t ext_k, const void *ext_omega_inv, const void *ext_divisor,
                            const void *zeta_inv) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a_host && ext_omega_inv && ext_divisor && zeta_inv && ext_k <= 28, "bad argument");
    Fr *d = nullptr;
    H2_CHK(stage_in(ctx, a_host, (size_t)1 << ext_k, &d));
    H2_CHK(h2hip_extended_to_coeff_dev(ctx, d, ext_k, ext_omega_inv, ext_divisor, zeta_inv));
    return stage_out(ctx, a_host, d, (size_t)1 << ext_k);
}

}  // extern "C"
