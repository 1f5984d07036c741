// permute_expression_pair of halo2's (permuted) lookup argument on the GPU [UPSTREAM halo2-axiom 0.5.3
// plonk/lookup/prover.rs; halo2-base creates the lookups at
// /root/reference/halo2-base/src/gates/range/mod.rs:131-150]:
//   A' = sort(A[..usable]) ;  S'[i] = A'[i] where A'[i] starts a run of equal values ;  the remaining rows (repeats
//   of A') receive the table elements not consumed that way, in ascending order, assigned from the LAST repeated
//   row backwards (upstream: BTreeMap iteration ascending, repeated_input_rows.pop()).
// The order of field elements is the numeric order of their canonical values (Fr: Ord compares to_repr() from the
// most significant byte).  Sorting is a bitonic network on 32-byte keys: stages with partner distance < 4096 run in
// LDS, the others as global passes; everything else is flags + prefix sums + binary searches.
#include <algorithm>

#include "internal.h"

namespace h2 {

struct alignas(16) Key256 {
    uint32_t l[8];
};
__device__ __forceinline__ bool key_less(const Key256 &a, const Key256 &b) {
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        if (a.l[i] != b.l[i]) return a.l[i] < b.l[i];
    }
    return false;
}
__device__ __forceinline__ bool key_eq(const Key256 &a, const Key256 &b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.l[i] ^ b.l[i];
    return d == 0;
}

// canonical keys, padded with +infinity (all ones > any canonical element) up to the power of two N
// *small_max (optional): max over the usable keys of (key < 2^32 ? key : 0xFFFFFFFF) — decides whether the counting sort applies
__global__ __launch_bounds__(256) void lk_keys_kernel(const Fr *__restrict__ in, Key256 *__restrict__ keys, uint32_t usable, uint32_t N,
                                                      uint32_t *__restrict__ small_max) {
    __shared__ uint32_t wg_max;
    if (threadIdx.x == 0) wg_max = 0;
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Key256 k;
    if (i < usable) {
        Fr c = fe_from_mont(in[i]);
        uint32_t hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            k.l[j] = c.l[j];
            if (j) hi |= c.l[j];
        }
        if (small_max) {
            uint32_t v = hi ? 0xFFFFFFFFu : c.l[0];
            if (v) atomicMax(&wg_max, v);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) k.l[j] = 0xFFFFFFFFu;
    }
    if (i < N) keys[i] = k;
    __syncthreads();
    if (small_max && threadIdx.x == 0 && wg_max) atomicMax(small_max, wg_max);
}

// ---- counting sort for small keys: halo2-base's lookups are range tables (values < 2^lookup_bits, range/mod.rs:154-170) and their inputs are
// table members (or 0 where q_lookup is off), so every key fits far below 2^32 — a histogram over the key values, a prefix sum and an
// expansion replace the ~180 compare-exchange passes of the bitonic network over 32-byte keys.
__global__ __launch_bounds__(256) void lk_count_kernel(const Key256 *__restrict__ keys, uint32_t usable, uint32_t *__restrict__ hist) {
    __shared__ uint32_t zeros;   // key 0 is by far the most frequent one (rows without a lookup): one global atomic per workgroup for it
    if (threadIdx.x == 0) zeros = 0;
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < usable) {
        uint32_t v = keys[i].l[0];
        if (v) atomicAdd(&hist[v], 1u);
        else atomicAdd(&zeros, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && zeros) atomicAdd(&hist[0], zeros);
}
// out[i] = the key whose [offsets[v], offsets[v+1]) range contains i (i < usable); +infinity padding beyond
__global__ __launch_bounds__(256) void lk_expand_kernel(const uint32_t *__restrict__ offsets, uint32_t bins, uint32_t usable, uint32_t N,
                                                        Key256 *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Key256 k;
    if (i < usable) {
        uint32_t lo = 0, hi = bins;   // largest v with offsets[v] <= i
        while (hi - lo > 1) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if (offsets[mid] <= i) lo = mid;
            else hi = mid;
        }
        k.l[0] = lo;
#pragma unroll
        for (int j = 1; j < 8; ++j) k.l[j] = 0;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) k.l[j] = 0xFFFFFFFFu;
    }
    out[i] = k;
}

constexpr uint32_t LK_MIN_TILE = 1024;   // keys per workgroup in the LDS stages: 1024 (32 KiB, 256 lanes) or, from 2^19 keys, 4096 (128 KiB, 1024 lanes)
// all stages (k, j) with j < LK_TILE for k in [k_lo, k_hi] (k_lo = 2: full presort of each tile; k_lo = k_hi = k: the
// tail j = LK_TILE/2 .. 1 of a larger merge step)
template <uint32_t LK_TILE, uint32_t LK_THREADS>
__global__ __launch_bounds__(LK_THREADS) void lk_bitonic_local_kernel(Key256 *__restrict__ keys, uint32_t k_lo, uint32_t k_hi) {
    HIP_DYNAMIC_SHARED(Key256, sh)   // LK_TILE keys
    const uint32_t base = blockIdx.x * LK_TILE, tid = threadIdx.x;
    for (uint32_t e = tid; e < LK_TILE; e += LK_THREADS) sh[e] = keys[base + e];
    __syncthreads();
    for (uint32_t k = k_lo; k <= k_hi; k <<= 1) {
        uint32_t j0 = (k >> 1) < LK_TILE ? (k >> 1) : (LK_TILE >> 1);
        for (uint32_t j = j0; j >= 1; j >>= 1) {
            for (uint32_t p = tid; p < LK_TILE / 2; p += LK_THREADS) {
                uint32_t lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;
                bool up = ((base + lo) & k) == 0;
                Key256 a = sh[lo], b = sh[hi];
                if (key_less(b, a) == up) {
                    sh[lo] = b;
                    sh[hi] = a;
                }
            }
            __syncthreads();
        }
        if (k == 0x80000000u) break;
    }
    for (uint32_t e = tid; e < LK_TILE; e += LK_THREADS) keys[base + e] = sh[e];
}
// one global stage (k, j) with j >= LK_TILE
__global__ __launch_bounds__(256) void lk_bitonic_global_kernel(Key256 *__restrict__ keys, uint32_t N, uint32_t j, uint32_t k) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N / 2) return;
    uint32_t lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;
    bool up = (lo & k) == 0;
    Key256 a = keys[lo], b = keys[hi];
    if (key_less(b, a) == up) {
        keys[lo] = b;
        keys[hi] = a;
    }
}

// first[i] = 1 where sorted A' starts a new run (i < usable), rep[i] = 1 - first[i]
__global__ __launch_bounds__(256) void lk_flags_kernel(const Key256 *__restrict__ ka, uint32_t usable, uint32_t *__restrict__ rep) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > usable) return;
    if (i == usable) {
        rep[i] = 0;
        return;
    }
    rep[i] = (i > 0 && key_eq(ka[i], ka[i - 1])) ? 1u : 0u;
}
// every run start of A' consumes one occurrence of its value from the sorted table: mark the first one
__global__ __launch_bounds__(256) void lk_mark_kernel(const Key256 *__restrict__ ka, const uint32_t *__restrict__ rep, const Key256 *__restrict__ ks,
                                                      uint32_t usable, uint32_t *__restrict__ unused, uint32_t *__restrict__ err) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= usable || rep[i]) return;
    Key256 v = ka[i];
    uint32_t lo = 0, hi = usable;   // lower_bound of v in ks[0..usable)
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (key_less(ks[mid], v)) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= usable || !key_eq(ks[lo], v)) {
        atomicMax(err, i + 1);   // input value missing from the table
        return;
    }
    unused[lo] = 0;
}
__global__ __launch_bounds__(256) void lk_fill_ones_kernel(uint32_t *__restrict__ a, uint32_t n, uint32_t tail_zero) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = 1;
    if (i == n && tail_zero) a[i] = 0;
}
// outputs: A' everywhere; S' = A' on run starts; leftover table elements (ascending) go to the repeated rows from the last one backwards
__global__ __launch_bounds__(256) void lk_emit_kernel(const Key256 *__restrict__ ka, const Key256 *__restrict__ ks, const uint32_t *__restrict__ rep,
                                                      const uint32_t *__restrict__ rep_rank, const uint32_t *__restrict__ unused,
                                                      const uint32_t *__restrict__ unused_rank, uint32_t usable, uint32_t *__restrict__ rep_rows,
                                                      Key256 *__restrict__ leftover, Fr *__restrict__ a_perm, Fr *__restrict__ s_perm) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= usable) return;
    Fr c;
#pragma unroll
    for (int j = 0; j < 8; ++j) c.l[j] = ka[i].l[j];
    Fr m = fe_to_mont(c);
    a_perm[i] = m;
    if (rep[i]) rep_rows[rep_rank[i]] = i;
    else s_perm[i] = m;
    if (unused[i]) leftover[unused_rank[i]] = ks[i];
}
__global__ __launch_bounds__(256) void lk_assign_kernel(const uint32_t *__restrict__ rep_rows, const Key256 *__restrict__ leftover, uint32_t m,
                                                        Fr *__restrict__ s_perm) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    Fr c;
#pragma unroll
    for (int t = 0; t < 8; ++t) c.l[t] = leftover[j].l[t];
    s_perm[rep_rows[m - 1 - j]] = fe_to_mont(c);
}

template <uint32_t LK_TILE, uint32_t LK_THREADS>
static int bitonic_sort_tiled(h2hip_ctx *ctx, Key256 *keys, uint32_t N) {
    hipStream_t st = ctx->stream;
    const uint32_t tiles = N / LK_TILE;
    const size_t lds = sizeof(Key256) * LK_TILE;
    if (!ctx->lookup_lds_attr_set && lds > (64u << 10)) {   // dynamic LDS above 64 KiB has to be enabled per kernel (and device) once
        H2_HIPCHK(hipFuncSetAttribute((const void *)lk_bitonic_local_kernel<LK_TILE, LK_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->lookup_lds_attr_set = true;
    }
    hipLaunchKernelGGL((lk_bitonic_local_kernel<LK_TILE, LK_THREADS>), dim3(tiles), dim3(LK_THREADS), lds, st, keys, 2u, LK_TILE);
    for (uint32_t k = LK_TILE << 1; k <= N && k != 0; k <<= 1) {
        for (uint32_t j = k >> 1; j >= LK_TILE; j >>= 1)
            hipLaunchKernelGGL(lk_bitonic_global_kernel, dim3((N / 2 + 255) / 256), dim3(256), 0, st, keys, N, j, k);
        hipLaunchKernelGGL((lk_bitonic_local_kernel<LK_TILE, LK_THREADS>), dim3(tiles), dim3(LK_THREADS), lds, st, keys, k, k);
    }
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
static int bitonic_sort(h2hip_ctx *ctx, Key256 *keys, uint32_t N) {
    if (N < LK_MIN_TILE) return H2HIP_ERR_INVALID;   // callers pad to at least one tile
    // measured: the 4096-key tile (fewer global passes) wins from 2^19 keys, the 1024-key tile (more workgroups) below
    if (N >= (1u << ctx->lookup_big_tile_bits)) return bitonic_sort_tiled<4096, 1024>(ctx, keys, N);
    return bitonic_sort_tiled<1024, 256>(ctx, keys, N);
}

constexpr uint32_t LK_COUNT_MAX_BINS = 1u << 22;   // 16 MiB of counters

static uint32_t padded_keys(uint32_t u) {
    uint32_t N = LK_MIN_TILE;
    while (N < u) N <<= 1;
    return N;
}
// canonical keys of in[0..u) sorted ascending into keys[0..N) (padding = +infinity): counting sort when every key is small, bitonic otherwise
static int sort_column_keys(h2hip_ctx *ctx, const Fr *in, Key256 *keys, uint32_t u, uint32_t N) {
    hipStream_t st = ctx->stream;
    uint32_t *small_max = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK5, 256, (void **)&small_max));
    H2_HIPCHK(hipMemsetAsync(small_max, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(lk_keys_kernel, dim3((std::max(N, u) + 255) / 256), dim3(256), 0, st, in, keys, u, N, small_max);
    uint32_t host_max = 0;
    H2_HIPCHK(hipMemcpyAsync(&host_max, small_max, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    H2_HIPCHK(hipStreamSynchronize(st));
    if (host_max >= LK_COUNT_MAX_BINS) return bitonic_sort(ctx, keys, N);
    const uint32_t bins = host_max + 1;
    uint32_t *hist = nullptr, *offsets = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_COUNTS, sizeof(uint32_t) * ((size_t)bins + 1), (void **)&hist));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OFFSETS, sizeof(uint32_t) * ((size_t)bins + 2), (void **)&offsets));
    H2_HIPCHK(hipMemsetAsync(hist, 0, sizeof(uint32_t) * ((size_t)bins + 1), st));
    hipLaunchKernelGGL(lk_count_kernel, dim3((u + 255) / 256), dim3(256), 0, st, (const Key256 *)keys, u, hist);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(exclusive_scan_u32(ctx, hist, offsets, bins + 1));
    hipLaunchKernelGGL(lk_expand_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (const uint32_t *)offsets, bins, u, N, keys);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

}  // namespace h2

using namespace h2;

// ks_sorted != nullptr: the table's sorted keys were prepared by h2hip_lookup_table_sort_dev (the table is a fixed column: sort it once)
static int lookup_permute_impl(h2hip_ctx *ctx, const void *a_dev, const void *s_dev, const void *ks_sorted, size_t usable_rows, void *a_perm_dev,
                               void *s_perm_dev) {
    H2_REQUIRE(ctx && (usable_rows == 0 || (a_dev && (s_dev || ks_sorted) && a_perm_dev && s_perm_dev)), "NULL argument");
    H2_REQUIRE(usable_rows < (1u << 28), "too many rows");
    if (!usable_rows) return H2HIP_OK;
    const uint32_t u = (uint32_t)usable_rows;
    const uint32_t N = padded_keys(u);
    hipStream_t st = ctx->stream;
    Key256 *ka, *ks, *leftover;
    uint32_t *flags, *ranks;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK0, sizeof(Key256) * N, (void **)&ka));
    if (ks_sorted) ks = (Key256 *)ks_sorted;
    else H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK1, sizeof(Key256) * N, (void **)&ks));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK2, sizeof(Key256) * (size_t)u, (void **)&leftover));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK3, sizeof(uint32_t) * (3 * (size_t)u + 8), (void **)&flags));   // rep | unused | rep_rows (+ err)
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK4, sizeof(uint32_t) * (2 * (size_t)u + 4), (void **)&ranks));   // rep_rank | unused_rank
    uint32_t *rep = flags, *unused = flags + (u + 1), *rep_rows = flags + 2 * (u + 1), *err = flags + 3 * (size_t)u + 4;
    uint32_t *rep_rank = ranks, *unused_rank = ranks + (u + 1);
    const dim3 gU((u + 256) / 256), blk(256);
    prof_begin(ctx, "lookup_permute_kernels");
    H2_HIPCHK(hipMemsetAsync(err, 0, sizeof(uint32_t), st));
    H2_CHK(sort_column_keys(ctx, (const Fr *)a_dev, ka, u, N));
    if (!ks_sorted) H2_CHK(sort_column_keys(ctx, (const Fr *)s_dev, ks, u, N));
    hipLaunchKernelGGL(lk_flags_kernel, gU, blk, 0, st, (const Key256 *)ka, u, rep);
    hipLaunchKernelGGL(lk_fill_ones_kernel, gU, blk, 0, st, unused, u, 1u);
    hipLaunchKernelGGL(lk_mark_kernel, gU, blk, 0, st, (const Key256 *)ka, (const uint32_t *)rep, (const Key256 *)ks, u, unused, err);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(exclusive_scan_u32(ctx, rep, rep_rank, u + 1));
    H2_CHK(exclusive_scan_u32(ctx, unused, unused_rank, u + 1));
    uint32_t host[3] = {0, 0, 0};
    H2_HIPCHK(hipMemcpyAsync(&host[0], err, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    H2_HIPCHK(hipMemcpyAsync(&host[1], rep_rank + u, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    H2_HIPCHK(hipMemcpyAsync(&host[2], unused_rank + u, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    H2_HIPCHK(hipStreamSynchronize(st));
    if (host[0]) {
        prof_end(ctx);
        set_error("h2hip_lookup_permute_dev: input value (sorted position %u) is missing from the table", host[0] - 1);
        return H2HIP_ERR_INVALID;
    }
    if (host[1] != host[2]) {
        prof_end(ctx);
        set_error("h2hip_lookup_permute_dev: %u repeated rows but %u leftover table elements", host[1], host[2]);
        return H2HIP_ERR_INVALID;
    }
    hipLaunchKernelGGL(lk_emit_kernel, gU, blk, 0, st, (const Key256 *)ka, (const Key256 *)ks, (const uint32_t *)rep, (const uint32_t *)rep_rank,
                       (const uint32_t *)unused, (const uint32_t *)unused_rank, u, rep_rows, leftover, (Fr *)a_perm_dev, (Fr *)s_perm_dev);
    if (host[1])
        hipLaunchKernelGGL(lk_assign_kernel, dim3((host[1] + 255) / 256), blk, 0, st, (const uint32_t *)rep_rows, (const Key256 *)leftover, host[1],
                           (Fr *)s_perm_dev);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

extern "C" {

// a_dev, s_dev: the compressed input / table expressions over the rows of the domain (only rows [0, usable) take part);
// a_perm_dev, s_perm_dev: outputs, rows [0, usable) are written (the caller appends the blinding rows).
// Returns H2HIP_ERR_INVALID ("input value missing from the table") where upstream returns ConstraintSystemFailure.
int h2hip_lookup_permute_dev(h2hip_ctx *ctx, const void *a_dev, const void *s_dev, size_t usable_rows, void *a_perm_dev, void *s_perm_dev) {
    H2_DEVICE_GUARD(ctx);
    return lookup_permute_impl(ctx, a_dev, s_dev, nullptr, usable_rows, a_perm_dev, s_perm_dev);
}
size_t h2hip_lookup_sorted_table_bytes(size_t usable_rows) { return usable_rows < (1u << 28) ? sizeof(Key256) * (size_t)padded_keys((uint32_t)usable_rows) : 0; }
int h2hip_lookup_table_sort_dev(h2hip_ctx *ctx, const void *s_dev, size_t usable_rows, void *sorted_out_dev) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && s_dev && sorted_out_dev && usable_rows >= 1 && usable_rows < (1u << 28), "bad argument");
    const uint32_t u = (uint32_t)usable_rows;
    return sort_column_keys(ctx, (const Fr *)s_dev, (Key256 *)sorted_out_dev, u, padded_keys(u));
}
int h2hip_lookup_permute_presorted_dev(h2hip_ctx *ctx, const void *a_dev, const void *sorted_table_dev, size_t usable_rows, void *a_perm_dev,
                                       void *s_perm_dev) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(sorted_table_dev || usable_rows == 0, "NULL argument");
    return lookup_permute_impl(ctx, a_dev, nullptr, sorted_table_dev, usable_rows, a_perm_dev, s_perm_dev);
}

}  // extern "C"
