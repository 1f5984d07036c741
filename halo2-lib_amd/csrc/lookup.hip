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
#include <vector>

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
// Every kernel of the permutation takes the lookup as blockIdx.y: a proof's lookups (all against the same table, dozens in a wide shape) go
// through each step in ONE launch.  Per-lookup inputs / outputs travel as pointer tables, scratch arrays are strided per lookup.
constexpr uint32_t LK_BATCH = 48;
struct LkIn {
    const Fr *p[LK_BATCH];
};
struct LkTables {
    const Key256 *ks[LK_BATCH];
};
struct LkOut {
    Fr *ap[LK_BATCH], *sp[LK_BATCH];
};
__global__ __launch_bounds__(256) void lk_keys_kernel(LkIn cols, Key256 *__restrict__ keys, uint32_t usable, uint32_t N,
                                                      uint32_t *__restrict__ small_max) {
    __shared__ uint32_t wg_max;
    const Fr *__restrict__ in = cols.p[blockIdx.y];
    keys += (size_t)blockIdx.y * N;
    if (small_max) small_max += blockIdx.y;
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
__global__ __launch_bounds__(256) void lk_count_kernel(const Key256 *__restrict__ keys, uint32_t usable, uint32_t N, uint32_t *__restrict__ hist,
                                                       size_t hist_stride) {
    __shared__ uint32_t zeros;   // key 0 is by far the most frequent one (rows without a lookup): one global atomic per workgroup for it
    keys += (size_t)blockIdx.y * N;
    hist += (size_t)blockIdx.y * hist_stride;
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
__global__ __launch_bounds__(256) void lk_expand_kernel(const uint32_t *__restrict__ offsets, size_t offsets_stride, uint32_t bins, uint32_t usable,
                                                        uint32_t N, Key256 *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    offsets += (size_t)blockIdx.y * offsets_stride;
    out += (size_t)blockIdx.y * N;
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
__global__ __launch_bounds__(256) void lk_flags_kernel(const Key256 *__restrict__ ka, uint32_t usable, uint32_t N, uint32_t *__restrict__ rep,
                                                       size_t flag_stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > usable) return;
    ka += (size_t)blockIdx.y * N;
    rep += (size_t)blockIdx.y * flag_stride;
    if (i == usable) {
        rep[i] = 0;
        rep[3 * (size_t)usable + 4] = 0;   // the lookup's missing-value marker (lk_mark_kernel's err) lives in the same scratch row
        return;
    }
    rep[i] = (i > 0 && key_eq(ka[i], ka[i - 1])) ? 1u : 0u;
}
// every run start of A' consumes one occurrence of its value from the sorted table: mark the first one
__global__ __launch_bounds__(256) void lk_mark_kernel(const Key256 *__restrict__ ka, const uint32_t *__restrict__ rep, LkTables tables,
                                                      uint32_t usable, uint32_t N, uint32_t *__restrict__ unused, uint32_t *__restrict__ err,
                                                      size_t flag_stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    ka += (size_t)blockIdx.y * N;
    rep += (size_t)blockIdx.y * flag_stride;
    unused += (size_t)blockIdx.y * flag_stride;
    err += (size_t)blockIdx.y * flag_stride;
    const Key256 *__restrict__ ks = tables.ks[blockIdx.y];
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
__global__ __launch_bounds__(256) void lk_fill_ones_kernel(uint32_t *__restrict__ a, uint32_t n, uint32_t tail_zero, size_t stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    a += (size_t)blockIdx.y * stride;
    if (i < n) a[i] = 1;
    if (i == n && tail_zero) a[i] = 0;
}
// outputs: A' everywhere; S' = A' on run starts; leftover table elements (ascending) go to the repeated rows from the last one backwards
__global__ __launch_bounds__(256) void lk_emit_kernel(const Key256 *__restrict__ ka, LkTables tables, const uint32_t *__restrict__ rep,
                                                      const uint32_t *__restrict__ rep_rank, const uint32_t *__restrict__ unused,
                                                      const uint32_t *__restrict__ unused_rank, uint32_t usable, uint32_t N,
                                                      uint32_t *__restrict__ rep_rows, Key256 *__restrict__ leftover, LkOut out, size_t flag_stride,
                                                      size_t rank_stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= usable) return;
    const uint32_t y = blockIdx.y;
    ka += (size_t)y * N;
    rep += (size_t)y * flag_stride;
    unused += (size_t)y * flag_stride;
    rep_rows += (size_t)y * flag_stride;
    rep_rank += (size_t)y * rank_stride;
    unused_rank += (size_t)y * rank_stride;
    leftover += (size_t)y * usable;
    const Key256 *__restrict__ ks = tables.ks[y];
    Fr *__restrict__ a_perm = out.ap[y];
    Fr *__restrict__ s_perm = out.sp[y];
    Fr c;
#pragma unroll
    for (int j = 0; j < 8; ++j) c.l[j] = ka[i].l[j];
    Fr m = fe_to_mont(c);
    a_perm[i] = m;
    if (rep[i]) rep_rows[rep_rank[i]] = i;
    else s_perm[i] = m;
    if (unused[i]) leftover[unused_rank[i]] = ks[i];
}
// (m = number of repeated rows of the lookup = rep_rank[usable], read on the device)
__global__ __launch_bounds__(256) void lk_assign_kernel(const uint32_t *__restrict__ rep_rows, const Key256 *__restrict__ leftover,
                                                        const uint32_t *__restrict__ rep_rank, uint32_t usable, LkOut out, size_t flag_stride,
                                                        size_t rank_stride) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y;
    const uint32_t m = rep_rank[(size_t)y * rank_stride + usable];
    if (j >= m) return;
    rep_rows += (size_t)y * flag_stride;
    leftover += (size_t)y * usable;
    Fr *__restrict__ s_perm = out.sp[y];
    Fr c;
#pragma unroll
    for (int t = 0; t < 8; ++t) c.l[t] = leftover[j].l[t];
    s_perm[rep_rows[m - 1 - j]] = fe_to_mont(c);
}

// out[3y .. 3y+3) = (missing-value marker, repeated rows, leftover table elements) of lookup y: one small copy for the host's checks
__global__ void lk_status_kernel(const uint32_t *__restrict__ err, const uint32_t *__restrict__ rep_rank, const uint32_t *__restrict__ unused_rank,
                                 uint32_t usable, uint32_t count, size_t flag_stride, size_t rank_stride, uint32_t *__restrict__ out) {
    uint32_t y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= count) return;
    out[3 * y] = err[(size_t)y * flag_stride];
    out[3 * y + 1] = rep_rank[(size_t)y * rank_stride + usable];
    out[3 * y + 2] = unused_rank[(size_t)y * rank_stride + usable];
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
// canonical keys of `count` columns in[j][0..u) sorted ascending into keys[j*N .. j*N + N) (padding = +infinity): ONE counting sort over all
// columns when every key is small (one host synchronisation for the batch), the bitonic network column by column otherwise
static int sort_columns_keys(h2hip_ctx *ctx, const Fr *const *in, uint32_t count, Key256 *keys, uint32_t u, uint32_t N) {
    hipStream_t st = ctx->stream;
    H2_REQUIRE(count >= 1 && count <= LK_BATCH, "1..48 columns per sort batch");
    uint32_t *small_max = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK5, 256 + sizeof(uint32_t) * 4 * LK_BATCH, (void **)&small_max));
    H2_HIPCHK(hipMemsetAsync(small_max, 0, sizeof(uint32_t) * count, st));
    LkIn cols;
    for (uint32_t j = 0; j < LK_BATCH; ++j) cols.p[j] = in[j < count ? j : 0];
    hipLaunchKernelGGL(lk_keys_kernel, dim3((std::max(N, u) + 255) / 256, count), dim3(256), 0, st, cols, keys, u, N, small_max);
    uint32_t host_max[LK_BATCH];
    H2_CHK(sync_results(ctx, host_max, small_max, sizeof(uint32_t) * count));
    uint32_t all_max = 0;
    for (uint32_t j = 0; j < count; ++j) all_max = std::max(all_max, host_max[j]);
    const uint32_t bins = all_max + 1;
    // one histogram per column: fall back to column-by-column work when the keys are large or the histograms would not fit comfortably
    const bool batched = all_max < LK_COUNT_MAX_BINS && (uint64_t)count * ((uint64_t)bins + 2) * 8 <= (1ull << 30);
    if (!batched) {
        for (uint32_t j = 0; j < count; ++j) {
            Key256 *kj = keys + (size_t)N * j;
            if (host_max[j] >= LK_COUNT_MAX_BINS) {
                H2_CHK(bitonic_sort(ctx, kj, N));
                continue;
            }
            const uint32_t bj = host_max[j] + 1;
            uint32_t *hist = nullptr, *offsets = nullptr;
            H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_COUNTS, sizeof(uint32_t) * ((size_t)bj + 1), (void **)&hist));
            H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OFFSETS, sizeof(uint32_t) * ((size_t)bj + 2), (void **)&offsets));
            H2_HIPCHK(hipMemsetAsync(hist, 0, sizeof(uint32_t) * ((size_t)bj + 1), st));
            hipLaunchKernelGGL(lk_count_kernel, dim3((u + 255) / 256, 1), dim3(256), 0, st, (const Key256 *)kj, u, N, hist, (size_t)0);
            H2_HIPCHK(hipGetLastError());
            H2_CHK(exclusive_scan_u32(ctx, hist, offsets, bj + 1));
            hipLaunchKernelGGL(lk_expand_kernel, dim3((N + 255) / 256, 1), dim3(256), 0, st, (const uint32_t *)offsets, (size_t)0, bj, u, N, kj);
            H2_HIPCHK(hipGetLastError());
        }
        return H2HIP_OK;
    }
    const size_t hs = (size_t)bins + 1, os = (size_t)bins + 2;
    uint32_t *hist = nullptr, *offsets = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_COUNTS, sizeof(uint32_t) * hs * count, (void **)&hist));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OFFSETS, sizeof(uint32_t) * os * count, (void **)&offsets));
    H2_HIPCHK(hipMemsetAsync(hist, 0, sizeof(uint32_t) * hs * count, st));
    hipLaunchKernelGGL(lk_count_kernel, dim3((u + 255) / 256, count), dim3(256), 0, st, (const Key256 *)keys, u, N, hist, hs);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(exclusive_scan_u32_segments(ctx, hist, offsets, bins + 1, count, hs, os));
    hipLaunchKernelGGL(lk_expand_kernel, dim3((N + 255) / 256, count), dim3(256), 0, st, (const uint32_t *)offsets, os, bins, u, N, keys);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
static int sort_column_keys(h2hip_ctx *ctx, const Fr *in, Key256 *keys, uint32_t u, uint32_t N) { return sort_columns_keys(ctx, &in, 1, keys, u, N); }

}  // namespace h2

using namespace h2;

// ks_sorted != nullptr: the table's sorted keys were prepared by h2hip_lookup_table_sort_dev (the table is a fixed column: sort it once)
// `count` (input, table) pairs: every step is one launch over all pairs (chunks of 48), with two host synchronisations per chunk (the sort's
// key-range probe, the multiset checks) instead of two per pair.  ks_sorted[j] (optional per pair): the table's sorted keys.
static int lookup_permute_many(h2hip_ctx *ctx, const void *const *a_dev, const void *const *s_dev, const void *const *ks_sorted, size_t count,
                               size_t usable_rows, void *const *a_perm_dev, void *const *s_perm_dev) {
    H2_REQUIRE(ctx && (count == 0 || usable_rows == 0 || (a_dev && (s_dev || ks_sorted) && a_perm_dev && s_perm_dev)), "NULL argument");
    H2_REQUIRE(usable_rows < (1u << 28), "too many rows");
    if (!usable_rows || !count) return H2HIP_OK;
    for (size_t j = 0; j < count; ++j)
        H2_REQUIRE(a_dev[j] && a_perm_dev[j] && s_perm_dev[j] && ((ks_sorted && ks_sorted[j]) || (s_dev && s_dev[j])), "NULL argument");
    const uint32_t u = (uint32_t)usable_rows;
    const uint32_t N = padded_keys(u);
    hipStream_t st = ctx->stream;
    const size_t flag_words = 3 * (size_t)u + 8, rank_words = 2 * (size_t)u + 4;
    const dim3 blk(256);
    prof_begin(ctx, "lookup_permute_kernels");
    for (size_t c0 = 0; c0 < count; c0 += LK_BATCH) {
        const uint32_t cc = (uint32_t)(count - c0 < LK_BATCH ? count - c0 : LK_BATCH);
        Key256 *ka, *ks_own = nullptr, *leftover;
        uint32_t *flags, *ranks, *status;
        bool any_unsorted = false;
        for (uint32_t j = 0; j < cc; ++j) any_unsorted |= !(ks_sorted && ks_sorted[c0 + j]);
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK0, sizeof(Key256) * N * cc, (void **)&ka));
        if (any_unsorted) H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK1, sizeof(Key256) * N * cc, (void **)&ks_own));
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK2, sizeof(Key256) * (size_t)u * cc, (void **)&leftover));
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK3, sizeof(uint32_t) * flag_words * cc, (void **)&flags));   // per pair: rep | unused | rep_rows (+ err)
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_LK4, sizeof(uint32_t) * (rank_words * cc + 3 * LK_BATCH), (void **)&ranks));   // per pair: rep_rank | unused_rank
        status = ranks + rank_words * cc;
        uint32_t *rep = flags, *unused = flags + (u + 1), *rep_rows = flags + 2 * (u + 1), *err = flags + 3 * (size_t)u + 4;
        uint32_t *rep_rank = ranks, *unused_rank = ranks + (u + 1);
        LkTables tables;
        LkOut outs;
        std::vector<const Fr *> ins(cc);
        for (uint32_t j = 0; j < LK_BATCH; ++j) {
            const size_t t = c0 + (j < cc ? j : 0);
            tables.ks[j] = (ks_sorted && ks_sorted[t]) ? (const Key256 *)ks_sorted[t] : ks_own + (size_t)N * (j < cc ? j : 0);
            outs.ap[j] = (Fr *)a_perm_dev[t];
            outs.sp[j] = (Fr *)s_perm_dev[t];
            if (j < cc) ins[j] = (const Fr *)a_dev[t];
        }
        H2_CHK(sort_columns_keys(ctx, ins.data(), cc, ka, u, N));
        for (uint32_t j = 0; j < cc; ++j)
            if (!(ks_sorted && ks_sorted[c0 + j])) H2_CHK(sort_column_keys(ctx, (const Fr *)s_dev[c0 + j], ks_own + (size_t)N * j, u, N));
        const dim3 gU((u + 256) / 256, cc);
        hipLaunchKernelGGL(lk_flags_kernel, gU, blk, 0, st, (const Key256 *)ka, u, N, rep, flag_words);
        hipLaunchKernelGGL(lk_fill_ones_kernel, gU, blk, 0, st, unused, u, 1u, flag_words);
        hipLaunchKernelGGL(lk_mark_kernel, gU, blk, 0, st, (const Key256 *)ka, (const uint32_t *)rep, tables, u, N, unused, err, flag_words);
        H2_HIPCHK(hipGetLastError());
        H2_CHK(exclusive_scan_u32_segments(ctx, rep, rep_rank, u + 1, cc, flag_words, rank_words));
        H2_CHK(exclusive_scan_u32_segments(ctx, unused, unused_rank, u + 1, cc, flag_words, rank_words));
        hipLaunchKernelGGL(lk_status_kernel, dim3(1), dim3(64), 0, st, (const uint32_t *)err, (const uint32_t *)rep_rank, (const uint32_t *)unused_rank, u,
                           cc, flag_words, rank_words, status);
        uint32_t host[3 * LK_BATCH];
        H2_CHK(sync_results(ctx, host, status, sizeof(uint32_t) * 3 * cc));
        for (uint32_t j = 0; j < cc; ++j) {
            if (host[3 * j]) {
                prof_end(ctx);
                set_error("h2hip_lookup_permute_dev: input value (sorted position %u) is missing from the table", host[3 * j] - 1);
                return H2HIP_ERR_INVALID;
            }
            if (host[3 * j + 1] != host[3 * j + 2]) {
                prof_end(ctx);
                set_error("h2hip_lookup_permute_dev: %u repeated rows but %u leftover table elements", host[3 * j + 1], host[3 * j + 2]);
                return H2HIP_ERR_INVALID;
            }
        }
        hipLaunchKernelGGL(lk_emit_kernel, gU, blk, 0, st, (const Key256 *)ka, tables, (const uint32_t *)rep, (const uint32_t *)rep_rank,
                           (const uint32_t *)unused, (const uint32_t *)unused_rank, u, N, rep_rows, leftover, outs, flag_words, rank_words);
        hipLaunchKernelGGL(lk_assign_kernel, gU, blk, 0, st, (const uint32_t *)rep_rows, (const Key256 *)leftover, (const uint32_t *)rep_rank, u, outs,
                           flag_words, rank_words);
        H2_HIPCHK(hipGetLastError());
    }
    prof_end(ctx);
    return H2HIP_OK;
}
static int lookup_permute_impl(h2hip_ctx *ctx, const void *a_dev, const void *s_dev, const void *ks_sorted, size_t usable_rows, void *a_perm_dev,
                               void *s_perm_dev) {
    H2_REQUIRE(ctx && (usable_rows == 0 || (a_dev && (s_dev || ks_sorted) && a_perm_dev && s_perm_dev)), "NULL argument");
    return lookup_permute_many(ctx, &a_dev, s_dev ? &s_dev : nullptr, ks_sorted ? &ks_sorted : nullptr, 1, usable_rows, &a_perm_dev, &s_perm_dev);
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

// several input columns against ONE presorted table (every range lookup of a halo2-base circuit reads the same table column): one host
// synchronisation for the whole batch.  a_dev / a_perm_dev / s_perm_dev: HOST arrays of `count` device pointers.
int h2hip_lookup_permute_presorted_batch_dev(h2hip_ctx *ctx, const void *const *a_dev, const void *sorted_table_dev, size_t usable_rows,
                                             void *const *a_perm_dev, void *const *s_perm_dev, size_t count) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(sorted_table_dev || usable_rows == 0 || count == 0, "NULL argument");
    std::vector<const void *> tables(count, sorted_table_dev);
    return lookup_permute_many(ctx, a_dev, nullptr, tables.data(), count, usable_rows, a_perm_dev, s_perm_dev);
}

}  // extern "C"
