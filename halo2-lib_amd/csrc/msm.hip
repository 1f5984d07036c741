// Pippenger multi-scalar multiplication over BN254 G1 for gfx950 — the device replacement for
// halo2_proofs::arithmetic::best_multiexp [UPSTREAM halo2-axiom 0.5.3 / halo2curves-axiom 0.7.3; the
// reference reaches it only through commit / commit_lagrange inside create_proof,
// /root/reference/halo2-base/src/utils/testing.rs:40-47; SURVEY.md §3.2 K1].
//
// The result (one group element) is unique, so the algorithm is free to differ from the CPU one:
//   1. msm_digits     one lane per scalar: Montgomery -> canonical, signed c-bit digits -> digits[w][i]
//   2. msm_hist       counting sort without global atomics: workgroup (w, g) histograms chunk g of window w
//                     in a *full-window* LDS histogram (2^15 x u32 = 128 KiB of the CU's 160 KiB).  Window w is
//                     pinned to XCD (w mod 8) so its sorted segment and cursors live in one L2.
//   3. msm_hist_scan  per (window, bucket) exclusive prefix over chunks + bucket totals; global scan -> offsets
//   4. msm_scatter    workgroup (w, g) loads its cursors into LDS, ranks entries with LDS atomics, writes
//                     (base index | sign) into bucket order
//   5. msm_accum      *distribution-oblivious* bucket accumulation: every lane owns K consecutive sorted
//                     entries (perfect balance for uniform and for 0/1-heavy circuit columns alike), walks the
//                     offsets table for bucket boundaries, adds in XYZZ on unsaturated 9x29-bit limbs (ec29.cuh),
//                     writes complete interior runs to their bucket and emits its first/last (possibly shared) runs
//                     as (key, XYZZ) partials.  Bases are gathered as aligned, packed 64-byte entries.
//   6. msm_merge      block-level segmented scan over the sorted partial list (early exit once no lane merges);
//                     runs closed inside a block go to buckets, block-crossing runs to the next, 128x shorter level
//   7. msm_presum/seg/winsum/fold   bucket reduction: per-window sum_j j*B_j over short segments (+ small
//                     scalar multiple), tree sums, 2^(c*w) fold (skipped for precomputed bases).  These stages are
//                     chains of dependent point additions; when few segments exist (precomputed bases) and for the
//                     fold they run on quad-lane point arithmetic (quad29.cuh: one point per 4 lanes, 4 concurrent
//                     field products per level, operands exchanged by DPP quad permutes).
// Bases stay resident in HBM (h2hip_bases); with H2HIP_BASES_PRECOMPUTE the table also holds 2^(c*w)*P_i for
// every window (16x the memory — sized for 288 GB HBM), so all windows share one bucket set per index and the
// serial 2^(c*w) fold disappears.  Signs are applied inside the mixed addition (no negated copy of y).
#include "internal.h"
#ifndef H2_HIPEMU
#include <hip/hip_ext.h>
#endif
#include "ec29.cuh"
#include "quad29.cuh"

namespace h2 {

// The latency-bound tail kernels (a few waves of dependent point additions) run next to another MSM's multiplier-bound
// accumulation when MSMs are pipelined over lanes: raise their wave priority so the SIMD arbiter issues them first.
// The LDS-histogram sort kernels take their 128 KiB as DYNAMIC shared memory: with a static array that leaves room for
// one workgroup per CU the compiler pads the kernel's register allocation to 96 VGPRs per lane ("occupancy is LDS-bound
// anyway"), and a 1024-lane workgroup (4 waves per SIMD) then never fits next to three resident accumulation waves
// (3 x 144 of 512 registers) — the sort of MSM i+1 waited for the accumulation of MSM i to drain (rocprofv3 timeline,
// profiles/archive/r01_pipeline_timeline_*.md).  With dynamic LDS the kernels allocate the 8-16 registers they use.
#ifdef H2_HIPEMU
#define H2_TAIL_PRIORITY() ((void)0)
#define H2_SORT_PRIORITY() ((void)0)
#else
#define H2_TAIL_PRIORITY() __builtin_amdgcn_s_setprio(3)
#define H2_SORT_PRIORITY() __builtin_amdgcn_s_setprio(2)   // digits / LDS-histogram sort of the next MSM: issue-light, latency-bound on LDS atomics
#endif


constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;
constexpr uint32_t MAX_LDS_BUCKETS = 1u << 15;   // 128 KiB of u32 counters

// ------------------------------------------------------------------ 1. digits
// blockIdx.y = column of a fused multi-column MSM (column col's digits follow column col-1's: window col*W + w)
struct DigitCols {
    const Fr *scalars[MSM_MAX_COLS];
};
// (the first workgroup also writes the sort's two sentinels — counts[nsort] = 0 and offsets[nsort + 1] = ~0, read by the scan and by the
// accumulation's boundary walk — which were two tiny memset launches per MSM on the lane's critical path)
// r06: a digit travels as a 16-BIT CODE — 0 = no entry, 1 .. B = +d, B + 1 .. 2B - 1 = -d (a negative digit is at most B - 1: v = B stays positive),
// 2B - 1 = 65535 at c = 16 — so the digits array, written once and read by the histogram and by the scatter, moves half the bytes
typedef uint16_t digit_t;
__device__ __forceinline__ uint32_t digit_bucket(uint32_t code, uint32_t B) { return code > B ? code - B : code; }   // 1-based bucket of a non-zero code
__global__ __launch_bounds__(256) void msm_digits_kernel(DigitCols cols, uint32_t n, uint32_t c, uint32_t W, digit_t *__restrict__ digits,
                                                         uint32_t *__restrict__ counts_tail, uint32_t *__restrict__ offsets_tail) {
    H2_SORT_PRIORITY();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        counts_tail[0] = 0;
        offsets_tail[0] = 0xFFFFFFFFu;
    }
    if (i >= n) return;
    digits += (size_t)blockIdx.y * W * n;
    Fr s = fe_from_mont(cols.scalars[blockIdx.y][i]);
    const uint32_t B = 1u << (c - 1);
    const uint64_t mask = (1ull << c) - 1;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; ++w) {
        uint32_t bit = w * c, limb = bit >> 5, off = bit & 31;
        uint64_t lo = 0, hi = 0;
        // static selection keeps s in registers (a runtime-indexed array would live in scratch)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((uint32_t)k == limb) lo = s.l[k];
            if ((uint32_t)k == limb + 1) hi = s.l[k];
        }
        uint32_t v = (uint32_t)((((hi << 32) | lo) >> off) & mask) + carry;
        uint32_t neg = v > B ? 1u : 0u;
        uint32_t d = neg ? (1u << c) - v : v;
        carry = neg;
        digits[(size_t)w * n + i] = (digit_t)((neg && d) ? B + d : d);   // (v = 2^c: digit 0 with a carry — no entry)
    }
}

// workgroup -> (window, chunk).  Workgroups are dealt round-robin to the 8 XCDs, each XCD works through its own sequence in order: the first
// 8 * floor(total / 8) windows are pinned — window w lives on XCD (w mod 8), its G chunk workgroups run together there, one window after the
// other — so that a window's slice of the sorted array stays in ONE 4 MiB L2 while it is written.  The total % 8 windows left over (17 windows:
// one) are dealt chunk by chunk over all XCDs: pinned too, the 17th window was a third round on XCD 0 alone with seven XCDs idle (2.1 rounds of
// work in the time of 3); its 4-byte writes now combine per XCD only (1/17 of the entries).
H2_HD uint32_t sort_grid_size(uint32_t total, uint32_t G) {
    const uint32_t full = total / 8u, rem = total - 8u * full;
    return 8u * (full * G + (rem * G + 7u) / 8u);
}
__device__ __forceinline__ void block_to_window_chunk(uint32_t L, uint32_t G, uint32_t total, uint32_t &w, uint32_t &g) {
    const uint32_t x = L & 7u, q = L >> 3, full = total / 8u;
    if (q < full * G) {
        w = x + 8u * (q / G);
        g = q % G;
        return;
    }
    const uint32_t p = (q - full * G) * 8u + x;   // pair index among the leftover windows' (window, chunk) pairs
    w = 8u * full + p / G;                        // >= total when p runs past the last pair: the caller returns
    g = p % G;
}

// ------------------------------------------------------------------ 2. per-(window, chunk) LDS histogram
// PACKED (r06): two 16-bit counters per LDS word (a chunk holds < 2^16 scalars, so no counter can carry into its neighbour): the histogram of a
// c = 15 window is 32 KiB instead of 64 — small enough for the slot ONE retiring accumulation workgroup leaves on a CU (160 - 2 x 36 KiB, its
// registers), where the 64 KiB histogram had to wait for a CU without any accumulation workgroup (one timed k = 19 proof: 105 us per launch
// against 16 us alone, profiles/r06_bench_proof_k19_kernels.md).
template <bool PACKED>
__global__ __launch_bounds__(1024) void msm_hist_kernel(const digit_t *__restrict__ digits, uint32_t n, uint32_t W, uint32_t B,
                                                        uint32_t G, uint32_t chunk, uint32_t S, uint32_t *__restrict__ bhist) {
    H2_SORT_PRIORITY();
    HIP_DYNAMIC_SHARED(uint32_t, hist)   // B / S counters (PACKED: half as many words)
    // S > 1 (r06): a workgroup counts one bucket SUB-RANGE of its (window, chunk) — the chunk's digits are read S times (2 bytes each), the histogram
    // is 1 / S the size: at c = 16 two sub-ranges of 32 KiB each fit the slot a retiring accumulation workgroup leaves, the whole 64 KiB did not
    uint32_t seg, g;
    block_to_window_chunk(blockIdx.x, G, W * S, seg, g);
    if (seg >= W * S) return;
    const uint32_t w = seg / S, h = seg - w * S, Bs = B / S, b0 = h * Bs;
    const uint32_t words = PACKED ? (Bs + 1) >> 1 : Bs;
    for (uint32_t b = threadIdx.x; b < words; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    uint32_t lo = g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const digit_t *dw = digits + (size_t)w * n;
    // eight independent loads in flight per lane, then the LDS atomics: one load -> one atomic per iteration left the kernel latency-bound
    const uint32_t T = blockDim.x;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 8 * T) {
        uint32_t d[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) d[k] = i + k * T < hi ? (uint32_t)dw[i + k * T] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t b = digit_bucket(d[k], B) - 1 - b0;   // code 0 wraps to a huge b: skipped like another sub-range's digit
            if (d[k] && b < Bs) {
                if (PACKED) atomicAdd(&hist[b >> 1], 1u << (16u * (b & 1u)));
                else atomicAdd(&hist[b], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *out = bhist + ((size_t)w * G + g) * B + b0;
    for (uint32_t b = threadIdx.x; b < Bs; b += blockDim.x) out[b] = PACKED ? (hist[b >> 1] >> (16u * (b & 1u))) & 0xFFFFu : hist[b];
}

// ------------------------------------------------------------------ 3. prefix over chunks per (window, bucket)
__global__ __launch_bounds__(256) void msm_hist_scan_kernel(uint32_t *__restrict__ bhist, uint32_t W, uint32_t B, uint32_t G, uint32_t *__restrict__ counts) {
    H2_SORT_PRIORITY();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * B) return;
    uint32_t w = t / B, b = t - w * B;
    uint32_t run = 0;
    for (uint32_t g0 = 0; g0 < G; g0 += 8) {   // eight independent loads in flight, then the stores (the array is read and written in place)
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) v[k] = g0 + k < G ? bhist[((size_t)w * G + g0 + k) * B + b] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            if (g0 + k < G) bhist[((size_t)w * G + g0 + k) * B + b] = run;
            run += v[k];
        }
    }
    counts[t] = run;   // key = w * B + b
}

// exclusive scan of u32 (3 kernels)
constexpr uint32_t SCAN_TILE = 1024;   // 256 lanes x 4
// (blockIdx.y = segment: independent scans of n elements each, in_stride / out_stride elements apart)
__global__ __launch_bounds__(256) void scan_tile_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                        uint32_t *__restrict__ tile_sums, uint32_t n, size_t in_stride, size_t out_stride) {
    H2_SORT_PRIORITY();
    __shared__ uint32_t sh[256];
    in += (size_t)blockIdx.y * in_stride;
    out += (size_t)blockIdx.y * out_stride;
    tile_sums += (size_t)blockIdx.y * gridDim.x;
    uint32_t tid = threadIdx.x, base = blockIdx.x * SCAN_TILE + tid * 4;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        sum += v[k];
    }
    sh[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t t = (tid >= d) ? sh[tid - d] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    uint32_t excl = sh[tid] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) out[base + k] = excl;
        excl += v[k];
    }
    if (tid == 255) tile_sums[blockIdx.x] = sh[255];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t *__restrict__ tile_sums, uint32_t ntiles) {
    H2_SORT_PRIORITY();
    __shared__ uint32_t sh[1024];
    tile_sums += (size_t)blockIdx.x * ntiles;   // one workgroup per segment
    uint32_t tid = threadIdx.x;
    uint32_t per = (ntiles + 1023) / 1024, lo = tid * per, hi = lo + per < ntiles ? lo + per : ntiles;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += tile_sums[k];
    sh[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t t = (tid >= d) ? sh[tid - d] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    uint32_t excl = sh[tid] - sum;
    for (uint32_t k = lo; k < hi; ++k) {
        uint32_t t = tile_sums[k];
        tile_sums[k] = excl;
        excl += t;
    }
}
__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ tile_sums, uint32_t n, size_t out_stride) {
    H2_SORT_PRIORITY();
    out += (size_t)blockIdx.y * out_stride;
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4, add = tile_sums[(size_t)blockIdx.y * gridDim.x + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) out[base + k] += add;
}
int exclusive_scan_u32_segments(h2hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t segments, size_t in_stride, size_t out_stride) {
    if (!n || !segments) return H2HIP_OK;
    H2_REQUIRE(segments <= 65535, "too many scan segments");
    uint32_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t *sums = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SCAN, sizeof(uint32_t) * ((size_t)ntiles * segments + 1), (void **)&sums));
    prof_begin(ctx, "scan_kernels");
    hipLaunchKernelGGL(scan_tile_kernel, dim3(ntiles, segments), dim3(256), 0, ctx->stream, in, out, sums, n, in_stride, out_stride);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(segments), dim3(1024), 0, ctx->stream, sums, ntiles);
    hipLaunchKernelGGL(scan_add_kernel, dim3(ntiles, segments), dim3(256), 0, ctx->stream, out, (const uint32_t *)sums, n, out_stride);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
int exclusive_scan_u32(h2hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n) { return exclusive_scan_u32_segments(ctx, in, out, n, 1, 0, 0); }

// ------------------------------------------------------------------ 4. scatter with LDS cursors
// Each workgroup scatters the entries of chunk g of window w whose bucket lies in sub-range h of S: all workgroups of
// one (window, sub-range) segment run together on one XCD (1 workgroup per CU because of the LDS cursors), and the
// segment's slice of the sorted array (n*4/S bytes) fits that XCD's 4 MiB L2, so the 4-byte writes combine there
// instead of each costing a 64-byte HBM write.
__global__ __launch_bounds__(1024) void msm_scatter_kernel(const digit_t *__restrict__ digits, uint32_t n, uint32_t W, uint32_t B,
                                                           uint32_t G, uint32_t chunk, uint32_t S, uint32_t table_stride, uint32_t Wcol,
                                                           const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ bhist,
                                                           uint32_t *__restrict__ sval) {
    H2_SORT_PRIORITY();
    HIP_DYNAMIC_SHARED(uint32_t, cursor)   // B / S cursors
    uint32_t seg, g;
    block_to_window_chunk(blockIdx.x, G, W * S, seg, g);
    if (seg >= W * S) return;
    const uint32_t w = seg / S, h = seg - w * S;
    const uint32_t Bs = B / S, b0 = h * Bs;   // this workgroup's bucket range [b0, b0 + Bs)
    const uint32_t *bh = bhist + ((size_t)w * G + g) * B + b0;
    for (uint32_t b = threadIdx.x; b < Bs; b += blockDim.x) cursor[b] = offsets[w * B + b0 + b] + bh[b];
    __syncthreads();
    uint32_t lo = g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const digit_t *dw = digits + (size_t)w * n;
    const uint32_t idx_base = (w % Wcol) * table_stride;   // precomputed bases: window w of a column reads table level w
    // eight entries per lane and round: the loads, then the returning LDS atomics, then the stores — each group independent, so the
    // latencies of a group overlap instead of adding up per entry
    const uint32_t T = blockDim.x;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 8 * T) {
        uint32_t dv[8], pos[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) dv[k] = i + k * T < hi ? (uint32_t)dw[i + k * T] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t d = digit_bucket(dv[k], B), b = d - 1 - b0;   // code 0 wraps to a huge b: skipped like another sub-range's entry
            pos[k] = (d && b < Bs) ? atomicAdd(&cursor[b], 1u) : KEY_INVALID;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            if (pos[k] != KEY_INVALID) sval[pos[k]] = (idx_base + i + k * T) | (dv[k] > B ? 0x80000000u : 0u);
    }
}

// ------------------------------------------------------------------ 5. chunked bucket accumulation
// largest k in [lo, nkeys) with offsets[k] <= e   (offsets has nkeys+1 entries, e < offsets[nkeys])
__device__ __forceinline__ uint32_t find_key(const uint32_t *__restrict__ offsets, uint32_t lo, uint32_t nkeys, uint32_t e) {
    uint32_t hi = nkeys;
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] <= e) lo = mid;
        else hi = mid;
    }
    return lo;
}

// One aligned 64-byte table entry.  The gather has no reuse (a 1 GiB table, one entry per addition): loaded NON-TEMPORALLY it streams past
// the L2 instead of evicting the lines the kernel does reuse — each lane's slice of the sorted entry list (32 entries per 128-byte line,
// touched over ~32 additions) and the bucket offsets.  rocprofv3 PMC (profiles/archive/r02_hbm_counter_calibration.md): with plain loads the
// kernel issued 2.0 memory-side line requests per addition, one of them a re-fetch of such an evicted line.
__device__ __forceinline__ G1Affine load_table_entry(const G1Affine *__restrict__ p) {
#ifdef H2_HIPEMU
    return *p;
#else
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u *q = reinterpret_cast<const v4u *>(p);
    v4u a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1), c = __builtin_nontemporal_load(q + 2),
        d = __builtin_nontemporal_load(q + 3);
    G1Affine r;
    r.x.l[0] = a.x; r.x.l[1] = a.y; r.x.l[2] = a.z; r.x.l[3] = a.w;
    r.x.l[4] = b.x; r.x.l[5] = b.y; r.x.l[6] = b.z; r.x.l[7] = b.w;
    r.y.l[0] = c.x; r.y.l[1] = c.y; r.y.l[2] = c.z; r.y.l[3] = c.w;
    r.y.l[4] = d.x; r.y.l[5] = d.y; r.y.l[6] = d.z; r.y.l[7] = d.w;
    return r;
#endif
}
// the pre-split entry (TableEntry29): 72 bytes of its 128-byte line, four 16-byte loads and one of 8, the limbs as the point addition takes them
__device__ __forceinline__ bool load_table_entry_split(const TableEntry29 *__restrict__ p, Fq29 &x, Fq29 &y) {
#ifdef H2_HIPEMU
    for (int i = 0; i < 9; ++i) {
        x.l[i] = p->x[i];
        y.l[i] = p->y[i];
    }
#else
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    const v4u *q = reinterpret_cast<const v4u *>(p);
    const v4u a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1), c = __builtin_nontemporal_load(q + 2),
              d = __builtin_nontemporal_load(q + 3);
    const v2u e = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(q + 4));
    x.l[0] = a.x; x.l[1] = a.y; x.l[2] = a.z; x.l[3] = a.w; x.l[4] = b.x; x.l[5] = b.y; x.l[6] = b.z; x.l[7] = b.w; x.l[8] = c.x;
    y.l[0] = c.y; y.l[1] = c.z; y.l[2] = c.w; y.l[3] = d.x; y.l[4] = d.y; y.l[5] = d.z; y.l[6] = d.w; y.l[7] = e.x; y.l[8] = e.y;
#endif
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= x.l[i] | y.l[i];
    return o != 0;   // false: the identity
}

// a lane's XYZZ29 from lane `src` of its wave (36 ds_bpermute; no LDS allocation)
__device__ __forceinline__ XYZZ29 xyzz29_shfl(const XYZZ29 &v, uint32_t src) {
    XYZZ29 r;
#ifdef H2_HIPEMU
    hipemu_shfl_words<9>(r.x.l, v.x.l, src);
    hipemu_shfl_words<9>(r.y.l, v.y.l, src);
    hipemu_shfl_words<9>(r.zz.l, v.zz.l, src);
    hipemu_shfl_words<9>(r.zzz.l, v.zzz.l, src);
#else
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.l[i] = __shfl(v.x.l[i], (int)src);
        r.y.l[i] = __shfl(v.y.l[i], (int)src);
        r.zz.l[i] = __shfl(v.zz.l[i], (int)src);
        r.zzz.l[i] = __shfl(v.zzz.l[i], (int)src);
    }
#endif
    return r;
}

// Every lane owns K consecutive entries [start, end) of the sorted list.  Runs (= buckets) that lie inside the lane go straight to their
// bucket.  The runs a lane shares with its neighbours — its first run L when that began in an earlier lane, its last run R when that goes
// on in a later one — are closed INSIDE THE WAVE before anything is written: a segmented scan over the lanes' R sums (shuffles, no LDS; it
// stops as soon as no lane has anything left to pull, which for uniform scalars is after one or two steps) gives every lane the sum of its
// run up to its own end, the lane in which a run ends adds its L and writes the bucket.  Only the (at most two) runs that cross the WAVE's
// boundaries leave as (key, XYZZ) partials — slot 2*wave (left) and 2*wave + 1 (right) of a sorted, hole-free list that msm_merge_kernel
// finishes: 2 slots per 64 lanes instead of 2 per lane (r02: 0.5 M partials of 144 B per 2^19-point MSM through HBM and a first merge level
// of 2048 workgroups).  No lane leaves early: lanes without entries take part in the shuffles with empty sums.
template <bool SPLIT>
__device__ __forceinline__ void msm_accum_body(const uint32_t *__restrict__ sval, const G1Affine *__restrict__ bases,
                                               const uint32_t *__restrict__ offsets, uint32_t nkeys, uint32_t K,
                                               XYZZ29 *__restrict__ buckets, uint32_t *__restrict__ out_keys, XYZZ29 *__restrict__ out_vals,
                                               uint32_t nthreads) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wave = t >> 6;
    const uint32_t total = offsets[nkeys];
    const uint64_t start64 = (uint64_t)t * K;
    const bool valid = t < nthreads && start64 < total;
    uint32_t start = 0, end = 0, cur = 0, next = 0, next2 = 0, hk = KEY_INVALID;
    bool first = true;
    bool empty = true;   // acc holds nothing yet (its registers are stale): emptiness is a flag, not 36 zeroed registers tested every step
    XYZZ29 acc = XYZZ29::identity();
    __shared__ XYZZ29 lsave[256];   // the lane's first-run sum waits here (not in 36 registers) while the lane walks the rest of its entries
    if (valid) {
        start = (uint32_t)start64;
        end = (start64 + K > total) ? total : start + K;
        cur = find_key(offsets, 0, nkeys, start);
        next = offsets[cur + 1];
        next2 = offsets[cur + 2];   // (offsets[nkeys + 1] is the ~0 sentinel)
    }
    const uint32_t first_key = valid ? cur : KEY_INVALID;   // key of the lane's first entry
    bool l_open = false;                                     // the lane's first run began before `start`
    if (valid) l_open = offsets[cur] < start;
    uint4 sv4 = {0u, 0u, 0u, 0u};
    auto entry = [&](uint32_t e) -> uint32_t {   // sorted entry e (called for consecutive e): 16 bytes at a time — a quarter of the loads
        if ((e & 3u) == 0 || e == start) sv4 = reinterpret_cast<const uint4 *>(sval)[e >> 2];
        const uint32_t sel = e & 3u;
        return sel == 0 ? sv4.x : sel == 1 ? sv4.y : sel == 2 ? sv4.z : sv4.w;
    };
    for (uint32_t e = start; e < end; ++e) {
        if (e >= next) {   // bucket boundary: close the run
            if (empty) acc = XYZZ29::identity();   // (a run of identity table entries, or one that cancelled)
            if (first) {
                lsave[threadIdx.x] = acc;
                hk = cur;
                first = false;
            } else {
                buckets[cur] = acc;
            }
            empty = true;
            // the next key: the end of the following bucket was requested at the previous boundary (no dependent load here in the common case)
            if (next2 > e) {
                cur = cur + 1;
                next = next2;
            } else {   // empty buckets in between
                cur = find_key(offsets, cur + 1, nkeys, e);
                next = offsets[cur + 1];
            }
            next2 = offsets[cur + 2];   // needed at the next boundary only
        }
        const uint32_t v = entry(e);
        if (SPLIT) {   // r05: 128-byte entries that already hold the 29-bit limbs (the same line of HBM traffic, no split)
            Fq29 px, py;
            if (load_table_entry_split(reinterpret_cast<const TableEntry29 *>(bases) + (v & 0x7fffffffu), px, py))
                xyzz29_add_affine_flag(acc, empty, px, py, (v >> 31) != 0);
        } else {
            const G1Affine p = load_table_entry(bases + (v & 0x7fffffffu));   // packed R'-domain point, one aligned 64-byte gather
            if (!p.is_identity()) xyzz29_add_affine_flag(acc, empty, f29_split<Q29P>(p.x), f29_split<Q29P>(p.y), (v >> 31) != 0);
        }
    }
    if (empty) acc = XYZZ29::identity();
    // ---- close the shared runs inside the wave
    const bool multi = valid && !first;               // the lane crossed at least one run boundary: L = (hk, lsave[lane]), R = (cur, acc)
    const bool r_open = valid && next > end;          // R goes on in the next lane
    bool head = !(valid && first && l_open);          // R's run starts inside this lane (lanes without entries: an empty, closed run)
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const bool need = !head && lane >= d;
        if (!(__any(need ? 1 : 0) != 0)) break;                // lanes still open after step d sit below lane d: nothing left to pull
        const XYZZ29 other = xyzz29_shfl(acc, lane >= d ? lane - d : lane);
        const uint32_t oh = __shfl(head ? 1u : 0u, (int)(lane >= d ? lane - d : lane));
        if (need) {
            xyzz29_add(acc, other);
            head = oh != 0;
        }
    }
    // acc = the sum of R's run from its start (head) or from the wave's left edge (!head) to this lane's end
    const uint32_t src1 = lane ? lane - 1 : 0;
    const uint32_t prev_head = __shfl(head ? 1u : 0u, (int)src1);
    const bool pull = multi && l_open && lane > 0;
    XYZZ29 lval = XYZZ29::identity();
    if (multi) lval = lsave[threadIdx.x];
    if (multi && !l_open) {                           // L began exactly at `start` and ended inside the lane: a complete run
        buckets[hk] = lval;
        lval = XYZZ29::identity();
    }
    if ((__any(pull ? 1 : 0) != 0)) {                          // the run that ends in this lane: L + (what the lanes before it summed)
        const XYZZ29 prev = xyzz29_shfl(acc, src1);
        if (pull) xyzz29_add(lval, prev);
    }
    bool left_mine = false, right_mine = false;
    uint32_t pkey = KEY_INVALID;
    XYZZ29 pv = XYZZ29::identity();
    if (multi && l_open) {
        if (lane > 0 && prev_head) {
            if (!lval.is_identity()) buckets[hk] = lval;
        } else {                                      // the chain reaches the wave's left edge
            left_mine = true;
            pkey = hk;
            pv = lval;
        }
    }
    const uint32_t next_valid = __shfl(valid ? 1u : 0u, (int)(lane < 63 ? lane + 1 : lane));
    const bool last_valid = valid && (lane == 63 || !next_valid);
    if (valid) {
        if (r_open && last_valid) {                   // crosses the wave's right edge (with !head: the whole wave lies inside one run)
            right_mine = true;
        } else if (!r_open) {                         // R ends with this lane
            if (head) {
                if (!acc.is_identity()) buckets[cur] = acc;
            } else {
                left_mine = true;
                pkey = cur;
                pv = acc;
            }
        }
    }
    // the wave's two slots: a partial, or an identity filler that keeps the list sorted and hole-free
    const bool any_left = (__any(left_mine ? 1 : 0) != 0);
    if (left_mine) {
        out_keys[2 * (size_t)wave] = pkey;
        out_vals[2 * (size_t)wave] = pv;
    } else if (lane == 0 && !any_left) {
        out_keys[2 * (size_t)wave] = first_key;
        if (first_key != KEY_INVALID) out_vals[2 * (size_t)wave] = XYZZ29::identity();
    }
    const bool any_valid = (__any(valid ? 1 : 0) != 0);
    if (right_mine) {
        out_keys[2 * (size_t)wave + 1] = cur;
        out_vals[2 * (size_t)wave + 1] = acc;
    } else if (last_valid) {
        out_keys[2 * (size_t)wave + 1] = cur;
        out_vals[2 * (size_t)wave + 1] = XYZZ29::identity();
    } else if (lane == 0 && !any_valid) {
        out_keys[2 * (size_t)wave + 1] = KEY_INVALID;
    }
}

// r04: the accumulator's emptiness is a flag (not 36 zeroed registers tested every step) and the next bucket's end offset is requested one
// boundary ahead: SQ_INSTS_VALU per 2^20-point launch 6.417e8 -> 6.322e8 (-1.5 %), time within noise (profiles/archive/r04_accum_flag_ab.log) — the
// loop is VALU-bound at ~2440 instructions per step, 1467 of them multiplies, ~560 the nine reductions' carries and quotient digits.
// r05: two waves per SIMD once more, this time WITH room made for the other lanes' sorts beside it (205 registers without spills, 72 KiB of LDS for
// its two workgroups per CU, the scatter on 64 KiB and 512-lane workgroups so that a histogram / scatter workgroup fits next to it): the k = 19
// proof 13.2-13.45 vs 13.1-13.2 ms, k = 21 50.7-53.0 vs 50.1-50.7 — slower in every combination (profiles/r05_accum_two_waves_ab.log), removed.
// r05, last: PMC + ISA showed a full memory drain behind every run boundary (the branch's offsets load merged with a move); builds without it, with
// the sorted entries a group ahead and with the next table entry requested a whole addition ahead (two waves per SIMD, no spills) run NO faster
// (0.682 / 0.707 ms per 2^19 points, proofs equal): the kernel does not wait for memory, its ~0.8 issue efficiency is dependent-issue latency that
// three waves cannot cover (profiles/r05_accum_wait_prefetch.log); reverted.
// r05, last: the addition's ten products issued as five side-by-side pairs (two dependency chains per lane, pinned with scheduling barriers) at two / three
// waves per SIMD: 0.702 / 0.680 against 0.690 ms, proofs equal — tools/probes/valu_rate.hip shows why: a dependent v_mad_u64_u32 chain issues as fast as
// independent ones (every 4 cycles per SIMD, like every instruction here except plain 32-bit adds / ands at 2).  profiles/r05_accum_pair_valu_rate.log; removed.
// three waves per SIMD (168 registers per lane); measured and left behind (profiles/archive/r03_msm_tune_*.log, r03_knob_ab.log): two waves per
// SIMD by launch bounds or by register padding (2 % slower / equal in isolation, nothing end to end), four (spills), the next table entry
// requested one addition ahead (5 % slower: the gather is not what the kernel waits for), plain instead of non-temporal table loads
template <bool SPLIT>
__global__ __launch_bounds__(256, 3) void msm_accum_kernel(const uint32_t *__restrict__ sval, const G1Affine *__restrict__ bases,
                                                          const uint32_t *__restrict__ offsets, uint32_t nkeys, uint32_t K,
                                                          XYZZ29 *__restrict__ buckets, uint32_t *__restrict__ out_keys,
                                                          XYZZ29 *__restrict__ out_vals, uint32_t nthreads) {
    msm_accum_body<SPLIT>(sval, bases, offsets, nkeys, K, buckets, out_keys, out_vals, nthreads);
}

// ------------------------------------------------------------------ 6. segmented merge of the partial list
// One slot per lane, 256 slots per workgroup.  keys are non-decreasing; KEY_INVALID only at the tail.
// Runs closed inside the workgroup (checked against the neighbouring workgroups' boundary keys) are written
// to buckets (identity totals are skipped: buckets start zeroed = identity, which also makes the filler slots
// harmless); the (at most two) runs that cross a workgroup boundary go to slots 2*blk, 2*blk+1 of the next level.
__global__ __launch_bounds__(256) void msm_merge_kernel(const uint32_t *__restrict__ kin, const XYZZ29 *__restrict__ vin, uint32_t len,
                                                        XYZZ29 *__restrict__ buckets, uint32_t *__restrict__ kout,
                                                        XYZZ29 *__restrict__ vout, uint32_t final_level) {
    H2_TAIL_PRIORITY();
    __shared__ XYZZ29 sv[256];
    __shared__ uint32_t sk[256];
    const uint32_t tid = threadIdx.x, blk = blockIdx.x;
    const uint32_t i = blk * 256 + tid;
    const uint32_t key = i < len ? kin[i] : KEY_INVALID;
    XYZZ29 val = XYZZ29::identity();
    if (key != KEY_INVALID) val = vin[i];
    sk[tid] = key;
    sv[tid] = val;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const bool m = tid >= d && key != KEY_INVALID && sk[tid - d] == key;
        XYZZ29 other = XYZZ29::identity();
        if (m) other = sv[tid - d];
        if (!__syncthreads_or(m ? 1 : 0)) break;   // also orders this step's LDS reads before its writes
        if (m) {
            xyzz29_add(val, other);
            sv[tid] = val;
        }
        __syncthreads();
    }
    const uint32_t first_key = sk[0], last_key = sk[255];
    if (!final_level && tid == 0) {   // default filler slots keep the next level sorted and hole-free
        kout[2 * (size_t)blk] = first_key;
        vout[2 * (size_t)blk] = XYZZ29::identity();
        kout[2 * (size_t)blk + 1] = last_key;
        if (last_key != KEY_INVALID) vout[2 * (size_t)blk + 1] = XYZZ29::identity();
    }
    __syncthreads();
    if (key == KEY_INVALID) return;
    const bool in_block_end = (tid == 255) || (sk[tid + 1] != key);
    if (!in_block_end) return;
    bool left_closed = true, right_closed = true;
    if (!final_level) {
        if (key == first_key && blk > 0) left_closed = kin[blk * 256 - 1] != key;
        if (tid == 255) right_closed = (i + 1 >= len) || (kin[i + 1] != key);
    }
    if (left_closed && right_closed) {
        if (!val.is_identity()) buckets[key] = val;
    } else if (key == first_key) {
        vout[2 * (size_t)blk] = val;   // slot 2*blk+1 keeps the (last_key == key, identity) filler if this is also the last run
    } else {
        vout[2 * (size_t)blk + 1] = val;
    }
}

// ------------------------------------------------------------------ 7. bucket reduction
__device__ __forceinline__ XYZZ29 xyzz_small_mul(const XYZZ29 &p, uint32_t k) {
    XYZZ29 r = XYZZ29::identity();
    for (int bit = 31 - __clz(k | 1u); bit >= 0; --bit) {
        r = xyzz29_double(r);
        if ((k >> bit) & 1u) xyzz29_add(r, p);
    }
    return r;
}
// precomputed bases: all windows carry weight 1, so fold them per bucket index first: out[b] = sum_w buckets[w][b]
// (rows = number of windows of `in`, summed in groups of `group`; launched twice: W -> ceil(W/4) -> 1 rows)
__global__ __launch_bounds__(64) void msm_presum_kernel(const XYZZ29 *__restrict__ in, XYZZ29 *__restrict__ out, uint32_t B, uint32_t rows,
                                                        uint32_t group) {
    H2_TAIL_PRIORITY();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t groups = (rows + group - 1) / group;
    if (t >= B * groups) return;
    in += (size_t)blockIdx.y * rows * B;     // blockIdx.y = column of a fused multi-column MSM
    out += (size_t)blockIdx.y * groups * B;
    uint32_t gi = t / B, b = t - gi * B;
    uint32_t w0 = gi * group, w1 = w0 + group < rows ? w0 + group : rows;
    XYZZ29 acc = in[(size_t)w0 * B + b];
    for (uint32_t w = w0 + 1; w < w1; ++w) xyzz29_add(acc, in[(size_t)w * B + b]);
    out[(size_t)gi * B + b] = acc;
}
// one lane per segment of L buckets: sum_{b in seg} (b+1) * bucket[b]
__global__ __launch_bounds__(64) void msm_seg_kernel(const XYZZ29 *__restrict__ buckets, XYZZ29 *__restrict__ seg_out, uint32_t B, uint32_t L,
                                                     uint32_t nseg_total) {
    H2_TAIL_PRIORITY();
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nseg_total) return;
    uint32_t per = B / L, w = g / per, lo = (g - w * per) * L;
    const XYZZ29 *bw = buckets + (size_t)w * B;
    XYZZ29 run = XYZZ29::identity(), acc = XYZZ29::identity();
    for (int b = (int)(lo + L) - 1; b >= (int)lo; --b) {
        xyzz29_add(run, bw[b]);
        xyzz29_add(acc, run);
    }
    if (lo) xyzz29_add(acc, xyzz_small_mul(run, lo));
    seg_out[g] = acc;
}
// one workgroup per window: tree sum of its segment results
__global__ __launch_bounds__(1024) void msm_winsum_kernel(const XYZZ29 *__restrict__ seg, XYZZ29 *__restrict__ win_out, uint32_t per) {
    H2_TAIL_PRIORITY();
    __shared__ XYZZ29 sh[1024];
    uint32_t tid = threadIdx.x, w = blockIdx.x;
    XYZZ29 acc = XYZZ29::identity();
    for (uint32_t i = tid; i < per; i += 1024) xyzz29_add(acc, seg[(size_t)w * per + i]);
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t d = 512; d >= 1; d >>= 1) {
        if (tid < d) {
            XYZZ29 a = sh[tid];
            xyzz29_add(a, sh[tid + d]);
            sh[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) win_out[w] = sh[0];
}
// out = sum_w 2^(c*w) * win[w]   (Wr <= 64 windows, one lane each, then a tree)
__global__ __launch_bounds__(64) void msm_fold_kernel(const XYZZ29 *__restrict__ win, uint32_t Wr, uint32_t c, XYZZ *__restrict__ out) {
    H2_TAIL_PRIORITY();
    __shared__ XYZZ29 sh[64];
    uint32_t tid = threadIdx.x;
    XYZZ29 p = XYZZ29::identity();
    if (tid < Wr) {
        p = win[tid];
        for (uint32_t i = 0; i < c * tid; ++i) p = xyzz29_double(p);
    }
    sh[tid] = p;
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        if (tid < d) {
            XYZZ29 a = sh[tid];
            xyzz29_add(a, sh[tid + d]);
            sh[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) out[0] = xyzz29_to_sat(sh[0]);   // back to saturated canonical limbs for the C ABI
}


// ---- quad-lane versions of the latency-bound tail (quad29.cuh): one point per 4 lanes ---------------------------
// one quad per segment of L buckets: sum_{b in seg} (b+1) * bucket[b]; with `tree` the 64 quads of a workgroup (all in
// the same window) are summed before leaving, so the window sum only has per/64 values left to add
__global__ __launch_bounds__(256) void msm_seg_quad_kernel(const XYZZ29 *__restrict__ buckets, XYZZ29 *__restrict__ seg_out, uint32_t B, uint32_t L,
                                                           uint32_t nseg_total, uint32_t lo_bits, uint32_t tree) {
    H2_TAIL_PRIORITY();
    __shared__ XYZZ29 sh[64];
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u, qi = threadIdx.x >> 2;
    uint32_t g = blockIdx.x * 64 + qi;
    const bool live = g < nseg_total;
    if (!live) g = nseg_total - 1;   // keep the quad in lock step on valid data; its result is discarded
    uint32_t per = B / L, w = g / per, lo = (g - w * per) * L;
    const XYZZ29 *bw = buckets + (size_t)w * B;
    Fq29 run = Fq29::zero(), acc = Fq29::zero();
    for (int b = (int)(lo + L) - 1; b >= (int)lo; --b) {
        run = quad_xyzz_add(run, quad_load(bw + b, q), lane);
        acc = quad_xyzz_add(acc, run, lane);
    }
    acc = quad_xyzz_add(acc, quad_xyzz_small_mul(run, lo, lo_bits, lane), lane);
    if (!tree) {
        if (live) quad_store(seg_out + g, q, acc);
        return;
    }
    if (!live) acc = Fq29::zero();
    quad_store(&sh[qi], q, acc);
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        Fq29 other = quad_load(&sh[(qi + d) & 63u], q);
        Fq29 sum = quad_xyzz_add(acc, other, lane);
        __syncthreads();
        if (qi < d) {
            acc = sum;
            quad_store(&sh[qi], q, acc);
        }
        __syncthreads();
    }
    if (qi == 0) quad_store(seg_out + blockIdx.x, q, acc);
}
// one workgroup (THREADS lanes = THREADS/4 quads) per window: tree sum of its `per` partial results.  The 256-lane
// build (one wave per SIMD) is the one used next to a pipelined accumulation: a 1024-lane workgroup needs four waves
// of 120 registers on every SIMD of a CU, i.e. an almost empty CU, and was measured waiting ~1 ms for one.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void msm_winsum_quad_kernel(const XYZZ29 *__restrict__ seg, XYZZ29 *__restrict__ win_out, uint32_t per) {
    H2_TAIL_PRIORITY();
    constexpr uint32_t NQ = THREADS / 4;
    __shared__ XYZZ29 sh[NQ];
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u, qi = threadIdx.x >> 2, w = blockIdx.x;
    Fq29 acc = Fq29::zero();
    const uint32_t rounds = (per + NQ - 1) / NQ;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t i = qi + NQ * r;
        Fq29 v = i < per ? quad_load(seg + (size_t)w * per + i, q) : Fq29::zero();
        acc = rounds == 1 ? v : quad_xyzz_add(acc, v, lane);
    }
    quad_store(&sh[qi], q, acc);
    __syncthreads();
    uint32_t d0 = NQ / 2;
    while (d0 > 1 && d0 >= per) d0 >>= 1;   // skip levels whose partners are all identity
    if (per <= 1) d0 = 0;
    for (uint32_t d = d0; d >= 1; d >>= 1) {
        Fq29 other = qi + d < NQ ? quad_load(&sh[qi + d], q) : Fq29::zero();
        Fq29 sum = quad_xyzz_add(acc, other, lane);
        __syncthreads();
        if (qi < d) {
            acc = sum;
            quad_store(&sh[qi], q, acc);
        }
        __syncthreads();
    }
    if (qi == 0) quad_store(win_out + w, q, acc);
}

// out = sum_w 2^(c*w) * win[w]   (Wr <= 64 windows, one quad each, then a tree); result in saturated limbs
__global__ __launch_bounds__(256) void msm_fold_quad_kernel(const XYZZ29 *__restrict__ win, uint32_t Wr, uint32_t c, XYZZ *__restrict__ out) {
    H2_TAIL_PRIORITY();
    __shared__ XYZZ29 sh[64];
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u, qi = threadIdx.x >> 2;
    Fq29 p = qi < Wr ? quad_load(win + qi, q) : Fq29::zero();
    const uint32_t max_dbl = c * (Wr - 1);
    for (uint32_t i = 0; i < max_dbl; ++i) {
        Fq29 d = quad_xyzz_double(p, lane);
        p = f29_select(i < c * qi && qi < Wr, d, p);
    }
    quad_store(&sh[qi], q, p);
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        Fq29 other = quad_load(&sh[(qi + d) & 63u], q);
        Fq29 sum = quad_xyzz_add(p, other, lane);
        __syncthreads();
        if (qi < d) {
            p = sum;
            quad_store(&sh[qi], q, p);
        }
        __syncthreads();
    }
    if (qi == 0) {   // back to saturated canonical limbs for the C ABI, coordinate by coordinate
        const bool id = quad_is_identity(p, lane);
        Fq v = id ? Fq::zero() : f29_to_sat(p);
        reinterpret_cast<Fq *>(out)[q] = v;
    }
}

// ------------------------------------------------------------------ host driver
// One MSM per scalar column over the same bases, all columns in ONE pass through the pipeline: the sort, accumulation and
// merge kernels see ncols * W windows (column-major), and the latency-bound bucket reduction runs once for all columns
// (its chains are as long as for one column, just ncols times wider).  ncols > 1 needs precomputed window tables
// (every column then owns ONE bucket set after the per-index presum).  out: ncols results.
// precomputed tables: every column's bucket set carries weight 1, the "fold" is only the conversion of its sum
__global__ __launch_bounds__(64) void msm_cols_out_kernel(const XYZZ29 *__restrict__ win, uint32_t ncols, XYZZ *__restrict__ out) {
    H2_TAIL_PRIORITY();
    uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < ncols) out[col] = xyzz29_to_sat(win[col]);
}

// Bucket reduction for ncols bucket sets laid out [col][Wcol][B] (plain bases: one column, Wcol weighted windows):
// per-index presum over a column's windows (precomputed tables), sum_b (b+1)*bucket[b] per set, conversion / fold.
// All columns go through the same launches: the dependent chains are as long as for one column.
int msm_reduce_cols(h2hip_ctx *ctx, const h2hip_bases *bases, uint32_t c, const XYZZ29 *buckets, uint32_t ncols, XYZZ *out) {
    hipStream_t st = ctx->stream;
    const bool precomp = bases->tables > 1;
    H2_REQUIRE(ncols >= 1 && ncols <= 64 && (ncols == 1 || precomp), "1..64 bucket sets (several need precomputed bases)");
    const uint32_t Wcol = (255 + c - 1) / c, B = 1u << (c - 1);
    uint32_t L = (uint32_t)ctx->msm_seg;
    if (L > B) L = B;
    const uint32_t Wr = precomp ? ncols : Wcol;   // bucket sets left after the optional per-index presum (one per column)
    const uint32_t nseg = Wr * (B / L);
    XYZZ29 *seg, *win, *presum = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SEG, sizeof(XYZZ29) * nseg, (void **)&seg));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_WIN, sizeof(XYZZ29) * 64, (void **)&win));
    const uint32_t rows = precomp ? Wcol : 1;   // bucket sets per column the accumulation left: [col][rows][B]
    const uint32_t pre_rows = (rows + 3) / 4;
    if (precomp && rows > 1) H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(XYZZ29) * B * (size_t)ncols * (pre_rows + 1), (void **)&presum));
    const XYZZ29 *red_in = buckets;
    if (precomp && rows > 1) {
        prof_begin(ctx, "msm_presum_kernel");
        // (a quad-lane tree version — four quads per bucket index, two LDS levels — measured slower: 0.075 vs 0.072 ms for one column and
        // 0.87 vs 0.53 ms per proof for the 3-4 column rounds: with 16Ki x columns independent chains this stage is throughput-bound, and a
        // quad addition spends 16 lane-products plus its DPP traffic on the 14 products of the addition)
        if (rows <= 4) {
            hipLaunchKernelGGL(msm_presum_kernel, dim3((B + 63) / 64, ncols), dim3(64), 0, st, buckets, presum, B, rows, rows);
        } else {
            XYZZ29 *stage1 = presum + (size_t)ncols * B;   // [ncols][pre_rows][B]; the final [ncols][B] sits in front of it
            hipLaunchKernelGGL(msm_presum_kernel, dim3((B * pre_rows + 63) / 64, ncols), dim3(64), 0, st, buckets, stage1, B, rows, 4u);
            hipLaunchKernelGGL(msm_presum_kernel, dim3((B + 63) / 64, ncols), dim3(64), 0, st, (const XYZZ29 *)stage1, presum, B, pre_rows, pre_rows);
        }
        prof_end(ctx);
        red_in = presum;
    }
    // quad-lane arithmetic pays when the stage is latency-bound (few segments: precomputed bases); with 16 windows'
    // worth of segments the stage is throughput-bound and the one-lane kernels win — the 2^(c*w) fold is always a chain
    const bool quad_reduce = ctx->msm_quad_tails && nseg <= (uint32_t)ctx->msm_quad_seg_max;
    if (quad_reduce) {
        uint32_t lo_bits = 0;   // bits needed for a segment's first bucket index (< B)
        while ((1u << lo_bits) < B) ++lo_bits;
        const uint32_t per = B / L;
        const uint32_t tree = (per % 64 == 0) ? 1u : 0u;   // a workgroup's 64 quads then belong to one window
        prof_begin(ctx, "msm_seg_kernel");
        hipLaunchKernelGGL(msm_seg_quad_kernel, dim3((nseg + 63) / 64), dim3(256), 0, st, red_in, seg, B, L, nseg, lo_bits, tree);
        prof_end(ctx);
        prof_begin(ctx, "msm_winsum_kernel");
        const uint32_t wper = tree ? per / 64 : per;
        if (wper <= 128)
            hipLaunchKernelGGL(msm_winsum_quad_kernel<256>, dim3(Wr), dim3(256), 0, st, (const XYZZ29 *)seg, win, wper);
        else
            hipLaunchKernelGGL(msm_winsum_quad_kernel<1024>, dim3(Wr), dim3(1024), 0, st, (const XYZZ29 *)seg, win, wper);
        prof_end(ctx);
    } else {
        prof_begin(ctx, "msm_seg_kernel");
        hipLaunchKernelGGL(msm_seg_kernel, dim3((nseg + 63) / 64), dim3(64), 0, st, red_in, seg, B, L, nseg);
        prof_end(ctx);
        prof_begin(ctx, "msm_winsum_kernel");
        hipLaunchKernelGGL(msm_winsum_kernel, dim3(Wr), dim3(1024), 0, st, (const XYZZ29 *)seg, win, B / L);
        prof_end(ctx);
    }
    prof_begin(ctx, "msm_fold_kernel");
    if (precomp) {
        hipLaunchKernelGGL(msm_cols_out_kernel, dim3(1), dim3(64), 0, st, (const XYZZ29 *)win, ncols, out);
    } else if (ctx->msm_quad_tails) {
        hipLaunchKernelGGL(msm_fold_quad_kernel, dim3(1), dim3(256), 0, st, (const XYZZ29 *)win, Wr, c, out);
    } else {
        hipLaunchKernelGGL(msm_fold_kernel, dim3(1), dim3(64), 0, st, (const XYZZ29 *)win, Wr, c, out);
    }
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// CONSUMES the pre-zeroed state: the caller is about to dirty the array, and only a completed buckets_clean_after_use re-arms it — a call
// that fails between the accumulation and its reduction must not leave the dirty array marked as zero
bool buckets_prezeroed(h2hip_ctx *ctx, int which, const void *buf, size_t bytes) {
    const bool ok = ctx->clean_ev && ctx->clean_ptr[which] == buf && ctx->clean_bytes[which] >= bytes;
    if (ok) ctx->clean_bytes[which] = 0;
    return ok;
}
// `on`: the stream the zero-fill is queued on (default: the context's own clean stream).  HIP serves all streams through four hardware queues:
// a fill that waits for the reduction blocks whatever is queued BEHIND it on a stream sharing its queue — the batch MSM therefore queues its fill
// after the tail hook's work and on its first lane's stream, which that work never uses (r04: profiles/archive/r04_timeline_k19.md)
int buckets_clean_after_use(h2hip_ctx *ctx, int which, void *buf, size_t bytes, hipStream_t on) {
    if (!ctx->clean_stream) {
        H2_HIPCHK(hipStreamCreateWithFlags(&ctx->clean_stream, hipStreamNonBlocking));
        H2_HIPCHK(hipEventCreateWithFlags(&ctx->clean_ev, hipEventDisableTiming));
        H2_HIPCHK(hipEventCreateWithFlags(&ctx->clean_ev1, hipEventDisableTiming));
        H2_HIPCHK(hipEventCreateWithFlags(&ctx->used_ev, hipEventDisableTiming));
    }
    const hipStream_t cs = on ? on : ctx->clean_stream;
    H2_HIPCHK(hipEventRecord(ctx->used_ev, ctx->stream));               // everything that reads the buckets is queued on the context's stream
    H2_HIPCHK(hipStreamWaitEvent(cs, ctx->used_ev, 0));
    H2_HIPCHK(hipMemsetAsync(buf, 0, bytes, cs));
    H2_HIPCHK(hipEventRecord(which ? ctx->clean_ev1 : ctx->clean_ev, cs));
    ctx->clean_ptr[which] = buf;
    ctx->clean_bytes[which] = bytes;
    return H2HIP_OK;
}

int msm_run_cols(h2hip_ctx *ctx, const h2hip_bases *bases, const Fr *const *scalars, uint32_t ncols, size_t n, XYZZ *out, XYZZ29 *ext_buckets,
                 bool ext_buckets_zeroed) {
    H2_REQUIRE(ncols >= 1 && ncols <= MSM_MAX_COLS, "1..32 columns per fused MSM");
    H2_REQUIRE(n <= bases->n, "more scalars than resident bases");
    H2_REQUIRE(bases->pts29 != nullptr || bases->n == 0, "bases are not prepared");
    H2_REQUIRE(n < (1u << 27), "n too large for 32-bit entry indices");
    hipStream_t st = ctx->stream;
    if (n == 0) {
        H2_REQUIRE(!ext_buckets, "empty MSM in a deferred-reduction batch");
        H2_HIPCHK(hipMemsetAsync(out, 0, sizeof(XYZZ) * ncols, st));
        return H2HIP_OK;
    }
    const bool precomp = bases->tables > 1;
    H2_REQUIRE(ncols == 1 || precomp, "a fused multi-column MSM needs precomputed bases");
    const uint32_t c = precomp ? bases->window_bits : (ctx->msm_window_bits ? (uint32_t)ctx->msm_window_bits : pick_window(n));
    H2_REQUIRE(c >= 2 && c <= 16, "window bits must be 2..16 (a window's bucket histogram lives in LDS)");
    const uint32_t Wcol = (255 + c - 1) / c;   // windows of one column
    H2_REQUIRE(Wcol <= 64, "too many windows");
    H2_REQUIRE(!precomp || bases->tables >= Wcol, "precomputed table has too few windows");
    const uint32_t W = Wcol * ncols;           // windows the sort / accumulation see
    const G1Affine *table = (const G1Affine *)bases->pts29;
    const uint32_t B = 1u << (c - 1);
    const uint32_t nkeys = W * B;              // keys of the counting sort = run keys of the accumulation = bucket slots
    const uint64_t emax = (uint64_t)n * W;
    H2_REQUIRE(emax < 0xFFFFFFF0ull, "n*W overflows 32 bits");
    uint32_t K1 = (uint32_t)ctx->msm_chunk;
    if (K1 == 0) {   // auto: as long as possible (fewer shared runs to merge) while the grid is still several waves per SIMD
        // Measured at 2^19 / 17 windows (tools/msm_r03.py): 34 entries per lane = 4096 waves 0.72 ms; 32 = 4352 waves 0.81 ms; 64 = 2176 waves
        // 0.98 ms; ONE exact round of two waves per SIMD (68 entries = 2048 waves) 0.72 ms although its wave-level merge is a single
        // addition per lane — with one round the kernel ends with its slowest wave, shorter lanes in several rounds balance themselves.
        // r06, re-measured in whole proofs with the sort kernels running beside the accumulations (profiles/r06_msm_chunk_ab.log): shorter lanes — more,
        // shorter accumulation workgroups, whose retiring gives the next column's sort its slots sooner — win up to 2^20 points: 24 entries per lane
        // -0.7 ... -1.3 % at k = 19, -1.6 % at k = 18, -3 % at k = 17 (four fused columns), -4 % at k = 16; 2^20 points: 32 (-1.5 ... -3 %; 48: +3 %); from 2^21
        // points the longest lanes stay best (56 / 48 / 32: +1 % / neutral / +1.3 %)
        uint64_t k = emax / 262144;
        K1 = k < 8 ? 8u : k > 64 ? 64u : (uint32_t)k;
        const bool lone = !ctx->is_lane && ctx->msm_chunk_lone != 0;   // a lone MSM (SHPLONK's W, W'): no other column's sort waits for its slots
        if (lone) {   // r05's rule stays: alone, 34 entries per lane are 1 - 3 % faster than 24 at 2^19 points (warm: sync MSM 1.09 - 1.12 vs 1.12 - 1.13 ms, accumulation 0.70 vs 0.72 - 0.78)
            if (ctx->msm_chunk_lone > 0) K1 = (uint32_t)ctx->msm_chunk_lone;
        } else if (emax >= 25000000ull) K1 = 64;
        else if (emax >= 12000000ull) K1 = 32;
        else if (emax >= 4000000ull) K1 = 24;
    }
    // chunking of the counting sort: about 32 chunks per window, 4Ki..64Ki scalars each.  32 = the CUs of an XCD: the (window, chunk)
    // workgroups of one window run together on the window's XCD, one per CU (tried: chunks sized for ONE round over the whole chip,
    // W * G <= CUs — 2^19: scatter 0.081 -> 0.145 ms, 2^20: 0.154 -> 0.242 ms: half of every XCD's CUs idle and three windows' slices
    // competing for one 4 MiB L2).
    // (r06 tried choosing the chunk count so that windows x chunks fills whole rounds of the chip's 2 x CUs sort-workgroup slots — 30 at 17 windows: 510
    // workgroups instead of 544 — on the theory that the 32 left-over workgroups cost a round: warm, a synchronous 2^19-point MSM is 1.08 ms either way
    // and whole proofs do not move (profiles/r06_sort_groups_ab.log; the first sweep's 1.20 -> 1.09 ms was the tool's cold first measurement).  The
    // knob stays: msm_sort_groups, 0 = 32.)
    const uint32_t sort_groups = ctx->msm_sort_groups > 0 ? (uint32_t)ctx->msm_sort_groups : 32u;
    uint32_t chunk = (uint32_t)((n + sort_groups - 1) / sort_groups);
    if (chunk < 4096) chunk = 4096;
    const uint32_t chunk_cap = ctx->msm_hist_packed ? 65535u : 65536u;   // (r06) a packed histogram counter holds at most 65535: 2^21 points sort as 33 chunks
    if (chunk > chunk_cap) chunk = chunk_cap;
    const uint32_t G = (uint32_t)((n + chunk - 1) / chunk);

    // The one-pass LDS-histogram counting sort is the only sort.  r03 also built a two-level sort with only coalesced writes (0.124 -> 0.087 ms
    // at 2^19 on uniform scalars) that lost inside the proofs — 5-9 % slower accumulation on its entry order, 15-20 % behind on 0/1-heavy columns
    // at k >= 20 — a bucket-major sort with one bucket set per column (msm_fold_windows: the wave-level merge of its 16x longer runs cost more
    // than the presum it saved) and a column's windows dealt to two lanes; all three were removed in r04, their A/B logs are
    // profiles/archive/r03_msm_sort_ab.log, r03_msm_reorder.log, r02_msm_fold_windows_ab.log, r03_msm_split_windows_ab.log.
    digit_t *digits;
    uint32_t *bhist, *counts, *offsets, *sval, *pkey[2];
    XYZZ29 *buckets, *pval[2];
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_DIGITS, sizeof(digit_t) * emax, (void **)&digits));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_CURSOR, sizeof(uint32_t) * (size_t)W * G * B, (void **)&bhist));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_COUNTS, sizeof(uint32_t) * ((size_t)nkeys + 1), (void **)&counts));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OFFSETS, sizeof(uint32_t) * ((size_t)nkeys + 2), (void **)&offsets));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SVAL, sizeof(uint32_t) * (emax + 4), (void **)&sval));   // + 4: the accumulation reads aligned 16-byte groups
    if (ext_buckets) buckets = ext_buckets;
    else H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_BUCKETS, sizeof(XYZZ29) * nkeys, (void **)&buckets));
    const uint32_t T1 = (uint32_t)((emax + K1 - 1) / K1);
    const uint32_t accum_blocks = (T1 + 255) / 256;
    const uint32_t len1 = 8 * accum_blocks, blocks1 = (len1 + 255) / 256;   // the accumulation leaves two partial slots per wave (4 waves per workgroup)
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PKEY0, sizeof(uint32_t) * (size_t)len1, (void **)&pkey[0]));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PVAL0, sizeof(XYZZ29) * (size_t)len1, (void **)&pval[0]));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PKEY1, sizeof(uint32_t) * 2 * (size_t)blocks1, (void **)&pkey[1]));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PVAL1, sizeof(XYZZ29) * 2 * (size_t)blocks1, (void **)&pval[1]));

    // ---- sort
    if (ext_buckets) {   // a batch's shared array: zeroed by the batch (after its previous use) or here
        if (!ext_buckets_zeroed) H2_HIPCHK(hipMemsetAsync(buckets, 0, sizeof(XYZZ29) * nkeys, st));
    } else if (buckets_prezeroed(ctx, 0, buckets, sizeof(XYZZ29) * nkeys)) {
        H2_HIPCHK(hipStreamWaitEvent(st, ctx->clean_ev, 0));
    } else {
        H2_HIPCHK(hipMemsetAsync(buckets, 0, sizeof(XYZZ29) * nkeys, st));
    }
    DigitCols dcols;
    for (uint32_t col = 0; col < MSM_MAX_COLS; ++col) {
        H2_REQUIRE(col >= ncols || scalars[col], "NULL scalar column");
        dcols.scalars[col] = col < ncols ? scalars[col] : nullptr;
    }
    prof_begin(ctx, "msm_digits_kernel");
    hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256), ncols), dim3(256), 0, st, dcols, (uint32_t)n, c, Wcol, digits, counts + nkeys,
                       offsets + nkeys + 1);   // + the sentinels counts[nkeys], offsets[nkeys + 1]
    prof_end(ctx);
    const uint32_t sort_threads = (uint32_t)ctx->msm_sort_threads;
    if (!ctx->msm_lds_attr_set) {   // dynamic LDS above 64 KiB has to be enabled per kernel (and device) once
        H2_HIPCHK(hipFuncSetAttribute((const void *)msm_hist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint32_t) * MAX_LDS_BUCKETS)));
        H2_HIPCHK(hipFuncSetAttribute((const void *)msm_hist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint32_t) * MAX_LDS_BUCKETS)));
        H2_HIPCHK(hipFuncSetAttribute((const void *)msm_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint32_t) * MAX_LDS_BUCKETS)));
        ctx->msm_lds_attr_set = true;
    }
    prof_begin(ctx, "msm_hist_kernel");
    const bool hist_packed = ctx->msm_hist_packed != 0 && chunk < 65536u && B >= 2;   // (a counter holds at most `chunk`)
    // bucket sub-ranges per window (msm_hist_split, default 1): two 32 KiB sub-ranges at c = 16 would fit one retiring accumulation workgroup's slot where
    // the 64 KiB window needs two — measured: k = 21 / 22 proofs unchanged, k = 20 +0.5 %, a synchronous 2^20-point MSM +4 % (profiles/r06_hist_split_ab.log)
    uint32_t HS = ctx->msm_hist_split > 0 ? (uint32_t)ctx->msm_hist_split : 1u;
    if (HS > B) HS = B;
    const uint32_t hist_grid = sort_grid_size(W * HS, G);
    if (hist_packed)
        hipLaunchKernelGGL(msm_hist_kernel<true>, dim3(hist_grid), dim3(sort_threads), sizeof(uint32_t) * ((B / HS + 1) / 2), st, (const digit_t *)digits, (uint32_t)n, W, B, G, chunk, HS, bhist);
    else
        hipLaunchKernelGGL(msm_hist_kernel<false>, dim3(hist_grid), dim3(sort_threads), sizeof(uint32_t) * (B / HS), st, (const digit_t *)digits, (uint32_t)n, W, B, G, chunk, HS, bhist);
    prof_end(ctx);
    prof_begin(ctx, "msm_hist_scan_kernel");
    hipLaunchKernelGGL(msm_hist_scan_kernel, dim3((nkeys + 255) / 256), dim3(256), 0, st, bhist, W, B, G, counts);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(exclusive_scan_u32(ctx, counts, offsets, nkeys + 1));
    prof_begin(ctx, "msm_scatter_kernel");
    uint32_t S = (uint32_t)ctx->msm_scatter_split;   // sub-ranges per window: keep a segment's output slice (n*4/S bytes) within ~2 MiB
    if (S == 0) {
        S = 1;
        while (S < 4 && S * 2 <= B && ((uint64_t)n * 4) / S > (2u << 20)) S *= 2;
    }
    if (S > B) S = B;
    const uint32_t scatter_grid = sort_grid_size(W * S, G);
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(scatter_grid), dim3(sort_threads), ctx->msm_scatter_full_lds ? sizeof(uint32_t) * MAX_LDS_BUCKETS : sizeof(uint32_t) * (B / S), st,   // full 128 KiB: one workgroup per CU keeps a segment's writes on one XCD
                       (const digit_t *)digits, (uint32_t)n, W, B, G, chunk, S, precomp ? (uint32_t)bases->n : 0u, Wcol, (const uint32_t *)offsets,
                       (const uint32_t *)bhist, sval);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    if (ctx->sorted_arm) {   // (msm_stagger_sorts) the next lane's sort may start
        ctx->sorted_arm = false;
        H2_HIPCHK(hipEventRecord(ctx->sorted_ev, st));
    }

    // ---- accumulate (+ wave-level merge), then the block-level merge of the wave-boundary partials
    {
        // (profiled launches carry their events themselves: hipExtLaunchKernelGGL — separate records around the twelve accumulations of a proof
        // cost it ~0.3 ms)
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
#ifndef H2_HIPEMU
        if (prof_launch_events(ctx, "msm_accum_kernel", &ev_a, &ev_b)) {
            if (bases->split)
                hipExtLaunchKernelGGL(msm_accum_kernel<true>, dim3(accum_blocks), dim3(256), 0, st, ev_a, ev_b, 0, (const uint32_t *)sval, table,
                                      (const uint32_t *)offsets, nkeys, K1, buckets, pkey[0], pval[0], T1);
            else
                hipExtLaunchKernelGGL(msm_accum_kernel<false>, dim3(accum_blocks), dim3(256), 0, st, ev_a, ev_b, 0, (const uint32_t *)sval, table,
                                      (const uint32_t *)offsets, nkeys, K1, buckets, pkey[0], pval[0], T1);
        } else
#endif
        {
            prof_begin(ctx, "msm_accum_kernel");   // (a no-op unless profiling without launch events: the emulated build)
            if (bases->split)
                hipLaunchKernelGGL(msm_accum_kernel<true>, dim3(accum_blocks), dim3(256), 0, st, (const uint32_t *)sval, table, (const uint32_t *)offsets, nkeys, K1,
                                   buckets, pkey[0], pval[0], T1);
            else
                hipLaunchKernelGGL(msm_accum_kernel<false>, dim3(accum_blocks), dim3(256), 0, st, (const uint32_t *)sval, table, (const uint32_t *)offsets, nkeys, K1,
                                   buckets, pkey[0], pval[0], T1);
            prof_end(ctx);
        }
        (void)ev_a;
        (void)ev_b;
    }
    H2_HIPCHK(hipGetLastError());
    uint32_t len = len1;
    int src = 0;
    for (;;) {
        const uint32_t blocks = (len + 255) / 256;
        const uint32_t final_level = blocks == 1 ? 1u : 0u;
        prof_begin(ctx, "msm_merge_kernel");
        hipLaunchKernelGGL(msm_merge_kernel, dim3(blocks), dim3(256), 0, st, (const uint32_t *)pkey[src], (const XYZZ29 *)pval[src], len, buckets,
                           pkey[src ^ 1], pval[src ^ 1], final_level);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
        if (final_level) break;
        len = 2 * blocks;
        src ^= 1;
    }

    if (ext_buckets) return H2HIP_OK;   // the caller reduces several MSMs' buckets together
    H2_CHK(msm_reduce_cols(ctx, bases, c, buckets, ncols, out));
    return buckets_clean_after_use(ctx, 0, buckets, sizeof(XYZZ29) * nkeys);
}

int msm_run(h2hip_ctx *ctx, const h2hip_bases *bases, const Fr *scalars, size_t n, XYZZ *out) {
    return msm_run_cols(ctx, bases, &scalars, 1, n, out, nullptr, false);
}

}  // namespace h2
