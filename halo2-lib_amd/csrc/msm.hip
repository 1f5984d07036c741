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
#include "ec29.cuh"
#include "quad29.cuh"

namespace h2 {

// The latency-bound tail kernels (a few waves of dependent point additions) run next to another MSM's multiplier-bound
// accumulation when MSMs are pipelined over lanes: raise their wave priority so the SIMD arbiter issues them first.
// The LDS-histogram sort kernels take their 128 KiB as DYNAMIC shared memory: with a static array that leaves room for
// one workgroup per CU the compiler pads the kernel's register allocation to 96 VGPRs per lane ("occupancy is LDS-bound
// anyway"), and a 1024-lane workgroup (4 waves per SIMD) then never fits next to three resident accumulation waves
// (3 x 144 of 512 registers) — the sort of MSM i+1 waited for the accumulation of MSM i to drain (rocprofv3 timeline,
// profiles/r01_pipeline_timeline_*.md).  With dynamic LDS the kernels allocate the 8-16 registers they use.
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
// (the first workgroup also writes the sort's two sentinels — counts[nsort] = 0 and offsets[nsort + 1 .. nsort + ks] = ~0, read by the scan and by
// the accumulation's boundary walk — which were two tiny memset launches per MSM on the lane's critical path)
// (w_lo: only the windows [w_lo, w_lo + W) are written — a column whose windows are dealt to two lanes; the signed-digit carry still runs
// through the windows below)
__global__ __launch_bounds__(256) void msm_digits_kernel(DigitCols cols, uint32_t n, uint32_t c, uint32_t W, uint32_t *__restrict__ digits,
                                                         uint32_t *__restrict__ counts_tail, uint32_t *__restrict__ offsets_tail, uint32_t ks,
                                                         uint32_t w_lo) {
    H2_SORT_PRIORITY();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x == 0) counts_tail[0] = 0;
        if (threadIdx.x < ks) offsets_tail[threadIdx.x] = 0xFFFFFFFFu;
    }
    if (i >= n) return;
    digits += (size_t)blockIdx.y * W * n;
    Fr s = fe_from_mont(cols.scalars[blockIdx.y][i]);
    const uint32_t B = 1u << (c - 1);
    const uint64_t mask = (1ull << c) - 1;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < w_lo + W; ++w) {
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
        if (w >= w_lo) digits[(size_t)(w - w_lo) * n + i] = d | (neg << 31);
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
__global__ __launch_bounds__(1024) void msm_hist_kernel(const uint32_t *__restrict__ digits, uint32_t n, uint32_t W, uint32_t B,
                                                        uint32_t G, uint32_t chunk, uint32_t *__restrict__ bhist) {
    H2_SORT_PRIORITY();
    HIP_DYNAMIC_SHARED(uint32_t, hist)   // B counters
    uint32_t w, g;
    block_to_window_chunk(blockIdx.x, G, W, w, g);
    if (w >= W) return;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    uint32_t lo = g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const uint32_t *dw = digits + (size_t)w * n;
    // eight independent loads in flight per lane, then the LDS atomics: one load -> one atomic per iteration left the kernel latency-bound
    const uint32_t T = blockDim.x;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 8 * T) {
        uint32_t d[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) d[k] = i + k * T < hi ? dw[i + k * T] & 0x7fffffffu : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            if (d[k]) atomicAdd(&hist[d[k] - 1], 1u);
    }
    __syncthreads();
    uint32_t *out = bhist + ((size_t)w * G + g) * B;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) out[b] = hist[b];
}

// ------------------------------------------------------------------ 3. prefix over chunks per (window, bucket)
// Precomputed bases: every window of a column carries weight 1, so groups of `fg` windows can share a bucket set.  The counting sort then
// lays its keys out BUCKET-major inside a group — key = ((col*NG + grp)*B + b)*fg + r for window grp*fg + r of column col (NG = groups per
// column; keys of windows past the last one stay empty) — so the sorted list holds the group's entries of one bucket index next to each
// other and the accumulation sums them into ONE bucket: the bucket array is [col][NG][B] instead of [col][Wcol][B].
struct FoldKeys {
    uint32_t fg, ng, wcol;   // fg == 0: plain window-major keys w*B + b
};
__device__ __forceinline__ uint32_t sort_key(uint32_t w, uint32_t b, uint32_t B, const FoldKeys &f) {
    if (!f.fg) return w * B + b;
    const uint32_t col = w / f.wcol, wc = w - col * f.wcol, grp = wc / f.fg, r = wc - grp * f.fg;
    return ((col * f.ng + grp) * B + b) * f.fg + r;
}
__global__ __launch_bounds__(256) void msm_hist_scan_kernel(uint32_t *__restrict__ bhist, uint32_t W, uint32_t B, uint32_t G, FoldKeys fold_w,
                                                            uint32_t *__restrict__ counts) {
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
    counts[sort_key(w, b, B, fold_w)] = run;
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
__global__ __launch_bounds__(1024) void msm_scatter_kernel(const uint32_t *__restrict__ digits, uint32_t n, uint32_t W, uint32_t B,
                                                           uint32_t G, uint32_t chunk, uint32_t S, uint32_t table_stride, uint32_t Wcol, FoldKeys fold_w,
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
    for (uint32_t b = threadIdx.x; b < Bs; b += blockDim.x) cursor[b] = offsets[sort_key(w, b0 + b, B, fold_w)] + bh[b];
    __syncthreads();
    uint32_t lo = g * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const uint32_t *dw = digits + (size_t)w * n;
    const uint32_t idx_base = (w % Wcol) * table_stride;   // precomputed bases: window w of a column reads table level w
    // eight entries per lane and round: the loads, then the returning LDS atomics, then the stores — each group independent, so the
    // latencies of a group overlap instead of adding up per entry
    const uint32_t T = blockDim.x;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 8 * T) {
        uint32_t dv[8], pos[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) dv[k] = i + k * T < hi ? dw[i + k * T] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t d = dv[k] & 0x7fffffffu, b = d - 1 - b0;   // d == 0 wraps to a huge b: skipped like another sub-range's entry
            pos[k] = (d && b < Bs) ? atomicAdd(&cursor[b], 1u) : KEY_INVALID;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            if (pos[k] != KEY_INVALID) sval[pos[k]] = (idx_base + i + k * T) | (dv[k] & 0x80000000u);
    }
}

// ------------------------------------------------------------------ 4b. two-level sort with coalesced writes (r03; selectable, not the default — see msm_run_cols)
// The one-pass counting sort above writes every entry on its own (a 4-byte store to its final position): 8.9 M scattered write transactions
// per 2^19-point MSM, which is what its 76 us are made of (the L2 retires them per transaction, not per byte), plus W*G full-window
// histograms (35 MB) written, prefixed in place and read back.  Here every write is a coalesced run:
//   A  msm_csort_hist      (window, chunk) workgroups count their entries per COARSE bucket (key >> LB: 128 per window)
//      exclusive scan over (window, coarse bucket, chunk)
//   B  msm_csort_scatter   the workgroup sorts its chunk by coarse bucket IN LDS and copies the runs out: a run of ~64 entries per
//                          (chunk, coarse bucket) is contiguous in the coarse array; entries carry (point index, sign, fine key)
//   C  msm_csort_fine      one workgroup per coarse bucket (~4Ki entries): histogram of the fine keys -> the window's bucket offsets (written
//                          directly: no global scan over W * 2^(c-1) counts), counting sort in LDS, one contiguous copy-out.  Buckets that
//                          do not fit the LDS buffer (0/1-heavy circuit columns put a quarter of a window into one key) are placed directly.
// Traffic: the digits are read twice, the entries written twice and read once — all of it coalesced.
constexpr uint32_t CS_THREADS = 512, CS_CHUNK = 8192, CS_MAX_NC = 128, CS_MAX_NF = 256, CS_BUF = 8192;
struct CSortGeom {
    uint32_t LB, NC, NF;   // fine bits, coarse buckets per window, fine keys per coarse bucket (NC * NF = B)
};
__global__ __launch_bounds__(CS_THREADS) void msm_csort_hist_kernel(const uint32_t *__restrict__ digits, uint32_t n, uint32_t G, CSortGeom cg,
                                                                    uint32_t *__restrict__ chist) {
    H2_SORT_PRIORITY();
    __shared__ uint32_t hist[CS_THREADS / 64][CS_MAX_NC];   // one sub-histogram per wave: no cross-wave contention on 128 counters
    const uint32_t w = blockIdx.x / G, g = blockIdx.x - w * G, tid = threadIdx.x, wave = tid >> 6;
    for (uint32_t k = tid; k < (CS_THREADS / 64) * CS_MAX_NC; k += CS_THREADS) (&hist[0][0])[k] = 0;
    __syncthreads();
    const uint32_t lo = g * CS_CHUNK, hi = lo + CS_CHUNK < n ? lo + CS_CHUNK : n;
    const uint32_t *dw = digits + (size_t)w * n;
    for (uint32_t i = lo + tid; i < hi; i += 8 * CS_THREADS) {
        uint32_t d[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) d[k] = i + k * CS_THREADS < hi ? dw[i + k * CS_THREADS] & 0x7fffffffu : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            if (d[k]) atomicAdd(&hist[wave][(d[k] - 1) >> cg.LB], 1u);
    }
    __syncthreads();
    for (uint32_t cb = tid; cb < cg.NC; cb += CS_THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t v = 0; v < CS_THREADS / 64; ++v) sum += hist[v][cb];
        chist[((size_t)w * cg.NC + cb) * G + g] = sum;
    }
}
__global__ __launch_bounds__(CS_THREADS) void msm_csort_scatter_kernel(const uint32_t *__restrict__ digits, uint32_t n, uint32_t G, CSortGeom cg,
                                                                       const uint32_t *__restrict__ cscan, uint32_t *__restrict__ centries) {
    H2_SORT_PRIORITY();
    __shared__ uint32_t lbase[CS_MAX_NC + 1], cursor[CS_MAX_NC], gbase[CS_MAX_NC];
    __shared__ uint32_t sorted[CS_CHUNK];
    const uint32_t w = blockIdx.x / G, g = blockIdx.x - w * G, tid = threadIdx.x;
    // this chunk's count per coarse bucket = the difference of neighbouring entries of the scanned array; local exclusive prefix by one wave
    if (tid < 64) {
        uint32_t run = 0;
        for (uint32_t c0 = 0; c0 < cg.NC; c0 += 64) {
            const uint32_t cb = c0 + tid;
            uint32_t cnt = 0, gb = 0;
            if (cb < cg.NC) {
                const size_t idx = ((size_t)w * cg.NC + cb) * G + g;
                gb = cscan[idx];
                cnt = cscan[idx + 1] - gb;
            }
            uint32_t incl = cnt;   // inclusive scan over the wave by shuffles
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl(incl, (int)(tid >= d ? tid - d : tid));
                if (tid >= d) incl += up;
            }
            if (cb < cg.NC) {
                lbase[cb] = run + incl - cnt;
                cursor[cb] = run + incl - cnt;
                gbase[cb] = gb;
            }
            run += __shfl(incl, 63);
        }
        if (tid == 0) lbase[cg.NC] = run;
    }
    __syncthreads();
    const uint32_t lo = g * CS_CHUNK, hi = lo + CS_CHUNK < n ? lo + CS_CHUNK : n;
    const uint32_t *dw = digits + (size_t)w * n;
    const uint32_t fmask = cg.NF - 1;
    for (uint32_t i = lo + tid; i < hi; i += 8 * CS_THREADS) {
        uint32_t dv[8], pos[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) dv[k] = i + k * CS_THREADS < hi ? dw[i + k * CS_THREADS] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t d = dv[k] & 0x7fffffffu;
            pos[k] = d ? atomicAdd(&cursor[(d - 1) >> cg.LB], 1u) : KEY_INVALID;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            if (pos[k] != KEY_INVALID)
                sorted[pos[k]] = ((i + k * CS_THREADS) << cg.LB) | (((dv[k] & 0x7fffffffu) - 1) & fmask) | (dv[k] & 0x80000000u);
    }
    __syncthreads();
    const uint32_t total = lbase[cg.NC];
    for (uint32_t p = tid; p < total; p += CS_THREADS) {
        uint32_t a = 0, b = cg.NC;   // largest cb with lbase[cb] <= p
        while (b - a > 1) {
            const uint32_t mid = (a + b) >> 1;
            if (lbase[mid] <= p) a = mid;
            else b = mid;
        }
        centries[gbase[a] + (p - lbase[a])] = sorted[p];
    }
}
__global__ __launch_bounds__(CS_THREADS) void msm_csort_fine_kernel(const uint32_t *__restrict__ centries, const uint32_t *__restrict__ cscan, uint32_t G,
                                                                    CSortGeom cg, uint32_t B, uint32_t Wcol, uint32_t table_stride, uint32_t nsort,
                                                                    uint32_t *__restrict__ sval, uint32_t *__restrict__ offsets) {
    H2_SORT_PRIORITY();
    __shared__ uint32_t fh[CS_MAX_NF], cur[CS_MAX_NF], buf[CS_BUF];
    const uint32_t tid = threadIdx.x, cbi = blockIdx.x, w = cbi / cg.NC, cb = cbi - w * cg.NC;
    const uint32_t cstart = cscan[(size_t)cbi * G], cend = cscan[(size_t)(cbi + 1) * G], size = cend - cstart;
    const uint32_t fmask = cg.NF - 1;
    for (uint32_t f = tid; f < cg.NF; f += CS_THREADS) fh[f] = 0;
    __syncthreads();
    for (uint32_t e = tid; e < size; e += CS_THREADS) atomicAdd(&fh[centries[cstart + e] & fmask], 1u);
    __syncthreads();
    // exclusive prefix over the NF <= 256 fine counts (Hillis-Steele over the first NF lanes)
    uint32_t mine = tid < cg.NF ? fh[tid] : 0u, incl = mine;
    if (tid < CS_MAX_NF) cur[tid] = incl;
    __syncthreads();
    for (uint32_t d = 1; d < cg.NF; d <<= 1) {
        uint32_t up = 0;
        if (tid < cg.NF && tid >= d) up = cur[tid - d];
        __syncthreads();
        if (tid < cg.NF) {
            incl += up;
            cur[tid] = incl;
        }
        __syncthreads();
    }
    if (tid < cg.NF) {
        const uint32_t excl = incl - mine;
        offsets[(size_t)w * B + cb * cg.NF + tid] = cstart + excl;   // the bucket boundaries the accumulation walks
        cur[tid] = excl;
    }
    if (cbi + 1 == gridDim.x && tid == 0) offsets[nsort] = cend;      // total number of entries
    __syncthreads();
    const uint32_t idx_base = (w % Wcol) * table_stride;   // precomputed bases: window w of a column reads table level w
    const bool fits = size <= CS_BUF;
    for (uint32_t e = tid; e < size; e += CS_THREADS) {
        const uint32_t v = centries[cstart + e];
        const uint32_t pos = atomicAdd(&cur[v & fmask], 1u);
        const uint32_t out = (idx_base + ((v & 0x7fffffffu) >> cg.LB)) | (v & 0x80000000u);
        if (fits) buf[pos] = out;
        else sval[cstart + pos] = out;
    }
    if (!fits) return;
    __syncthreads();
    for (uint32_t p = tid; p < size; p += CS_THREADS) sval[cstart + p] = buf[p];
}

// diagnostics (msm_debug_reorder): reorder the entries INSIDE every bucket — 1: ascending point index, 2: hashed (no order at all).  The
// sum of a bucket does not depend on it; the accumulation's table gather does (tools/msm_r03.py measures how much).  One thread per
// bucket, insertion sort in place; buckets above 1024 entries are left alone.
__global__ __launch_bounds__(256) void msm_debug_reorder_kernel(uint32_t *__restrict__ sval, const uint32_t *__restrict__ offsets, uint32_t nkeys, int mode) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nkeys) return;
    const uint32_t lo = offsets[k], hi = offsets[k + 1];
    if (hi - lo > 1024) return;
    for (uint32_t i = lo + 1; i < hi; ++i) {
        const uint32_t v = sval[i], kv = mode == 1 ? (v & 0x7fffffffu) : (v & 0x7fffffffu) * 2654435761u;
        uint32_t j = i;
        while (j > lo) {
            const uint32_t u = sval[j - 1], ku = mode == 1 ? (u & 0x7fffffffu) : (u & 0x7fffffffu) * 2654435761u;
            if (ku <= kv) break;
            sval[j] = u;
            --j;
        }
        sval[j] = v;
    }
}

// ------------------------------------------------------------------ 5. chunked bucket accumulation
// largest k in [lo, nkeys) with offsets[k] <= e   (offsets has nkeys+1 entries, e < offsets[nkeys])
// (ks = stride between consecutive run keys in `offsets`: 1, or the number of windows folded into one bucket)
__device__ __forceinline__ uint32_t find_key(const uint32_t *__restrict__ offsets, uint32_t ks, uint32_t lo, uint32_t nkeys, uint32_t e) {
    uint32_t hi = nkeys;
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (offsets[(size_t)mid * ks] <= e) lo = mid;
        else hi = mid;
    }
    return lo;
}

// One aligned 64-byte table entry.  The gather has no reuse (a 1 GiB table, one entry per addition): loaded NON-TEMPORALLY it streams past
// the L2 instead of evicting the lines the kernel does reuse — each lane's slice of the sorted entry list (32 entries per 128-byte line,
// touched over ~32 additions) and the bucket offsets.  rocprofv3 PMC (profiles/r02_hbm_counter_calibration.md): with plain loads the
// kernel issued 2.0 memory-side line requests per addition, one of them a re-fetch of such an evicted line.
template <bool NT>
__device__ __forceinline__ G1Affine load_table_entry(const G1Affine *__restrict__ p) {
#ifdef H2_HIPEMU
    return *p;
#else
    if (!NT) return *p;
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
template <bool NT, bool PF = false>
__device__ __forceinline__ void msm_accum_body(const uint32_t *__restrict__ sval, const G1Affine *__restrict__ bases,
                                               const uint32_t *__restrict__ offsets, uint32_t nkeys, uint32_t ks, uint32_t K,
                                               XYZZ29 *__restrict__ buckets, uint32_t *__restrict__ out_keys, XYZZ29 *__restrict__ out_vals,
                                               uint32_t nthreads) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wave = t >> 6;
    const uint32_t total = offsets[(size_t)nkeys * ks];
    const uint64_t start64 = (uint64_t)t * K;
    const bool valid = t < nthreads && start64 < total;
    uint32_t start = 0, end = 0, cur = 0, next = 0, hk = KEY_INVALID;
    bool first = true;
    XYZZ29 acc = XYZZ29::identity();
    __shared__ XYZZ29 lsave[256];   // the lane's first-run sum waits here (not in 36 registers) while the lane walks the rest of its entries
    if (valid) {
        start = (uint32_t)start64;
        end = (start64 + K > total) ? total : start + K;
        cur = find_key(offsets, ks, 0, nkeys, start);
        next = offsets[(size_t)(cur + 1) * ks];
    }
    const uint32_t first_key = valid ? cur : KEY_INVALID;   // key of the lane's first entry
    bool l_open = false;                                     // the lane's first run began before `start`
    if (valid) l_open = offsets[(size_t)cur * ks] < start;
    uint4 sv4 = {0u, 0u, 0u, 0u};
    auto entry = [&](uint32_t e) -> uint32_t {   // sorted entry e (called for consecutive e)
        if (NT) {   // the lane's entries 16 bytes at a time: a quarter of the loads (and of the chances to find the line evicted)
            if ((e & 3u) == 0 || e == start) sv4 = reinterpret_cast<const uint4 *>(sval)[e >> 2];
            const uint32_t sel = e & 3u;
            return sel == 0 ? sv4.x : sel == 1 ? sv4.y : sel == 2 ? sv4.z : sv4.w;
        }
        return sval[e];
    };
    // PF: the table entry of the NEXT addition is requested before this one is computed (the gather's ~2 us then overlap ~2400 instructions)
    uint32_t v_nxt = 0;
    G1Affine p_nxt;
    if (PF && start < end) {
        v_nxt = entry(start);
        p_nxt = load_table_entry<NT>(bases + (v_nxt & 0x7fffffffu));
    }
    for (uint32_t e = start; e < end; ++e) {
        uint32_t v;
        G1Affine p;
        if (PF) {
            v = v_nxt;
            p = p_nxt;
            if (e + 1 < end) {
                v_nxt = entry(e + 1);
                p_nxt = load_table_entry<NT>(bases + (v_nxt & 0x7fffffffu));
            }
        }
        if (e >= next) {   // bucket boundary: close the run
            if (first) {
                lsave[threadIdx.x] = acc;
                hk = cur;
                first = false;
            } else {
                buckets[cur] = acc;
            }
            acc = XYZZ29::identity();
            cur = (offsets[(size_t)(cur + 2) * ks] > e) ? cur + 1 : find_key(offsets, ks, cur + 1, nkeys, e);
            next = offsets[(size_t)(cur + 1) * ks];
        }
        if (!PF) {
            v = entry(e);
            p = load_table_entry<NT>(bases + (v & 0x7fffffffu));   // packed R'-domain point, one aligned 64-byte gather
        }
        if (!p.is_identity()) xyzz29_add_affine(acc, f29_split<Q29P>(p.x), f29_split<Q29P>(p.y), (v >> 31) != 0);
    }
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

template <int MINW, bool NT, bool PF = false>
__global__ __launch_bounds__(256, MINW) void msm_accum_kernel(const uint32_t *__restrict__ sval, const G1Affine *__restrict__ bases,
                                                        const uint32_t *__restrict__ offsets, uint32_t nkeys, uint32_t ks, uint32_t K,
                                                        XYZZ29 *__restrict__ buckets, uint32_t *__restrict__ out_keys,
                                                        XYZZ29 *__restrict__ out_vals, uint32_t nthreads) {
    msm_accum_body<NT, PF>(sval, bases, offsets, nkeys, ks, K, buckets, out_keys, out_vals, nthreads);
}
// Same body with the register allocation padded to 176 per lane: two waves per SIMD instead of three, which leaves a
// third of every SIMD's register file free at all times for the tail / sort kernels of the neighbouring pipelined MSMs
// (msm_accum_variant = 2).
__global__ __launch_bounds__(256) void msm_accum_w2_kernel(const uint32_t *__restrict__ sval, const G1Affine *__restrict__ bases,
                                                           const uint32_t *__restrict__ offsets, uint32_t nkeys, uint32_t ks, uint32_t K,
                                                           XYZZ29 *__restrict__ buckets, uint32_t *__restrict__ out_keys,
                                                           XYZZ29 *__restrict__ out_vals, uint32_t nthreads) {
#ifndef H2_HIPEMU
    asm volatile("" ::: "v119");
#endif
    msm_accum_body<false>(sval, bases, offsets, nkeys, ks, K, buckets, out_keys, out_vals, nthreads);
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

// ------------------------------------------------------------------ precomputed tables (H2HIP_BASES_PRECOMPUTE)
// level w holds 2^(c*w) * P_i.  Step 1: Jacobian doublings of the previous level; step 2: batch normalisation
// (Montgomery's trick over runs of NORM_RUN points, one Fermat inversion per run).
constexpr uint32_t NORM_RUN = 32;
__global__ __launch_bounds__(256) void table_double_kernel(const G1Affine *__restrict__ prev, G1Jac *__restrict__ tmp, uint32_t n, uint32_t c) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = prev[i];
    XYZZ a = XYZZ::from_affine(p);
    for (uint32_t k = 0; k < c; ++k) a = xyzz_double(a);
    tmp[i] = xyzz_to_jacobian(a);
}
__global__ __launch_bounds__(64) void table_normalize_kernel(const G1Jac *__restrict__ tmp, Fq *__restrict__ prefix, G1Affine *__restrict__ out,
                                                             uint32_t n) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo = r * NORM_RUN;
    if (lo >= n) return;
    uint32_t hi = lo + NORM_RUN < n ? lo + NORM_RUN : n;
    Fq acc = Fq::one();
    for (uint32_t i = lo; i < hi; ++i) {
        prefix[i] = acc;
        Fq z = tmp[i].z;
        if (!z.is_zero()) acc = fe_mul(acc, z);
    }
    acc = fe_inv(acc);
    for (uint32_t i = hi; i-- > lo;) {
        G1Jac p = tmp[i];
        G1Affine a;
        if (p.z.is_zero()) {
            a.x = Fq::zero();
            a.y = Fq::zero();
        } else {
            Fq zi = fe_mul(acc, prefix[i]);
            acc = fe_mul(acc, p.z);
            Fq zi2 = fe_sqr(zi);
            a.x = fe_mul(p.x, zi2);
            a.y = fe_mul(p.y, fe_mul(zi2, zi));
        }
        out[i] = a;
    }
}

// Jacobian -> affine for n points (tmp is read, out written; they may not alias)
int batch_normalize_jac(h2hip_ctx *ctx, const G1Jac *tmp, G1Affine *out, uint32_t n) {
    Fq *prefix = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP2, sizeof(Fq) * (size_t)(n ? n : 1), (void **)&prefix));
    uint32_t runs = (n + NORM_RUN - 1) / NORM_RUN;
    if (!runs) return H2HIP_OK;
    prof_begin(ctx, "table_normalize_kernel");
    hipLaunchKernelGGL(table_normalize_kernel, dim3((runs + 63) / 64), dim3(64), 0, ctx->stream, tmp, prefix, out, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// saturated affine table -> unsaturated 29-bit layout used by msm_accum_kernel
// The table is stored PACKED: 64 B per point = canonical 8x32-bit limbs of x*2^261 and y*2^261 (the unsaturated
// domain's Montgomery form), 64-byte aligned so that one gather touches exactly one half cache line; lanes unpack to
// 9x29-bit limbs with shifts only (f29_split).  Identity stays all-zero.
__global__ __launch_bounds__(256) void bases_to_29_kernel(const G1Affine *__restrict__ in, G1Affine *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = in[i], r;
    if (p.is_identity()) {
        r = p;
    } else {
        r.x = f29_pack_canonical<FqP>(f29_from_sat(p.x));
        r.y = f29_pack_canonical<FqP>(f29_from_sat(p.y));
    }
    out[i] = r;
}

// Window size by a cost model in field multiplications.  Plain bases: every window has its own bucket set, reduced at
// ~28 multiplications per bucket.  Precomputed tables: ONE bucket set, but its reduction is a chain of dependent
// additions whose latency is worth ~150 multiplications of the (parallel) accumulation per bucket — fitted to the
// measured optimum c = 13/14 at 2^16, 15/16 at 2^18, 16 at 2^19 and above (tools/c_sweep.sh).
static uint32_t pick_window(size_t n, bool precomp = false) {
    uint32_t best = 4;
    double best_cost = 1e300;
    for (uint32_t c = 4; c <= 16; ++c) {
        double W = (double)((255 + c - 1) / c);
        // precomputed tables: W*n mixed additions; every (window, bucket) pair costs a full addition in the per-index presum plus its share of
        // the run boundaries, zero fill and merge (~34 products' worth, fitted on proofs of 2^14..2^17-row shapes: tools/prove_time.py
        // --param=msm_window_bits=..); the running sums over one bucket set per column come last
        const double B = (double)(1u << (c - 1));
        double cost = precomp ? W * 10.0 * (double)n + W * B * 34.0 + 60.0 * B
                              : W * (10.0 * (double)n + 28.0 * B + 400.0 * c);
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

// (re)builds bases->pts29, the unsaturated copy every MSM reads; with `precompute` it holds W levels
// 2^(c*w) * P_i (level w at offset w*n), built level by level in saturated arithmetic and converted.
int msm_prepare_bases(h2hip_ctx *ctx, h2hip_bases *b, bool precompute) {
    const uint32_t n = (uint32_t)b->n;
    uint32_t c = 0, W = 1;
    if (precompute && n) {
        c = ctx->msm_window_bits ? (uint32_t)ctx->msm_window_bits : pick_window(b->n, true);
        H2_REQUIRE(c >= 2 && c <= 16, "window bits out of range for precomputed bases");
        W = (255 + c - 1) / c;
        H2_REQUIRE((uint64_t)b->n * W < (1ull << 31), "precomputed table too large for 31-bit indices");
    }
    G1Affine *t29 = nullptr;
    hipError_t e = hipMalloc((void **)&t29, sizeof(G1Affine) * (size_t)(n ? n : 1) * W);
    if (e != hipSuccess) {
        set_error("hipMalloc for %zu bases x %u windows failed: %s", b->n, W, hipGetErrorString(e));
        return H2HIP_ERR_NOMEM;
    }
    auto convert = [&](const G1Affine *src, uint32_t level) -> int {
        if (!n) return H2HIP_OK;
        prof_begin(ctx, "bases_to_29_kernel");
        hipLaunchKernelGGL(bases_to_29_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, src, t29 + (size_t)level * n, (size_t)n);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
        return H2HIP_OK;
    };
    int rc = convert(b->pts, 0);
    if (rc == H2HIP_OK && W > 1) {
        G1Jac *tmp = nullptr;
        G1Affine *lvl[2] = {nullptr, nullptr};
        rc = ws_reserve(ctx, h2hip_ctx::WS_TMP1, sizeof(G1Jac) * b->n, (void **)&tmp);
        if (rc == H2HIP_OK) rc = ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(G1Affine) * 2 * b->n, (void **)&lvl[0]);
        lvl[1] = lvl[0] + b->n;
        const G1Affine *prev = b->pts;
        for (uint32_t w = 1; rc == H2HIP_OK && w < W; ++w) {
            G1Affine *cur = lvl[w & 1];
            prof_begin(ctx, "table_double_kernel");
            hipLaunchKernelGGL(table_double_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, prev, tmp, n, c);
            prof_end(ctx);
            if (hipGetLastError() != hipSuccess) rc = H2HIP_ERR_HIP;
            if (rc == H2HIP_OK) rc = batch_normalize_jac(ctx, tmp, cur, n);
            if (rc == H2HIP_OK) rc = convert(cur, w);
            prev = cur;
        }
    }
    if (rc == H2HIP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = H2HIP_ERR_HIP;
    if (rc != H2HIP_OK) {
        hipFree(t29);
        return rc;
    }
    if (b->pts29) hipFree(b->pts29);
    b->pts29 = t29;
    b->tables = W;
    b->window_bits = c;
    return H2HIP_OK;
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

// windows per shared bucket set (0: none) for a context's msm_fold_windows setting
static uint32_t fold_group(const h2hip_ctx *ctx, bool precomp, uint32_t Wcol) {
    if (!precomp || ctx->msm_fold_windows <= 1) return 0;
    return (uint32_t)ctx->msm_fold_windows < Wcol ? (uint32_t)ctx->msm_fold_windows : Wcol;
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
    const uint32_t fg = fold_group(ctx, precomp, Wcol);
    const uint32_t rows = precomp ? (fg ? (Wcol + fg - 1) / fg : Wcol) : 1;   // bucket sets per column the accumulation left: [col][rows][B]
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
int buckets_clean_after_use(h2hip_ctx *ctx, int which, void *buf, size_t bytes) {
    if (!ctx->clean_stream) {
        H2_HIPCHK(hipStreamCreateWithFlags(&ctx->clean_stream, hipStreamNonBlocking));
        H2_HIPCHK(hipEventCreateWithFlags(&ctx->clean_ev, hipEventDisableTiming));
        H2_HIPCHK(hipEventCreateWithFlags(&ctx->used_ev, hipEventDisableTiming));
    }
    H2_HIPCHK(hipEventRecord(ctx->used_ev, ctx->stream));               // everything that reads the buckets is queued on the context's stream
    H2_HIPCHK(hipStreamWaitEvent(ctx->clean_stream, ctx->used_ev, 0));
    H2_HIPCHK(hipMemsetAsync(buf, 0, bytes, ctx->clean_stream));
    H2_HIPCHK(hipEventRecord(ctx->clean_ev, ctx->clean_stream));
    ctx->clean_ptr[which] = buf;
    ctx->clean_bytes[which] = bytes;
    return H2HIP_OK;
}

int msm_run_cols(h2hip_ctx *ctx, const h2hip_bases *bases, const Fr *const *scalars, uint32_t ncols, size_t n, XYZZ *out, XYZZ29 *ext_buckets,
                 uint32_t phases, bool ext_buckets_zeroed, uint32_t w_lo, uint32_t w_cnt) {
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
    const uint32_t Wfull = (255 + c - 1) / c;   // windows of one column
    H2_REQUIRE(Wfull <= 64, "too many windows");
    H2_REQUIRE(!precomp || bases->tables >= Wfull, "precomputed table has too few windows");
    // w_cnt != 0: only the windows [w_lo, w_lo + w_cnt) of the one column, their buckets into ext_buckets (= the column's array + w_lo * B):
    // table level w_lo + w is reached by handing the kernels the table from that level on
    H2_REQUIRE(w_cnt == 0 || (ncols == 1 && precomp && ext_buckets && w_lo + w_cnt <= Wfull), "a window sub-range needs one column, precomputed bases and a shared bucket array");
    const uint32_t Wcol = w_cnt ? w_cnt : Wfull;
    const uint32_t W = Wcol * ncols;           // windows the sort / accumulation see
    const G1Affine *table = (const G1Affine *)bases->pts29 + (w_cnt ? (size_t)w_lo * bases->n : 0);
    const uint32_t B = 1u << (c - 1);
    FoldKeys fold_w;
    fold_w.fg = w_cnt ? 0u : fold_group(ctx, precomp, Wcol);
    fold_w.ng = fold_w.fg ? (Wcol + fold_w.fg - 1) / fold_w.fg : 0u;
    fold_w.wcol = Wcol;
    const uint32_t ks = fold_w.fg ? fold_w.fg : 1u;
    const uint32_t nsort = fold_w.fg ? ncols * fold_w.ng * fold_w.fg * B : W * B;   // keys of the counting sort (>= W*B: a short last group is padded)
    const uint32_t nkeys = nsort / ks;     // run keys of the accumulation = bucket slots
    const uint64_t emax = (uint64_t)n * W;
    H2_REQUIRE(emax < 0xFFFFFFF0ull, "n*W overflows 32 bits");
    uint32_t K1 = (uint32_t)ctx->msm_chunk;
    if (K1 == 0) {   // auto: as long as possible (fewer shared runs to merge) while the grid is still several waves per SIMD
        // Measured at 2^19 / 17 windows (tools/msm_r03.py): 34 entries per lane = 4096 waves 0.72 ms; 32 = 4352 waves 0.81 ms; 64 = 2176 waves
        // 0.98 ms; ONE exact round of two waves per SIMD (68 entries = 2048 waves) 0.72 ms although its wave-level merge is a single
        // addition per lane — with one round the kernel ends with its slowest wave, shorter lanes in several rounds balance themselves.
        uint64_t k = emax / 262144;
        K1 = k < 8 ? 8u : k > 64 ? 64u : (uint32_t)k;
    }
    // chunking of the counting sort: about 32 chunks per window, 4Ki..64Ki scalars each.  32 = the CUs of an XCD: the (window, chunk)
    // workgroups of one window run together on the window's XCD, one per CU (tried: chunks sized for ONE round over the whole chip,
    // W * G <= CUs — 2^19: scatter 0.081 -> 0.145 ms, 2^20: 0.154 -> 0.242 ms: half of every XCD's CUs idle and three windows' slices
    // competing for one 4 MiB L2).
    uint32_t chunk = (uint32_t)((n + 31) / 32);
    if (chunk < 4096) chunk = 4096;
    if (chunk > 65536) chunk = 65536;
    const uint32_t G = (uint32_t)((n + chunk - 1) / chunk);
    const uint32_t sort_grid = sort_grid_size(W, G);

    uint32_t *digits, *bhist, *counts, *offsets, *sval, *pkey[2];
    XYZZ29 *buckets, *pval[2];
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_DIGITS, sizeof(uint32_t) * emax, (void **)&digits));
    // two-level sort (r03, the default): coarse buckets of a window = the top 7 bits of the key (fewer for small windows), fine keys below
    CSortGeom cg;
    cg.LB = c - 1 > 7 ? c - 1 - 7 : (c - 1) / 2;
    cg.NF = 1u << cg.LB;
    cg.NC = B >> cg.LB;
    const uint32_t Gc = (uint32_t)((n + CS_CHUNK - 1) / CS_CHUNK);
    const size_t ncoarse = (size_t)W * cg.NC * Gc;
    // measured (profiles/r03_msm_sort_ab.log, r03_msm_reorder.log): on uniform scalars the sort itself is ahead from 2^18 points on (2^19:
    // 0.124 -> 0.087 ms per MSM, 2^20: 0.24 -> 0.18; 2^16: many tiny coarse buckets, 0.03 -> 0.10 ms), but at 2^19 the accumulation that
    // follows runs 5-9 % slower on its output (the entry order inside a bucket moves the gather by that much: ascending 0.70, hashed 0.74 ms)
    // and the k = 19 proof does not gain (15.4 -> 15.5-15.7 ms); on a circuit's 0/1-heavy columns ONE workgroup walks the coarse bucket that
    // holds a quarter of a window (0.5 M entries at 2^21) and the proofs at k >= 20 lose 15-20 % (k = 21: 62 -> 76 ms).  So the one-pass sort
    // stays the default at every size; msm_sort_mode 2 selects this one (it would need heavy coarse buckets dealt over several workgroups).
    const bool two_level = ctx->msm_sort_mode == 2 && !fold_w.fg && n <= ((size_t)1 << (31 - cg.LB)) && cg.NC <= CS_MAX_NC && cg.NF <= CS_MAX_NF &&
                           ncoarse + 1 <= (size_t)nsort + 1 + (1u << 20);
    uint32_t *centries = nullptr, *cscan = nullptr;
    if (two_level) {
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_CURSOR, sizeof(uint32_t) * (ncoarse + 2), (void **)&cscan));
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SKEY, sizeof(uint32_t) * (emax + 4), (void **)&centries));
        bhist = nullptr;
    } else {
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_CURSOR, sizeof(uint32_t) * (size_t)W * G * B, (void **)&bhist));
    }
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_COUNTS, sizeof(uint32_t) * ((size_t)nsort + 1 + (1u << 20)), (void **)&counts));   // (+ room for the two-level sort's coarse counts)
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OFFSETS, sizeof(uint32_t) * ((size_t)nsort + ks + 1), (void **)&offsets));
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

    // `phases` lets a caller issue the latency-bound sort and merge of one MSM and its multiplier-bound accumulation on different
    // streams (all derived sizes and scratch pointers are recomputed identically on every call for the same arguments)
    if (phases & MSM_PHASE_SORT) {
    if (nsort != W * B) H2_HIPCHK(hipMemsetAsync(counts, 0, sizeof(uint32_t) * nsort, st));   // padded keys: no window writes their counts
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
    H2_REQUIRE(ks <= 256, "fold group too large");
    hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256), ncols), dim3(256), 0, st, dcols, (uint32_t)n, c, Wcol, digits,
                       two_level ? counts + ncoarse : counts + nsort, offsets + nsort + 1, ks, w_lo);   // + the sentinels counts[last], offsets[(nkeys + 1) * ks]
    prof_end(ctx);
    if (two_level) {
        prof_begin(ctx, "msm_hist_kernel");
        hipLaunchKernelGGL(msm_csort_hist_kernel, dim3(W * Gc), dim3(CS_THREADS), 0, st, (const uint32_t *)digits, (uint32_t)n, Gc, cg, counts);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
        H2_CHK(exclusive_scan_u32(ctx, counts, cscan, (uint32_t)(ncoarse + 1)));   // (window, coarse bucket, chunk) order; the last entry is the total
        prof_begin(ctx, "msm_scatter_kernel");
        hipLaunchKernelGGL(msm_csort_scatter_kernel, dim3(W * Gc), dim3(CS_THREADS), 0, st, (const uint32_t *)digits, (uint32_t)n, Gc, cg, (const uint32_t *)cscan,
                           centries);
        hipLaunchKernelGGL(msm_csort_fine_kernel, dim3(W * cg.NC), dim3(CS_THREADS), 0, st, (const uint32_t *)centries, (const uint32_t *)cscan, Gc, cg, B, Wcol,
                           precomp ? (uint32_t)bases->n : 0u, nsort, sval, offsets);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
    } else {
    const uint32_t sort_threads = (uint32_t)ctx->msm_sort_threads;
    if (!ctx->msm_lds_attr_set) {   // dynamic LDS above 64 KiB has to be enabled per kernel (and device) once
        H2_HIPCHK(hipFuncSetAttribute((const void *)msm_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint32_t) * MAX_LDS_BUCKETS)));
        H2_HIPCHK(hipFuncSetAttribute((const void *)msm_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint32_t) * MAX_LDS_BUCKETS)));
        ctx->msm_lds_attr_set = true;
    }
    prof_begin(ctx, "msm_hist_kernel");
    hipLaunchKernelGGL(msm_hist_kernel, dim3(sort_grid), dim3(sort_threads), sizeof(uint32_t) * B, st, (const uint32_t *)digits, (uint32_t)n, W, B, G, chunk, bhist);
    prof_end(ctx);
    prof_begin(ctx, "msm_hist_scan_kernel");
    hipLaunchKernelGGL(msm_hist_scan_kernel, dim3((nsort + 255) / 256), dim3(256), 0, st, bhist, W, B, G, fold_w, counts);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(exclusive_scan_u32(ctx, counts, offsets, nsort + 1));
    prof_begin(ctx, "msm_scatter_kernel");
    uint32_t S = (uint32_t)ctx->msm_scatter_split;   // sub-ranges per window: keep a segment's output slice (n*4/S bytes) within ~2 MiB
    if (S == 0) {
        S = 1;
        while (S < 4 && S * 2 <= B && ((uint64_t)n * 4) / S > (2u << 20)) S *= 2;
    }
    if (S > B) S = B;
    const uint32_t scatter_grid = sort_grid_size(W * S, G);
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(scatter_grid), dim3(sort_threads), ctx->msm_scatter_full_lds ? sizeof(uint32_t) * MAX_LDS_BUCKETS : sizeof(uint32_t) * (B / S), st,   // full 128 KiB: one workgroup per CU keeps a segment's writes on one XCD
                       (const uint32_t *)digits, (uint32_t)n, W, B, G, chunk, S,
                       precomp ? (uint32_t)bases->n : 0u, Wcol, fold_w, (const uint32_t *)offsets, (const uint32_t *)bhist, sval);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    }   // one-pass sort
    if (ctx->msm_debug_reorder && ks == 1) {
        prof_begin(ctx, "msm_debug_reorder_kernel");
        hipLaunchKernelGGL(msm_debug_reorder_kernel, dim3((nkeys + 255) / 256), dim3(256), 0, st, sval, (const uint32_t *)offsets, nkeys, ctx->msm_debug_reorder);
        prof_end(ctx);
    }
    }   // MSM_PHASE_SORT

    if (phases & MSM_PHASE_ACCUM) {
    prof_begin(ctx, "msm_accum_kernel");
    if (ctx->msm_accum_variant == 6 || ctx->msm_accum_variant == 7) {   // software-prefetched table gather at 2 / 3 waves per SIMD
        if (ctx->msm_accum_variant == 6)
            hipLaunchKernelGGL((msm_accum_kernel<2, true, true>), dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)sval, table,
                               (const uint32_t *)offsets, nkeys, ks, K1, buckets, pkey[0], pval[0], T1);
        else
            hipLaunchKernelGGL((msm_accum_kernel<3, true, true>), dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)sval, table,
                               (const uint32_t *)offsets, nkeys, ks, K1, buckets, pkey[0], pval[0], T1);
    } else if (ctx->msm_accum_variant == 5)   // two waves per SIMD by launch bounds (256 registers: the wave-level merge's epilogue then spills nothing)
        hipLaunchKernelGGL((msm_accum_kernel<2, true>), dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)sval, table,
                           (const uint32_t *)offsets, nkeys, ks, K1, buckets, pkey[0], pval[0], T1);
    else if (ctx->msm_accum_variant == 2)
        hipLaunchKernelGGL(msm_accum_w2_kernel, dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)sval, table,
                           (const uint32_t *)offsets, nkeys, ks, K1, buckets, pkey[0], pval[0], T1);
    else if (ctx->msm_accum_variant == 4)
        hipLaunchKernelGGL((msm_accum_kernel<4, false>), dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)sval, table,
                           (const uint32_t *)offsets, nkeys, ks, K1, buckets, pkey[0], pval[0], T1);
    else if (ctx->msm_table_nontemporal)
        hipLaunchKernelGGL((msm_accum_kernel<3, true>), dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)sval, table,
                           (const uint32_t *)offsets, nkeys, ks, K1, buckets, pkey[0], pval[0], T1);
    else
        hipLaunchKernelGGL((msm_accum_kernel<3, false>), dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)sval, table,
                           (const uint32_t *)offsets, nkeys, ks, K1, buckets, pkey[0], pval[0], T1);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    }   // MSM_PHASE_ACCUM
    if (phases & MSM_PHASE_MERGE) {
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
    }   // MSM_PHASE_MERGE

    if (ext_buckets || !(phases & MSM_PHASE_REDUCE)) return H2HIP_OK;   // the caller reduces several MSMs' buckets together
    H2_CHK(msm_reduce_cols(ctx, bases, c, buckets, ncols, out));
    return buckets_clean_after_use(ctx, 0, buckets, sizeof(XYZZ29) * nkeys);
}

int msm_run(h2hip_ctx *ctx, const h2hip_bases *bases, const Fr *scalars, size_t n, XYZZ *out) {
    return msm_run_cols(ctx, bases, &scalars, 1, n, out, nullptr, MSM_PHASE_ALL, false, 0, 0);
}

}  // namespace h2
