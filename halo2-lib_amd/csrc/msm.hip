// Pippenger multi-scalar multiplication over BN254 G1 for gfx950 — the device replacement for
// halo2_proofs::arithmetic::best_multiexp [UPSTREAM halo2-axiom 0.5.3 / halo2curves-axiom 0.7.3; the
// reference reaches it only through commit / commit_lagrange inside create_proof,
// /root/reference/halo2-base/src/utils/testing.rs:40-47; SURVEY.md §3.2 K1].
//
// The result (one group element) is unique, so the algorithm is free to differ from the CPU one:
//   1. msm_digits    one lane per scalar: Montgomery -> canonical, signed c-bit digits, histogram (atomics)
//   2. scan          exclusive prefix sum of the histogram -> bucket offsets
//   3. msm_scatter   counting sort: (bucket key, base index | sign) pairs grouped by bucket
//   4. msm_accum     *distribution-oblivious* bucket accumulation: every lane owns K consecutive sorted
//                    entries (perfect balance for uniform and for 0/1-heavy circuit columns alike), adds
//                    them in XYZZ coordinates, writes complete interior runs straight to their bucket and
//                    hands its first/last (possibly shared) runs to the next level as (key, XYZZ) partials;
//                    levels repeat on the partial list until one lane remains.  No atomics on points.
//   5. msm_seg/winsum/fold   per-window running sums sum_j j*B_j over short segments (+ small scalar
//                    multiple), tree sum per window, 2^(c*w) fold.
// Bases stay resident in HBM (h2hip_bases); signs are applied by negating y on load.
#include "internal.h"

namespace h2 {

constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;

// ------------------------------------------------------------------ 1. digits + histogram
__global__ __launch_bounds__(256) void msm_digits_kernel(const Fr *__restrict__ scalars, uint32_t n, uint32_t c, uint32_t W,
                                                         uint32_t *__restrict__ digits, uint32_t *__restrict__ counts,
                                                         uint32_t keys_per_window) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = fe_from_mont(scalars[i]);
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
        digits[(size_t)w * n + i] = d | (neg << 31);
        if (d) atomicAdd(&counts[w * keys_per_window + d - 1], 1u);
    }
}

// ------------------------------------------------------------------ 2. exclusive scan of u32 (3 kernels)
constexpr uint32_t SCAN_TILE = 1024;   // 256 lanes x 4
__global__ __launch_bounds__(256) void scan_tile_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                        uint32_t *__restrict__ tile_sums, uint32_t n) {
    __shared__ uint32_t sh[256];
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
    __shared__ uint32_t sh[1024];
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
__global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ tile_sums, uint32_t n) {
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4, add = tile_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) out[base + k] += add;
}
static int exclusive_scan_u32(h2hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n) {
    uint32_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t *sums = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SCAN, sizeof(uint32_t) * (ntiles + 1), (void **)&sums));
    prof_begin(ctx, "scan_kernels");
    hipLaunchKernelGGL(scan_tile_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, in, out, sums, n);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums, ntiles);
    hipLaunchKernelGGL(scan_add_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, out, (const uint32_t *)sums, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// ------------------------------------------------------------------ 3. counting-sort scatter
__global__ __launch_bounds__(256) void msm_scatter_kernel(const uint32_t *__restrict__ digits, uint32_t n, uint32_t W,
                                                          uint32_t keys_per_window, uint32_t precomp,
                                                          const uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor,
                                                          uint32_t *__restrict__ skey, uint32_t *__restrict__ sval) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)n * W) return;
    uint32_t dv = digits[g];
    uint32_t d = dv & 0x7fffffffu;
    if (!d) return;
    uint32_t w = (uint32_t)(g / n), i = (uint32_t)(g - (size_t)w * n);
    uint32_t key = w * keys_per_window + d - 1;
    uint32_t pos = offsets[key] + atomicAdd(&cursor[key], 1u);
    skey[pos] = key;
    sval[pos] = (precomp ? (uint32_t)g : i) | (dv & 0x80000000u);
}

// ------------------------------------------------------------------ 4. chunked bucket accumulation
// AFFINE = true : level 1, vals = (base index | sign<<31), entries added with the mixed XYZZ+affine formula
// AFFINE = false: level >= 2, vals = XYZZ partial sums
template <bool AFFINE>
__global__ __launch_bounds__(256) void msm_accum_kernel(const uint32_t *__restrict__ keys, const void *__restrict__ vals_,
                                                        const G1Affine *__restrict__ bases, const uint32_t *__restrict__ total_ptr,
                                                        uint32_t total_fixed, uint32_t K, XYZZ *__restrict__ buckets,
                                                        uint32_t *__restrict__ out_keys, XYZZ *__restrict__ out_vals,
                                                        uint32_t nthreads, uint32_t final_level) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    const uint32_t total = total_ptr ? *total_ptr : total_fixed;
    uint64_t start = (uint64_t)t * K, end = start + K;
    if (end > total) end = total;
    uint32_t cur = KEY_INVALID, hk = KEY_INVALID, tk = KEY_INVALID;
    bool first = true;
    XYZZ acc = XYZZ::identity();
    for (uint64_t e = start; e < end; ++e) {
        uint32_t k = keys[e];
        if (k == KEY_INVALID) continue;
        if (k != cur) {
            if (cur != KEY_INVALID) {
                if (first && !final_level) {
                    out_vals[2 * (size_t)t] = acc;
                    hk = cur;
                } else {
                    buckets[cur] = acc;
                }
                first = false;
            }
            cur = k;
            acc = XYZZ::identity();
        }
        if (AFFINE) {
            uint32_t v = ((const uint32_t *)vals_)[e];
            G1Affine p = bases[v & 0x7fffffffu];
            if (!p.is_identity()) {
                if (v >> 31) p.y = fe_neg(p.y);
                xyzz_add_affine(acc, p.x, p.y);
            }
        } else {
            XYZZ p = ((const XYZZ *)vals_)[e];
            xyzz_add(acc, p);
        }
    }
    if (cur != KEY_INVALID) {
        if (final_level) {
            buckets[cur] = acc;
        } else if (first) {
            out_vals[2 * (size_t)t] = acc;
            hk = cur;
        } else {
            out_vals[2 * (size_t)t + 1] = acc;
            tk = cur;
        }
    }
    if (!final_level) {
        out_keys[2 * (size_t)t] = hk;
        out_keys[2 * (size_t)t + 1] = tk;
    }
}

// ------------------------------------------------------------------ 5. bucket reduction
__device__ __forceinline__ XYZZ xyzz_small_mul(const XYZZ &p, uint32_t k) {
    XYZZ r = XYZZ::identity();
    for (int bit = 31 - __clz(k | 1u); bit >= 0; --bit) {
        r = xyzz_double(r);
        if ((k >> bit) & 1u) xyzz_add(r, p);
    }
    return r;
}
// one lane per segment of L buckets: sum_{b in seg} (b+1) * bucket[b]
__global__ __launch_bounds__(64) void msm_seg_kernel(const XYZZ *__restrict__ buckets, XYZZ *__restrict__ seg_out, uint32_t B, uint32_t L,
                                                     uint32_t nseg_total) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nseg_total) return;
    uint32_t per = B / L, w = g / per, lo = (g - w * per) * L;
    const XYZZ *bw = buckets + (size_t)w * B;
    XYZZ run = XYZZ::identity(), acc = XYZZ::identity();
    for (int b = (int)(lo + L) - 1; b >= (int)lo; --b) {
        xyzz_add(run, bw[b]);
        xyzz_add(acc, run);
    }
    if (lo) xyzz_add(acc, xyzz_small_mul(run, lo));
    seg_out[g] = acc;
}
// one workgroup per window: tree sum of its segment results
__global__ __launch_bounds__(256) void msm_winsum_kernel(const XYZZ *__restrict__ seg, XYZZ *__restrict__ win_out, uint32_t per) {
    __shared__ XYZZ sh[256];
    uint32_t tid = threadIdx.x, w = blockIdx.x;
    XYZZ acc = XYZZ::identity();
    for (uint32_t i = tid; i < per; i += 256) xyzz_add(acc, seg[(size_t)w * per + i]);
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        if (tid < d) {
            XYZZ a = sh[tid];
            xyzz_add(a, sh[tid + d]);
            sh[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) win_out[w] = sh[0];
}
// out = sum_w 2^(c*w) * win[w]   (Wr <= 64 windows, one lane each, then a tree)
__global__ __launch_bounds__(64) void msm_fold_kernel(const XYZZ *__restrict__ win, uint32_t Wr, uint32_t c, XYZZ *__restrict__ out) {
    __shared__ XYZZ sh[64];
    uint32_t tid = threadIdx.x;
    XYZZ p = XYZZ::identity();
    if (tid < Wr) {
        p = win[tid];
        for (uint32_t i = 0; i < c * tid; ++i) p = xyzz_double(p);
    }
    sh[tid] = p;
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

// ------------------------------------------------------------------ host driver
static uint32_t pick_window(size_t n) {
    uint32_t best = 4;
    double best_cost = 1e300;
    for (uint32_t c = 4; c <= 20; ++c) {
        double W = (double)((255 + c - 1) / c);
        double cost = W * (10.0 * (double)n + 28.0 * (double)(1u << (c - 1)) + 400.0 * c);
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

int msm_run(h2hip_ctx *ctx, const h2hip_bases *bases, const Fr *scalars, size_t n, XYZZ *out) {
    H2_REQUIRE(n <= bases->n, "more scalars than resident bases");
    H2_REQUIRE(n < (1u << 27), "n too large for 32-bit entry indices");
    hipStream_t st = ctx->stream;
    if (n == 0) {
        H2_HIPCHK(hipMemsetAsync(out, 0, sizeof(XYZZ), st));
        return H2HIP_OK;
    }
    const bool precomp = bases->tables > 1;
    const uint32_t c = precomp ? bases->window_bits : (ctx->msm_window_bits ? (uint32_t)ctx->msm_window_bits : pick_window(n));
    H2_REQUIRE(c >= 2 && c <= 23, "window bits out of range");
    const uint32_t W = (255 + c - 1) / c;
    H2_REQUIRE(!precomp || bases->tables >= W, "precomputed table has too few windows");
    H2_REQUIRE(!precomp || n == bases->n, "precomputed bases require n == table size");
    const uint32_t B = 1u << (c - 1);
    const uint32_t Wr = precomp ? 1 : W;            // windows present in the bucket array
    const uint32_t nkeys = Wr * B;
    const uint32_t kpw = precomp ? 0 : B;           // key stride per window
    const uint64_t emax = (uint64_t)n * W;
    H2_REQUIRE(emax < 0xFFFFFFF0ull, "n*W overflows 32 bits");
    const uint32_t K1 = (uint32_t)ctx->msm_chunk, K2 = (uint32_t)ctx->msm_chunk2;
    uint32_t L = (uint32_t)ctx->msm_seg;
    if (L > B) L = B;

    uint32_t *digits, *counts, *offsets, *cursor, *skey, *sval, *pkey[2];
    XYZZ *buckets, *pval[2], *seg, *win;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_DIGITS, sizeof(uint32_t) * emax, (void **)&digits));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_COUNTS, sizeof(uint32_t) * (nkeys + 1), (void **)&counts));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OFFSETS, sizeof(uint32_t) * (nkeys + 1), (void **)&offsets));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_CURSOR, sizeof(uint32_t) * nkeys, (void **)&cursor));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SKEY, sizeof(uint32_t) * emax, (void **)&skey));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SVAL, sizeof(uint32_t) * emax, (void **)&sval));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_BUCKETS, sizeof(XYZZ) * nkeys, (void **)&buckets));
    const uint32_t T1 = (uint32_t)((emax + K1 - 1) / K1);
    const uint32_t T2 = (2 * T1 + K2 - 1) / K2;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PKEY0, sizeof(uint32_t) * 2 * (size_t)T1, (void **)&pkey[0]));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PVAL0, sizeof(XYZZ) * 2 * (size_t)T1, (void **)&pval[0]));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PKEY1, sizeof(uint32_t) * 2 * (size_t)T2, (void **)&pkey[1]));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_PVAL1, sizeof(XYZZ) * 2 * (size_t)T2, (void **)&pval[1]));
    const uint32_t nseg = Wr * (B / L);
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SEG, sizeof(XYZZ) * nseg, (void **)&seg));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_WIN, sizeof(XYZZ) * 64, (void **)&win));
    H2_REQUIRE(Wr <= 64, "too many windows");

    H2_HIPCHK(hipMemsetAsync(counts, 0, sizeof(uint32_t) * (nkeys + 1), st));
    H2_HIPCHK(hipMemsetAsync(cursor, 0, sizeof(uint32_t) * nkeys, st));
    H2_HIPCHK(hipMemsetAsync(buckets, 0, sizeof(XYZZ) * nkeys, st));

    prof_begin(ctx, "msm_digits_kernel");
    hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, scalars, (uint32_t)n, c, W, digits, counts, kpw);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(exclusive_scan_u32(ctx, counts, offsets, nkeys + 1));
    prof_begin(ctx, "msm_scatter_kernel");
    hipLaunchKernelGGL(msm_scatter_kernel, dim3((uint32_t)((emax + 255) / 256)), dim3(256), 0, st, (const uint32_t *)digits, (uint32_t)n, W,
                       kpw, precomp ? 1u : 0u, (const uint32_t *)offsets, cursor, skey, sval);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());

    // level 1 over the sorted entries (true count lives on the device: offsets[nkeys])
    prof_begin(ctx, "msm_accum_kernel<affine>");
    hipLaunchKernelGGL(msm_accum_kernel<true>, dim3((T1 + 255) / 256), dim3(256), 0, st, (const uint32_t *)skey, (const void *)sval,
                       (const G1Affine *)bases->pts, (const uint32_t *)(offsets + nkeys), 0u, K1, buckets, pkey[0], pval[0], T1, 0u);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    uint32_t len = 2 * T1;
    int src = 0;
    while (len > K2) {
        uint32_t T = (len + K2 - 1) / K2;
        prof_begin(ctx, "msm_accum_kernel<xyzz>");
        hipLaunchKernelGGL(msm_accum_kernel<false>, dim3((T + 255) / 256), dim3(256), 0, st, (const uint32_t *)pkey[src],
                           (const void *)pval[src], (const G1Affine *)nullptr, (const uint32_t *)nullptr, len, K2, buckets, pkey[src ^ 1],
                           pval[src ^ 1], T, 0u);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
        len = 2 * T;
        src ^= 1;
    }
    prof_begin(ctx, "msm_accum_kernel<xyzz>");
    hipLaunchKernelGGL(msm_accum_kernel<false>, dim3(1), dim3(64), 0, st, (const uint32_t *)pkey[src], (const void *)pval[src],
                       (const G1Affine *)nullptr, (const uint32_t *)nullptr, len, len, buckets, pkey[src ^ 1], pval[src ^ 1], 1u, 1u);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());

    prof_begin(ctx, "msm_seg_kernel");
    hipLaunchKernelGGL(msm_seg_kernel, dim3((nseg + 63) / 64), dim3(64), 0, st, (const XYZZ *)buckets, seg, B, L, nseg);
    prof_end(ctx);
    prof_begin(ctx, "msm_winsum_kernel");
    hipLaunchKernelGGL(msm_winsum_kernel, dim3(Wr), dim3(256), 0, st, (const XYZZ *)seg, win, B / L);
    prof_end(ctx);
    prof_begin(ctx, "msm_fold_kernel");
    hipLaunchKernelGGL(msm_fold_kernel, dim3(1), dim3(64), 0, st, (const XYZZ *)win, Wr, c, out);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

}  // namespace h2
