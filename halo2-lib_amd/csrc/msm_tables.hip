// Resident base tables for the MSM (msm.hip): conversion of the SRS points to packed 64-byte 9x29-limb entries and, with
// H2HIP_BASES_PRECOMPUTE, the window table 2^(c*w) * P_i for every window w — the device side of what halo2's
// ParamsKZG keeps as `g` / `g_lagrange` [UPSTREAM halo2-axiom 0.5.3 poly/kzg/commitment.rs; the reference holds them behind
// gen_srs, /root/reference/halo2-base/src/utils/mod.rs:441].  Built once per SRS (h2hip_bases_create), never in a proof.
#include "internal.h"
#include "ec29.cuh"

namespace h2 {

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
// ... or into 128-byte entries that already hold the 9 x 29-bit limbs (TableEntry29, internal.h)
__global__ __launch_bounds__(256) void bases_to_29_split_kernel(const G1Affine *__restrict__ in, TableEntry29 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = in[i];
    TableEntry29 e;
#pragma unroll
    for (int j = 0; j < 14; ++j) e.pad[j] = 0;
    if (p.is_identity()) {
#pragma unroll
        for (int j = 0; j < 9; ++j) e.x[j] = e.y[j] = 0;
    } else {
        // the SAME limbs the packed format yields after f29_split: canonical value of x*2^261, split at 29-bit boundaries
        const Fq29 x = f29_split<Q29P>(f29_pack_canonical<FqP>(f29_from_sat(p.x))), y = f29_split<Q29P>(f29_pack_canonical<FqP>(f29_from_sat(p.y)));
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            e.x[j] = x.l[j];
            e.y[j] = y.l[j];
        }
    }
    out[i] = e;
}

// Window size by a cost model in field multiplications.  Plain bases: every window has its own bucket set, reduced at
// ~28 multiplications per bucket.  Precomputed tables: ONE bucket set, but its reduction is a chain of dependent
// additions whose latency is worth ~150 multiplications of the (parallel) accumulation per bucket — fitted to the
// measured optimum c = 13/14 at 2^16, 15/16 at 2^18, 16 at 2^19 and above (tools/c_sweep.sh).
uint32_t pick_window(size_t n, bool precomp) {
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
    bool split = ctx->msm_table_split != 0;
    G1Affine *t29 = nullptr;   // (typed as the packed format; the split format is addressed through TableEntry29 *)
    hipError_t e = hipMalloc((void **)&t29, (split ? sizeof(TableEntry29) : sizeof(G1Affine)) * (size_t)(n ? n : 1) * W);
    if (e != hipSuccess && split) {   // the 128-byte pre-split entries double the table; when they do not fit (several keys / contexts
        (void)hipGetLastError();      // in flight, a smaller part), the 64-byte packed format msm_accum_kernel<false> reads is tried before giving up
        split = false;
        t29 = nullptr;
        e = hipMalloc((void **)&t29, sizeof(G1Affine) * (size_t)(n ? n : 1) * W);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc for %zu bases x %u windows failed: %s", b->n, W, hipGetErrorString(e));
        return H2HIP_ERR_NOMEM;
    }
    auto convert = [&](const G1Affine *src, uint32_t level) -> int {
        if (!n) return H2HIP_OK;
        prof_begin(ctx, "bases_to_29_kernel");
        if (split)
            hipLaunchKernelGGL(bases_to_29_split_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, src, (TableEntry29 *)t29 + (size_t)level * n, (size_t)n);
        else
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
    b->split = split;
    b->tables = W;
    b->window_bits = c;
    return H2HIP_OK;
}

}  // namespace h2
