// Multi-scalar multiplication over BN254 G2 (the sextic twist y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2+1)).
// north_star lists G1/G2; in the reference G2 appears only as the two verifier-side elements of ParamsKZG (g2, s*g2:
// halo2-base/src/utils/mod.rs:401-443 loads them, halo2-base/src/utils/testing.rs:64-88 consumes them in verify_proof) — no prover call site
// commits over G2, so this path is built for completeness, not tuned like the G1 MSM: 8-bit unsigned windows, one
// workgroup per (chunk of points, window) with one lane per bucket (no atomics: a lane scans the chunk's digits and adds the points that fall
// into ITS bucket), per-window reduction sum_b b*B_b by bit decomposition (8 LDS tree sums), the 2^(8w) fold on the host.
#include "internal.h"

namespace h2 {

struct F2 {
    Fq c0, c1;
};
H2_HD F2 f2_zero() { return {Fq::zero(), Fq::zero()}; }
H2_HD F2 f2_one() { return {Fq::one(), Fq::zero()}; }
H2_HD bool f2_is_zero(const F2 &a) { return a.c0.is_zero() && a.c1.is_zero(); }
H2_HD bool f2_eq(const F2 &a, const F2 &b) { return a.c0 == b.c0 && a.c1 == b.c1; }
H2_HD F2 f2_add(const F2 &a, const F2 &b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
H2_HD F2 f2_sub(const F2 &a, const F2 &b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
H2_HD F2 f2_neg(const F2 &a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
H2_HD F2 f2_dbl(const F2 &a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
H2_HD F2 f2_mul(const F2 &a, const F2 &b) {   // Karatsuba: 3 base-field products
    Fq t0 = fe_mul(a.c0, b.c0), t1 = fe_mul(a.c1, b.c1);
    return {fe_sub(t0, t1), fe_sub(fe_sub(fe_mul(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1)), t0), t1)};
}
H2_HD F2 f2_sqr(const F2 &a) { return {fe_mul(fe_add(a.c0, a.c1), fe_sub(a.c0, a.c1)), fe_dbl(fe_mul(a.c0, a.c1))}; }
H2_HD F2 f2_inv(const F2 &a) {
    Fq d = fe_inv(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return {fe_mul(a.c0, d), fe_neg(fe_mul(a.c1, d))};
}

struct alignas(16) G2Affine {   // halo2curves G2Affine{x: Fq2, y: Fq2}, 128 B, identity = all zero
    F2 x, y;
    H2_HD bool is_identity() const { return f2_is_zero(x) && f2_is_zero(y); }
};
struct alignas(16) G2Jac {      // x = X/Z^2, y = Y/Z^3, identity Z = 0
    F2 x, y, z;
    H2_HD static G2Jac identity() { return {f2_zero(), f2_one(), f2_zero()}; }
    H2_HD bool is_identity() const { return f2_is_zero(z); }
};
H2_HD G2Jac g2_double(const G2Jac &p) {   // dbl-2009-l (a = 0)
    if (p.is_identity()) return p;
    F2 A = f2_sqr(p.x), B = f2_sqr(p.y), C = f2_sqr(B);
    F2 D = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(p.x, B)), A), C));
    F2 E = f2_add(f2_dbl(A), A), F = f2_sqr(E);
    G2Jac r;
    r.x = f2_sub(F, f2_dbl(D));
    r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), f2_dbl(f2_dbl(f2_dbl(C))));
    r.z = f2_dbl(f2_mul(p.y, p.z));
    return r;
}
H2_HD G2Jac g2_add(const G2Jac &p, const G2Jac &q) {   // add-2007-bl with every exceptional case
    if (p.is_identity()) return q;
    if (q.is_identity()) return p;
    F2 Z1Z1 = f2_sqr(p.z), Z2Z2 = f2_sqr(q.z);
    F2 U1 = f2_mul(p.x, Z2Z2), U2 = f2_mul(q.x, Z1Z1);
    F2 S1 = f2_mul(f2_mul(p.y, q.z), Z2Z2), S2 = f2_mul(f2_mul(q.y, p.z), Z1Z1);
    F2 H = f2_sub(U2, U1), rr = f2_sub(S2, S1);
    if (f2_is_zero(H)) return f2_is_zero(rr) ? g2_double(p) : G2Jac::identity();
    F2 I = f2_sqr(f2_dbl(H)), J = f2_mul(H, I);
    rr = f2_dbl(rr);
    F2 V = f2_mul(U1, I);
    G2Jac r;
    r.x = f2_sub(f2_sub(f2_sqr(rr), J), f2_dbl(V));
    r.y = f2_sub(f2_mul(rr, f2_sub(V, r.x)), f2_dbl(f2_mul(S1, J)));
    r.z = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}
H2_HD G2Jac g2_add_affine(const G2Jac &p, const G2Affine &q) {
    if (q.is_identity()) return p;
    G2Jac qq = {q.x, q.y, f2_one()};
    return g2_add(p, qq);
}

constexpr uint32_t G2_C = 8, G2_W = 32, G2_B = 256;   // 8-bit unsigned windows over the 256-bit canonical scalar
// workgroup (chunk, window): lane b accumulates the chunk's points whose window digit equals b
__global__ __launch_bounds__(256) void g2_bucket_kernel(const G2Affine *__restrict__ pts, const Fr *__restrict__ scalars, uint32_t n, uint32_t chunk,
                                                        G2Jac *__restrict__ partial) {
    HIP_DYNAMIC_SHARED(uint8_t, digit)   // chunk bytes
    const uint32_t w = blockIdx.y, lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n, b = threadIdx.x;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        Fr s = fe_from_mont(scalars[i]);
        digit[i - lo] = (uint8_t)((s.l[w >> 2] >> (8 * (w & 3))) & 0xffu);
    }
    __syncthreads();
    G2Jac acc = G2Jac::identity();
    if (b)
        for (uint32_t i = lo; i < hi; ++i)
            if (digit[i - lo] == b) acc = g2_add_affine(acc, pts[i]);
    partial[((size_t)w * gridDim.x + blockIdx.x) * G2_B + b] = acc;
}
// lane (w, b): sum over the chunks
__global__ __launch_bounds__(256) void g2_merge_kernel(const G2Jac *__restrict__ partial, uint32_t chunks, G2Jac *__restrict__ buckets) {
    const uint32_t w = blockIdx.x, b = threadIdx.x;
    G2Jac acc = G2Jac::identity();
    for (uint32_t c = 0; c < chunks; ++c) acc = g2_add(acc, partial[((size_t)w * chunks + c) * G2_B + b]);
    buckets[(size_t)w * G2_B + b] = acc;
}
// workgroup w: out[w][t] = sum of the buckets whose index has bit t set (t < 8), so that sum_b b*B_b = sum_t 2^t out[w][t]
__global__ __launch_bounds__(256) void g2_bit_sums_kernel(const G2Jac *__restrict__ buckets, G2Jac *__restrict__ out) {
    __shared__ G2Jac sh[256];
    const uint32_t w = blockIdx.x, b = threadIdx.x;
    const G2Jac mine = buckets[(size_t)w * G2_B + b];
    for (uint32_t t = 0; t < G2_C; ++t) {
        sh[b] = ((b >> t) & 1u) ? mine : G2Jac::identity();
        __syncthreads();
        for (uint32_t d = 128; d >= 1; d >>= 1) {
            if (b < d) sh[b] = g2_add(sh[b], sh[b + d]);
            __syncthreads();
        }
        if (b == 0) out[(size_t)w * G2_C + t] = sh[0];
        __syncthreads();
    }
}

static int msm_g2_run(h2hip_ctx *ctx, const G2Affine *pts_dev, const Fr *scalars_dev, size_t n, void *out_affine_host) {
    G2Affine result;
    memset(&result, 0, sizeof(result));
    if (n) {
        const uint32_t chunk = n <= 4096 ? 256u : 2048u;
        const uint32_t chunks = (uint32_t)((n + chunk - 1) / chunk);
        G2Jac *partial = nullptr, *buckets = nullptr, *bits = nullptr;
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_BUCKETS, sizeof(G2Jac) * (size_t)G2_W * chunks * G2_B, (void **)&partial));
        if (ctx->clean_stream) H2_HIPCHK(hipStreamSynchronize(ctx->clean_stream));   // a G1 MSM's zero-fill of this slot may still be pending
        ctx->clean_bytes[0] = 0;                                                     // ... and the slot no longer holds zeros
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_SEG, sizeof(G2Jac) * G2_W * G2_B, (void **)&buckets));
        H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_WIN, sizeof(G2Jac) * G2_W * G2_C, (void **)&bits));
        prof_begin(ctx, "g2_msm_kernels");
        hipLaunchKernelGGL(g2_bucket_kernel, dim3(chunks, G2_W), dim3(256), chunk, ctx->stream, pts_dev, scalars_dev, (uint32_t)n, chunk, partial);
        hipLaunchKernelGGL(g2_merge_kernel, dim3(G2_W), dim3(256), 0, ctx->stream, (const G2Jac *)partial, chunks, buckets);
        hipLaunchKernelGGL(g2_bit_sums_kernel, dim3(G2_W), dim3(256), 0, ctx->stream, (const G2Jac *)buckets, bits);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
        std::vector<G2Jac> host_bits(G2_W * G2_C);
        H2_HIPCHK(hipMemcpyAsync(host_bits.data(), bits, sizeof(G2Jac) * host_bits.size(), hipMemcpyDeviceToHost, ctx->stream));
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));
        // fold on the host: window sums by Horner over the bits, then over the windows (256 doublings in all)
        G2Jac acc = G2Jac::identity();
        for (int w = (int)G2_W - 1; w >= 0; --w) {
            G2Jac sw = G2Jac::identity();
            for (int t = (int)G2_C - 1; t >= 0; --t) sw = g2_add(g2_double(sw), host_bits[(size_t)w * G2_C + t]);
            for (uint32_t d = 0; d < G2_C; ++d) acc = g2_double(acc);
            acc = g2_add(acc, sw);
        }
        if (!acc.is_identity()) {
            const F2 zi = f2_inv(acc.z), zi2 = f2_sqr(zi);
            result.x = f2_mul(acc.x, zi2);
            result.y = f2_mul(acc.y, f2_mul(zi2, zi));
        }
    }
    memcpy(out_affine_host, &result, sizeof(result));
    return H2HIP_OK;
}

}  // namespace h2

using namespace h2;

extern "C" {

// out = sum_i scalars[i] * points[i] over G2; points: n x 128 B affine (x.c0, x.c1, y.c0, y.c1 Montgomery limbs, identity all-zero),
// scalars: n x 32 B Montgomery Fr; out_affine_host: 128 B (all-zero = identity)
int h2hip_msm_g2_dev(h2hip_ctx *ctx, const void *g2_affine_dev, const void *scalars_dev, size_t n, void *out_affine_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && out_affine_host && (n == 0 || (g2_affine_dev && scalars_dev)) && n < (1u << 26), "bad argument");
    return msm_g2_run(ctx, (const G2Affine *)g2_affine_dev, (const Fr *)scalars_dev, n, out_affine_host);
}
int h2hip_msm_g2(h2hip_ctx *ctx, const void *g2_affine_host, const void *scalars_host, size_t n, void *out_affine_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && out_affine_host && (n == 0 || (g2_affine_host && scalars_host)) && n < (1u << 26), "bad argument");
    char *stage = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_STAGE, (sizeof(G2Affine) + sizeof(Fr)) * (n ? n : 1), (void **)&stage));
    if (n) {
        H2_HIPCHK(hipMemcpyAsync(stage, g2_affine_host, sizeof(G2Affine) * n, hipMemcpyHostToDevice, ctx->stream));
        H2_HIPCHK(hipMemcpyAsync(stage + sizeof(G2Affine) * n, scalars_host, sizeof(Fr) * n, hipMemcpyHostToDevice, ctx->stream));
    }
    return msm_g2_run(ctx, (const G2Affine *)stage, (const Fr *)(stage + sizeof(G2Affine) * n), n, out_affine_host);
}

}  // extern "C"
