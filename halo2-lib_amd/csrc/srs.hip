// KZG SRS generation on the GPU — the device side of ParamsKZG::<Bn256>::setup(k, rng) [UPSTREAM
// halo2-axiom 0.5.3 poly::kzg::commitment; reached from the reference at halo2-base/src/utils/mod.rs:439-443
// (gen_srs) and halo2-base/benches/mul.rs:39]:
//     g[i]          = s^i * G1                                   (monomial basis)
//     g_lagrange[i] = L_i(s) * G1,  L_i(s) = (s^n - 1)/n * w^i / (s - w^i)   (Lagrange basis, no group FFT needed
//                                                                             because s is known at setup)
// Both are batches of fixed-base scalar multiplications: an 8-bit window table of the base (32 x 255 affine
// points) is built once, then every lane does <= 32 mixed XYZZ additions and the batch is normalised.
// The G2 half of the SRS (g2, s*g2) is verifier-side and not on the prover path.
#include "internal.h"

namespace h2 {

constexpr uint32_t FB_WIN = 8, FB_WINDOWS = 32, FB_PER = 255;

__device__ __forceinline__ XYZZ xyzz_small_mul_affine(const G1Affine &p, uint32_t k) {
    XYZZ r = XYZZ::identity();
    if (p.is_identity()) return r;
    for (int bit = 31 - __clz(k | 1u); bit >= 0; --bit) {
        r = xyzz_double(r);
        if ((k >> bit) & 1u) xyzz_add_affine(r, p.x, p.y);
    }
    return r;
}
// lane w: Q_w = 2^(8w) * P
__global__ __launch_bounds__(64) void fb_window_bases_kernel(G1Affine base, G1Jac *__restrict__ out) {
    uint32_t w = threadIdx.x;
    if (w >= FB_WINDOWS) return;
    XYZZ a = XYZZ::from_affine(base);
    for (uint32_t i = 0; i < FB_WIN * w; ++i) a = xyzz_double(a);
    out[w] = xyzz_to_jacobian(a);
}
// lane (w, d): T[w][d] = (d+1) * Q_w
__global__ __launch_bounds__(256) void fb_table_kernel(const G1Affine *__restrict__ qw, G1Jac *__restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_WINDOWS * FB_PER) return;
    uint32_t w = t / FB_PER, d = t - w * FB_PER;
    out[t] = xyzz_to_jacobian(xyzz_small_mul_affine(qw[w], d + 1));
}
__global__ __launch_bounds__(256) void fb_mul_kernel(const G1Affine *__restrict__ table, const Fr *__restrict__ scalars, uint32_t n,
                                                     G1Jac *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = fe_from_mont(scalars[i]);
    XYZZ acc = XYZZ::identity();
#pragma unroll
    for (int limb = 0; limb < 8; ++limb) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t d = (s.l[limb] >> (8 * b)) & 0xffu;
            if (d) {
                G1Affine p = table[(limb * 4 + b) * FB_PER + d - 1];
                if (!p.is_identity()) xyzz_add_affine(acc, p.x, p.y);
            }
        }
    }
    out[i] = xyzz_to_jacobian(acc);
}

// scalars for ParamsKZG::setup: mono[i] = s^i ; den[i] = s - w^i ; num[i] = mult * w^i, mult = (s^n-1)/n
__global__ __launch_bounds__(256) void kzg_setup_scalars_kernel(Fr s, Fr omega, Fr mult, uint32_t n, Fr *__restrict__ mono, Fr *__restrict__ num,
                                                                Fr *__restrict__ den) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mono[i] = fe_pow_u64(s, i);
    Fr wi = fe_pow_u64(omega, i);
    num[i] = fe_mul(mult, wi);
    den[i] = fe_sub(s, wi);
}

static int fixed_base_table(h2hip_ctx *ctx, const G1Affine &base, G1Affine **table_out) {
    const uint32_t total = FB_WINDOWS * FB_PER;
    char *buf = nullptr;
    // layout: [Jac tmp: total] [affine Q_w: 32] [affine table: total]
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_FBTABLE, sizeof(G1Jac) * total + sizeof(G1Affine) * (FB_WINDOWS + total), (void **)&buf));
    G1Jac *jtmp = (G1Jac *)buf;
    G1Affine *qw = (G1Affine *)(buf + sizeof(G1Jac) * total);
    G1Affine *table = qw + FB_WINDOWS;
    prof_begin(ctx, "fb_table_kernels");
    hipLaunchKernelGGL(fb_window_bases_kernel, dim3(1), dim3(64), 0, ctx->stream, base, jtmp);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(batch_normalize_jac(ctx, jtmp, qw, FB_WINDOWS));
    prof_begin(ctx, "fb_table_kernels");
    hipLaunchKernelGGL(fb_table_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, (const G1Affine *)qw, jtmp);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(batch_normalize_jac(ctx, jtmp, table, total));
    *table_out = table;
    return H2HIP_OK;
}

static int fixed_base_mul(h2hip_ctx *ctx, const G1Affine *table, const Fr *scalars, uint32_t n, G1Affine *out) {
    G1Jac *jtmp = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP1, sizeof(G1Jac) * (size_t)(n ? n : 1), (void **)&jtmp));
    if (!n) return H2HIP_OK;
    prof_begin(ctx, "fb_mul_kernel");
    hipLaunchKernelGGL(fb_mul_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, table, scalars, n, jtmp);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return batch_normalize_jac(ctx, jtmp, out, n);
}

// ------------------------------------------------------------------ g_to_lagrange: the group-element inverse FFT
// ParamsKZG without Lagrange bases (a ceremony file, ParamsKZG::from_parts(.., None, ..), downsize) derives them as
//   g_lagrange[i] = n^-1 * sum_j omega^(-i j) * g[j]
// [UPSTREAM poly/kzg/commitment.rs g_to_lagrange: best_fft over G1 with omega^-1, then * n^-1, then batch_normalize].
// Radix-2 decimation in time on XYZZ points: bit-reversed load, log n stages of butterflies whose twiddle product is a
// 254-bit double-and-add scalar multiplication (one lane per butterfly; throughput-bound, ~0.2 s at k = 19 — a one-off
// per SRS next to upstream's minutes on the CPU).
__device__ __forceinline__ XYZZ29 xyzz29_scalar_mul(const XYZZ29 &p, const Fr &canonical) {
    XYZZ29 r = XYZZ29::identity();
    for (int bit = 253; bit >= 0; --bit) {
        r = xyzz29_double(r);
        if ((canonical.l[bit >> 5] >> (bit & 31)) & 1u) xyzz29_add(r, p);
    }
    return r;
}
__global__ __launch_bounds__(256) void gl_load_kernel(const G1Affine *__restrict__ g, XYZZ29 *__restrict__ a, uint32_t k) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << k)) return;
    G1Affine p = g[i];
    XYZZ29 v = XYZZ29::identity();
    if (!p.is_identity()) {
        G1Affine29 q = g1affine29_from_sat(p);
        v.x = q.x;
        v.y = q.y;
        v.zz = Fq29::one();
        v.zzz = Fq29::one();
    }
    a[k ? (__brev(i) >> (32 - k)) : 0] = v;
}
// tw[j] = canonical limbs of omega_inv^j, j < n/2
__global__ __launch_bounds__(256) void gl_twiddle_kernel(Fr *__restrict__ tw, Fr omega_inv, uint32_t count) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) tw[j] = fe_from_mont(fe_pow_u64(omega_inv, j));
}
__global__ __launch_bounds__(256) void gl_stage_kernel(XYZZ29 *__restrict__ a, const Fr *__restrict__ tw, uint32_t k, uint32_t s) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (1u << (k - 1))) return;
    const uint32_t half = 1u << s, j = b & (half - 1);
    const uint32_t i = ((b >> s) << (s + 1)) + j, e = j << (k - 1 - s);
    XYZZ29 t = a[i + half];
    if (e) t = xyzz29_scalar_mul(t, tw[e]);
    XYZZ29 u = a[i], v = u;
    xyzz29_add(u, t);
    if (!t.is_identity()) t.y = f29_neg<4>(t.y);
    xyzz29_add(v, t);
    a[i] = u;
    a[i + half] = v;
}
__global__ __launch_bounds__(256) void gl_scale_kernel(const XYZZ29 *__restrict__ a, Fr n_inv_canonical, uint32_t n, G1Jac *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = xyzz_to_jacobian(xyzz29_to_sat(xyzz29_scalar_mul(a[i], n_inv_canonical)));
}

}  // namespace h2

using namespace h2;

// ---- SRS files (ParamsKZG::read, reference halo2-base/src/utils/mod.rs:401-435): point validation and decompression ------------------
struct Exp256 {
    uint32_t e[8];
};
__device__ __forceinline__ bool fq_is_canonical(const Fq &a) {   // limbs < q
    unsigned br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) subb32(a.l[j], FqP::m(j), br);
    return br != 0;
}
__device__ __forceinline__ bool g1_on_curve_mont(const Fq &x, const Fq &y) {
    Fq b3 = fe_add(fe_add(Fq::one(), Fq::one()), Fq::one());
    return fe_sqr(y) == fe_add(fe_mul(fe_sqr(x), x), b3);
}
// SerdeFormat::RawBytes points: both coordinates must be canonical Montgomery limbs (< q) and satisfy y^2 = x^3 + 3, or be (0, 0)
__global__ __launch_bounds__(256) void g1_validate_kernel(const G1Affine *__restrict__ p, size_t n, uint32_t *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = p[i];
    bool ok = fq_is_canonical(a.x) && fq_is_canonical(a.y) && (a.is_identity() || g1_on_curve_mont(a.x, a.y));
    if (!ok) atomicAdd(bad, 1u);
}
// SerdeFormat::Processed points: 32 bytes, x little-endian canonical, sign(y) = y mod 2 and the identity flag in the top byte
__global__ __launch_bounds__(256) void g1_decompress_kernel(const uint32_t *__restrict__ in, size_t n, G1Affine *__restrict__ out, uint32_t sign_bit,
                                                            uint32_t inf_bit, Exp256 sqrt_exp, uint32_t *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fq x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x.l[j] = in[i * 8 + j];
    const uint32_t top = x.l[7] >> 24;
    const bool inf = (top >> inf_bit) & 1u, sign = (top >> sign_bit) & 1u;
    x.l[7] &= ~(((1u << sign_bit) | (1u << inf_bit)) << 24);
    G1Affine r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    if (inf) {
        if (!x.is_zero() || sign) atomicAdd(bad, 1u);   // non-canonical identity encoding
        out[i] = r;
        return;
    }
    if (!fq_is_canonical(x)) {
        atomicAdd(bad, 1u);
        out[i] = r;
        return;
    }
    const Fq xm = fe_to_mont(x);
    const Fq b3 = fe_add(fe_add(Fq::one(), Fq::one()), Fq::one());
    const Fq y2 = fe_add(fe_mul(fe_sqr(xm), xm), b3);
    Fq y = fe_pow(y2, sqrt_exp.e);   // q = 3 mod 4: a square root of y2, if one exists, is y2^((q+1)/4)
    if (!(fe_sqr(y) == y2)) {
        atomicAdd(bad, 1u);
        out[i] = r;
        return;
    }
    if ((fe_from_mont(y).l[0] & 1u) != (sign ? 1u : 0u)) y = fe_neg(y);
    r.x = xm;
    r.y = y;
    out[i] = r;
}

extern "C" {

// out[i] = scalars[i] * base   (device arrays; base is a host G1Affine)
int h2hip_g1_fixed_base_mul_batch_dev(h2hip_ctx *ctx, const void *base_affine, const void *scalars_dev, size_t n, void *out_affine_dev) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && base_affine && (n == 0 || (scalars_dev && out_affine_dev)), "NULL argument");
    H2_REQUIRE(n < (1u << 31), "n too large");
    G1Affine base;
    memcpy(&base, base_affine, sizeof(base));
    G1Affine *table = nullptr;
    H2_CHK(fixed_base_table(ctx, base, &table));
    return fixed_base_mul(ctx, table, (const Fr *)scalars_dev, (uint32_t)n, (G1Affine *)out_affine_dev);
}

int h2hip_g1_to_lagrange(h2hip_ctx *ctx, const h2hip_bases *g, uint32_t k, uint32_t flags, h2hip_bases **g_lagrange_out) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && g && g_lagrange_out, "NULL argument");
    H2_REQUIRE(k <= 26 && g->n >= ((size_t)1 << k), "need at least 2^k monomial bases, k <= 26");
    const uint32_t n = 1u << k;
    Fr rou_canon = {{0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du, 0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u}};
    Fr omega = fe_to_mont(rou_canon);
    for (uint32_t i = k; i < 28; ++i) omega = fe_sqr(omega);
    const Fr omega_inv = fe_inv(omega);
    Fr n_fr = Fr::zero();
    n_fr.l[0] = n;
    const Fr n_inv = fe_from_mont(fe_inv(fe_to_mont(n_fr)));
    XYZZ29 *a = nullptr;
    Fr *tw = nullptr;
    G1Jac *jac = nullptr;
    G1Affine *aff = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(XYZZ29) * (size_t)n, (void **)&a));
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP1, sizeof(G1Jac) * (size_t)n + sizeof(Fr) * (size_t)(n / 2 + 1), (void **)&jac));
    tw = (Fr *)(jac + n);
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_FBTABLE, sizeof(G1Affine) * (size_t)n, (void **)&aff));
    const dim3 blk(256), grid_n((n + 255) / 256), grid_h((n / 2 + 255) / 256);
    prof_begin(ctx, "g_to_lagrange_kernels");
    hipLaunchKernelGGL(gl_load_kernel, grid_n, blk, 0, ctx->stream, (const G1Affine *)g->pts, a, k);
    if (n > 1) hipLaunchKernelGGL(gl_twiddle_kernel, grid_h, blk, 0, ctx->stream, tw, omega_inv, n / 2);
    for (uint32_t s = 0; s < k; ++s) hipLaunchKernelGGL(gl_stage_kernel, grid_h, blk, 0, ctx->stream, a, (const Fr *)tw, k, s);
    hipLaunchKernelGGL(gl_scale_kernel, grid_n, blk, 0, ctx->stream, (const XYZZ29 *)a, n_inv, n, jac);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(batch_normalize_jac(ctx, jac, aff, n));
    return h2hip_bases_from_device(ctx, aff, n, flags, g_lagrange_out);
}

int h2hip_params_kzg_setup(h2hip_ctx *ctx, uint32_t k, const void *s_fr, uint32_t flags, h2hip_bases **g_out, h2hip_bases **g_lagrange_out) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && s_fr && g_out && g_lagrange_out, "NULL argument");
    H2_REQUIRE(k <= 26, "k too large");
    const uint32_t n = 1u << k;
    Fr s;
    memcpy(&s, s_fr, sizeof(Fr));
    // domain constants on the host (a handful of field operations)
    // ROOT_OF_UNITY = 7^((r-1)/2^28), canonical 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    Fr rou_canon = {{0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du, 0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u}};
    Fr omega = fe_to_mont(rou_canon);
    for (uint32_t i = k; i < 28; ++i) omega = fe_sqr(omega);
    Fr sn = fe_pow_u64(s, n);
    Fr n_fr = Fr::zero();
    n_fr.l[0] = n;
    Fr mult = fe_mul(fe_sub(sn, Fr::one()), fe_inv(fe_to_mont(n_fr)));
    // s must not be an n-th root of unity (then some L_i has a pole); the real setup samples s at random
    H2_REQUIRE(!fe_sub(sn, Fr::one()).is_zero(), "s is an n-th root of unity");

    Fr *mono = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(Fr) * 3 * (size_t)n, (void **)&mono));
    Fr *num = mono + n, *den = num + n;
    prof_begin(ctx, "kzg_setup_scalars_kernel");
    hipLaunchKernelGGL(kzg_setup_scalars_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, s, omega, mult, n, mono, num, den);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_CHK(h2hip_fr_batch_invert_dev(ctx, den, n));
    H2_CHK(h2hip_fr_mul_batch_dev(ctx, num, num, den, n));   // num[i] = L_i(s)

    G1Affine gen;   // G1 generator (1, 2)
    Fq one_c = Fq::zero(), two_c = Fq::zero();
    one_c.l[0] = 1;
    two_c.l[0] = 2;
    gen.x = fe_to_mont(one_c);
    gen.y = fe_to_mont(two_c);
    G1Affine *table = nullptr;
    H2_CHK(fixed_base_table(ctx, gen, &table));
    G1Affine *pts = nullptr;
    H2_HIPCHK(hipMalloc((void **)&pts, sizeof(G1Affine) * 2 * (size_t)n));
    int rc = fixed_base_mul(ctx, table, mono, n, pts);
    if (rc == H2HIP_OK) rc = fixed_base_mul(ctx, table, num, n, pts + n);
    if (rc == H2HIP_OK) rc = h2hip_bases_from_device(ctx, pts, n, flags, g_out);
    if (rc == H2HIP_OK) {
        rc = h2hip_bases_from_device(ctx, pts + n, n, flags, g_lagrange_out);
        if (rc != H2HIP_OK) {
            h2hip_bases_free(ctx, *g_out);
            *g_out = nullptr;
        }
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(pts);
    return rc;
}

// copies the (level-0) affine points of a resident base set back to the host
int h2hip_bases_download(h2hip_ctx *ctx, const h2hip_bases *bases, void *out_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && bases && out_host, "NULL argument");
    if (!bases->n) return H2HIP_OK;
    H2_HIPCHK(hipMemcpyAsync(out_host, bases->pts, sizeof(G1Affine) * bases->n, hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    return H2HIP_OK;
}

// ParamsKZG::read for SerdeFormat::RawBytes: *invalid = number of points that are not canonical on-curve points (or the identity)
int h2hip_g1_validate_dev(h2hip_ctx *ctx, const void *points_dev, size_t n, size_t *invalid) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && invalid && (n == 0 || points_dev), "NULL argument");
    *invalid = 0;
    if (!n) return H2HIP_OK;
    uint32_t *bad = nullptr, host_bad = 0;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OUT, 1024, (void **)&bad));
    H2_HIPCHK(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
    prof_begin(ctx, "g1_validate_kernel");
    hipLaunchKernelGGL(g1_validate_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const G1Affine *)points_dev, n, bad);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_HIPCHK(hipMemcpyAsync(&host_bad, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    *invalid = host_bad;
    return H2HIP_OK;
}
// ParamsKZG::read for SerdeFormat::Processed: n compressed points (32 B each) -> affine Montgomery points; H2HIP_ERR_INVALID if any
// encoding is non-canonical or not on the curve.  sign_bit / inf_bit: flag positions in the top byte (halo2curves: 6 / 7).
int h2hip_g1_decompress_batch_dev(h2hip_ctx *ctx, const void *compressed_dev, size_t n, void *out_affine_dev, uint32_t sign_bit, uint32_t inf_bit) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (n == 0 || (compressed_dev && out_affine_dev)), "NULL argument");
    H2_REQUIRE(sign_bit < 8 && inf_bit < 8 && sign_bit != inf_bit && sign_bit >= 6 && inf_bit >= 6, "flag bits must be the two spare bits (6, 7) of the top byte");
    if (!n) return H2HIP_OK;
    Exp256 e;   // (q + 1) / 4
    {
        uint64_t carry = 1;
        uint32_t t[8];
        for (int j = 0; j < 8; ++j) {
            uint64_t v = (uint64_t)FqP::m(j) + carry;
            t[j] = (uint32_t)v;
            carry = v >> 32;
        }
        for (int j = 0; j < 8; ++j) e.e[j] = (t[j] >> 2) | (j < 7 ? t[j + 1] << 30 : 0u);
    }
    uint32_t *bad = nullptr, host_bad = 0;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_OUT, 1024, (void **)&bad));
    H2_HIPCHK(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
    prof_begin(ctx, "g1_decompress_kernel");
    hipLaunchKernelGGL(g1_decompress_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *)compressed_dev, n,
                       (G1Affine *)out_affine_dev, sign_bit, inf_bit, e, bad);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    H2_HIPCHK(hipMemcpyAsync(&host_bad, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    if (host_bad) {
        set_error("h2hip_g1_decompress_batch_dev: %u of %zu point encodings are invalid (non-canonical x, not on the curve, or a malformed identity)", host_bad, n);
        return H2HIP_ERR_INVALID;
    }
    return H2HIP_OK;
}

}  // extern "C"
