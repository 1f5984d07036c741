// Radix-2^m Stockham NTT over BN254 F_r for gfx950 — the device replacement for
// halo2_proofs::arithmetic::best_fft and EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff}
// [UPSTREAM halo2-axiom 0.5.3; the reference reaches them only through create_proof,
// /root/reference/halo2-base/src/utils/testing.rs:40-47; semantics restated in SURVEY.md A.2].
//
// Natural order in, natural order out, no bit-reversal pass: log_n is split into P passes of m_i bits.
// Pass i (stride s = 2^(m_1+..+m_{i-1})) lets one 256-thread workgroup own a tile of C = 2^cb adjacent
// "columns" j: it reads rows x[j + t*N/R] (each row a contiguous C*32 B segment), runs the R-point
// decimation-in-time butterflies in LDS on unsaturated 9x29-bit limbs (fq29.cuh), multiplies by the inter-pass
// twiddle omega^((j-q)u) and writes y[(j-q)R + q + u*s], q = j mod s (again contiguous in q).  All passes but the last are
// out-of-place (ping-pong with a context-owned scratch buffer); the last pass touches the same index set
// it reads, so it runs in place.  Input scaling by zeta^(i mod 3) with implicit zero padding
// (coeff_to_extended) is fused into the first pass, output scaling (ifft divisor, zeta^-(i mod 3)) into
// the last.  Twiddles come from a two-level table omega^e = T2[e >> lo] * T1[e & mask] built once per
// (log_n, omega) and cached in the context.
#include "internal.h"
#include "fq29.cuh"

namespace h2 {

struct NttScale {
    Fr in3[3];
    Fr out3[3];
};

// saturated Montgomery (x*2^256) -> unsaturated R' = 2^261 form, normalised, < 1.01 r
__device__ __forceinline__ Fr29 fr29_from_sat(const Fr &s) {
    Fr29 k;
#pragma unroll
    for (int i = 0; i < 9; ++i) k.l[i] = R29P::conv_in(i);
    return f29_mul(f29_split<R29P>(s), k);
}

__global__ void ntt_twiddle_kernel(Fr29 *t1, Fr29 *t2, Fr omega, uint32_t lo_bits, uint32_t hi_count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo_count = 1u << lo_bits;
    if (i < lo_count) t1[i] = fr29_from_sat(fe_pow_u64(omega, (uint64_t)i));
    if (i < hi_count) t2[i] = fr29_from_sat(fe_pow_u64(omega, (uint64_t)i << lo_bits));
}

__device__ __forceinline__ Fr29 tw_lookup(const Fr29 *__restrict__ t1, const Fr29 *__restrict__ t2, uint32_t lo_bits, uint64_t e) {
    uint32_t lo = (uint32_t)(e & ((1ull << lo_bits) - 1));
    uint32_t hi = (uint32_t)(e >> lo_bits);
    return f29_mul(t2[hi], t1[lo]);
}

__device__ __forceinline__ uint32_t bitrev_m(uint32_t x, uint32_t m) { return m ? (__brev(x) >> (32 - m)) : 0; }

// One pass; grid = number of tiles, block = 256.  Data moves through HBM as saturated canonical Montgomery limbs
// (the caller's format); inside the workgroup it lives in LDS as unsaturated 9 x 29-bit limbs (36 B, a 9-word stride
// is bank-conflict free), where the integer value*2^256 is kept, only weakly reduced.  Stage twiddles and inter-pass
// twiddles are stored in the R' = 2^261 Montgomery form, so mont29(x, w) keeps the integer's 2^256 scaling and no
// conversion multiply is ever needed.  Butterflies are decimation-in-time (t = w*b; a + t, a - t + 2r): the bound of
// a lane's value grows by at most 2r per stage (<= 21 r after 10 stages) instead of doubling.
__global__ __launch_bounds__(256) void ntt_pass_kernel(const Fr *__restrict__ x, Fr *__restrict__ y, uint32_t log_n, uint32_t m,
                                                       uint32_t log_s, uint32_t cb, const Fr29 *__restrict__ t1,
                                                       const Fr29 *__restrict__ t2, uint32_t lo_bits, uint64_t in_len, int in_mul,
                                                       int out_mul, NttScale sc) {
    HIP_DYNAMIC_SHARED(Fr29, lds)
    const uint32_t tid = threadIdx.x;
    const uint32_t R = 1u << m, C = 1u << cb;
    const uint32_t elems = R << cb;
    Fr29 *tw_s = lds + elems;                           // omega_R^k, k < R/2
    const uint64_t rows_stride = 1ull << (log_n - m);   // N/R
    const uint64_t j0 = (uint64_t)blockIdx.x << cb;

    for (uint32_t k = tid; k < (R >> 1); k += 256) tw_s[k] = tw_lookup(t1, t2, lo_bits, (uint64_t)k << (log_n - m));

    Fr29 in3[3];
    if (in_mul) {
#pragma unroll
        for (int k = 0; k < 3; ++k) in3[k] = fr29_from_sat(sc.in3[k]);
    }
    for (uint32_t e = tid; e < elems; e += 256) {
        uint32_t t = e >> cb, c = e & (C - 1);
        uint64_t idx = j0 + c + (uint64_t)t * rows_stride;
        Fr29 v = Fr29::zero();
        if (idx < in_len) {
            v = f29_split<R29P>(x[idx]);
            if (in_mul) {
                uint32_t r3 = (uint32_t)(idx % 3);
                v = f29_mul(v, r3 == 0 ? in3[0] : r3 == 1 ? in3[1] : in3[2]);
            }
        }
        lds[(bitrev_m(t, m) << cb) + c] = v;   // DIT: bit-reversed rows in, natural rows out
    }

    for (uint32_t st = 0; st < m; ++st) {
        const uint32_t h = 1u << st;
        __syncthreads();
        for (uint32_t b = tid; b < (elems >> 1); b += 256) {
            uint32_t c = b & (C - 1), p = b >> cb;
            uint32_t i = p & (h - 1), blk = p >> st;
            uint32_t e0 = (((blk << (st + 1)) + i) << cb) + c, e1 = e0 + (h << cb);
            Fr29 a = lds[e0], bb = lds[e1];
            Fr29 t = st ? f29_mul(bb, tw_s[i << (m - 1 - st)]) : bb;   // omega_{2h}^i = omega_R^(i * R/(2h)); stage 0: w = 1
            if (!st) t = f29_norm(t);
            lds[e0] = f29_norm(f29_add(a, t));
            lds[e1] = f29_sub<2>(a, t);
        }
    }
    __syncthreads();

    const bool has_tw = (log_s + m) < log_n;   // the last pass has j - q == 0 everywhere
    const uint64_t smask = (1ull << log_s) - 1;
    Fr29 out3[3];
    if (out_mul) {
#pragma unroll
        for (int k = 0; k < 3; ++k) out3[k] = fr29_from_sat(sc.out3[k]);
    }
    for (uint32_t e = tid; e < elems; e += 256) {
        uint32_t u, c;
        if (log_s == 0) {   // first pass: output (j0+c)*R + u is contiguous in u
            c = e >> m;
            u = e & (R - 1);
        } else {            // later passes: contiguous in q (i.e. in c)
            u = e >> cb;
            c = e & (C - 1);
        }
        uint64_t j = j0 + c, q = j & smask, jq = j - q;
        Fr29 v = lds[(u << cb) + c];
        uint64_t oidx = (jq << m) + q + ((uint64_t)u << log_s);
        if (has_tw) v = f29_mul(v, tw_lookup(t1, t2, lo_bits, jq * u));
        if (out_mul) {
            uint32_t r3 = (uint32_t)(oidx % 3);
            v = f29_mul(v, r3 == 0 ? out3[0] : r3 == 1 ? out3[1] : out3[2]);
        }
        if (!has_tw && !out_mul) v = f29_mul(v, Fr29::one());   // weak bound (<= 21 r) -> < 1.2 r before packing
        y[oidx] = f29_pack_canonical<FrP>(v);
    }
}

static int get_twiddles(h2hip_ctx *ctx, uint32_t log_n, const Fr &omega, TwiddleSet **out) {
    for (auto &t : ctx->twiddles)
        if (t.log_n == log_n && t.omega == omega) {
            *out = &t;
            return H2HIP_OK;
        }
    TwiddleSet t;
    t.log_n = log_n;
    t.omega = omega;
    t.lo_bits = (log_n + 1) / 2;
    uint32_t lo_count = 1u << t.lo_bits, hi_count = 1u << (log_n - t.lo_bits);
    H2_HIPCHK(hipMalloc((void **)&t.t1, sizeof(Fr29) * lo_count));
    H2_HIPCHK(hipMalloc((void **)&t.t2, sizeof(Fr29) * hi_count));
    uint32_t cnt = lo_count > hi_count ? lo_count : hi_count;
    prof_begin(ctx, "ntt_twiddle_kernel");
    hipLaunchKernelGGL(ntt_twiddle_kernel, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, (Fr29 *)t.t1, (Fr29 *)t.t2, omega, t.lo_bits, hi_count);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    if (ctx->twiddles.size() >= 16) {   // bounded cache: drop the oldest table
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));
        hipFree(ctx->twiddles.front().t1);
        hipFree(ctx->twiddles.front().t2);
        ctx->twiddles.erase(ctx->twiddles.begin());
    }
    ctx->twiddles.push_back(t);
    *out = &ctx->twiddles.back();
    return H2HIP_OK;
}

// a: N = 2^log_n device elements (result lands here).  in_override (optional): read the input from there
// (in_len valid elements, implicit zeros beyond).  in_scale3 / out_scale3 (optional, host pointers to 3 Fr):
// multiply input / output element i by scale[i mod 3].
int ntt_run(h2hip_ctx *ctx, Fr *a, uint32_t log_n, const Fr &omega, const Fr *in_override, uint64_t in_len, const Fr *in_scale3,
            const Fr *out_scale3) {
    H2_REQUIRE(log_n <= 28, "log_n exceeds the 2-adicity of F_r (28)");
    const uint64_t N = 1ull << log_n;
    if (!in_override) in_len = N;
    H2_REQUIRE(in_len <= N, "input longer than the transform");
    NttScale sc;
    for (int i = 0; i < 3; ++i) {
        sc.in3[i] = in_scale3 ? in_scale3[i] : Fr::one();
        sc.out3[i] = out_scale3 ? out_scale3[i] : Fr::one();
    }
    const uint32_t LT = (uint32_t)ctx->ntt_tile_bits;
    uint32_t mlist[8], P;
    if (log_n <= LT) {
        P = 1;
        mlist[0] = log_n;
    } else {
        uint32_t maxm = LT - 3;
        P = (log_n + maxm - 1) / maxm;
        for (uint32_t i = 0; i < P; ++i) mlist[i] = log_n / P + (i < log_n % P ? 1 : 0);
    }
    TwiddleSet *tw = nullptr;
    H2_CHK(get_twiddles(ctx, log_n, omega, &tw));
    Fr *scratch = nullptr;
    if (P > 1) H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_NTT, sizeof(Fr) * N, (void **)&scratch));

    const Fr *cur = in_override ? in_override : a;
    uint32_t log_s = 0;
    for (uint32_t i = 0; i < P; ++i) {
        const uint32_t m = mlist[i];
        Fr *dst = (i == P - 1) ? a : (cur == scratch ? a : scratch);
        uint32_t cb = LT - m;
        if (cb > log_n - m) cb = log_n - m;
        if (i > 0 && cb > log_s) cb = log_s;
        const uint32_t tiles = 1u << (log_n - m - cb);
        const size_t shmem = sizeof(Fr29) * (((size_t)1 << (m + cb)) + ((size_t)1 << (m ? m - 1 : 0)));
        const bool first = (i == 0), last = (i == P - 1);
        prof_begin(ctx, "ntt_pass_kernel");
        hipLaunchKernelGGL(ntt_pass_kernel, dim3(tiles), dim3(256), shmem, ctx->stream, cur, dst, log_n, m, log_s, cb, (const Fr29 *)tw->t1, (const Fr29 *)tw->t2,
                           tw->lo_bits, first ? in_len : N, (first && in_scale3) ? 1 : 0, (last && out_scale3) ? 1 : 0, sc);
        prof_end(ctx);
        H2_HIPCHK(hipGetLastError());
        cur = dst;
        log_s += m;
    }
    return H2HIP_OK;
}

}  // namespace h2
