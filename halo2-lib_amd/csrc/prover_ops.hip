// Device steps of create_proof that sit between the big kernels: the factors of the permutation
// and lookup grand products, and the resolution of Assigned::Rational advice cells.  What they compute is fixed by halo2-base's
// constraint system — the equality-enabled columns of FlexGateConfig / RangeConfig (reference halo2-base/src/gates/flex_gate/mod.rs:
// 69,124-128, halo2-base/src/gates/range/mod.rs:104) and RangeConfig's lookups (range/mod.rs:131-150) — the argument formulas are
// upstream halo2's (permutation/prover.rs, lookup/prover.rs [UPSTREAM]).
#include "internal.h"
#include "fr29.cuh"

namespace h2 {

int ntt_pow_table(h2hip_ctx *ctx, uint32_t log_n, const Fr &omega, OmegaTable *out);   // ntt.hip

constexpr int PP_MAX_COLS = 8;
struct PermProductArgs {
    const Fr *cols[PP_MAX_COLS], *sigmas[PP_MAX_COLS];
    uint32_t ncols;
    Fr beta, gamma, delta;
    Fr x0;      // beta * delta^(first column index of the set): the identity-permutation term at row 0
    Fr omega;   // row generator
    Fr xstep;   // omega^(grid stride)
};
// num[i] = prod_j (v_j[i] + beta*delta^(c0+j)*omega^i + gamma),  den[i] = prod_j (v_j[i] + beta*sigma_j[i] + gamma),  i < rows
__global__ __launch_bounds__(256) void perm_product_terms_kernel(Fr *__restrict__ num, Fr *__restrict__ den, PermProductArgs g, size_t rows) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= rows) return;
    Fr xbase = fe_mul(g.x0, fe_pow_u64(g.omega, (uint64_t)i0));
    for (size_t i = i0; i < rows; i += stride, xbase = fe_mul(xbase, g.xstep)) {
        Fr nu = Fr::one(), de = Fr::one(), xterm = xbase;
        for (uint32_t j = 0; j < g.ncols; ++j) {
            Fr v = g.cols[j][i];
            Fr a = fe_add(fe_add(v, xterm), g.gamma);
            Fr b = fe_add(fe_add(v, fe_mul(g.beta, g.sigmas[j][i])), g.gamma);
            nu = j ? fe_mul(nu, a) : a;
            de = j ? fe_mul(de, b) : b;
            xterm = fe_mul(xterm, g.delta);
        }
        num[i] = nu;
        den[i] = de;
    }
}

// The same factors for SEVERAL consecutive sets per launch (a wide shape has ~80 sets of three 2^14-row columns): set s of the launch owns
// columns [s*chunk, (s+1)*chunk) of the launch's column table and rows [s*rows, (s+1)*rows) of num / den; omega^i is computed once per row.
constexpr uint32_t PP_BATCH_COLS = 64;
struct PermProductBatchArgs {
    const Fr *cols[PP_BATCH_COLS], *sigmas[PP_BATCH_COLS];
    uint32_t ncols, chunk;
    Fr beta, gamma, delta;
    Fr x0;      // beta * delta^(index of the launch's first column)
    Fr omega, xstep;
};
__global__ __launch_bounds__(256) void perm_product_terms_batch_kernel(Fr *__restrict__ num, Fr *__restrict__ den, PermProductBatchArgs g, size_t rows) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= rows) return;
    Fr xbase = fe_mul(g.x0, fe_pow_u64(g.omega, (uint64_t)i0));
    for (size_t i = i0; i < rows; i += stride, xbase = fe_mul(xbase, g.xstep)) {
        Fr xterm = xbase;
        for (uint32_t c0 = 0, set = 0; c0 < g.ncols; c0 += g.chunk, ++set) {
            const uint32_t c1 = c0 + g.chunk < g.ncols ? c0 + g.chunk : g.ncols;
            Fr nu = Fr::one(), de = Fr::one();
            for (uint32_t j = c0; j < c1; ++j) {
                Fr v = g.cols[j][i];
                Fr a = fe_add(fe_add(v, xterm), g.gamma);
                Fr b = fe_add(fe_add(v, fe_mul(g.beta, g.sigmas[j][i])), g.gamma);
                nu = j > c0 ? fe_mul(nu, a) : a;
                de = j > c0 ? fe_mul(de, b) : b;
                xterm = fe_mul(xterm, g.delta);
            }
            num[(size_t)set * rows + i] = nu;
            den[(size_t)set * rows + i] = de;
        }
    }
}

// the same factors on unsaturated limbs (fr29.cuh): every factor is formed 32-fold — 32 v from the split, 32 gamma and 32 beta from the constants, the
// X term's chain started at 32 beta delta^j0 omega^i — which is what a product of two stored-domain values needs; products start from 1
struct PermProductConsts29 {
    Fr29 beta32, delta, xstep;          // R' form of 32 beta, delta, omega^(grid stride)
    Fr29 x0_32;                         // raw split of 32 beta delta^(first column of the launch) (omega^row0 included): the chain's start = omega^i0 (table, R' form) x this
    Fr29 gamma32, one;                  // raw splits of 32 gamma and of 1
    OmegaTable pw;                        // omega^e, e < rows (r06: replaces a per-lane square-and-multiply of ~28 saturated products — more than the
                                        // four rows a lane then processes cost)
};
__global__ __launch_bounds__(256) void perm_product_terms_batch29_kernel(Fr *__restrict__ num, Fr *__restrict__ den, PermProductBatchArgs g, PermProductConsts29 k29,
                                                                         size_t rows) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= rows) return;
    Fr29 xbase = f29_mul(pow_lookup(k29.pw.t1, k29.pw.t2, k29.pw.lo_bits, (uint64_t)i0), k29.x0_32);
    for (size_t i = i0; i < rows; i += stride, xbase = f29_mul(xbase, k29.xstep)) {
        Fr29 xterm = xbase;
        for (uint32_t c0 = 0, set = 0; c0 < g.ncols; c0 += g.chunk, ++set) {
            const uint32_t c1 = c0 + g.chunk < g.ncols ? c0 + g.chunk : g.ncols;
            Fr29 nu = k29.one, de = k29.one;
            for (uint32_t j = c0; j < c1; ++j) {
                const Fr29 v32 = f29_add(r29_load32(g.cols[j][i]), k29.gamma32);                                    // lazy
                nu = f29_mul(nu, f29_norm(f29_add(v32, xterm)));                                                    // < 1.25 x 34.02
                de = f29_mul(de, f29_norm(f29_add(v32, f29_mul(r29_load(g.sigmas[j][i]), k29.beta32))));
                xterm = f29_mul(xterm, k29.delta);
            }
            num[(size_t)set * rows + i] = r29_store(nu);
            den[(size_t)set * rows + i] = r29_store(de);
        }
    }
}

// num[i] = (a[i] + beta)(s[i] + gamma),  den[i] = (a'[i] + beta)(s'[i] + gamma)
__global__ __launch_bounds__(256) void lookup_product_terms_kernel(Fr *__restrict__ num, Fr *__restrict__ den, const Fr *__restrict__ a,
                                                                   const Fr *__restrict__ s, const Fr *__restrict__ ap, const Fr *__restrict__ sp,
                                                                   Fr beta, Fr gamma, size_t rows) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
        num[i] = fe_mul(fe_add(a[i], beta), fe_add(s[i], gamma));
        den[i] = fe_mul(fe_add(ap[i], beta), fe_add(sp[i], gamma));
    }
}

struct SmallPoly {
    Fr c[8];
};
__global__ void fr_sub_low_kernel(Fr *__restrict__ y, SmallPoly p, uint32_t m) {
    uint32_t i = threadIdx.x;
    if (i < m) y[i] = fe_sub(y[i], p.c[i]);
}

static uint32_t grid_rows(h2hip_ctx *ctx, size_t n, size_t per_lane) {
    size_t blocks = (n / per_lane + 255) / 256, cap = (size_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    return (uint32_t)(blocks ? blocks : 1);
}
static Fr ld(const void *p) {
    Fr r;
    memcpy(&r, p, sizeof(Fr));
    return r;
}

}  // namespace h2

using namespace h2;

// ---------------------------------------------------------------------------------------------- cosets of the extended domain (multi-GPU)
// The extended domain zeta * <omega_e> of size 2^ek splits into 2^(ek-k) cosets of the ORIGINAL domain: rows i = (j << (ek-k)) + c hold the
// evaluations at s_c * omega^j with s_c = zeta * omega_e^c.  Every identity of h(X) is pointwise up to rotations by omega, which stay inside
// a coset — so a rank of the sharded prover evaluates whole cosets with the ordinary 2^k-point kernels (ek := k, zeta := s_c, omega_e := omega).
constexpr uint32_t COSET_BATCH = 32;
struct CosetScaleArgs {
    const Fr *in[COSET_BATCH];
    Fr *out[COSET_BATCH];
    Fr sblock;   // s^(elements per lane)
};
// out[col][t] = in[col][t] * s^t: the coefficients of f(s X).  Four consecutive coefficients per lane; s^(4 i0) by square-and-multiply.
__global__ __launch_bounds__(256) void coset_scale_kernel(CosetScaleArgs g, Fr s, size_t n) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    const Fr *in = g.in[blockIdx.y];
    Fr *out = g.out[blockIdx.y];
    Fr p = fe_pow_u64(s, (uint64_t)i0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j < n) out[i0 + j] = fe_mul(in[i0 + j], p);
        p = fe_mul(p, s);
    }
}
// out[m * n + j] = in[(j << log_c) + cosets[m]]   (the rank's cosets of a 2^ek-point array, each contiguous)
struct CosetList {
    uint32_t c[16];
    uint32_t count;
};
__global__ __launch_bounds__(256) void coset_gather_kernel(Fr *__restrict__ out, const Fr *__restrict__ in, CosetList cl, uint32_t log_c, size_t n) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    for (uint32_t m = 0; m < cl.count; ++m) out[(size_t)m * n + j] = in[(j << log_c) + cl.c[m]];
}
// out[(j << log_c) + c] = in[slot(c) * n + j] for every coset c < 2^log_c; slot(c) = position of coset c in the all-gathered buffer
struct CosetSlots {
    uint32_t slot[16];
};
__global__ __launch_bounds__(256) void coset_interleave_kernel(Fr *__restrict__ out, const Fr *__restrict__ in, CosetSlots sl, uint32_t log_c, size_t n) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t nc = 1u << log_c;
    for (uint32_t c = 0; c < nc; ++c) out[(j << log_c) + c] = in[(size_t)sl.slot[c] * n + j];
}

// Coefficients of a polynomial of degree < C * n (C = 2^LOGC) from its PER-COSET inverse transforms: coset c of the extended domain is s_c * <omega>,
// s_c = zeta * omega_e^c, and the inverse coset transform of its n evaluations (iNTT of size n, then the scaling by s_c^-t) is
//     P_c[t] = sum_q h[q n + t] * (s_c^n)^q = sum_q (h[q n + t] * zeta^(n q)) * rho^(c q),     rho = omega_e^n (a primitive C-th root of unity)
// — a C-point DFT over q for every t.  So  h[q n + t] = zeta^(-n q) / C * sum_c P_c[t] * rho^(-c q):  one lane per t, a decimation-in-frequency
// network in registers (results in bit-reversed order), C scalings.  The sharded prover runs the size-n transforms where the cosets are (1 / C of
// extended_to_coeff's work each) and only this pointwise step on every rank after the all-gather.
struct CosetCombineArgs {
    uint32_t slot[16];
    Fr tw[8];       // rho^-j, j < C / 2
    Fr scale[16];   // zeta^(-n q) / C
};
template <int LOGC>
__global__ __launch_bounds__(256) void coset_combine_kernel(Fr *__restrict__ out, const Fr *__restrict__ in, CosetCombineArgs a, size_t n) {
    constexpr int C = 1 << LOGC;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Fr x[C];
    static_for<C>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        x[c] = in[(size_t)a.slot[c] * n + t];
    });
    static_for<LOGC>([&](auto sc) {
        constexpr int st = decltype(sc)::value, len = (C / 2) >> st;
        static_for<C / 2>([&](auto pc) {   // pair pr of the stage: block pr / len, offset pr % len
            constexpr int pr = decltype(pc)::value, j = pr % len, lo = (pr / len) * 2 * len + j;
            const Fr u = x[lo], v = x[lo + len];
            x[lo] = fe_add(u, v);
            const Fr d = fe_sub(u, v);
            x[lo + len] = j == 0 ? d : fe_mul(d, a.tw[j << st]);
        });
    });
    static_for<C>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        int rev = 0;
        for (int bit = 0; bit < LOGC; ++bit) rev |= ((q >> bit) & 1) << (LOGC - 1 - bit);
        out[(size_t)q * n + t] = fe_mul(x[rev], a.scale[q]);
    });
}

extern "C" {

int h2hip_permutation_product_terms_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *const *cols_dev, const void *const *sigmas_dev,
                                        uint32_t ncols, uint32_t first_col_index, size_t rows, const void *beta, const void *gamma, const void *delta,
                                        const void *omega) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && beta && gamma && delta && omega && cols_dev && sigmas_dev && (rows == 0 || (num_dev && den_dev)), "NULL argument");
    H2_REQUIRE(ncols >= 1 && ncols <= PP_MAX_COLS, "1..8 columns per permutation set");
    if (!rows) return H2HIP_OK;
    PermProductArgs g;
    memset(&g, 0, sizeof(g));
    for (uint32_t j = 0; j < ncols; ++j) {
        H2_REQUIRE(cols_dev[j] && sigmas_dev[j], "NULL column");
        g.cols[j] = (const Fr *)cols_dev[j];
        g.sigmas[j] = (const Fr *)sigmas_dev[j];
    }
    g.ncols = ncols;
    g.beta = ld(beta); g.gamma = ld(gamma); g.delta = ld(delta); g.omega = ld(omega);
    g.x0 = fe_mul(g.beta, fe_pow_u64(g.delta, first_col_index));
    uint32_t grid = grid_rows(ctx, rows, 4);
    g.xstep = fe_pow_u64(g.omega, (uint64_t)grid * 256);
    prof_begin(ctx, "perm_product_terms_kernel");
    hipLaunchKernelGGL(perm_product_terms_kernel, dim3(grid), dim3(256), 0, ctx->stream, (Fr *)num_dev, (Fr *)den_dev, g, rows);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// every set of the permutation argument: set s owns columns [s*chunk_len, (s+1)*chunk_len) and rows [s*rows, (s+1)*rows) of num / den
int h2hip_permutation_product_terms_sets_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *const *cols_dev, const void *const *sigmas_dev,
                                             uint32_t num_columns, uint32_t chunk_len, size_t rows, const void *beta, const void *gamma,
                                             const void *delta, const void *omega) {
    return h2hip_permutation_product_terms_rows_dev(ctx, num_dev, den_dev, cols_dev, sigmas_dev, num_columns, chunk_len, 0, rows, beta, gamma, delta, omega);
}
// the same for the ROW RANGE [row0, row0 + rows) of the columns (the multi-GPU prover: a rank forms the factors of its rows only); num / den: set s
// at [s * rows, (s + 1) * rows)
int h2hip_permutation_product_terms_rows_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *const *cols_dev, const void *const *sigmas_dev,
                                             uint32_t num_columns, uint32_t chunk_len, size_t row0, size_t rows, const void *beta, const void *gamma,
                                             const void *delta, const void *omega) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && beta && gamma && delta && omega && (num_columns == 0 || (cols_dev && sigmas_dev)) && (rows == 0 || num_columns == 0 || (num_dev && den_dev)),
               "NULL argument");
    H2_REQUIRE(chunk_len >= 1 && chunk_len <= PP_MAX_COLS, "1..8 columns per permutation set");
    if (!rows || !num_columns) return H2HIP_OK;
    for (uint32_t c = 0; c < num_columns; ++c) H2_REQUIRE(cols_dev[c] && sigmas_dev[c], "NULL column");
    PermProductBatchArgs g;
    memset(&g, 0, sizeof(g));
    g.chunk = chunk_len;
    g.beta = ld(beta); g.gamma = ld(gamma); g.delta = ld(delta); g.omega = ld(omega);
    const uint32_t grid = grid_rows(ctx, rows, rows >= ((size_t)1 << 16) ? 4 : 1);   // long columns: four rows per lane amortise the omega^i start-up
    g.xstep = fe_pow_u64(g.omega, (uint64_t)grid * 256);
    OmegaTable pw = {nullptr, nullptr, 0};
    if (ctx->quotient_29) {
        uint32_t log_rows = 0;
        while (((size_t)1 << log_rows) < rows) ++log_rows;
        H2_CHK(ntt_pow_table(ctx, log_rows, g.omega, &pw));
    }
    const uint32_t per_launch = PP_BATCH_COLS / chunk_len * chunk_len;   // whole sets
    Fr x0 = fe_mul(g.beta, fe_pow_u64(g.omega, (uint64_t)row0));
    for (uint32_t c0 = 0; c0 < num_columns; c0 += per_launch) {
        g.ncols = num_columns - c0 < per_launch ? num_columns - c0 : per_launch;
        g.x0 = x0;
        for (uint32_t j = 0; j < g.ncols; ++j) {
            g.cols[j] = (const Fr *)cols_dev[c0 + j] + row0;
            g.sigmas[j] = (const Fr *)sigmas_dev[c0 + j] + row0;
            x0 = fe_mul(x0, g.delta);
        }
        const size_t first_set = c0 / chunk_len;
        prof_begin(ctx, "perm_product_terms_batch_kernel");
        if (ctx->quotient_29) {
            PermProductConsts29 k29;
            k29.beta32 = r29_const(fe_x32(g.beta));
            k29.delta = r29_const(g.delta);
            k29.xstep = r29_const(g.xstep);
            k29.x0_32 = r29_load(fe_x32(g.x0));
            k29.gamma32 = r29_load(fe_x32(g.gamma));
            k29.one = r29_load(Fr::one());
            k29.pw = pw;
            hipLaunchKernelGGL(perm_product_terms_batch29_kernel, dim3(grid), dim3(256), 0, ctx->stream, (Fr *)num_dev + first_set * rows,
                               (Fr *)den_dev + first_set * rows, g, k29, rows);
        } else {
            hipLaunchKernelGGL(perm_product_terms_batch_kernel, dim3(grid), dim3(256), 0, ctx->stream, (Fr *)num_dev + first_set * rows,
                               (Fr *)den_dev + first_set * rows, g, rows);
        }
        prof_end(ctx);
    }
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

int h2hip_lookup_product_terms_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *a_dev, const void *s_dev, const void *a_perm_dev,
                                   const void *s_perm_dev, size_t rows, const void *beta, const void *gamma) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && beta && gamma && (rows == 0 || (num_dev && den_dev && a_dev && s_dev && a_perm_dev && s_perm_dev)), "NULL argument");
    if (!rows) return H2HIP_OK;
    prof_begin(ctx, "lookup_product_terms_kernel");
    hipLaunchKernelGGL(lookup_product_terms_kernel, dim3(grid_rows(ctx, rows, 1)), dim3(256), 0, ctx->stream, (Fr *)num_dev, (Fr *)den_dev,
                       (const Fr *)a_dev, (const Fr *)s_dev, (const Fr *)a_perm_dev, (const Fr *)s_perm_dev, ld(beta), ld(gamma), rows);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// y[i] -= low[i] for i < m <= 8: subtracting the low-degree interpolant r(X) of an opening set from a resident polynomial
// (ProverSHPLONK: P(X) - r(X) before the division by the set's vanishing polynomial); `low_host`: m Montgomery elements
int h2hip_fr_sub_low_dev(h2hip_ctx *ctx, void *y_dev, const void *low_host, uint32_t m) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && m <= 8 && (m == 0 || (y_dev && low_host)), "bad argument (m <= 8)");
    if (!m) return H2HIP_OK;
    SmallPoly p;
    memset(&p, 0, sizeof(p));
    memcpy(p.c, low_host, sizeof(Fr) * m);
    hipLaunchKernelGGL(fr_sub_low_kernel, dim3(1), dim3(64), 0, ctx->stream, (Fr *)y_dev, p, m);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// Assigned::Rational(num, den) -> num * den^-1 with 0^-1 := 0 (Assigned::evaluate / batch_invert_assigned [UPSTREAM]): one batch
// inversion of the denominators and one product, the column never leaves HBM.
int h2hip_assigned_resolve_dev(h2hip_ctx *ctx, void *out_dev, const void *num_dev, const void *den_dev, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (n == 0 || (out_dev && num_dev && den_dev)), "NULL argument");
    H2_REQUIRE(out_dev != num_dev || n == 0, "out must not alias num");
    if (!n) return H2HIP_OK;
    if (out_dev != den_dev) H2_HIPCHK(hipMemcpyAsync(out_dev, den_dev, sizeof(Fr) * n, hipMemcpyDeviceToDevice, ctx->stream));
    H2_CHK(h2hip_fr_batch_invert_dev(ctx, out_dev, n));
    return h2hip_fr_mul_batch_dev(ctx, out_dev, out_dev, num_dev, n);
}

// f(X) -> f(s X) in coefficient form for `count` columns of n coefficients (out may alias in): the first half of a coset evaluation
int h2hip_fr_coset_scale_batch_dev(h2hip_ctx *ctx, void *const *outs_dev, const void *const *ins_dev, size_t count, size_t n, const void *s) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && s && (count == 0 || (outs_dev && ins_dev)), "NULL argument");
    if (!n) return H2HIP_OK;
    Fr sv;
    memcpy(&sv, s, sizeof(Fr));
    for (size_t c0 = 0; c0 < count; c0 += COSET_BATCH) {
        const uint32_t cc = (uint32_t)(count - c0 < COSET_BATCH ? count - c0 : COSET_BATCH);
        CosetScaleArgs g;
        for (uint32_t j = 0; j < COSET_BATCH; ++j) {
            g.in[j] = j < cc ? (const Fr *)ins_dev[c0 + j] : nullptr;
            g.out[j] = j < cc ? (Fr *)outs_dev[c0 + j] : nullptr;
            H2_REQUIRE(j >= cc || (g.in[j] && g.out[j]), "NULL column");
        }
        g.sblock = Fr::one();
        prof_begin(ctx, "coset_scale_kernel");
        hipLaunchKernelGGL(coset_scale_kernel, dim3((uint32_t)(((n + 3) / 4 + 255) / 256), cc), dim3(256), 0, ctx->stream, g, sv, n);
        prof_end(ctx);
    }
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
// the listed cosets (count <= 16, each < 2^log_cosets) of a (n << log_cosets)-point array, one after the other in out (count * n elements)
int h2hip_fr_coset_gather_dev(h2hip_ctx *ctx, void *out_dev, const void *in_dev, const uint32_t *cosets, uint32_t count, uint32_t log_cosets, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (count == 0 || (out_dev && in_dev && cosets)) && count <= 16 && log_cosets <= 4, "bad argument");
    if (!count || !n) return H2HIP_OK;
    CosetList cl;
    cl.count = count;
    for (uint32_t m = 0; m < 16; ++m) {
        cl.c[m] = m < count ? cosets[m] : 0;
        H2_REQUIRE(cl.c[m] < (1u << log_cosets), "coset index out of range");
    }
    prof_begin(ctx, "coset_gather_kernel");
    hipLaunchKernelGGL(coset_gather_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (Fr *)out_dev, (const Fr *)in_dev, cl, log_cosets, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
// the inverse: coset c of the (n << log_cosets)-point array out comes from in[slots[c] * n ...]
int h2hip_fr_coset_interleave_dev(h2hip_ctx *ctx, void *out_dev, const void *in_dev, const uint32_t *slots, uint32_t log_cosets, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && out_dev && in_dev && slots && log_cosets <= 4, "bad argument");
    if (!n) return H2HIP_OK;
    CosetSlots sl;
    for (uint32_t c = 0; c < 16; ++c) sl.slot[c] = c < (1u << log_cosets) ? slots[c] : 0;
    prof_begin(ctx, "coset_interleave_kernel");
    hipLaunchKernelGGL(coset_interleave_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (Fr *)out_dev, (const Fr *)in_dev, sl, log_cosets, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// out[q * n + t] = zeta_n_inv^q / C * sum_c in[slots[c] * n + t] * rho_inv^(c q),  C = 2^log_cosets (1 <= log_cosets <= 4): the coefficients of the
// degree < C * n polynomial whose per-coset inverse transforms P_c sit at in[slots[c] * n ...] (see coset_combine_kernel); out must not alias in
int h2hip_fr_coset_combine_dev(h2hip_ctx *ctx, void *out_dev, const void *in_dev, const uint32_t *slots, uint32_t log_cosets, size_t n, const void *rho_inv,
                               const void *zeta_n_inv) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && out_dev && in_dev && slots && rho_inv && zeta_n_inv && log_cosets >= 1 && log_cosets <= 4 && out_dev != in_dev, "bad argument");
    if (!n) return H2HIP_OK;
    const uint32_t C = 1u << log_cosets;
    CosetCombineArgs a;
    memset((void *)&a, 0, sizeof(a));
    for (uint32_t c = 0; c < C; ++c) a.slot[c] = slots[c];
    const Fr r = ld(rho_inv), zn = ld(zeta_n_inv);
    Fr p = Fr::one();
    for (uint32_t j = 0; j < C / 2; ++j) {
        a.tw[j] = p;
        p = fe_mul(p, r);
    }
    Fr cf = Fr::zero();
    for (uint32_t i = 0; i < C; ++i) cf = fe_add(cf, Fr::one());
    Fr sc = fe_inv(cf);
    for (uint32_t q = 0; q < C; ++q) {
        a.scale[q] = sc;
        sc = fe_mul(sc, zn);
    }
    const dim3 grid((uint32_t)((n + 255) / 256)), blk(256);
    prof_begin(ctx, "coset_combine_kernel");
    if (log_cosets == 1) hipLaunchKernelGGL(coset_combine_kernel<1>, grid, blk, 0, ctx->stream, (Fr *)out_dev, (const Fr *)in_dev, a, n);
    if (log_cosets == 2) hipLaunchKernelGGL(coset_combine_kernel<2>, grid, blk, 0, ctx->stream, (Fr *)out_dev, (const Fr *)in_dev, a, n);
    if (log_cosets == 3) hipLaunchKernelGGL(coset_combine_kernel<3>, grid, blk, 0, ctx->stream, (Fr *)out_dev, (const Fr *)in_dev, a, n);
    if (log_cosets == 4) hipLaunchKernelGGL(coset_combine_kernel<4>, grid, blk, 0, ctx->stream, (Fr *)out_dev, (const Fr *)in_dev, a, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

}  // extern "C"
