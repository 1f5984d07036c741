// Pointwise F_r kernels around the MSM/NTT core: the "witness-column Montgomery mul"
// batches of halo2-base's GateInstructions (reference halo2-base/src/gates/flex_gate/mod.rs:158-277: add /
// sub / mul / mul_add on column values) plus the diagnostic multiplier micro-benchmark that defines the
// integer roofline quoted by bench.py.
#include "internal.h"
#include "fr29.cuh"
#include "fq29.cuh"

namespace h2 {

int ntt_pow_table(h2hip_ctx *ctx, uint32_t log_n, const Fr &omega, OmegaTable *out);   // ntt.hip

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2 };

template <int OP>
__global__ __launch_bounds__(256) void fr_binop_kernel(Fr *__restrict__ out, const Fr *__restrict__ a, const Fr *__restrict__ b, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr x = a[i], y = b[i];
        out[i] = OP == OP_ADD ? fe_add(x, y) : OP == OP_SUB ? fe_sub(x, y) : fe_mul(x, y);
    }
}

// out[i] = a[i]*b[i] + c[i]   (GateInstructions::mul_add, flex_gate/mod.rs:262-277)
__global__ __launch_bounds__(256) void fr_mul_add_kernel(Fr *__restrict__ out, const Fr *__restrict__ a, const Fr *__restrict__ b,
                                                         const Fr *__restrict__ c, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = fe_add(fe_mul(a[i], b[i]), c[i]);
}

// y[i] = y[i]*s + a*x[i]: the polynomial linear combinations of the multiopen argument (Polynomial * F, += of scaled
// polynomials in SHPLONK's rotation-set quotients); x may be null (pure scaling)
__global__ __launch_bounds__(256) void fr_axpby_kernel(Fr *__restrict__ y, Fr s, bool scale_y, const Fr *__restrict__ x, Fr a, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr v = y[i];
        if (scale_y) v = fe_mul(v, s);
        if (x) v = fe_add(v, fe_mul(a, x[i]));
        y[i] = v;
    }
}

// Multiplier roofline probe: every lane runs CHAINS independent dependent-multiply chains of `iters` steps.
// out[i] = (accumulate ? out[i] : 0) + sum_j c_j * p_j[i], up to LINCOMB_MAX polynomials per pass: the multiopen argument's sum_j y^j P_j(X)
// and its linearisation in ONE read of every operand instead of an axpy (read y, read x, write y) per term.  The data stays in its
// saturated Montgomery form x*2^256 as a 9x29-bit integer (f29_split: no multiplication), the coefficients come in the 2^261 form, so a
// product c'*x*2^-261 is again x-form; five products share one Montgomery reduction (f29_dot).
constexpr uint32_t LINCOMB_MAX = 15;
struct LinCombArgs {
    const Fr *p[LINCOMB_MAX];
    Fr29 c[LINCOMB_MAX];   // coefficient * 2^261, normalised, < 1.01 r
    uint32_t count, accumulate;
};
__global__ __launch_bounds__(256) void fr_lincomb_kernel(Fr *out, LinCombArgs a, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr29 tot = a.accumulate ? f29_split<R29P>(out[i]) : Fr29::zero();
        for (uint32_t j0 = 0; j0 < a.count; j0 += 5) {
            Fr29 x[5], c[5];
#pragma unroll
            for (uint32_t t = 0; t < 5; ++t) {
                const bool live = j0 + t < a.count;
                x[t] = live ? f29_split<R29P>(a.p[j0 + t][i]) : Fr29::zero();
                c[t] = live ? a.c[j0 + t] : Fr29::zero();
            }
            tot = f29_add(tot, f29_dot<5>(x, c));   // lazy: at most 1 + 3 normalised terms, limbs < 2^31
        }
        out[i] = f29_pack_canonical<FrP>(f29_weak_reduce(f29_norm(tot)));   // < r + 3 * 1.03 r -> < 2 r -> canonical
    }
}

template <int CHAINS>
__global__ __launch_bounds__(256) void modmul_bench_kernel(Fr *__restrict__ io, uint32_t iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr x[CHAINS];
    Fr y = io[i];
    y.l[7] &= 0x0fffffffu;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        x[k] = y;
        x[k].l[0] ^= (uint32_t)k;
    }
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) x[k] = fe_mul(x[k], y);
    }
    Fr acc = x[0];
#pragma unroll
    for (int k = 1; k < CHAINS; ++k) acc = fe_add(acc, x[k]);
    io[i] = acc;
}


// the same probe on the unsaturated 9 x 29-bit representation (fq29.cuh) the MSM and NTT kernels multiply in
template <int CHAINS>
__global__ __launch_bounds__(256) void modmul29_bench_kernel(Fr *__restrict__ io, uint32_t iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr y0 = io[i];
    y0.l[7] &= 0x0fffffffu;
    Fr29 y = f29_split<R29P>(y0), x[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        x[k] = y;
        x[k].l[0] ^= (uint32_t)k;
    }
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) x[k] = f29_mul(x[k], y);
    }
    Fr29 acc = x[0];
#pragma unroll
    for (int k = 1; k < CHAINS; ++k) acc = f29_norm(f29_add(acc, x[k]));
    io[i] = f29_pack_canonical<FrP>(f29_mul(acc, Fr29::one()));
}

// r04 probe: ONE radix-4 round of the NTT pass kernel (ntt.hip: two stages, four products per lane, the lazy adds / subs / norms between
// them) in a loop, with the parts of the real kernel switched on one at a time — MODE 0: registers only; 1: the four elements and three
// twiddles come from LDS and go back to it every iteration (48-byte elements, conflict-free addresses of a later round); 2: + a block
// barrier per iteration; 3: like 2 with the first round's 4-way conflicting addresses.  Modes >= 1 declare the real kernel's LDS footprint
// (three workgroups per CU).  Reported as products/s (4 per lane and iteration): against h2hip_bench_modmul29's rate it says what the
// round's own instruction stream, its LDS round trip and its barrier each cost (tools/issue_probe.py).
struct alignas(16) ProbeElem {
    Fr29 v;
    uint32_t pad[3];
};
template <int MODE>
__global__ __launch_bounds__(256, 3) void ntt_round_probe_kernel(Fr *__restrict__ io, uint32_t iters) {
    HIP_DYNAMIC_SHARED(ProbeElem, plds)
    const uint32_t tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * blockDim.x + tid;
    Fr y0 = io[i];
    y0.l[7] &= 0x0fffffffu;
    const Fr29 y = f29_split<R29P>(y0);
    Fr29 x0 = y, x1 = y, x2 = y, x3 = y;
    x1.l[0] ^= 1u;
    x2.l[0] ^= 2u;
    x3.l[0] ^= 3u;
    // addresses of a radix-4 group in a 1024-element tile with 4 columns: a later round (st = 2: conflict-free) or the first (st = 0)
    const uint32_t c = tid & 3u, p = tid >> 2;
    const uint32_t st = MODE == 3 ? 0u : 2u, h = 1u << st;
    const uint32_t e0 = ((((p >> st) << (st + 2)) + (p & (h - 1))) << 2) + c, stride = h << 2;
    ProbeElem *tw = plds + 1024;
    if (MODE >= 1) {
        plds[e0].v = x0;
        plds[e0 + stride].v = x1;
        plds[e0 + 2 * stride].v = x2;
        plds[e0 + 3 * stride].v = x3;
        if (tid < 128) tw[tid].v = y;
        __syncthreads();
    }
    Fr29 w1 = y, w2 = y, w3 = y;
    w2.l[1] ^= 5u;
    w3.l[1] ^= 9u;
    for (uint32_t it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            x0 = plds[e0].v;
            x1 = plds[e0 + stride].v;
            x2 = plds[e0 + 2 * stride].v;
            x3 = plds[e0 + 3 * stride].v;
            w1 = tw[(tid + it) & 127u].v;
            w2 = tw[(tid + 2 * it + 1) & 127u].v;
            w3 = tw[(tid + 3 * it + 2) & 127u].v;
        }
        x1 = f29_mul(x1, w1);
        x3 = f29_mul(x3, w1);
        const Fr29 a0 = f29_add(x0, x1), a1 = f29_sub_lazy<2>(x0, x1);
        const Fr29 a2 = f29_mul_wide(f29_add(x2, x3), w2);
        const Fr29 a3 = f29_mul_wide(f29_sub_lazy<2>(x2, x3), w3);
        x0 = f29_norm(f29_add(a0, a2));
        x2 = f29_sub<2>(a0, a2);
        x1 = f29_norm(f29_add(a1, a3));
        x3 = f29_sub<2>(a1, a3);
        // keep the values inside the products' input bounds over many iterations (the real kernel runs <= 5 rounds per tile)
        x0 = f29_weak_reduce(x0);
        x1 = f29_weak_reduce(x1);
        x2 = f29_weak_reduce(x2);
        x3 = f29_weak_reduce(x3);
        if (MODE >= 1) {
            plds[e0].v = x0;
            plds[e0 + stride].v = x1;
            plds[e0 + 2 * stride].v = x2;
            plds[e0 + 3 * stride].v = x3;
        }
        if (MODE >= 2) __syncthreads();
    }
    const Fr29 acc = f29_norm(f29_add(f29_norm(f29_add(x0, x1)), f29_norm(f29_add(x2, x3))));
    io[i] = f29_pack_canonical<FrP>(f29_mul(acc, Fr29::one()));
}

// ------------------------------------------------------------------ K4: BatchInvert (0 -> 0)
// Montgomery's trick over strided runs: lane t owns a[t], a[t+T], a[t+2T], ... (coalesced), one inversion
// (division steps, modinv.cuh) per lane.  [UPSTREAM ff::BatchInvert / halo2 batch_invert_assigned; denominators come from
// reference halo2-base/src/gates/flex_gate/mod.rs:677-681,791-795]
// (r06) src -> dst: out of place when they differ — the grand products invert their denominators straight out of the factor array instead of
// copying it first (a 32 MiB device copy per proof at k = 19, on the critical path between the factors and the prefix products)
__global__ __launch_bounds__(256) void fr_batch_invert_kernel(const Fr *__restrict__ src, Fr *__restrict__ dst, Fr *__restrict__ scratch, size_t n) {
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Fr acc = Fr::one();
    size_t last = t;
    for (size_t i = t; i < n; i += T) {
        Fr v = src[i];
        scratch[i] = acc;
        if (!v.is_zero()) acc = fe_mul(acc, v);
        last = i;
    }
    Fr inv = fe_inv(acc);
    for (size_t i = last;; i -= T) {
        Fr v = src[i];
        if (!v.is_zero()) {
            dst[i] = fe_mul(inv, scratch[i]);
            inv = fe_mul(inv, v);
        } else if (dst != src) {
            dst[i] = v;   // 0 -> 0
        }
        if (i < T) break;
    }
}

// (r05 built the same kernel on unsaturated 9 x 29-bit limbs with the loads issued four elements ahead: bit-exact, 5 - 8 % slower alone — a lane's chain is
// the division steps, not the products — and equal inside proofs, where this kernel's 0.4 ms are the transforms running beside it: 0.129 ms alone for 2^20
// elements.  profiles/r05_invert29_ab.log; removed.)

// ------------------------------------------------------------------ K5: prefix product (grand product core)
// inclusive prefix product over tiles of 256 lanes x SCAN_J consecutive elements
constexpr uint32_t SCAN_J = 8;
// (blockIdx.y = segment: independent products over equal-length segments `seg_stride` elements apart go through the same three launches)
__global__ __launch_bounds__(256) void fr_prefix_prod_tile_kernel(const Fr *__restrict__ in, Fr *__restrict__ out, Fr *__restrict__ tile_prod,
                                                                  size_t n, size_t seg_stride) {
    __shared__ Fr sh[256];
    in += (size_t)blockIdx.y * seg_stride;
    out += (size_t)blockIdx.y * seg_stride;
    tile_prod += (size_t)blockIdx.y * gridDim.x;
    const uint32_t tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * SCAN_J;
    Fr acc = Fr::one();
    for (uint32_t k = 0; k < SCAN_J; ++k)
        if (base + k < n) acc = fe_mul(acc, in[base + k]);
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        Fr o = Fr::one();
        if (tid >= d) o = sh[tid - d];
        __syncthreads();
        if (tid >= d) sh[tid] = fe_mul(sh[tid], o);
        __syncthreads();
    }
    Fr run = tid ? sh[tid - 1] : Fr::one();   // exclusive prefix of this lane inside the tile
    for (uint32_t k = 0; k < SCAN_J; ++k)
        if (base + k < n) {
            run = fe_mul(run, in[base + k]);
            out[base + k] = run;
        }
    if (tid == 255) tile_prod[blockIdx.x] = sh[255];
}
// exclusive prefix product of the tile totals, one workgroup (tiles <= 1024 * per)
__global__ __launch_bounds__(1024) void fr_prefix_prod_sums_kernel(Fr *__restrict__ tile_prod, uint32_t ntiles) {
    __shared__ Fr sh[1024];
    tile_prod += (size_t)blockIdx.x * ntiles;   // one workgroup per segment
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (ntiles + 1023) / 1024, lo = tid * per, hi = lo + per < ntiles ? lo + per : ntiles;
    Fr acc = Fr::one();
    for (uint32_t k = lo; k < hi; ++k) acc = fe_mul(acc, tile_prod[k]);
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        Fr o = Fr::one();
        if (tid >= d) o = sh[tid - d];
        __syncthreads();
        if (tid >= d) sh[tid] = fe_mul(sh[tid], o);
        __syncthreads();
    }
    Fr run = tid ? sh[tid - 1] : Fr::one();
    for (uint32_t k = lo; k < hi; ++k) {
        Fr t = tile_prod[k];
        tile_prod[k] = run;
        run = fe_mul(run, t);
    }
}
__global__ __launch_bounds__(256) void fr_prefix_prod_apply_kernel(Fr *__restrict__ out, const Fr *__restrict__ tile_excl, size_t n, size_t seg_stride) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_J;
    if (blockIdx.x == 0) return;
    out += (size_t)blockIdx.y * seg_stride;
    Fr m = tile_excl[(size_t)blockIdx.y * gridDim.x + blockIdx.x];
    for (uint32_t k = 0; k < SCAN_J; ++k)
        if (base + k < n) out[base + k] = fe_mul(out[base + k], m);
}
// z[0] = 1 (written by the host wrapper), t[i] = num[i] * den_inv[i]
__global__ __launch_bounds__(256) void fr_set_one_kernel(Fr *__restrict__ z) {
    if (threadIdx.x == 0 && blockIdx.x == 0) z[0] = Fr::one();
}
// several grand products at once: the factors num[g] * den_inv[g] of `segments` products of seg_len factors each, laid out for ONE
// (segmented) prefix product.  chained: [1, all factors] — the prefix product over the concatenation IS the chain z_i(0) = z_{i-1}(last);
// otherwise every segment gets its own leading 1: [1, factors of segment 0][1, factors of segment 1]...
__global__ __launch_bounds__(256) void fr_ratio_rows_kernel(Fr *__restrict__ r, const Fr *__restrict__ num, const Fr *__restrict__ den_inv, size_t total,
                                                            size_t seg_len, int chained) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t seg = g / seg_len, i = g - seg * seg_len;
        const size_t dst = chained ? g + 1 : seg * (seg_len + 1) + i + 1;
        r[dst] = fe_mul(num[g], den_inv[g]);
        if (i == 0 && (!chained || seg == 0)) r[dst - 1] = Fr::one();
    }
}
// dst.p[y][i] = src[y * src_stride + i], i < len: rows of a resident matrix out to separately allocated columns (32 per launch)
struct RowPtrs {
    Fr *p[32];
};
__global__ __launch_bounds__(256) void fr_scatter_rows_kernel(RowPtrs dst, const Fr *__restrict__ src, size_t src_stride, size_t len) {
    Fr *__restrict__ d = dst.p[blockIdx.y];
    const Fr *__restrict__ sp = src + (size_t)blockIdx.y * src_stride;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) d[i] = sp[i];
}

// ------------------------------------------------------------------ K7: eval_polynomial / kate_division
struct PowTable {
    Fr p[24];
};
// stage 1: each workgroup evaluates its 256*EVAL_J coefficients relative to its first one:
// lane Horner over EVAL_J coefficients, then a tree with x^(EVAL_J * 2^l)
constexpr uint32_t EVAL_J = 32;   // (r06: 8 -> 32: the 8-level workgroup tree costs a product per level on every lane, as much as 8 Horner steps; PMC: 446 -> see profiles/r06_quotient_pmc.md)
__global__ __launch_bounds__(256) void fr_eval_tile_kernel(const Fr *__restrict__ coeffs, size_t n, Fr x, PowTable pw, Fr *__restrict__ tile_val) {
    __shared__ Fr sh[256];
    const uint32_t tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * EVAL_J;
    Fr acc = Fr::zero();
    for (int k = EVAL_J - 1; k >= 0; --k) {
        acc = fe_mul(acc, x);
        if (base + k < n) acc = fe_add(acc, coeffs[base + k]);
    }
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t d = 1, l = 0; d < 256; d <<= 1, ++l) {
        if ((tid & (2 * d - 1)) == 0) sh[tid] = fe_add(sh[tid], fe_mul(sh[tid + d], pw.p[l]));   // p[l] = x^(EVAL_J*2^l)
        __syncthreads();
    }
    if (tid == 0) tile_val[blockIdx.x] = sh[0];
}
// stage 2: one workgroup combines the tile values: sum_b tile_val[b] * X^b with X = x^(256*EVAL_J) = pw.p[8]
__global__ __launch_bounds__(256) void fr_eval_final_kernel(const Fr *__restrict__ tile_val, uint32_t ntiles, PowTable pw, Fr *__restrict__ out) {
    __shared__ Fr sh[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (ntiles + 255) / 256, lo = tid * per;
    const Fr X = pw.p[8];
    Fr acc = Fr::zero();
    for (int k = (int)per - 1; k >= 0; --k) {
        acc = fe_mul(acc, X);
        if (lo + k < ntiles) acc = fe_add(acc, tile_val[lo + k]);
    }
    // lane value must be scaled by X^(per*tid): tree with Y^(2^l), Y = X^per
    sh[tid] = acc;
    __syncthreads();
    Fr Y = fe_pow_u64(X, per);
    for (uint32_t d = 1; d < 256; d <<= 1) {
        if ((tid & (2 * d - 1)) == 0) sh[tid] = fe_add(sh[tid], fe_mul(sh[tid + d], Y));
        Y = fe_sqr(Y);
        __syncthreads();
    }
    if (tid == 0) out[0] = sh[0];
}

// Batched form: `count` (polynomial, point) pairs in two launches and one copy-out — the prover's evaluation round (every queried
// polynomial at x and its rotations) instead of one launch pair + one synchronising copy per evaluation.
struct EvalJob {
    const Fr *coeffs;
    size_t n;
    Fr x;
    PowTable pw;
    Fr29 x29, one29, pw29[8];   // R' form of x, 1 and x^(EVAL_J * 2^l): the tile kernel on unsaturated limbs (fr29.cuh)
};
__global__ __launch_bounds__(256) void fr_eval_tile_batch_kernel(const EvalJob *__restrict__ jobs, uint32_t ntiles_max, Fr *__restrict__ tile_val) {
    __shared__ Fr sh[256];
    const EvalJob &job = jobs[blockIdx.y];
    const size_t n = job.n;
    const uint32_t tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * EVAL_J;
    if ((size_t)blockIdx.x * 256 * EVAL_J >= n && blockIdx.x) return;   // tile past the end of this polynomial (uniform per workgroup)
    const Fr x = job.x;
    const Fr *__restrict__ coeffs = job.coeffs;
    Fr acc = Fr::zero();
    for (int k = EVAL_J - 1; k >= 0; --k) {
        acc = fe_mul(acc, x);
        if (base + k < n) acc = fe_add(acc, coeffs[base + k]);
    }
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t d = 1, l = 0; d < 256; d <<= 1, ++l) {
        if ((tid & (2 * d - 1)) == 0) sh[tid] = fe_add(sh[tid], fe_mul(sh[tid + d], job.pw.p[l]));
        __syncthreads();
    }
    if (tid == 0) tile_val[(size_t)blockIdx.y * ntiles_max + blockIdx.x] = sh[0];
}
// the tile kernel on unsaturated limbs: the point is a per-job constant (R' form), coefficients enter as raw splits, the Horner value stays lazy
// between products; the tile's value (< 11 r after the tree) leaves through one product with R'(1)
__global__ __launch_bounds__(256) void fr_eval_tile_batch29_kernel(const EvalJob *__restrict__ jobs, uint32_t ntiles_max, Fr *__restrict__ tile_val) {
    __shared__ Fr29 sh[256];
    const EvalJob &job = jobs[blockIdx.y];
    const size_t n = job.n;
    const uint32_t tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * EVAL_J;
    if ((size_t)blockIdx.x * 256 * EVAL_J >= n && blockIdx.x) return;   // tile past the end of this polynomial (uniform per workgroup)
    const Fr29 x = job.x29;
    const Fr *__restrict__ coeffs = job.coeffs;
    Fr29 acc = Fr29::zero();
#pragma unroll 1
    for (int k = EVAL_J - 1; k >= 0; --k) {
        acc = f29_mul(acc, x);
        if (base + k < n) acc = f29_add(acc, r29_load(coeffs[base + k]));
    }
    sh[tid] = f29_norm(acc);
    __syncthreads();
    for (uint32_t d = 1, l = 0; d < 256; d <<= 1, ++l) {
        if ((tid & (2 * d - 1)) == 0) sh[tid] = f29_norm(f29_add(sh[tid], f29_mul(sh[tid + d], job.pw29[l])));
        __syncthreads();
    }
    if (tid == 0) tile_val[(size_t)blockIdx.y * ntiles_max + blockIdx.x] = r29_store(f29_mul(sh[0], job.one29));
}
__global__ __launch_bounds__(256) void fr_eval_final_batch_kernel(const EvalJob *__restrict__ jobs, uint32_t ntiles_max, const Fr *__restrict__ tile_val,
                                                                  Fr *__restrict__ out) {
    __shared__ Fr sh[256];
    const EvalJob &job = jobs[blockIdx.x];
    const Fr *tv = tile_val + (size_t)blockIdx.x * ntiles_max;
    uint32_t ntiles = (uint32_t)((job.n + 256 * EVAL_J - 1) / (256 * EVAL_J));
    if (!ntiles) ntiles = 1;
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (ntiles + 255) / 256, lo = tid * per;
    const Fr X = job.pw.p[8];
    Fr acc = Fr::zero();
    for (int k = (int)per - 1; k >= 0; --k) {
        acc = fe_mul(acc, X);
        if (lo + k < ntiles) acc = fe_add(acc, tv[lo + k]);
    }
    sh[tid] = acc;
    __syncthreads();
    Fr Y = fe_pow_u64(X, per);
    for (uint32_t d = 1; d < 256; d <<= 1) {
        if ((tid & (2 * d - 1)) == 0) sh[tid] = fe_add(sh[tid], fe_mul(sh[tid + d], Y));
        Y = fe_sqr(Y);
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = sh[0];
}

// kate_division: q[m] = sum_{j>m} c_j b^(j-m-1), m = 0..n-2  (suffix Horner).  Stage 1 computes every
// workgroup's head H = sum_{j in tile} c_j b^(j-lo); stage 2 turns heads into carries
// carry[blk] = sum_{blk'>blk} H[blk'] * (b^TILE)^(blk'-blk-1); stage 3 replays the tile with its carry.
// (the head of one tile of 256 * J coefficients, saturated arithmetic; `top`: a virtual coefficient of index n, see KateJob::top)
template <uint32_t J>
__device__ __forceinline__ Fr kate_tile_head(const Fr *__restrict__ c, size_t n, size_t lo, Fr b, const PowTable &pw, Fr *sh, const Fr *top) {
    const uint32_t tid = threadIdx.x;
    const size_t base = lo + (size_t)tid * J;
    Fr h = Fr::zero();
    for (int k = (int)J - 1; k >= 0; --k) {
        h = fe_mul(h, b);
        if (base + k < n) h = fe_add(h, c[base + k]);
        else if (top && base + k == n) h = fe_add(h, *top);
    }
    sh[tid] = h;
    __syncthreads();
    // inclusive suffix scan: I_t = h_t + b^J * I_{t+1}
    for (uint32_t d = 1, l = 0; d < 256; d <<= 1, ++l) {
        Fr o = Fr::zero();
        if (tid + d < 256) o = sh[tid + d];
        __syncthreads();
        if (tid + d < 256) sh[tid] = fe_add(sh[tid], fe_mul(o, pw.p[l]));   // p[l] = b^(J*2^l)
        __syncthreads();
    }
    return sh[0];
}
// Division by the vanishing polynomial of SEVERAL points in one pass (ProverSHPLONK's per-rotation-set quotient): by partial fractions,
//   (f(X) - r(X)) / prod_j (X - b_j)  =  sum_j w_j * (f(X) - f(b_j)) / (X - b_j),   w_j = 1 / prod_{i != j} (b_j - b_i),
// where r is the interpolant of f on the b_j — so the quotient is a weighted sum of independent kate divisions of the SAME polynomial: the
// heads / carries of all points are computed side by side and one pass over f writes the combined quotient (instead of one
// heads-carry-apply triple per root on a shrinking intermediate).
struct KateJob {
    Fr b, w;
    Fr top;   // range division (h2hip_fr_kate_division_range_dev): sum_{i >= n} f_i b^(i - n) over the coefficients ABOVE the range held here,
              // which enters the suffix Horner as one more coefficient of index n; zero otherwise
    PowTable pw;
};
// (J = coefficients per lane: a tile is 256 * J coefficients.  The multi-point kernels pick J by the polynomial's length — when there are fewer
// waves than SIMDs, a wave's instruction count IS the kernel's time, and short tiles spread a short polynomial over more waves)
template <uint32_t J>
__global__ __launch_bounds__(256) void fr_kate_heads_multi_kernel(const Fr *__restrict__ c, size_t n, const KateJob *__restrict__ jobs, uint32_t ntiles,
                                                                  Fr *__restrict__ heads) {
    __shared__ Fr sh[256];
    const KateJob &job = jobs[blockIdx.y];
    Fr h = kate_tile_head<J>(c, n, (size_t)blockIdx.x * (256 * J), job.b, job.pw, sh, &job.top);
    if (threadIdx.x == 0) heads[(size_t)blockIdx.y * (ntiles + 1) + blockIdx.x] = h;
}
__global__ __launch_bounds__(256) void fr_kate_carry_multi_kernel(const Fr *__restrict__ heads, Fr *__restrict__ carry, uint32_t ntiles,
                                                                  const KateJob *__restrict__ jobs) {
    __shared__ Fr sh[256];
    const uint32_t tid = threadIdx.x;
    const Fr *hd = heads + (size_t)blockIdx.x * (ntiles + 1);
    Fr *cr = carry + (size_t)blockIdx.x * (ntiles + 1);
    const PowTable &pw = jobs[blockIdx.x].pw;
    const uint32_t per = (ntiles + 255) / 256, lo = tid * per;
    const Fr B = pw.p[8];
    Fr h = Fr::zero();
    for (int k = (int)per - 1; k >= 0; --k) {
        h = fe_mul(h, B);
        if (lo + k < ntiles) h = fe_add(h, hd[lo + k]);
    }
    sh[tid] = h;
    __syncthreads();
    Fr Y = fe_pow_u64(B, per);
    for (uint32_t d = 1; d < 256; d <<= 1) {
        Fr o = Fr::zero();
        if (tid + d < 256) o = sh[tid + d];
        __syncthreads();
        if (tid + d < 256) sh[tid] = fe_add(sh[tid], fe_mul(o, Y));
        Y = fe_sqr(Y);
        __syncthreads();
    }
    Fr car = (tid + 1 < 256) ? sh[tid + 1] : Fr::zero();
    for (int k = (int)per - 1; k >= 0; --k) {
        if (lo + k < ntiles) {
            cr[lo + k] = car;
            car = fe_add(hd[lo + k], fe_mul(car, B));
        }
    }
}
// All M points of the set advance together through one pass over the tile: their Horner values, their suffix scans (one pair of barriers
// per doubling step for all points) and their quotient chains; the tile's incoming carry sits in an extra scan slot (index 256), which the scan
// multiplies by the right power of b^J on its own.  Points beyond m (padding up to the compiled M) carry weight 0.
// (static_for, field.cuh: with `for (j < M)` + `#pragma unroll` the compiler leaves the loops around two field multiplications per point rolled
// and the per-point arrays in scratch memory)
template <int M, uint32_t J, bool TOP>
__global__ __launch_bounds__(256) void fr_kate_apply_multi_kernel(const Fr *__restrict__ c, size_t n, const KateJob *__restrict__ jobs, uint32_t m,
                                                                  uint32_t ntiles, const Fr *__restrict__ carry, Fr *__restrict__ q, int accumulate) {
    __shared__ Fr sh[M][257];
    const uint32_t tid = threadIdx.x;
    const size_t lo = (size_t)blockIdx.x * (256 * J), base = lo + (size_t)tid * J;
    const int ktop = TOP && n >= base && n < base + J ? (int)(n - base) : -1;   // the lane (one in the grid) that holds the virtual coefficient n
    const size_t n_out = n + (TOP ? 1 : 0);
    // (no per-lane arrays over k either: the k loops stay rolled, the coefficients are read again in the second pass — the tile was just read,
    // they come from the caches.  The job list is padded with zero jobs up to M: b, w are wave-uniform loads, no select)
    Fr b[M], h[M];
    static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        b[j] = jobs[j].b;
        h[j] = Fr::zero();
    });
#pragma unroll 1
    for (int k = (int)J - 1; k >= 0; --k) {
        const Fr cvk = base + k < n ? c[base + k] : Fr::zero();   // coefficients past n are zero
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            h[j] = fe_add(fe_mul(h[j], b[j]), cvk);
            if (TOP && k == ktop) h[j] = fe_add(h[j], jobs[j].top);
        });
    }
    static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        sh[j][tid] = h[j];
        if (tid == 0) sh[j][256] = (uint32_t)j < m ? carry[(size_t)j * (ntiles + 1) + blockIdx.x] : Fr::zero();
    });
    __syncthreads();
    for (uint32_t d = 1, l = 0; d <= 256; d <<= 1, ++l) {   // inclusive suffix scan over 257 slots: I_t = h_t + b^J * I_{t+1}, I_256 = carry
        Fr o[M];
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            o[j] = tid + d <= 256 ? sh[j][tid + d] : Fr::zero();
        });
        __syncthreads();
        if (tid + d <= 256) {
            static_for<M>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if ((uint32_t)j < m) sh[j][tid] = fe_add(sh[j][tid], fe_mul(o[j], jobs[j].pw.p[l]));
            });
        }
        __syncthreads();
    }
    Fr tmp[M], w[M];
    static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        tmp[j] = sh[j][tid + 1];
        w[j] = jobs[j].w;
    });
#pragma unroll 1
    for (int k = (int)J - 1; k >= 0; --k) {
        const Fr cvk = base + k < n ? c[base + k] : Fr::zero();
        Fr acc = Fr::zero();
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            tmp[j] = fe_add(cvk, fe_mul(tmp[j], b[j]));             // = quotient coefficient of index base + k - 1 for point j
            if (TOP && k == ktop) tmp[j] = fe_add(tmp[j], jobs[j].top);
            acc = fe_add(acc, fe_mul(w[j], tmp[j]));
        });
        if (base + k < n_out && base + k >= 1) q[base + k - 1] = accumulate ? fe_add(q[base + k - 1], acc) : acc;
    }
}

// ---- the multi-point division on unsaturated limbs (fr29.cuh).  Every product of these kernels has a per-root CONSTANT operand (b, the scan's
// powers of b, the weights), which arrive in R' form: stored coefficients enter as raw splits and everything stays in the stored domain.  Horner
// values are kept lazy (h b + c: limbs < 2^30) where the next product takes them, normalised where LDS or a dot product needs it; the scan's
// values grow by about r per doubling step (< 12 r: far inside the product's input range); a tile's head leaves through one product with R' (1).
struct KateJob29 {
    Fr29 b, w, one;        // R' form of the root, the weight and 1
    Fr29 top;              // raw split of KateJob::top
    Fr29 pw[9];            // R' form of b^(J * 2^l), l <= 8
};
template <uint32_t J>
__global__ __launch_bounds__(256) void fr_kate_heads_multi29_kernel(const Fr *__restrict__ c, size_t n, const KateJob29 *__restrict__ jobs, uint32_t ntiles,
                                                                    Fr *__restrict__ heads, int with_top) {
    __shared__ Fr29 sh[256];
    const KateJob29 &job = jobs[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * (256 * J) + (size_t)tid * J;
    const Fr29 b = job.b;
    Fr29 h = Fr29::zero();
#pragma unroll 1
    for (int k = (int)J - 1; k >= 0; --k) {
        h = f29_mul(h, b);
        if (base + k < n) h = f29_add(h, r29_load(c[base + k]));
        else if (with_top && base + k == n) h = f29_add(h, job.top);
    }
    sh[tid] = f29_norm(h);
    __syncthreads();
    for (uint32_t d = 1, l = 0; d < 256; d <<= 1, ++l) {   // inclusive suffix scan: I_t = h_t + b^J * I_{t+1}
        Fr29 o = Fr29::zero();
        if (tid + d < 256) o = sh[tid + d];
        __syncthreads();
        if (tid + d < 256) sh[tid] = f29_norm(f29_add(sh[tid], f29_mul(o, job.pw[l])));
        __syncthreads();
    }
    if (tid == 0) heads[(size_t)blockIdx.y * (ntiles + 1) + blockIdx.x] = r29_store(f29_mul(sh[0], job.one));
}
template <int M, uint32_t J, bool TOP>
__global__ __launch_bounds__(256) void fr_kate_apply_multi29_kernel(const Fr *__restrict__ c, size_t n, const KateJob29 *__restrict__ jobs, uint32_t m,
                                                                    uint32_t ntiles, const Fr *__restrict__ carry, Fr *__restrict__ q, int accumulate) {
    __shared__ Fr29 sh[M][257];
    const uint32_t tid = threadIdx.x;
    const size_t lo = (size_t)blockIdx.x * (256 * J), base = lo + (size_t)tid * J;
    const int ktop = TOP && n >= base && n < base + J ? (int)(n - base) : -1;
    const size_t n_out = n + (TOP ? 1 : 0);
    Fr29 b[M], h[M];
    static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        b[j] = jobs[j].b;   // (zero jobs behind the last point: weight 0)
        h[j] = Fr29::zero();
    });
#pragma unroll 1
    for (int k = (int)J - 1; k >= 0; --k) {
        const Fr29 cvk = base + k < n ? r29_load(c[base + k]) : Fr29::zero();
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            h[j] = f29_add(f29_mul(h[j], b[j]), cvk);                              // lazy: limbs < 2^30, value < 2.02 r
            if (TOP && k == ktop) h[j] = f29_norm(f29_add(h[j], jobs[j].top));
        });
    }
    static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        sh[j][tid] = f29_norm(h[j]);
        if (tid == 0) sh[j][256] = (uint32_t)j < m ? r29_load(carry[(size_t)j * (ntiles + 1) + blockIdx.x]) : Fr29::zero();
    });
    __syncthreads();
    for (uint32_t d = 1, l = 0; d <= 256; d <<= 1, ++l) {   // inclusive suffix scan over 257 slots: I_t = h_t + b^J * I_{t+1}, I_256 = carry
        Fr29 o[M];
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            o[j] = tid + d <= 256 ? sh[j][tid + d] : Fr29::zero();
        });
        __syncthreads();
        if (tid + d <= 256) {
            static_for<M>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if ((uint32_t)j < m) sh[j][tid] = f29_norm(f29_add(sh[j][tid], f29_mul(o[j], jobs[j].pw[l])));
            });
        }
        __syncthreads();
    }
    Fr29 tmp[M], w[M];
    static_for<M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        tmp[j] = sh[j][tid + 1];
        w[j] = jobs[j].w;
    });
#pragma unroll 1
    for (int k = (int)J - 1; k >= 0; --k) {
        const Fr29 cvk = base + k < n ? r29_load(c[base + k]) : Fr29::zero();
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            tmp[j] = f29_add(cvk, f29_mul(tmp[j], b[j]));                          // = quotient coefficient of index base + k - 1 for point j
            if (TOP && k == ktop) tmp[j] = f29_add(tmp[j], jobs[j].top);
            tmp[j] = f29_norm(tmp[j]);                                             // N, < 3.03 r
        });
        const Fr acc = r29_store(f29_dot<M>(w, tmp));                              // sum_j w_j q_j with one reduction: < 1 + M * 3.06 / 169
        if (base + k < n_out && base + k >= 1) q[base + k - 1] = accumulate ? fe_add(q[base + k - 1], acc) : acc;
    }
}

// ------------------------------------------------------------------ K8: Poseidon permutation batches
// One lane per instance, textbook rounds (ARK, x^5, MDS) with the caller's spec — algebraically equal to
// halo2-base's optimised PoseidonState::permutation (reference halo2-base/src/poseidon/hasher/state.rs:35-83,
// absorb rule :124-160: inputs added to s[1..], a padding 1 after the last input when fewer than RATE).
// One lane per permutation, state in the unsaturated R' = 2^261 domain for the whole permutation (converted on entry and
// exit: 2T of the ~400 products): the S-box is two squarings and a product on lazy sums, and an MDS row is ONE dual...
// T-fold product with a single Montgomery reduction (f29_dot) instead of T products and T reductions.  rc / mds arrive
// pre-converted (R' form, packed 8 x 32 bit) from h2hip_poseidon_set_spec.
template <int T>
__global__ __launch_bounds__(256) void poseidon_permute_kernel(Fr *__restrict__ states, const Fr *__restrict__ inputs, uint32_t num_inputs,
                                                               size_t n, const Fr *__restrict__ rc, const Fr *__restrict__ mds, uint32_t r_f,
                                                               uint32_t r_p) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr29 s[T];
#pragma unroll
    for (int k = 0; k < T; ++k) s[k] = fr29_from_sat(states[i * T + k]);
#pragma unroll
    for (int k = 0; k < T - 1; ++k) {
        if ((uint32_t)k < num_inputs) s[k + 1] = f29_norm(f29_add(s[k + 1], fr29_from_sat(inputs[i * num_inputs + k])));
        else if ((uint32_t)k == num_inputs) s[k + 1] = f29_norm(f29_add(s[k + 1], Fr29::one()));
    }
    const uint32_t half = r_f / 2;
    for (uint32_t r = 0; r < r_f + r_p; ++r) {
        const bool full = r < half || r >= half + r_p;
#pragma unroll
        for (int k = 0; k < T; ++k) {
            Fr29 v = f29_add(s[k], f29_split<R29P>(rc[r * T + k]));   // lazy: value < 3.1 r, limbs <= 2^30
            if (full || k == 0) {
                Fr29 v2 = f29_sqr(v);
                v = f29_mul(f29_sqr(v2), v);                            // x^5, N, < 1.03 r
            } else {
                v = f29_norm(v);
            }
            s[k] = v;
        }
        Fr29 o[T];
#pragma unroll
        for (int a = 0; a < T; ++a) {
            Fr29 row[T];
#pragma unroll
            for (int b = 0; b < T; ++b) row[b] = f29_split<R29P>(mds[a * T + b]);
            o[a] = f29_dot<T>(row, s);                                  // sum_b X_a*X_b <= 5 * 1.01 * 3.1 -> < 1.1 r
        }
#pragma unroll
        for (int k = 0; k < T; ++k) s[k] = o[k];
    }
#pragma unroll
    for (int k = 0; k < T; ++k) states[i * T + k] = fr29_to_sat(s[k]);
}

// ------------------------------------------------------------------ K6: halo2-base gate term of the quotient
// acc[i] = acc[i]*y + q[i] * (a[i] + a[i+s]*a[i+2s] - a[i+3s])  on the extended domain (indices mod n_ext,
// s = 2^(ext_k-k) = one row): the single custom gate of halo2-base, q*(a + b*c - d) at rotations 0..3
// (reference halo2-base/src/gates/flex_gate/mod.rs:80-91), folded into h's numerator by powers of y.
__global__ __launch_bounds__(256) void quotient_flex_gate_kernel(Fr *__restrict__ acc, const Fr *__restrict__ q, const Fr *__restrict__ a,
                                                                 size_t n_ext, uint32_t rot_step, Fr y) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = n_ext - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ext; i += stride) {
        Fr a0 = a[i], a1 = a[(i + rot_step) & mask], a2 = a[(i + 2 * (size_t)rot_step) & mask], a3 = a[(i + 3 * (size_t)rot_step) & mask];
        Fr g = fe_mul(q[i], fe_sub(fe_add(a0, fe_mul(a1, a2)), a3));
        acc[i] = fe_add(fe_mul(acc[i], y), g);
    }
}

// ------------------------------------------------------------------ K6: lookup and permutation identities of h(X)
// Pointwise over the extended domain (ne = 2^ext_k points, one circuit row = `step` = 2^(ext_k-k) points), every
// identity folded into the numerator as acc = acc*y + term in upstream's order [UPSTREAM evaluation.rs, SURVEY.md A.4/A.5;
// halo2-base creates these arguments at halo2-base/src/gates/range/mod.rs:131-150 (lookup) and
// halo2-base/src/gates/flex_gate/mod.rs:69,124-128 (equality-enabled columns)].
struct LookupArgs {
    const Fr *z, *a, *s, *ap, *sp, *l0, *l_last, *l_blind;
    Fr beta, gamma, y;
};
// terms: l0*(1-z) ; l_last*(z^2-z) ; active*(z(wX)(a'+beta)(s'+gamma) - z(X)(a+beta)(s+gamma)) ; l0*(a'-s') ;
//        active*(a'-s')*(a'-a'(w^-1 X)),   active = 1 - (l_last + l_blind)
__global__ __launch_bounds__(256) void quotient_lookup_kernel(Fr *__restrict__ acc, LookupArgs g, size_t ne, uint32_t step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ne - 1;
    const Fr one = Fr::one();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += stride) {
        const size_t inext = (i + step) & mask, iprev = (i + ne - step) & mask;
        Fr z = g.z[i], a = g.a[i], sv = g.s[i], ap = g.ap[i], sp = g.sp[i], l0 = g.l0[i], ll = g.l_last[i];
        Fr active = fe_sub(one, fe_add(ll, g.l_blind[i]));
        Fr v = acc[i];
        v = fe_add(fe_mul(v, g.y), fe_mul(l0, fe_sub(one, z)));
        v = fe_add(fe_mul(v, g.y), fe_mul(ll, fe_sub(fe_sqr(z), z)));
        Fr left = fe_mul(fe_mul(g.z[inext], fe_add(ap, g.beta)), fe_add(sp, g.gamma));
        Fr right = fe_mul(fe_mul(z, fe_add(a, g.beta)), fe_add(sv, g.gamma));
        v = fe_add(fe_mul(v, g.y), fe_mul(active, fe_sub(left, right)));
        Fr d = fe_sub(ap, sp);
        v = fe_add(fe_mul(v, g.y), fe_mul(l0, d));
        v = fe_add(fe_mul(v, g.y), fe_mul(active, fe_mul(d, fe_sub(ap, g.ap[iprev]))));
        acc[i] = v;
    }
}

constexpr int PERM_MAX_COLS = 8;
struct PermArgs {
    const Fr *z, *z_prev, *l0, *l_last, *l_blind;
    const Fr *cols[PERM_MAX_COLS], *sigmas[PERM_MAX_COLS];
    uint32_t ncols, terms, last_rot_points;   // terms: H2HIP_PERM_* mask; last_rot_points = last_rotation * step (already reduced mod ne)
    Fr beta, gamma, delta, y;
    Fr x0_delta;   // beta * zeta * delta^(first column index of this set): the X-term coefficient at extended point 0
    Fr ext_omega;
    Fr xstep;      // ext_omega^(grid stride), computed on the host
};
// terms for one permutation set i: [first set] l0*(1-z) ; [last set] l_last*(z^2-z) ; [i>0] l0*(z_i - z_{i-1}(w^last X)) ;
//        active*( z(wX) prod_j(p_j + beta*s_j + gamma) - z(X) prod_j(p_j + delta^j*beta*X + gamma) ),  X = zeta*w_ext^i
__global__ __launch_bounds__(256) void quotient_permutation_kernel(Fr *__restrict__ acc, PermArgs g, size_t ne, uint32_t step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ne - 1;
    const Fr one = Fr::one();
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr xbase = fe_mul(g.x0_delta, fe_pow_u64(g.ext_omega, (uint64_t)i0));   // beta * delta^j0 * zeta * w_ext^i
    const Fr xstep = g.xstep;
    for (size_t i = i0; i < ne; i += stride, xbase = fe_mul(xbase, xstep)) {
        const size_t inext = (i + step) & mask;
        Fr z = g.z[i], l0 = g.l0[i], ll = g.l_last[i];
        Fr active = fe_sub(one, fe_add(ll, g.l_blind[i]));
        Fr v = acc[i];
        if (g.terms & H2HIP_PERM_FIRST) v = fe_add(fe_mul(v, g.y), fe_mul(l0, fe_sub(one, z)));
        if (g.terms & H2HIP_PERM_LAST) v = fe_add(fe_mul(v, g.y), fe_mul(ll, fe_sub(fe_sqr(z), z)));
        if (g.terms & H2HIP_PERM_CHAIN) v = fe_add(fe_mul(v, g.y), fe_mul(l0, fe_sub(z, g.z_prev[(i + g.last_rot_points) & mask])));
        if (g.terms & H2HIP_PERM_PRODUCT) {
            Fr left = g.z[inext], right = z;
            Fr xterm = xbase;
            for (uint32_t j = 0; j < g.ncols; ++j) {
                Fr p = g.cols[j][i];
                left = fe_mul(left, fe_add(fe_add(p, fe_mul(g.beta, g.sigmas[j][i])), g.gamma));
                right = fe_mul(right, fe_add(fe_add(p, xterm), g.gamma));
                xterm = fe_mul(xterm, g.delta);
            }
            v = fe_add(fe_mul(v, g.y), fe_mul(active, fe_sub(left, right)));
        }
        acc[i] = v;
    }
}

// The same kernel on unsaturated limbs (fr29.cuh): every product at the 9 x 29 rate, acc*y + term with ONE reduction.  One operand of every
// data x data product carries the factor 32 — taken when a stored element is split (r29_load32) or folded into the constants of the factor
// it is built from: the X term's start value, beta and gamma arrive as 32 beta zeta delta^j0, 32 beta, 32 gamma.  Bounds (multiples of r):
// acc < 1.64, the factors 32 p + 32 beta s + 32 gamma < 34.02, left / right < 1.25, every term's operands 32 x 3.3 at most.
struct PermArgs29 {
    const Fr *z, *z_prev, *l0, *l_last, *l_blind;
    const Fr *cols[PERM_MAX_COLS], *sigmas[PERM_MAX_COLS];
    uint32_t ncols, terms, last_rot_points;
    Fr29 beta32, delta, y, xstep;               // R' form (r29_const) of 32 beta, delta, y, ext_omega^(grid stride)
    Fr29 x0_delta32;                            // raw split of 32 beta zeta delta^j0 (stored domain): the X term's start = ext_omega^i0 (table, R' form) x this
    Fr29 gamma32;                               // raw split of (32 gamma mod r) in the stored domain
    OmegaTable pw;                              // ext_omega^e, e < 2^ext_k: the coset transforms' twiddle set (r06: was ~28 saturated products per lane)
};
__global__ __launch_bounds__(256, 3) void quotient_permutation29_kernel(Fr *__restrict__ acc, PermArgs29 g, size_t ne, uint32_t step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ne - 1;
    const Fr one_sat = Fr::one();
    const Fr29 one = r29_load(one_sat);
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= ne) return;   // (the table holds ext_omega^e for e < ne only)
    Fr29 xbase = f29_mul(pow_lookup(g.pw.t1, g.pw.t2, g.pw.lo_bits, (uint64_t)i0), g.x0_delta32);   // 32 beta delta^j0 zeta w_ext^i, < 1.01
    for (size_t i = i0; i < ne; i += stride, xbase = f29_mul(xbase, g.xstep)) {
        const size_t inext = (i + step) & mask;
        const Fr z_sat = g.z[i], ll_sat = g.l_last[i];
        const Fr29 z = r29_load(z_sat);
        Fr29 v = r29_load(acc[i]);
        if (g.terms & H2HIP_PERM_FIRST) v = f29_mul2(v, g.y, r29_load32(g.l0[i]), f29_sub<2>(one, z));            // 1.64 + 32 * 3
        if (g.terms & H2HIP_PERM_LAST) {
            const Fr29 zz = f29_mul(r29_load32(z_sat), f29_sub<2>(z, one));                                       // z (z - 1) < 1.57
            v = f29_mul2(v, g.y, r29_load32(ll_sat), zz);
        }
        if (g.terms & H2HIP_PERM_CHAIN)
            v = f29_mul2(v, g.y, r29_load32(g.l0[i]), f29_sub<2>(z, r29_load(g.z_prev[(i + g.last_rot_points) & mask])));
        if (g.terms & H2HIP_PERM_PRODUCT) {
            Fr29 left = r29_load(g.z[inext]), right = z, xterm = xbase;
            for (uint32_t j = 0; j < g.ncols; ++j) {
                const Fr29 p32 = f29_add(r29_load32(g.cols[j][i]), g.gamma32);                                    // lazy, limbs < 2^30
                const Fr29 fl = f29_norm(f29_add(p32, f29_mul(r29_load(g.sigmas[j][i]), g.beta32)));              // 32 (p + beta s + gamma) < 34.02
                const Fr29 fr = f29_norm(f29_add(p32, xterm));
                left = f29_mul(left, fl);
                right = f29_mul(right, fr);
                xterm = f29_mul(xterm, g.delta);
            }
            const Fr active = fe_sub(one_sat, fe_add(ll_sat, g.l_blind[i]));
            v = f29_mul2(v, g.y, r29_load32(active), f29_sub<2>(left, right));                                    // 1.64 + 32 * 3.25
        }
        acc[i] = r29_store(v);
    }
}

// ---- the same identities for MANY columns / sets / lookups per launch (wide shapes: hundreds of columns of a few thousand rows): every
// launch reads and writes the accumulator once and folds its jobs in order, acc = acc*y + term per job — the same values as one launch per
// job, without a few-hundred-workgroup launch (and an accumulator round trip) per column.  Job tables travel as kernel arguments.
constexpr uint32_t GATE_BATCH = 64, LOOKUP_BATCH = 32, PERM_BATCH = 12;
struct GateBatchArgs {
    const Fr *q[GATE_BATCH], *a[GATE_BATCH];
    uint32_t count;
    Fr y;
};
__global__ __launch_bounds__(256) void quotient_flex_gate_batch_kernel(Fr *__restrict__ acc, GateBatchArgs g, size_t n_ext, uint32_t rot_step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = n_ext - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ext; i += stride) {
        const size_t i1 = (i + rot_step) & mask, i2 = (i + 2 * (size_t)rot_step) & mask, i3 = (i + 3 * (size_t)rot_step) & mask;
        Fr v = acc[i];
        for (uint32_t j = 0; j < g.count; ++j) {
            const Fr *__restrict__ a = g.a[j];
            Fr t = fe_mul(g.q[j][i], fe_sub(fe_add(a[i], fe_mul(a[i1], a[i2])), a[i3]));
            v = fe_add(fe_mul(v, g.y), t);
        }
        acc[i] = v;
    }
}
struct LookupJob {
    const Fr *z, *a, *s, *ap, *sp;
};
struct LookupBatchArgs {
    const Fr *l0, *l_last, *l_blind;
    Fr beta, gamma, y;
    uint32_t count;
    LookupJob jobs[LOOKUP_BATCH];
};
__global__ __launch_bounds__(256) void quotient_lookup_batch_kernel(Fr *__restrict__ acc, LookupBatchArgs g, size_t ne, uint32_t step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ne - 1;
    const Fr one = Fr::one();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += stride) {
        const size_t inext = (i + step) & mask, iprev = (i + ne - step) & mask;
        const Fr l0 = g.l0[i], ll = g.l_last[i];
        const Fr active = fe_sub(one, fe_add(ll, g.l_blind[i]));
        Fr v = acc[i];
        for (uint32_t j = 0; j < g.count; ++j) {
            const LookupJob &q = g.jobs[j];
            Fr z = q.z[i], a = q.a[i], sv = q.s[i], ap = q.ap[i], sp = q.sp[i];
            v = fe_add(fe_mul(v, g.y), fe_mul(l0, fe_sub(one, z)));
            v = fe_add(fe_mul(v, g.y), fe_mul(ll, fe_sub(fe_sqr(z), z)));
            Fr left = fe_mul(fe_mul(q.z[inext], fe_add(ap, g.beta)), fe_add(sp, g.gamma));
            Fr right = fe_mul(fe_mul(z, fe_add(a, g.beta)), fe_add(sv, g.gamma));
            v = fe_add(fe_mul(v, g.y), fe_mul(active, fe_sub(left, right)));
            Fr d = fe_sub(ap, sp);
            v = fe_add(fe_mul(v, g.y), fe_mul(l0, d));
            v = fe_add(fe_mul(v, g.y), fe_mul(active, fe_mul(d, fe_sub(ap, q.ap[iprev]))));
        }
        acc[i] = v;
    }
}
struct PermJob {
    const Fr *z, *z_prev;
    const Fr *cols[PERM_MAX_COLS], *sigmas[PERM_MAX_COLS];
    Fr x0_delta;   // beta * zeta * delta^(first column index of the set)
    uint32_t ncols, terms;
};
struct PermBatchArgs {
    const Fr *l0, *l_last, *l_blind;
    Fr beta, gamma, delta, y, ext_omega, xstep;
    uint32_t last_rot_points, njobs;
    PermJob jobs[PERM_BATCH];
};
__global__ __launch_bounds__(256) void quotient_permutation_batch_kernel(Fr *__restrict__ acc, PermBatchArgs g, size_t ne, uint32_t step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ne - 1;
    const Fr one = Fr::one();
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr wpow = fe_pow_u64(g.ext_omega, (uint64_t)i0);   // w_ext^i
    for (size_t i = i0; i < ne; i += stride, wpow = fe_mul(wpow, g.xstep)) {
        const size_t inext = (i + step) & mask;
        const Fr l0 = g.l0[i], ll = g.l_last[i];
        const Fr active = fe_sub(one, fe_add(ll, g.l_blind[i]));
        Fr v = acc[i];
        for (uint32_t jb = 0; jb < g.njobs; ++jb) {
            const PermJob &q = g.jobs[jb];
            const Fr z = q.z[i];
            if (q.terms & H2HIP_PERM_FIRST) v = fe_add(fe_mul(v, g.y), fe_mul(l0, fe_sub(one, z)));
            if (q.terms & H2HIP_PERM_LAST) v = fe_add(fe_mul(v, g.y), fe_mul(ll, fe_sub(fe_sqr(z), z)));
            if (q.terms & H2HIP_PERM_CHAIN) v = fe_add(fe_mul(v, g.y), fe_mul(l0, fe_sub(z, q.z_prev[(i + g.last_rot_points) & mask])));
            if (q.terms & H2HIP_PERM_PRODUCT) {
                Fr left = q.z[inext], right = z;
                Fr xterm = fe_mul(q.x0_delta, wpow);
                for (uint32_t j = 0; j < q.ncols; ++j) {
                    Fr p = q.cols[j][i];
                    left = fe_mul(left, fe_add(fe_add(p, fe_mul(g.beta, q.sigmas[j][i])), g.gamma));
                    right = fe_mul(right, fe_add(fe_add(p, xterm), g.gamma));
                    xterm = fe_mul(xterm, g.delta);
                }
                v = fe_add(fe_mul(v, g.y), fe_mul(active, fe_sub(left, right)));
            }
        }
        acc[i] = v;
    }
}

// ---- the batched kernels on unsaturated limbs (fr29.cuh; see quotient_permutation29_kernel for the factor-of-32 bookkeeping).  Differences of
// stored elements that enter a data x data product are formed in saturated arithmetic first (an add-with-carry chain) and split with the factor.
__global__ __launch_bounds__(256) void quotient_flex_gate_batch29_kernel(Fr *__restrict__ acc, GateBatchArgs g, Fr29 y, size_t n_ext, uint32_t rot_step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = n_ext - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ext; i += stride) {
        const size_t i1 = (i + rot_step) & mask, i2 = (i + 2 * (size_t)rot_step) & mask, i3 = (i + 3 * (size_t)rot_step) & mask;
        Fr29 v = r29_load(acc[i]);
        for (uint32_t j = 0; j < g.count; ++j) {
            const Fr *__restrict__ a = g.a[j];
            const Fr29 bc = f29_mul(r29_load32(a[i1]), r29_load(a[i2]));                        // < 1.19
            const Fr29 t = f29_sub<2>(f29_add(r29_load(a[i]), bc), r29_load(a[i3]));          // a + b c - d + 2 r < 4.2
            v = f29_mul2(v, y, r29_load32(g.q[j][i]), t);                                     // 1.8 + 32 * 4.2 = 136.2 -> < 1.81
        }
        acc[i] = r29_store(v);
    }
}
struct LookupConsts29 {
    Fr29 y;                  // R' form
    Fr29 beta32, gamma32;    // raw splits of 32 beta, 32 gamma (stored domain)
};
__global__ __launch_bounds__(256, 3) void quotient_lookup_batch29_kernel(Fr *__restrict__ acc, LookupBatchArgs g, LookupConsts29 k29, size_t ne, uint32_t step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ne - 1;
    const Fr one_sat = Fr::one();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += stride) {
        const size_t inext = (i + step) & mask, iprev = (i + ne - step) & mask;
        const Fr ll_sat = g.l_last[i];
        const Fr29 l0 = r29_load32(g.l0[i]), ll = r29_load32(ll_sat), active = r29_load32(fe_sub(one_sat, fe_add(ll_sat, g.l_blind[i])));
        Fr29 v = r29_load(acc[i]);
        for (uint32_t j = 0; j < g.count; ++j) {
            const LookupJob &q = g.jobs[j];
            const Fr z_sat = q.z[i], ap_sat = q.ap[i], sp_sat = q.sp[i];
            const Fr29 z = r29_load(z_sat);
            v = f29_mul2(v, k29.y, l0, r29_load(fe_sub(one_sat, z_sat)));                                          // l0 (1 - z): 1.7 + 32
            v = f29_mul2(v, k29.y, ll, f29_mul(r29_load32(z_sat), r29_load(fe_sub(z_sat, one_sat))));              // l_last z (z - 1)
            Fr29 left = f29_mul(r29_load(q.z[inext]), f29_add(r29_load32(ap_sat), k29.beta32));                    // 1 x 33
            left = f29_mul(left, f29_add(r29_load32(sp_sat), k29.gamma32));                                        // 1.2 x 33
            Fr29 right = f29_mul(z, f29_add(r29_load32(q.a[i]), k29.beta32));
            right = f29_mul(right, f29_add(r29_load32(q.s[i]), k29.gamma32));
            v = f29_mul2(v, k29.y, active, f29_sub<2>(left, right));                                               // 1.7 + 32 * 3.25
            const Fr d_sat = fe_sub(ap_sat, sp_sat);
            v = f29_mul2(v, k29.y, l0, r29_load(d_sat));                                                           // l0 (a' - s')
            v = f29_mul2(v, k29.y, active, f29_mul(r29_load32(d_sat), r29_load(fe_sub(ap_sat, q.ap[iprev]))));     // active (a' - s')(a' - a'(w^-1 X))
        }
        acc[i] = r29_store(v);
    }
}
struct PermConsts29 {
    Fr29 beta32, delta, y, xstep;   // R' form of 32 beta, delta, y, ext_omega^(grid stride)
    Fr29 gamma32;                   // raw split of 32 gamma
    Fr29 x0_delta32[PERM_BATCH];    // raw split of 32 beta zeta delta^(first column of the job's set) (the X term = w_ext^i in R' form x this)
    OmegaTable pw;                  // ext_omega^e, e < 2^ext_k
};
__global__ __launch_bounds__(256, 3) void quotient_permutation_batch29_kernel(Fr *__restrict__ acc, PermBatchArgs g, PermConsts29 k29, size_t ne, uint32_t step) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, mask = ne - 1;
    const Fr one_sat = Fr::one();
    const Fr29 one = r29_load(one_sat);
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= ne) return;   // (the table holds ext_omega^e for e < ne only)
    Fr29 wpow = pow_lookup(k29.pw.t1, k29.pw.t2, k29.pw.lo_bits, (uint64_t)i0);   // w_ext^i in R' form (the chain stays there: xstep is an R' constant)
    for (size_t i = i0; i < ne; i += stride, wpow = f29_mul(wpow, k29.xstep)) {
        const size_t inext = (i + step) & mask;
        const Fr ll_sat = g.l_last[i];
        const Fr29 l0 = r29_load32(g.l0[i]), ll = r29_load32(ll_sat), active = r29_load32(fe_sub(one_sat, fe_add(ll_sat, g.l_blind[i])));
        Fr29 v = r29_load(acc[i]);
        for (uint32_t jb = 0; jb < g.njobs; ++jb) {
            const PermJob &q = g.jobs[jb];
            const Fr z_sat = q.z[i];
            const Fr29 z = r29_load(z_sat);
            if (q.terms & H2HIP_PERM_FIRST) v = f29_mul2(v, k29.y, l0, f29_sub<2>(one, z));
            if (q.terms & H2HIP_PERM_LAST) v = f29_mul2(v, k29.y, ll, f29_mul(r29_load32(z_sat), f29_sub<2>(z, one)));
            if (q.terms & H2HIP_PERM_CHAIN) v = f29_mul2(v, k29.y, l0, f29_sub<2>(z, r29_load(q.z_prev[(i + g.last_rot_points) & mask])));
            if (q.terms & H2HIP_PERM_PRODUCT) {
                Fr29 left = r29_load(q.z[inext]), right = z;
                Fr29 xterm = f29_mul(wpow, k29.x0_delta32[jb]);
                for (uint32_t j = 0; j < q.ncols; ++j) {
                    const Fr29 p32 = f29_add(r29_load32(q.cols[j][i]), k29.gamma32);
                    const Fr29 fl = f29_norm(f29_add(p32, f29_mul(r29_load(q.sigmas[j][i]), k29.beta32)));
                    const Fr29 fr = f29_norm(f29_add(p32, xterm));
                    left = f29_mul(left, fl);
                    right = f29_mul(right, fr);
                    xterm = f29_mul(xterm, k29.delta);
                }
                v = f29_mul2(v, k29.y, active, f29_sub<2>(left, right));
            }
        }
        acc[i] = r29_store(v);
    }
}

// HBM-counter calibration probes (profiles/archive/r02_*_pmc_*.md): a random gather of aligned ENTRY-byte table entries — the access pattern
// of msm_accum_kernel's base-table reads (one aligned 64-byte entry per mixed addition out of a table far larger than the 256 MiB
// Infinity Cache) — with an exactly known useful byte count, and a coalesced stream of the same volume.
template <int ENTRY>
__global__ __launch_bounds__(256) void gather_probe_kernel(const uint4 *__restrict__ table, uint64_t entries, uint32_t per_lane, uint4 *__restrict__ out) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t state = lane * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    uint4 acc = {0u, 0u, 0u, 0u};
    for (uint32_t k = 0; k < per_lane; ++k) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t idx = (state >> 20) % entries;
        const uint4 *e = table + idx * (ENTRY / 16);
#pragma unroll
        for (int q = 0; q < ENTRY / 16; ++q) {
            uint4 v = e[q];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    out[lane] = acc;
}
__global__ __launch_bounds__(256) void stream_probe_kernel(const uint4 *__restrict__ table, uint64_t vec16, uint4 *__restrict__ out) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    uint4 acc = {0u, 0u, 0u, 0u};
    for (uint64_t i = lane; i < vec16; i += stride) {
        uint4 v = table[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    out[lane] = acc;
}

static uint32_t grid_for(h2hip_ctx *ctx, size_t n) {
    size_t blocks = (n + 255) / 256, cap = (size_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    return (uint32_t)(blocks ? blocks : 1);
}

}  // namespace h2

using namespace h2;

extern "C" {

static int binop(h2hip_ctx *ctx, int op, void *out, const void *a, const void *b, size_t n) {
    H2_REQUIRE(ctx && (n == 0 || (out && a && b)), "NULL argument");
    if (!n) return H2HIP_OK;
    dim3 g(grid_for(ctx, n)), blk(256);
    prof_begin(ctx, "fr_binop_kernel");
    if (op == OP_ADD) hipLaunchKernelGGL(fr_binop_kernel<OP_ADD>, g, blk, 0, ctx->stream, (Fr *)out, (const Fr *)a, (const Fr *)b, n);
    if (op == OP_SUB) hipLaunchKernelGGL(fr_binop_kernel<OP_SUB>, g, blk, 0, ctx->stream, (Fr *)out, (const Fr *)a, (const Fr *)b, n);
    if (op == OP_MUL) hipLaunchKernelGGL(fr_binop_kernel<OP_MUL>, g, blk, 0, ctx->stream, (Fr *)out, (const Fr *)a, (const Fr *)b, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
int h2hip_fr_add_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, size_t n) {
    H2_DEVICE_GUARD(ctx); return binop(ctx, OP_ADD, out, a, b, n); }
int h2hip_fr_sub_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, size_t n) {
    H2_DEVICE_GUARD(ctx); return binop(ctx, OP_SUB, out, a, b, n); }
int h2hip_fr_mul_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, size_t n) {
    H2_DEVICE_GUARD(ctx); return binop(ctx, OP_MUL, out, a, b, n); }
int h2hip_fr_mul_add_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, const void *c, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (n == 0 || (out && a && b && c)), "NULL argument");
    if (!n) return H2HIP_OK;
    prof_begin(ctx, "fr_mul_add_kernel");
    hipLaunchKernelGGL(fr_mul_add_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (Fr *)out, (const Fr *)a, (const Fr *)b,
                       (const Fr *)c, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
int h2hip_fr_axpy_dev(h2hip_ctx *ctx, void *y, const void *a, const void *x, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a && (n == 0 || (y && x)), "NULL argument");
    if (!n) return H2HIP_OK;
    Fr av;
    memcpy(&av, a, sizeof(Fr));
    prof_begin(ctx, "fr_axpby_kernel");
    hipLaunchKernelGGL(fr_axpby_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (Fr *)y, Fr::one(), false, (const Fr *)x, av, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
// y[i] = s*y[i] + a*x[i]
int h2hip_fr_axpby_dev(h2hip_ctx *ctx, void *y, const void *s, const void *a, const void *x, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a && s && (n == 0 || (y && x)), "NULL argument");
    if (!n) return H2HIP_OK;
    Fr av, sv;
    memcpy(&av, a, sizeof(Fr));
    memcpy(&sv, s, sizeof(Fr));
    prof_begin(ctx, "fr_axpby_kernel");
    hipLaunchKernelGGL(fr_axpby_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (Fr *)y, sv, true, (const Fr *)x, av, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
int h2hip_fr_scale_dev(h2hip_ctx *ctx, void *y, const void *s, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && s && (n == 0 || y), "NULL argument");
    if (!n) return H2HIP_OK;
    Fr sv;
    memcpy(&sv, s, sizeof(Fr));
    prof_begin(ctx, "fr_axpby_kernel");
    hipLaunchKernelGGL(fr_axpby_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (Fr *)y, sv, true, (const Fr *)nullptr, Fr::one(), n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// out[i] = sum_j coeffs[j] * polys[j][i]
int h2hip_fr_linear_combination_dev(h2hip_ctx *ctx, void *out, const void *const *polys, const void *coeffs, size_t count, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (n == 0 || out) && (count == 0 || (polys && coeffs)), "NULL argument");
    if (!n) return H2HIP_OK;
    for (size_t j = 0; j < count; ++j) H2_REQUIRE(polys[j], "NULL polynomial");
    if (!count) {
        H2_HIPCHK(hipMemsetAsync(out, 0, sizeof(Fr) * n, ctx->stream));
        return H2HIP_OK;
    }
    const Fr *cs = (const Fr *)coeffs;
    for (size_t j0 = 0; j0 < count; j0 += LINCOMB_MAX) {
        LinCombArgs a;
        a.count = (uint32_t)(count - j0 < LINCOMB_MAX ? count - j0 : LINCOMB_MAX);
        a.accumulate = j0 ? 1u : 0u;
        for (uint32_t t = 0; t < LINCOMB_MAX; ++t) {
            const bool live = t < a.count;
            Fr c;
            if (live) memcpy(&c, cs + j0 + t, sizeof(Fr));
            a.p[t] = live ? (const Fr *)polys[j0 + t] : nullptr;
            a.c[t] = live ? fr29_from_sat(c) : Fr29::zero();
        }
        prof_begin(ctx, "fr_lincomb_kernel");
        hipLaunchKernelGGL(fr_lincomb_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (Fr *)out, a, n);
        prof_end(ctx);
    }
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// ------------------------------------------------------------------ K4 / K5
}  // extern "C"
namespace h2 {
int fr_batch_invert_to(h2hip_ctx *ctx, const Fr *src, Fr *dst, size_t n) {   // dst == src: in place
    if (!n) return H2HIP_OK;
    Fr *scratch = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP1, sizeof(Fr) * n, (void **)&scratch));
    // elements per lane: the inversion (division steps, ~50 products' worth of instructions) is cheap enough to spend one
    // per 8 elements — the dependent chain per lane (3 products per element + the inversion) is what the kernel waits for
    size_t per_lane = (size_t)ctx->fr_invert_run;
    if (per_lane < 1) {   // auto: about 2^16 lanes (measured optimum: 4 at 2^16, 8 at 2^19, 16-32 at 2^21; tools/invert_sweep.py)
        per_lane = n >> 16;
        per_lane = per_lane < 4 ? 4 : per_lane > 32 ? 32 : per_lane;
    }
    size_t lanes = (n + per_lane - 1) / per_lane;
    uint32_t blocks = (uint32_t)((lanes + 255) / 256);
    prof_begin(ctx, "fr_batch_invert_kernel");
    hipLaunchKernelGGL(fr_batch_invert_kernel, dim3(blocks), dim3(256), 0, ctx->stream, src, dst, scratch, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
}  // namespace h2
extern "C" {
int h2hip_fr_batch_invert_dev(h2hip_ctx *ctx, void *a, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (n == 0 || a), "NULL argument");
    return fr_batch_invert_to(ctx, (const Fr *)a, (Fr *)a, n);
}

// inclusive prefix products of `segments` independent runs of n elements, `seg_stride` elements apart (in -> out, same layout)
static int prefix_product_segments(h2hip_ctx *ctx, const Fr *in, Fr *out, size_t n, size_t segments, size_t seg_stride) {
    const uint32_t tile = 256 * SCAN_J;
    uint32_t ntiles = (uint32_t)((n + tile - 1) / tile);
    H2_REQUIRE(segments >= 1 && segments <= 65535, "1..65535 segments");
    Fr *tp = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP2, sizeof(Fr) * ((size_t)ntiles * segments + 1), (void **)&tp));
    prof_begin(ctx, "fr_prefix_prod_kernels");
    hipLaunchKernelGGL(fr_prefix_prod_tile_kernel, dim3(ntiles, (uint32_t)segments), dim3(256), 0, ctx->stream, in, out, tp, n, seg_stride);
    hipLaunchKernelGGL(fr_prefix_prod_sums_kernel, dim3((uint32_t)segments), dim3(1024), 0, ctx->stream, tp, ntiles);
    hipLaunchKernelGGL(fr_prefix_prod_apply_kernel, dim3(ntiles, (uint32_t)segments), dim3(256), 0, ctx->stream, out, (const Fr *)tp, n, seg_stride);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
static int prefix_product_inplace(h2hip_ctx *ctx, const Fr *in, Fr *out, size_t n) { return prefix_product_segments(ctx, in, out, n, 1, 0); }
}  // extern "C"
namespace h2 {
// dst[j][i] = src[j * src_stride + i], i < len, for `count` separately allocated destinations (32 per launch)
int fr_scatter_rows(h2hip_ctx *ctx, Fr *const *dst, size_t count, const Fr *src, size_t src_stride, size_t len) {
    if (!count || !len) return H2HIP_OK;
    prof_begin(ctx, "fr_scatter_rows_kernel");
    for (size_t s0 = 0; s0 < count; s0 += 32) {
        const uint32_t g = (uint32_t)(count - s0 < 32 ? count - s0 : 32);
        RowPtrs rows;
        for (uint32_t j = 0; j < 32; ++j) rows.p[j] = dst[s0 + (j < g ? j : 0)];
        hipLaunchKernelGGL(fr_scatter_rows_kernel, dim3(grid_for(ctx, len), g), dim3(256), 0, ctx->stream, rows, src + s0 * src_stride, src_stride, len);
    }
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
}  // namespace h2
extern "C" {
int h2hip_fr_prefix_product_dev(h2hip_ctx *ctx, void *out, const void *in, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (n == 0 || (out && in)), "NULL argument");
    if (!n) return H2HIP_OK;
    return prefix_product_inplace(ctx, (const Fr *)in, (Fr *)out, n);
}
// z[0] = 1, z[i+1] = z[i] * num[i] / den[i], i < n  (z has n+1 elements; 0 denominators count as 0^-1 := 0)
int h2hip_fr_grand_product_dev(h2hip_ctx *ctx, void *z, const void *num, const void *den, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && z && (n == 0 || (num && den)), "NULL argument");
    Fr *zz = (Fr *)z;
    hipLaunchKernelGGL(fr_set_one_kernel, dim3(1), dim3(64), 0, ctx->stream, zz);
    if (!n) return H2HIP_OK;
    Fr *t = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(Fr) * n, (void **)&t));
    H2_CHK(fr_batch_invert_to(ctx, (const Fr *)den, t, n));
    H2_CHK(binop(ctx, OP_MUL, t, t, num, n));
    return prefix_product_inplace(ctx, t, zz + 1, n);
}

// `segments` grand products of seg_len factors each in a handful of launches: num / den hold the factors of all segments back to back,
// z[s] receives seg_len + 1 values.  chained != 0: z[s][0] = z[s-1][seg_len] (z[0][0] = 1) — the permutation argument's sets; otherwise every
// z[s][0] = 1 — the lookup arguments.  0 denominators count as 0^-1 := 0.
int h2hip_fr_grand_products_dev(h2hip_ctx *ctx, void *const *z, const void *num, const void *den, size_t segments, size_t seg_len, int chained) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (segments == 0 || z) && (segments == 0 || seg_len == 0 || (num && den)), "NULL argument");
    if (!segments) return H2HIP_OK;
    for (size_t s2 = 0; s2 < segments; ++s2) H2_REQUIRE(z[s2], "NULL product column");
    H2_REQUIRE(segments <= 65535 && seg_len < ((size_t)1 << 40), "too many segments");
    const size_t total = segments * seg_len;
    const size_t rlen = chained ? total + 1 : segments * (seg_len + 1);
    Fr *t = nullptr, *r = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(Fr) * (total + 2 * rlen), (void **)&t));
    r = t + total;
    Fr *e = r + rlen;
    if (!total) {
        for (size_t s2 = 0; s2 < segments; ++s2) hipLaunchKernelGGL(fr_set_one_kernel, dim3(1), dim3(64), 0, ctx->stream, (Fr *)z[s2]);
        H2_HIPCHK(hipGetLastError());
        return H2HIP_OK;
    }
    H2_CHK(fr_batch_invert_to(ctx, (const Fr *)den, t, total));   // (out of place: no copy of the denominators first)
    prof_begin(ctx, "fr_ratio_rows_kernel");
    hipLaunchKernelGGL(fr_ratio_rows_kernel, dim3(grid_for(ctx, total)), dim3(256), 0, ctx->stream, r, (const Fr *)num, (const Fr *)t, total, seg_len,
                       chained ? 1 : 0);
    prof_end(ctx);
    if (segments == 1) return prefix_product_segments(ctx, r, (Fr *)z[0], rlen, 1, 0);   // one product: straight into its column
    if (chained) H2_CHK(prefix_product_segments(ctx, r, e, rlen, 1, 0));
    else H2_CHK(prefix_product_segments(ctx, r, e, seg_len + 1, segments, seg_len + 1));
    // chained: consecutive products share one element (the last value of one is the first of the next): rows seg_len apart, seg_len + 1 long
    return fr_scatter_rows(ctx, (Fr *const *)z, segments, e, chained ? seg_len : seg_len + 1, seg_len + 1);
}

// ------------------------------------------------------------------ K7
static void pow_table(const Fr &x, uint32_t j, PowTable &pw) {
    // p[l] = x^(j*2^l) for l = 0..23
    Fr v = fe_pow_u64(x, j);
    for (int l = 0; l < 24; ++l) {
        pw.p[l] = v;
        v = fe_sqr(v);
    }
}
int h2hip_fr_eval_polynomial_dev(h2hip_ctx *ctx, const void *coeffs, size_t n, const void *x, void *out_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && out_host && x && (n == 0 || coeffs), "NULL argument");
    Fr xv;
    memcpy(&xv, x, sizeof(Fr));
    const uint32_t tile = 256 * EVAL_J;
    uint32_t ntiles = (uint32_t)((n + tile - 1) / tile);
    if (!ntiles) ntiles = 1;
    PowTable pw;
    pow_table(xv, EVAL_J, pw);   // p[8] = x^(EVAL_J*256) = x^tile
    Fr *tv = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP2, sizeof(Fr) * (ntiles + 1), (void **)&tv));
    prof_begin(ctx, "fr_eval_kernels");
    hipLaunchKernelGGL(fr_eval_tile_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, (const Fr *)coeffs, n, xv, pw, tv);
    hipLaunchKernelGGL(fr_eval_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const Fr *)tv, ntiles, pw, tv + ntiles);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return sync_results(ctx, out_host, tv + ntiles, sizeof(Fr));
}
int h2hip_fr_eval_polynomial_batch_dev(h2hip_ctx *ctx, const void *const *coeffs_dev, const size_t *lens, const void *points, size_t count,
                                       void *out_host) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (count == 0 || (coeffs_dev && lens && points && out_host)), "NULL argument");
    if (!count) return H2HIP_OK;
    H2_REQUIRE(count <= 4096, "too many evaluations in one batch");
    std::vector<EvalJob> jobs(count);
    std::vector<size_t> distinct;   // jobs holding the first table of each point
    size_t nmax = 0;
    for (size_t j = 0; j < count; ++j) {
        H2_REQUIRE(lens[j] == 0 || coeffs_dev[j], "NULL polynomial");
        jobs[j].coeffs = (const Fr *)coeffs_dev[j];
        jobs[j].n = lens[j];
        memcpy(&jobs[j].x, (const char *)points + sizeof(Fr) * j, sizeof(Fr));
        // a proof asks for hundreds of evaluations at a handful of points (x and its rotations): one power table per distinct point
        size_t seen = j;
        for (size_t t = 0; t < distinct.size(); ++t)
            if (jobs[distinct[t]].x == jobs[j].x) {
                seen = distinct[t];
                break;
            }
        if (seen != j) {
            jobs[j].pw = jobs[seen].pw;
            jobs[j].x29 = jobs[seen].x29;
            jobs[j].one29 = jobs[seen].one29;
            for (int l = 0; l < 8; ++l) jobs[j].pw29[l] = jobs[seen].pw29[l];
        } else {
            pow_table(jobs[j].x, EVAL_J, jobs[j].pw);
            jobs[j].x29 = r29_const(jobs[j].x);
            jobs[j].one29 = r29_const(Fr::one());
            for (int l = 0; l < 8; ++l) jobs[j].pw29[l] = r29_const(jobs[j].pw.p[l]);
            if (distinct.size() < 16) distinct.push_back(j);
        }
        if (lens[j] > nmax) nmax = lens[j];
    }
    const uint32_t tile = 256 * EVAL_J;
    uint32_t ntiles = (uint32_t)((nmax + tile - 1) / tile);
    if (!ntiles) ntiles = 1;
    char *buf = nullptr;
    const size_t jobs_bytes = (sizeof(EvalJob) * count + 255) / 256 * 256;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP2, jobs_bytes + sizeof(Fr) * ((size_t)ntiles * count + count), (void **)&buf));
    EvalJob *djobs = (EvalJob *)buf;
    Fr *tv = (Fr *)(buf + jobs_bytes), *res = tv + (size_t)ntiles * count;
    H2_HIPCHK(hipMemcpyAsync(djobs, jobs.data(), sizeof(EvalJob) * count, hipMemcpyHostToDevice, ctx->stream));
    prof_begin(ctx, "fr_eval_kernels");
    if (ctx->kate_29)
        hipLaunchKernelGGL(fr_eval_tile_batch29_kernel, dim3(ntiles, (uint32_t)count), dim3(256), 0, ctx->stream, (const EvalJob *)djobs, ntiles, tv);
    else
        hipLaunchKernelGGL(fr_eval_tile_batch_kernel, dim3(ntiles, (uint32_t)count), dim3(256), 0, ctx->stream, (const EvalJob *)djobs, ntiles, tv);
    hipLaunchKernelGGL(fr_eval_final_batch_kernel, dim3((uint32_t)count), dim3(256), 0, ctx->stream, (const EvalJob *)djobs, ntiles, (const Fr *)tv, res);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return sync_results(ctx, out_host, res, sizeof(Fr) * count);   // (the wait also keeps `jobs` alive until the upload has been consumed)
}
}  // extern "C"
// q[0..n-1) = sum_j weights[j] * (f(X) - f(points[j])) / (X - points[j]),  m <= 8 points; q_dev must not alias coeffs_dev
// one quotient of the kind above per SET: sets[i] = (coefficients, points, weights, tops or NULL, m, output, add to the output?), all of n coefficients.
// The sets share one job table, one upload and ONE launch of the latency-bound carry kernel (a workgroup per job); heads and apply run per set.
struct KateSet {
    const void *coeffs, *points, *weights, *tops;
    uint32_t m;
    void *q;
    bool add_to_q;
};
template <uint32_t J>
static int kate_division_sets_run(h2hip_ctx *ctx, const KateSet *sets, size_t nsets, size_t n) {
    const uint32_t tile = 256 * J;
    bool any_top = false;
    for (size_t i = 0; i < nsets; ++i) any_top |= sets[i].tops != nullptr;
    const uint32_t ntiles = (uint32_t)((n + (any_top ? 1 : 0) + tile - 1) / tile);   // the virtual coefficient n may open a tile of its own
    // a pass handles up to four points and reads that many jobs: every set's jobs are padded with zero jobs (b = w = top = 0) to a multiple of four
    std::vector<uint32_t> first(nsets);
    uint32_t total = 0;
    for (size_t i = 0; i < nsets; ++i) {
        first[i] = total;
        total += (sets[i].m + 3) / 4 * 4;
    }
    std::vector<KateJob> jobs(total);
    memset((void *)jobs.data(), 0, sizeof(KateJob) * total);
    const bool k29 = ctx->kate_29 != 0;
    std::vector<KateJob29> jobs29(k29 ? total : 0);
    if (k29) memset((void *)jobs29.data(), 0, sizeof(KateJob29) * total);
    for (size_t i = 0; i < nsets; ++i)
        for (uint32_t j = 0; j < sets[i].m; ++j) {
            KateJob &jb = jobs[first[i] + j];
            memcpy(&jb.b, (const char *)sets[i].points + sizeof(Fr) * j, sizeof(Fr));
            memcpy(&jb.w, (const char *)sets[i].weights + sizeof(Fr) * j, sizeof(Fr));
            if (sets[i].tops) memcpy(&jb.top, (const char *)sets[i].tops + sizeof(Fr) * j, sizeof(Fr));
            pow_table(jb.b, J, jb.pw);   // p[l] = b^(J * 2^l): p[8] = b^tile
            if (k29) {   // the same job for the kernels on unsaturated limbs: constants in R' form
                KateJob29 &j9 = jobs29[first[i] + j];
                j9.b = r29_const(jb.b);
                j9.w = r29_const(jb.w);
                j9.one = r29_const(Fr::one());
                j9.top = r29_load(jb.top);
                for (int l = 0; l < 9; ++l) j9.pw[l] = r29_const(jb.pw.p[l]);
            }
        }
    char *buf = nullptr;
    const size_t jobs_bytes = (sizeof(KateJob) * total + 255) / 256 * 256, jobs29_bytes = (sizeof(KateJob29) * jobs29.size() + 255) / 256 * 256;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP2, jobs_bytes + jobs29_bytes + sizeof(Fr) * 2 * (size_t)total * (ntiles + 1), (void **)&buf));
    KateJob *djobs = (KateJob *)buf;
    KateJob29 *djobs29 = (KateJob29 *)(buf + jobs_bytes);
    Fr *heads = (Fr *)(buf + jobs_bytes + jobs29_bytes), *carry = heads + (size_t)total * (ntiles + 1);
    H2_CHK(upload_jobs(ctx, djobs, jobs.data(), sizeof(KateJob) * total));   // through the pinned ring: no synchronisation per call
    if (k29) H2_CHK(upload_jobs(ctx, djobs29, jobs29.data(), sizeof(KateJob29) * total));
    prof_begin(ctx, "fr_kate_kernels");
    if (nsets > 1) H2_HIPCHK(hipMemsetAsync(heads, 0, sizeof(Fr) * (size_t)total * (ntiles + 1), ctx->stream));   // (the padding jobs' rows: the carry launch reads them)
    for (size_t i = 0; i < nsets; ++i) {
        const uint32_t g0 = first[i];
        Fr *hd = heads + (size_t)g0 * (ntiles + 1);
        if (k29)
            hipLaunchKernelGGL(fr_kate_heads_multi29_kernel<J>, dim3(ntiles, sets[i].m), dim3(256), 0, ctx->stream, (const Fr *)sets[i].coeffs, n,
                               (const KateJob29 *)(djobs29 + g0), ntiles, hd, sets[i].tops ? 1 : 0);
        else
            hipLaunchKernelGGL(fr_kate_heads_multi_kernel<J>, dim3(ntiles, sets[i].m), dim3(256), 0, ctx->stream, (const Fr *)sets[i].coeffs, n,
                               (const KateJob *)(djobs + g0), ntiles, hd);
    }
    hipLaunchKernelGGL(fr_kate_carry_multi_kernel, dim3(nsets > 1 ? total : sets[0].m), dim3(256), 0, ctx->stream, (const Fr *)heads, carry, ntiles,
                       (const KateJob *)djobs);
    for (size_t i = 0; i < nsets; ++i) {
        const uint32_t m = sets[i].m;
        const int with_top = sets[i].tops != nullptr;
        for (uint32_t j0 = 0; j0 < m; j0 += 4) {   // four points per pass (the scans of a pass share the workgroup's LDS); halo2-base's sets stop at 4
            const uint32_t mm = m - j0 < 4 ? m - j0 : 4, g0 = first[i] + j0;
            const KateJob *jb = djobs + g0;
            const Fr *cr = carry + (size_t)g0 * (ntiles + 1);
            const int accumulate = (j0 || sets[i].add_to_q) ? 1 : 0;
            const Fr *cf = (const Fr *)sets[i].coeffs;
            Fr *qo = (Fr *)sets[i].q;
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), 0, ctx->stream, cf, n, jb, mm, ntiles, cr, qo, accumulate); };
            const KateJob29 *jb29 = djobs29 + g0;
            auto go29 = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), 0, ctx->stream, cf, n, jb29, mm, ntiles, cr, qo, accumulate); };
            if (k29 && with_top) {
                if (mm == 1) go29(fr_kate_apply_multi29_kernel<1, J, true>);
                else if (mm == 2) go29(fr_kate_apply_multi29_kernel<2, J, true>);
                else go29(fr_kate_apply_multi29_kernel<4, J, true>);
            } else if (k29) {
                if (mm == 1) go29(fr_kate_apply_multi29_kernel<1, J, false>);
                else if (mm == 2) go29(fr_kate_apply_multi29_kernel<2, J, false>);
                else go29(fr_kate_apply_multi29_kernel<4, J, false>);
            } else if (with_top) {
                if (mm == 1) go(fr_kate_apply_multi_kernel<1, J, true>);
                else if (mm == 2) go(fr_kate_apply_multi_kernel<2, J, true>);
                else go(fr_kate_apply_multi_kernel<4, J, true>);
            } else {
                if (mm == 1) go(fr_kate_apply_multi_kernel<1, J, false>);
                else if (mm == 2) go(fr_kate_apply_multi_kernel<2, J, false>);
                else go(fr_kate_apply_multi_kernel<4, J, false>);
            }
        }
    }
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
static int kate_division_sets_pick(h2hip_ctx *ctx, const KateSet *sets, size_t nsets, size_t n) {
    uint32_t j = ctx->kate_coeffs_per_lane;
    if (j != 1 && j != 2 && j != 4 && j != 8) j = n >= ((size_t)1 << 20) ? 8 : n >= ((size_t)1 << 18) ? 4 : n >= ((size_t)1 << 17) ? 2 : 1;   // (2^19: 4 and 8 within noise, 4 ahead by 0.04 ms per proof; 2^21: 8 ahead by 0.4 ms — profiles/archive/r04_kate_tile_ab.log)
    if (j == 8) return kate_division_sets_run<8>(ctx, sets, nsets, n);
    if (j == 4) return kate_division_sets_run<4>(ctx, sets, nsets, n);
    if (j == 2) return kate_division_sets_run<2>(ctx, sets, nsets, n);
    return kate_division_sets_run<1>(ctx, sets, nsets, n);
}
// coefficients per lane: a tile is 256 * J coefficients; about one wave per SIMD or more (ctx->kate_coeffs_per_lane overrides: 1, 2, 4, 8)
static int kate_division_multi_pick(h2hip_ctx *ctx, void *q, const void *coeffs, size_t n, const void *points, const void *weights, uint32_t m, const void *tops,
                                    bool add_to_q = false) {
    const KateSet one = {coeffs, points, weights, tops, m, q, add_to_q};
    return kate_division_sets_pick(ctx, &one, 1, n);
}
extern "C" {
int h2hip_fr_kate_division_multi_dev(h2hip_ctx *ctx, void *q, const void *coeffs, size_t n, const void *points, const void *weights, uint32_t m) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && points && weights && n >= 1 && coeffs && (n == 1 || q) && m >= 1 && m <= 8, "bad argument (1..8 points)");
    H2_REQUIRE(q != coeffs, "q must not alias coeffs");
    if (n == 1) return H2HIP_OK;
    return kate_division_multi_pick(ctx, q, coeffs, n, points, weights, m, nullptr);
}
// q[0..n-1) += the same sum: SHPLONK adds the rotation sets' quotients up with weights v^i — folded into weights[], the sum lands in its
// accumulator without a pass of its own
int h2hip_fr_kate_division_multi_acc_dev(h2hip_ctx *ctx, void *q, const void *coeffs, size_t n, const void *points, const void *weights, uint32_t m) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && points && weights && n >= 1 && coeffs && (n == 1 || q) && m >= 1 && m <= 8, "bad argument (1..8 points)");
    H2_REQUIRE(q != coeffs, "q must not alias coeffs");
    if (n == 1) return H2HIP_OK;
    return kate_division_multi_pick(ctx, q, coeffs, n, points, weights, m, nullptr, true);
}
// q[0..n-1) (+)= sum over `nsets` polynomials of that sum: coeffs_dev[i] with set_sizes[i] points / weights taken from the flat arrays in order (every
// set 1..8 points, all polynomials of n coefficients).  SHPLONK's whole v-weighted sum over the rotation sets in one call: one job table, one upload,
// ONE launch of the latency-bound carry kernel for all (set, point) pairs.  accumulate = 0: q is overwritten (by the first set).
int h2hip_fr_kate_division_sets_dev(h2hip_ctx *ctx, void *q, const void *const *coeffs, size_t n, const void *points, const void *weights,
                                    const uint32_t *set_sizes, size_t nsets, int accumulate) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && n >= 1 && (nsets == 0 || (coeffs && points && weights && set_sizes)) && (n == 1 || q) && nsets <= 64, "bad argument");
    if (n == 1 || !nsets) return H2HIP_OK;
    std::vector<KateSet> sets(nsets);
    size_t off = 0;
    for (size_t i = 0; i < nsets; ++i) {
        H2_REQUIRE(coeffs[i] && coeffs[i] != q && set_sizes[i] >= 1 && set_sizes[i] <= 8, "bad set (1..8 points, q must not alias a polynomial)");
        sets[i] = {coeffs[i], (const char *)points + sizeof(Fr) * off, (const char *)weights + sizeof(Fr) * off, nullptr, set_sizes[i], q, accumulate != 0 || i > 0};
        off += set_sizes[i];
    }
    return kate_division_sets_pick(ctx, sets.data(), nsets, n);
}
// q[0..n-1) = (f(X) - f(b)) / (X - b)   [UPSTREAM arithmetic::kate_division]: the one-point case of the kernels above (weight 1)
int h2hip_fr_kate_division_dev(h2hip_ctx *ctx, void *q, const void *coeffs, size_t n, const void *b) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && b && n >= 1 && coeffs && (n == 1 || q), "bad argument");
    H2_REQUIRE(q != coeffs, "q must not alias coeffs");
    if (n == 1) return H2HIP_OK;
    const Fr one = Fr::one();
    return kate_division_multi_pick(ctx, q, coeffs, n, b, &one, 1, nullptr);
}
// The same division for ONE COEFFICIENT RANGE [lo, lo + n) of f (the multi-GPU prover: a rank holds the range of its SRS slice): coeffs_dev = that
// range, carries[j] = sum_{i >= lo + n} f_i points[j]^(i - lo - n) — what the ranges above contribute, assembled by the caller from the ranks'
// partial evaluations (zero for the top range) — and q_dev[0..n) = the quotient's coefficients lo .. lo + n - 1 (n values, one more than the
// whole-polynomial call writes: the quotient coefficient lo + n - 1 is the carry itself).
int h2hip_fr_kate_division_range_dev(h2hip_ctx *ctx, void *q, const void *coeffs, size_t n, const void *points, const void *weights, const void *carries,
                                     uint32_t m) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && points && weights && carries && n >= 1 && coeffs && q && m >= 1 && m <= 8, "bad argument (1..8 points)");
    H2_REQUIRE(q != coeffs, "q must not alias coeffs");
    return kate_division_multi_pick(ctx, q, coeffs, n, points, weights, m, carries);
}

// ------------------------------------------------------------------ K8 Poseidon
int h2hip_poseidon_set_spec(h2hip_ctx *ctx, uint32_t t, uint32_t r_f, uint32_t r_p, const void *round_constants, const void *mds) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && round_constants && mds, "NULL argument");
    H2_REQUIRE(t == 3 || t == 5, "state width t must be 3 or 5");
    H2_REQUIRE(r_f >= 2 && (r_f % 2) == 0 && r_f <= 16 && r_p <= 256, "round numbers out of range");
    size_t nrc = (size_t)(r_f + r_p) * t, nm = (size_t)t * t;
    Fr *buf = nullptr;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_POSEIDON, sizeof(Fr) * (nrc + nm), (void **)&buf));
    // the kernel multiplies in the unsaturated R' = 2^261 domain: store c * 2^261 (packed 8 x 32 bit) once here
    std::vector<Fr> conv(nrc + nm);
    for (size_t j = 0; j < nrc + nm; ++j) {
        Fr c;
        memcpy(&c, (const char *)(j < nrc ? round_constants : mds) + sizeof(Fr) * (j < nrc ? j : j - nrc), sizeof(Fr));
        conv[j] = f29_pack_canonical<FrP>(fr29_from_sat(c));
    }
    H2_HIPCHK(hipMemcpyAsync(buf, conv.data(), sizeof(Fr) * (nrc + nm), hipMemcpyHostToDevice, ctx->stream));
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->pos_t = t;
    ctx->pos_rf = r_f;
    ctx->pos_rp = r_p;
    return H2HIP_OK;
}
int h2hip_poseidon_permute_batch_dev(h2hip_ctx *ctx, void *states, const void *inputs, uint32_t num_inputs, size_t n) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && (n == 0 || states), "NULL argument");
    H2_REQUIRE(ctx->pos_t != 0, "call h2hip_poseidon_set_spec first");
    H2_REQUIRE(num_inputs < ctx->pos_t, "num_inputs must be <= RATE = t-1");
    H2_REQUIRE(num_inputs == 0 || inputs || n == 0, "inputs is NULL");
    if (!n) return H2HIP_OK;
    const Fr *rc = (const Fr *)ctx->ws[h2hip_ctx::WS_POSEIDON].p;
    const Fr *mds = rc + (size_t)(ctx->pos_rf + ctx->pos_rp) * ctx->pos_t;
    dim3 g((uint32_t)((n + 255) / 256)), blk(256);
    prof_begin(ctx, "poseidon_permute_kernel");
    if (ctx->pos_t == 3)
        hipLaunchKernelGGL(poseidon_permute_kernel<3>, g, blk, 0, ctx->stream, (Fr *)states, (const Fr *)inputs, num_inputs, n, rc, mds, ctx->pos_rf,
                           ctx->pos_rp);
    else
        hipLaunchKernelGGL(poseidon_permute_kernel<5>, g, blk, 0, ctx->stream, (Fr *)states, (const Fr *)inputs, num_inputs, n, rc, mds, ctx->pos_rf,
                           ctx->pos_rp);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// ------------------------------------------------------------------ K6 (halo2-base gate term)
int h2hip_quotient_flex_gate_dev(h2hip_ctx *ctx, void *acc, const void *q, const void *a, uint32_t ext_k, uint32_t k, const void *y) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && acc && q && a && y, "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    H2_REQUIRE(acc != a && acc != q, "acc must not alias an input");
    Fr yv;
    memcpy(&yv, y, sizeof(Fr));
    size_t n_ext = (size_t)1 << ext_k;
    prof_begin(ctx, "quotient_flex_gate_kernel");
    hipLaunchKernelGGL(quotient_flex_gate_kernel, dim3(grid_for(ctx, n_ext)), dim3(256), 0, ctx->stream, (Fr *)acc, (const Fr *)q, (const Fr *)a, n_ext,
                       1u << (ext_k - k), yv);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// EvaluationDomain::divide_by_vanishing_poly [UPSTREAM poly/domain.rs, SURVEY.md A.2]: t(X) = X^n - 1 takes only
// L = 2^(ext_k-k) distinct values on the coset {zeta * ext_omega^i}: t_i = zeta^n * (ext_omega^n)^i - 1, period L.
__global__ __launch_bounds__(64) void vanishing_inverses_kernel(Fr *__restrict__ tinv, uint32_t L, Fr zeta_n, Fr step) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    Fr t = fe_sub(fe_mul(zeta_n, fe_pow_u64(step, i)), Fr::one());
    tinv[i] = fe_inv(t);   // t != 0: the coset avoids the n-th roots of unity
}
__global__ __launch_bounds__(256) void divide_by_vanishing_kernel(Fr *__restrict__ a, const Fr *__restrict__ tinv, size_t n_ext, uint32_t mask) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ext; i += stride) a[i] = fe_mul(a[i], tinv[i & mask]);
}
struct VanishSmall {
    Fr v[8];
};
__global__ __launch_bounds__(256) void divide_by_vanishing_small_kernel(Fr *__restrict__ a, VanishSmall t, size_t n_ext, uint32_t mask) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ext; i += stride) {
        Fr f = t.v[0];
#pragma unroll
        for (uint32_t k = 1; k < 8; ++k)
            if ((i & mask) == k) f = t.v[k];
        a[i] = fe_mul(a[i], f);
    }
}
int h2hip_divide_by_vanishing_poly_dev(h2hip_ctx *ctx, void *a, uint32_t ext_k, uint32_t k, const void *ext_omega, const void *zeta) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && a && ext_omega && zeta, "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28 && ext_k - k <= 16, "need k <= ext_k <= 28 and ext_k - k <= 16");
    Fr w, z;
    memcpy(&w, ext_omega, sizeof(Fr));
    memcpy(&z, zeta, sizeof(Fr));
    const uint64_t n = 1ull << k;
    const uint32_t L = 1u << (ext_k - k);
    Fr *tinv;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_VANISH, sizeof(Fr) * L, (void **)&tinv));
    prof_begin(ctx, "divide_by_vanishing_kernels");
    const Fr zn = fe_pow_u64(z, n), wn = fe_pow_u64(w, n);
    size_t n_ext = (size_t)1 << ext_k;
    if (L <= 8) {   // the usual case (ext_k - k <= 3): the few inverses are computed on the host and travel as kernel arguments
        VanishSmall t;
        Fr cur = zn;
        for (uint32_t i = 0; i < 8; ++i) {
            t.v[i] = i < L ? fe_inv(fe_sub(cur, Fr::one())) : Fr::zero();
            cur = fe_mul(cur, wn);
        }
        hipLaunchKernelGGL(divide_by_vanishing_small_kernel, dim3(grid_for(ctx, n_ext)), dim3(256), 0, ctx->stream, (Fr *)a, t, n_ext, L - 1);
    } else {
        hipLaunchKernelGGL(vanishing_inverses_kernel, dim3((L + 63) / 64), dim3(64), 0, ctx->stream, tinv, L, zn, wn);
        hipLaunchKernelGGL(divide_by_vanishing_kernel, dim3(grid_for(ctx, n_ext)), dim3(256), 0, ctx->stream, (Fr *)a, (const Fr *)tinv, n_ext, L - 1);
    }
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// workgroups of the permutation-identity kernels: 8 points per lane from 2^21 extended points, 4 / 2 / 1 for 2^20 / 2^19 / smaller domains
static uint32_t perm_grid(size_t ne) {
    size_t per_lane = ne >> 18;
    per_lane = per_lane < 1 ? 1 : per_lane > 8 ? 8 : per_lane;
    const size_t g = (ne / per_lane + 255) / 256;
    return g < 1 ? 1u : (uint32_t)g;
}
static Fr ld_fr(const void *p) {
    Fr r;
    memcpy(&r, p, sizeof(Fr));
    return r;
}
int h2hip_quotient_lookup_dev(h2hip_ctx *ctx, void *acc, const void *z, const void *a, const void *s, const void *a_perm, const void *s_perm,
                              const void *l0, const void *l_last, const void *l_blind, uint32_t ext_k, uint32_t k, const void *beta,
                              const void *gamma, const void *y) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && acc && z && a && s && a_perm && s_perm && l0 && l_last && l_blind && beta && gamma && y, "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    LookupArgs g;
    g.z = (const Fr *)z; g.a = (const Fr *)a; g.s = (const Fr *)s; g.ap = (const Fr *)a_perm; g.sp = (const Fr *)s_perm;
    g.l0 = (const Fr *)l0; g.l_last = (const Fr *)l_last; g.l_blind = (const Fr *)l_blind;
    g.beta = ld_fr(beta); g.gamma = ld_fr(gamma); g.y = ld_fr(y);
    size_t ne = (size_t)1 << ext_k;
    prof_begin(ctx, "quotient_lookup_kernel");
    hipLaunchKernelGGL(quotient_lookup_kernel, dim3(grid_for(ctx, ne)), dim3(256), 0, ctx->stream, (Fr *)acc, g, ne, 1u << (ext_k - k));
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
int h2hip_quotient_permutation_set_dev(h2hip_ctx *ctx, void *acc, const void *z, const void *z_prev, const void *const *cols,
                                       const void *const *sigmas, uint32_t ncols, uint32_t first_col_index, const void *l0, const void *l_last,
                                       const void *l_blind, uint32_t ext_k, uint32_t k, uint32_t terms, int32_t last_rotation,
                                       const void *beta, const void *gamma, const void *delta, const void *zeta, const void *ext_omega, const void *y) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && acc && z && l0 && l_last && l_blind && beta && gamma && delta && zeta && ext_omega && y, "NULL argument");
    H2_REQUIRE(terms != 0 && (terms & ~15u) == 0, "terms must be a non-empty mask of H2HIP_PERM_*");
    H2_REQUIRE(!(terms & H2HIP_PERM_CHAIN) || z_prev, "H2HIP_PERM_CHAIN needs z_prev_dev");
    if (!(terms & H2HIP_PERM_PRODUCT)) ncols = 0;
    H2_REQUIRE(!(terms & H2HIP_PERM_PRODUCT) || (cols && sigmas && ncols >= 1 && ncols <= PERM_MAX_COLS), "1..8 columns per permutation set");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    PermArgs g;
    memset(&g, 0, sizeof(g));
    g.z = (const Fr *)z; g.z_prev = (const Fr *)z_prev; g.l0 = (const Fr *)l0; g.l_last = (const Fr *)l_last; g.l_blind = (const Fr *)l_blind;
    for (uint32_t j = 0; j < ncols; ++j) {
        H2_REQUIRE(cols[j] && sigmas[j], "NULL column");
        g.cols[j] = (const Fr *)cols[j];
        g.sigmas[j] = (const Fr *)sigmas[j];
    }
    g.ncols = ncols; g.terms = terms;
    const size_t ne = (size_t)1 << ext_k;
    const uint32_t step = 1u << (ext_k - k);
    const int64_t n = (int64_t)1 << k;
    int64_t rot = ((int64_t)last_rotation % n + n) % n;
    g.last_rot_points = (uint32_t)(((uint64_t)rot * step) & (ne - 1));
    g.beta = ld_fr(beta); g.gamma = ld_fr(gamma); g.delta = ld_fr(delta); g.y = ld_fr(y); g.ext_omega = ld_fr(ext_omega);
    g.x0_delta = fe_mul(fe_mul(g.beta, ld_fr(zeta)), fe_pow_u64(g.delta, first_col_index));
    // 8 extended points per lane from 2^21 points (fewer below: small domains need the lanes): the per-lane start-up (ext_omega^i0,
    // ~28 products) is amortised, the stride power is one host-side exponentiation
    const uint32_t pgrid = perm_grid(ne);
    g.xstep = fe_pow_u64(g.ext_omega, (uint64_t)pgrid * 256);
    OmegaTable pw = {nullptr, nullptr, 0};
    if (ctx->quotient_29) H2_CHK(ntt_pow_table(ctx, ext_k, g.ext_omega, &pw));   // (before the bracket: a new table launches its own profiled kernel)
    prof_begin(ctx, "quotient_permutation_kernel");
    if (ctx->quotient_29) {
        PermArgs29 h;
        memset((void *)&h, 0, sizeof(h));
        h.z = g.z; h.z_prev = g.z_prev; h.l0 = g.l0; h.l_last = g.l_last; h.l_blind = g.l_blind;
        for (uint32_t j = 0; j < ncols; ++j) {
            h.cols[j] = g.cols[j];
            h.sigmas[j] = g.sigmas[j];
        }
        h.ncols = g.ncols; h.terms = g.terms; h.last_rot_points = g.last_rot_points;
        h.beta32 = r29_const(fe_x32(g.beta));
        h.delta = r29_const(g.delta);
        h.y = r29_const(g.y);
        h.x0_delta32 = r29_load(fe_x32(g.x0_delta));
        h.xstep = r29_const(g.xstep);
        h.gamma32 = r29_load(fe_x32(g.gamma));
        h.pw = pw;
        hipLaunchKernelGGL(quotient_permutation29_kernel, dim3(pgrid), dim3(256), 0, ctx->stream, (Fr *)acc, h, ne, step);
    } else {
        hipLaunchKernelGGL(quotient_permutation_kernel, dim3(pgrid), dim3(256), 0, ctx->stream, (Fr *)acc, g, ne, step);
    }
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

// ---- batched forms: all gate columns / all permutation sets / all lookups of a proof
int h2hip_quotient_flex_gate_batch_dev(h2hip_ctx *ctx, void *acc, const void *const *q, const void *const *a, size_t count, uint32_t ext_k, uint32_t k,
                                       const void *y) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && acc && y && (count == 0 || (q && a)), "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    const size_t n_ext = (size_t)1 << ext_k;
    for (size_t j0 = 0; j0 < count; j0 += GATE_BATCH) {
        GateBatchArgs g;
        memset(&g, 0, sizeof(g));
        g.count = (uint32_t)(count - j0 < GATE_BATCH ? count - j0 : GATE_BATCH);
        g.y = ld_fr(y);
        for (uint32_t j = 0; j < g.count; ++j) {
            H2_REQUIRE(q[j0 + j] && a[j0 + j], "NULL column");
            g.q[j] = (const Fr *)q[j0 + j];
            g.a[j] = (const Fr *)a[j0 + j];
        }
        prof_begin(ctx, "quotient_flex_gate_batch_kernel");
        if (ctx->quotient_29)
            hipLaunchKernelGGL(quotient_flex_gate_batch29_kernel, dim3(grid_for(ctx, n_ext)), dim3(256), 0, ctx->stream, (Fr *)acc, g, r29_const(g.y), n_ext,
                               1u << (ext_k - k));
        else
            hipLaunchKernelGGL(quotient_flex_gate_batch_kernel, dim3(grid_for(ctx, n_ext)), dim3(256), 0, ctx->stream, (Fr *)acc, g, n_ext, 1u << (ext_k - k));
        prof_end(ctx);
    }
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
int h2hip_quotient_lookups_dev(h2hip_ctx *ctx, void *acc, const void *const *z, const void *const *a, const void *const *s, const void *const *a_perm,
                               const void *const *s_perm, size_t count, const void *l0, const void *l_last, const void *l_blind, uint32_t ext_k, uint32_t k,
                               const void *beta, const void *gamma, const void *y) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && acc && l0 && l_last && l_blind && beta && gamma && y && (count == 0 || (z && a && s && a_perm && s_perm)), "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    const size_t ne = (size_t)1 << ext_k;
    for (size_t j0 = 0; j0 < count; j0 += LOOKUP_BATCH) {
        LookupBatchArgs g;
        memset(&g, 0, sizeof(g));
        g.l0 = (const Fr *)l0; g.l_last = (const Fr *)l_last; g.l_blind = (const Fr *)l_blind;
        g.beta = ld_fr(beta); g.gamma = ld_fr(gamma); g.y = ld_fr(y);
        g.count = (uint32_t)(count - j0 < LOOKUP_BATCH ? count - j0 : LOOKUP_BATCH);
        for (uint32_t j = 0; j < g.count; ++j) {
            const size_t t = j0 + j;
            H2_REQUIRE(z[t] && a[t] && s[t] && a_perm[t] && s_perm[t], "NULL column");
            g.jobs[j].z = (const Fr *)z[t]; g.jobs[j].a = (const Fr *)a[t]; g.jobs[j].s = (const Fr *)s[t];
            g.jobs[j].ap = (const Fr *)a_perm[t]; g.jobs[j].sp = (const Fr *)s_perm[t];
        }
        prof_begin(ctx, "quotient_lookup_batch_kernel");
        if (ctx->quotient_29) {
            LookupConsts29 k29;
            k29.y = r29_const(g.y);
            k29.beta32 = r29_load(fe_x32(g.beta));
            k29.gamma32 = r29_load(fe_x32(g.gamma));
            hipLaunchKernelGGL(quotient_lookup_batch29_kernel, dim3(grid_for(ctx, ne)), dim3(256), 0, ctx->stream, (Fr *)acc, g, k29, ne, 1u << (ext_k - k));
        } else {
            hipLaunchKernelGGL(quotient_lookup_batch_kernel, dim3(grid_for(ctx, ne)), dim3(256), 0, ctx->stream, (Fr *)acc, g, ne, 1u << (ext_k - k));
        }
        prof_end(ctx);
    }
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}
// the whole permutation argument in evaluate_h's order: FIRST (set 0), LAST (last set), CHAIN (sets 1..), PRODUCT (all sets)
int h2hip_quotient_permutation_sets_dev(h2hip_ctx *ctx, void *acc, const void *const *z, uint32_t num_sets, const void *const *cols, const void *const *sigmas,
                                        uint32_t num_columns, uint32_t chunk_len, const void *l0, const void *l_last, const void *l_blind, uint32_t ext_k,
                                        uint32_t k, int32_t last_rotation, const void *beta, const void *gamma, const void *delta, const void *zeta,
                                        const void *ext_omega, const void *y) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && acc && l0 && l_last && l_blind && beta && gamma && delta && zeta && ext_omega && y, "NULL argument");
    H2_REQUIRE(k <= ext_k && ext_k <= 28, "need k <= ext_k <= 28");
    if (!num_sets) return H2HIP_OK;
    H2_REQUIRE(z && cols && sigmas, "NULL argument");
    H2_REQUIRE(chunk_len >= 1 && chunk_len <= PERM_MAX_COLS, "1..8 columns per permutation set");
    H2_REQUIRE(num_columns > (uint64_t)(num_sets - 1) * chunk_len && num_columns <= (uint64_t)num_sets * chunk_len, "num_sets must be ceil(num_columns / chunk_len)");
    for (uint32_t s2 = 0; s2 < num_sets; ++s2) H2_REQUIRE(z[s2], "NULL product column");
    for (uint32_t c = 0; c < num_columns; ++c) H2_REQUIRE(cols[c] && sigmas[c], "NULL column");
    if (num_sets == 1)   // one set: the dedicated kernel (no job loop, the X term without the per-job product)
        return h2hip_quotient_permutation_set_dev(ctx, acc, z[0], nullptr, cols, sigmas, num_columns, 0, l0, l_last, l_blind, ext_k, k,
                                                  H2HIP_PERM_FIRST | H2HIP_PERM_LAST | H2HIP_PERM_PRODUCT, last_rotation, beta, gamma, delta, zeta, ext_omega, y);
    const size_t ne = (size_t)1 << ext_k;
    const uint32_t step = 1u << (ext_k - k);
    PermBatchArgs g;
    memset(&g, 0, sizeof(g));
    g.l0 = (const Fr *)l0; g.l_last = (const Fr *)l_last; g.l_blind = (const Fr *)l_blind;
    g.beta = ld_fr(beta); g.gamma = ld_fr(gamma); g.delta = ld_fr(delta); g.y = ld_fr(y); g.ext_omega = ld_fr(ext_omega);
    const int64_t n = (int64_t)1 << k;
    const int64_t rot = ((int64_t)last_rotation % n + n) % n;
    g.last_rot_points = (uint32_t)(((uint64_t)rot * step) & (ne - 1));
    const uint32_t pgrid = perm_grid(ne);
    g.xstep = fe_pow_u64(g.ext_omega, (uint64_t)pgrid * 256);
    const Fr beta_zeta = fe_mul(g.beta, ld_fr(zeta));
    // the job list in upstream's order
    struct Item {
        uint32_t set, terms;
    };
    std::vector<Item> items;
    if (num_sets == 1) {
        items.push_back({0, H2HIP_PERM_FIRST | H2HIP_PERM_LAST | H2HIP_PERM_PRODUCT});
    } else {
        items.push_back({0, H2HIP_PERM_FIRST});
        items.push_back({num_sets - 1, H2HIP_PERM_LAST});
        for (uint32_t s2 = 1; s2 < num_sets; ++s2) items.push_back({s2, H2HIP_PERM_CHAIN});
        for (uint32_t s2 = 0; s2 < num_sets; ++s2) items.push_back({s2, H2HIP_PERM_PRODUCT});
    }
    Fr dpow = Fr::one();   // delta^(first column of set s), kept per set
    std::vector<Fr> set_x0(num_sets);
    for (uint32_t s2 = 0; s2 < num_sets; ++s2) {
        set_x0[s2] = fe_mul(beta_zeta, dpow);
        for (uint32_t c = 0; c < chunk_len; ++c) dpow = fe_mul(dpow, g.delta);
    }
    OmegaTable pw = {nullptr, nullptr, 0};
    if (ctx->quotient_29) H2_CHK(ntt_pow_table(ctx, ext_k, g.ext_omega, &pw));
    for (size_t j0 = 0; j0 < items.size(); j0 += PERM_BATCH) {
        g.njobs = (uint32_t)(items.size() - j0 < PERM_BATCH ? items.size() - j0 : PERM_BATCH);
        for (uint32_t j = 0; j < g.njobs; ++j) {
            const Item &it = items[j0 + j];
            PermJob &q = g.jobs[j];
            memset(&q, 0, sizeof(q));
            q.z = (const Fr *)z[it.set];
            q.z_prev = it.set ? (const Fr *)z[it.set - 1] : nullptr;
            q.terms = it.terms;
            q.x0_delta = set_x0[it.set];
            if (it.terms & H2HIP_PERM_PRODUCT) {
                const uint32_t c0 = it.set * chunk_len, c1 = c0 + chunk_len < num_columns ? c0 + chunk_len : num_columns;
                q.ncols = c1 - c0;
                for (uint32_t c = c0; c < c1; ++c) {
                    q.cols[c - c0] = (const Fr *)cols[c];
                    q.sigmas[c - c0] = (const Fr *)sigmas[c];
                }
            }
        }
        prof_begin(ctx, "quotient_permutation_batch_kernel");
        if (ctx->quotient_29) {
            PermConsts29 k29;
            k29.beta32 = r29_const(fe_x32(g.beta));
            k29.delta = r29_const(g.delta);
            k29.y = r29_const(g.y);
            k29.xstep = r29_const(g.xstep);
            k29.gamma32 = r29_load(fe_x32(g.gamma));
            for (uint32_t j = 0; j < PERM_BATCH; ++j) k29.x0_delta32[j] = j < g.njobs ? r29_load(fe_x32(g.jobs[j].x0_delta)) : Fr29::zero();
            k29.pw = pw;
            hipLaunchKernelGGL(quotient_permutation_batch29_kernel, dim3(pgrid), dim3(256), 0, ctx->stream, (Fr *)acc, g, k29, ne, step);
        } else {
            hipLaunchKernelGGL(quotient_permutation_batch_kernel, dim3(pgrid), dim3(256), 0, ctx->stream, (Fr *)acc, g, ne, step);
        }
        prof_end(ctx);
    }
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

int h2hip_bench_modmul29(h2hip_ctx *ctx, uint32_t blocks, uint32_t iters, uint32_t chains, double *elapsed_ms, double *modmuls) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && elapsed_ms && modmuls && blocks && iters, "bad argument");
    H2_REQUIRE(chains == 1 || chains == 2 || (chains >= 16 && chains <= 19), "chains must be 1 or 2 (16..19: the NTT round probe, mode = chains - 16)");
    Fr *buf = nullptr;
    size_t lanes = (size_t)blocks * 256;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(Fr) * lanes, (void **)&buf));
    H2_HIPCHK(hipMemsetAsync(buf, 0x5a, sizeof(Fr) * lanes, ctx->stream));
    const size_t probe_lds = sizeof(ProbeElem) * (1024 + 128 + 8);
    for (int rep = 0; rep < 2; ++rep) {   // rep 0 = warm-up
        H2_CHK(h2hip_timer_start(ctx));
        if (chains == 1) hipLaunchKernelGGL(modmul29_bench_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, buf, iters);
        if (chains == 2) hipLaunchKernelGGL(modmul29_bench_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, buf, iters);
        if (chains == 16) hipLaunchKernelGGL(ntt_round_probe_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, buf, iters);
        if (chains == 17) hipLaunchKernelGGL(ntt_round_probe_kernel<1>, dim3(blocks), dim3(256), probe_lds, ctx->stream, buf, iters);
        if (chains == 18) hipLaunchKernelGGL(ntt_round_probe_kernel<2>, dim3(blocks), dim3(256), probe_lds, ctx->stream, buf, iters);
        if (chains == 19) hipLaunchKernelGGL(ntt_round_probe_kernel<3>, dim3(blocks), dim3(256), probe_lds, ctx->stream, buf, iters);
        H2_HIPCHK(hipGetLastError());
        H2_CHK(h2hip_timer_stop(ctx, elapsed_ms));
    }
    *modmuls = (double)lanes * iters * (chains >= 16 ? 4 : chains);
    return H2HIP_OK;
}

int h2hip_bench_modmul(h2hip_ctx *ctx, uint32_t blocks, uint32_t iters, uint32_t chains, double *elapsed_ms, double *modmuls) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && elapsed_ms && modmuls && blocks && iters, "bad argument");
    H2_REQUIRE(chains == 1 || chains == 2 || chains == 4, "chains must be 1, 2 or 4");
    Fr *buf = nullptr;
    size_t lanes = (size_t)blocks * 256;
    H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_TMP0, sizeof(Fr) * lanes, (void **)&buf));
    H2_HIPCHK(hipMemsetAsync(buf, 0x5a, sizeof(Fr) * lanes, ctx->stream));
    for (int rep = 0; rep < 2; ++rep) {   // rep 0 = warm-up
        H2_CHK(h2hip_timer_start(ctx));
        if (chains == 1) hipLaunchKernelGGL(modmul_bench_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, buf, iters);
        if (chains == 2) hipLaunchKernelGGL(modmul_bench_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, buf, iters);
        if (chains == 4) hipLaunchKernelGGL(modmul_bench_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, buf, iters);
        H2_HIPCHK(hipGetLastError());
        H2_CHK(h2hip_timer_stop(ctx, elapsed_ms));
    }
    *modmuls = (double)lanes * iters * chains;
    return H2HIP_OK;
}

// kind 0: coalesced stream of table_bytes; kind 64 / 128: lanes * per_lane random gathers of aligned 64- / 128-byte entries from a
// table of table_bytes.  *useful_bytes = the bytes the lanes asked for.
int h2hip_bench_gather(h2hip_ctx *ctx, uint32_t kind, size_t table_bytes, uint32_t lanes, uint32_t per_lane, double *elapsed_ms, double *useful_bytes) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && elapsed_ms && useful_bytes && table_bytes >= 4096 && lanes >= 256 && per_lane >= 1, "bad argument");
    H2_REQUIRE(kind == 0 || kind == 64 || kind == 128, "kind must be 0 (stream), 64 or 128");
    void *table = nullptr, *out = nullptr;
    H2_HIPCHK(hipMalloc(&table, table_bytes));
    if (hipMalloc(&out, sizeof(uint4) * (size_t)lanes) != hipSuccess) {
        hipFree(table);
        set_error("hipMalloc failed");
        return H2HIP_ERR_NOMEM;
    }
    hipMemsetAsync(table, 0x5a, table_bytes, ctx->stream);
    const uint32_t blocks = lanes / 256;
    int rc = H2HIP_OK;
    for (int rep = 0; rep < 2 && rc == H2HIP_OK; ++rep) {   // rep 0 = warm-up
        rc = h2hip_timer_start(ctx);
        if (kind == 0) hipLaunchKernelGGL(stream_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4 *)table, (uint64_t)(table_bytes / 16), (uint4 *)out);
        if (kind == 64) hipLaunchKernelGGL(gather_probe_kernel<64>, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4 *)table, (uint64_t)(table_bytes / 64), per_lane, (uint4 *)out);
        if (kind == 128) hipLaunchKernelGGL(gather_probe_kernel<128>, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4 *)table, (uint64_t)(table_bytes / 128), per_lane, (uint4 *)out);
        if (rc == H2HIP_OK) rc = h2hip_timer_stop(ctx, elapsed_ms);
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(table);
    hipFree(out);
    *useful_bytes = kind == 0 ? (double)table_bytes : (double)blocks * 256.0 * per_lane * kind;
    return rc;
}

}  // extern "C"
