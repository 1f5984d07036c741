// Pointwise F_r kernels around the MSM/NTT core: the "witness-column Montgomery mul"
// batches of halo2-base's GateInstructions (reference halo2-base/src/gates/flex_gate/mod.rs:158-277: add /
// sub / mul / mul_add on column values) plus the diagnostic multiplier micro-benchmark that defines the
// integer roofline quoted by bench.py.
#include "internal.h"

namespace h2 {

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

// Multiplier roofline probe: every lane runs CHAINS independent dependent-multiply chains of `iters` steps.
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
int h2hip_fr_add_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, size_t n) { return binop(ctx, OP_ADD, out, a, b, n); }
int h2hip_fr_sub_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, size_t n) { return binop(ctx, OP_SUB, out, a, b, n); }
int h2hip_fr_mul_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, size_t n) { return binop(ctx, OP_MUL, out, a, b, n); }
int h2hip_fr_mul_add_batch_dev(h2hip_ctx *ctx, void *out, const void *a, const void *b, const void *c, size_t n) {
    H2_REQUIRE(ctx && (n == 0 || (out && a && b && c)), "NULL argument");
    if (!n) return H2HIP_OK;
    prof_begin(ctx, "fr_mul_add_kernel");
    hipLaunchKernelGGL(fr_mul_add_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (Fr *)out, (const Fr *)a, (const Fr *)b,
                       (const Fr *)c, n);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

int h2hip_bench_modmul(h2hip_ctx *ctx, uint32_t blocks, uint32_t iters, uint32_t chains, double *elapsed_ms, double *modmuls) {
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

}  // extern "C"
