// r03's FP64 multiplier probe — the 254-bit Montgomery product on 5 x 52-bit limbs through the FP64 FMA unit (hi / lo halves of every partial product
// by the 2^102 / 2^52 rounding constants, round toward zero) — moved out of libh2hip.so in r06.  Measured 0.70x of the 9 x 29-bit
// integer form (profiles/archive: r03 modmul_repr), so nothing in the product uses it.  Standalone: builds against the product's headers only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I halo2-lib_amd/csrc -I include tools/probes/modmul52_probe.hip -o /tmp/modmul52_probe && /tmp/modmul52_probe
// prints products/s and the limbs of lane 0 after `iters` steps of x <- x * y * 2^-260 mod r from x_0 = y (check them against big-int arithmetic:
// tests/test_gpu_parity.py held that check while the probe was part of the library).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define H2_HD __host__ __device__
// the THIRD multiplier representation — 5 x 52-bit limbs held as doubles, products by v_fma_f64.  With the FP64
// unit rounding toward zero, hi = fma(a, b, 2^104) carries floor(a*b / 2^52) in its mantissa and lo = fma(a, b, 2^104 + 2^52 - hi) the low 52
// bits (both exact); column sums are integer additions of the raw bit patterns (the exponent fields are cancelled by the initial column
// values).  Montgomery reduction (R = 2^260, one 52-bit digit m_k per step, m_k * r by the same hi/lo pairs) interleaved as in fq29.cuh.
// Per product: 25 + 25 limb products = 100 FMAs + 50 exact FP subtractions + 100 64-bit integer additions + the five m_k digits, carries
// and int <-> double conversions — against 171 v_mad_u64_u32 + 57 others for the 9 x 29-bit form, whose multiply-add already accumulates
// 64 bits in the same instruction (profiles/archive/r01_modmul_repr.md has the measured rates side by side).
struct Mod52 {
    // r in 52-bit limbs and -r^-1 mod 2^52
    static constexpr uint64_t P0 = 0x1f593f0000001ull, P1 = 0x4879b9709143eull, P2 = 0x181585d2833e8ull, P3 = 0xa029b85045b68ull, P4 = 0x30644e72e131ull;
    static constexpr uint64_t PINV = 0x1f593efffffffull;
};
__device__ __forceinline__ uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
__device__ __forceinline__ double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
__device__ __forceinline__ uint64_t add64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_HIPEMU)
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));   // one instruction per 64-bit addition (gfx940+)
    return r;
#else
    return a + b;
#endif
}
__device__ __forceinline__ void mul52(double (&x)[5], const double (&y)[5]) {
    constexpr uint64_t KLO = 0x4330000000000000ull, KHI = 0x4670000000000000ull, MASK = (1ull << 52) - 1;
    const double C1 = 5.0706024009129176e30 /* 2^102 */ * 4.0, TWO52 = 4503599627370496.0, C2 = C1 + TWO52;
    const double p[5] = {(double)Mod52::P0, (double)Mod52::P1, (double)Mod52::P2, (double)Mod52::P3, (double)Mod52::P4};
    // column t receives 2*cnt(t) low halves and 2*cnt(t-1) high halves (product + reduction), cnt(t) = #{(i, j): i + j = t}
    uint64_t col[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        const int cl = t <= 8 ? (t < 4 ? t : (8 - t < 4 ? 8 - t : 4)) + 1 : 0, ch = t >= 1 ? ((t - 1) < 4 ? (t - 1) : (9 - t < 4 ? 9 - t : 4)) + 1 : 0;
        col[t] = 0ull - (2ull * cl * KLO + 2ull * ch * KHI);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double hi = __builtin_fma(x[i], y[j], C1);
            const double lo = __builtin_fma(x[i], y[j], C2 - hi);
            col[i + j] = add64(col[i + j], d2u(lo));
            col[i + j + 1] = add64(col[i + j + 1], d2u(hi));
        }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double c = u2d((col[k] & MASK) | KLO) - TWO52;                   // the column's low 52 bits as a double
        const double th = __builtin_fma(c, (double)Mod52::PINV, C1);
        const double m = __builtin_fma(c, (double)Mod52::PINV, C2 - th) - TWO52;   // m_k = c * (-r^-1) mod 2^52
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double hi = __builtin_fma(m, p[j], C1);
            const double lo = __builtin_fma(m, p[j], C2 - hi);
            col[k + j] = add64(col[k + j], d2u(lo));
            col[k + j + 1] = add64(col[k + j + 1], d2u(hi));
        }
        col[k + 1] = add64(col[k + 1], col[k] >> 52);                          // the column is now a multiple of 2^52
    }
    uint64_t carry = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const uint64_t v = col[5 + t] + carry;
        carry = v >> 52;
        x[t] = u2d((t < 4 ? (v & MASK) : v) | KLO) - TWO52;
    }
}
template <int CHAINS>
__global__ __launch_bounds__(256) void modmul52_bench_kernel(uint64_t *__restrict__ out, uint32_t iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double y[5] = {(double)(0x9e3779b97f4a7ull ^ (i & 0xffff)), (double)0x3c6ef372fe94full, (double)0x54ff53a5f1d36ull, (double)0x10e527fade682ull, (double)0x1f83d9abfb41ull};
    double x[CHAINS][5];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k)
#pragma unroll
        for (int t = 0; t < 5; ++t) x[k][t] = y[t] + (t == 0 ? (double)k : 0.0);
    for (uint32_t it = 0; it < iters; ++it) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_HIPEMU)
        // MODE.FP_ROUND[3:2] (f64 / f16) := toward zero.  Set here, as opaque assembly, every iteration: the compiler's mode-register pass
        // knows nothing of a mode set by hand and puts its own "back to round-to-nearest" after the int -> double conversions above
        // (the operands tie every product of the iteration to the instruction: without them the FMAs are scheduled above it)
#pragma unroll
        for (int k = 0; k < CHAINS; ++k)
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" : "+v"(x[k][0]), "+v"(x[k][1]), "+v"(x[k][2]), "+v"(x[k][3]), "+v"(x[k][4]));
#endif
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) mul52(x[k], y);
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_HIPEMU)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0" : "+v"(x[k][0]), "+v"(x[k][1]), "+v"(x[k][2]), "+v"(x[k][3]), "+v"(x[k][4]));
#endif
#pragma unroll
    for (int k = 0; k < CHAINS; ++k)
#pragma unroll
        for (int t = 0; t < 5; ++t) out[(i * CHAINS + k) * 5 + t] = (uint64_t)x[k][t];
}


int main(int argc, char **argv) {
    const uint32_t blocks = argc > 1 ? (uint32_t)atoi(argv[1]) : 16384, iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 256;
    uint64_t *buf = nullptr;
    const size_t lanes = (size_t)blocks * 256;
    if (hipMalloc(&buf, sizeof(uint64_t) * 5 * 2 * lanes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int chains = 1; chains <= 2; ++chains) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (chains == 1) hipLaunchKernelGGL(modmul52_bench_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, iters);
            else hipLaunchKernelGGL(modmul52_bench_kernel<2>, dim3(blocks), dim3(256), 0, 0, buf, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        uint64_t limbs[5];
        hipMemcpy(limbs, buf, sizeof(limbs), hipMemcpyDeviceToHost);
        printf("chains %d: %.4g products/s; lane 0: %013llx %013llx %013llx %013llx %013llx\n", chains, (double)lanes * iters * chains / (ms * 1e-3),
               (unsigned long long)limbs[0], (unsigned long long)limbs[1], (unsigned long long)limbs[2], (unsigned long long)limbs[3], (unsigned long long)limbs[4]);
    }
    hipFree(buf);
    return 0;
}
