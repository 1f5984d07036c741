// r06, last: a probe of SHOUP-form products for the NTT's twiddle multiplications — not part of libh2hip.so.  Measured: +15 % products/s
// here, and NO gain when built into ntt_tile_kernel's butterflies (profiles/r06_ntt_shoup.log: the pass is not bound by its multiply count) — so only the probe stays.
// A butterfly multiplies a data element x by a CONSTANT w.  With w stored as the pair (w, w' = floor(w * 2^261 / r)):
//     q = floor(x * w' / 2^261)            only the top nine limbs of the product: columns 7..17 (two guard columns, q is off by <= 2)
//     z = (x * w + q * (2^261 - r)) mod 2^261 = x * w - q * r   in [0, 3 r)
// 53 + 45 + 45 = 143 multiplies against Montgomery's 81 + 9 + 81 = 171, no quotient digits, and x * w keeps the domain x is stored in (w is plain).
// (w, w') from the R'-form table entry W = w * 2^261 mod r (canonical): w = W * 1 * 2^-261 (a Montgomery product by the integer 1),
// w' = W * (-r^-1) mod 2^261 (one low-half product) — because w * 2^261 = w' * r + W.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I halo2-lib_amd/csrc -I include tools/probes/shoup_probe.hip -o /tmp/shoup_probe && /tmp/shoup_probe
// prints: a correctness check (2^20 random pairs: canonical(shoup(x, w)) == canonical(montgomery(x, W))) and the products/s of both forms in the
// dependent-chain loop the library's multiplier roof is measured with (modmul29_bench_kernel in csrc/fr_ops.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "fr29.cuh"

using namespace h2;

struct Shoup29 {
    static constexpr __host__ __device__ uint32_t negpinv(int i) { constexpr uint32_t v[9] = {0x0fffffffu, 0x170fac9fu, 0x1a446cf0u, 0x0d0c9698u, 0x02391658u, 0x0c144c83u, 0x06cb8e6au, 0x03a1b068u, 0x1273f82fu}; return v[i]; }   // -r^-1 mod 2^261
    static constexpr __host__ __device__ uint32_t pbar(int i) { constexpr uint32_t v[9] = {0x0fffffffu, 0x00f05360u, 0x11a3dbafu, 0x182f6f0cu, 0x0a7a2d7cu, 0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u}; return v[i]; }      // 2^261 - r
};

// low nine limbs of a * b (mod 2^261), a and b normalised
__device__ __forceinline__ Fr29 low_product(const Fr29 &a, const uint32_t (&b)[9]) {
    uint64_t d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; i + j < 9; ++j) d[i + j] += (uint64_t)a.l[i] * b[j];
    Fr29 r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint64_t v = d[k] + carry;
        r.l[k] = (uint32_t)v & MASK29;
        carry = v >> 29;
    }
    return r;
}
// x limbs < 2^31 (a lazy sum), w and wq normalised: z = x * w mod r, normalised, value < 3 r
__device__ __forceinline__ Fr29 f29_mul_shoup(const Fr29 &x, const Fr29 &w, const Fr29 &wq) {
    uint64_t c[18];
#pragma unroll
    for (int k = 7; k < 18; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (i + j >= 7) c[i + j] += (uint64_t)x.l[i] * wq.l[j];
    uint32_t q[9];
    uint64_t carry = (c[7] >> 29);
    carry = (c[8] + carry) >> 29;
#pragma unroll
    for (int k = 9; k < 17; ++k) {
        const uint64_t v = c[k] + carry;
        q[k - 9] = (uint32_t)v & MASK29;
        carry = v >> 29;
    }
    q[8] = (uint32_t)(c[17] + carry);   // (c[17] holds no product: the top limb is the last carry)
    uint64_t d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; i + j < 9; ++j) d[i + j] += (uint64_t)x.l[i] * w.l[j] + (uint64_t)q[i] * Shoup29::pbar(j);
    Fr29 r;
    carry = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint64_t v = d[k] + carry;
        r.l[k] = (uint32_t)v & MASK29;
        carry = v >> 29;
    }
    return r;
}
__device__ __forceinline__ Fr29 canonical29(const Fr29 &v) { return f29_split<R29P>(f29_pack_canonical<FrP>(v)); }   // value < 2 r in

__device__ __forceinline__ void shoup_pair(const Fr29 &W, Fr29 &w, Fr29 &wq) {   // W: R' form, any weakly reduced value below 2 r
    const Fr29 Wc = canonical29(W);
    Fr29 one_int = Fr29::zero();
    one_int.l[0] = 1;
    w = canonical29(f29_mul(Wc, one_int));
    uint32_t npi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) npi[i] = Shoup29::negpinv(i);
    wq = low_product(Wc, npi);
}

__global__ __launch_bounds__(256) void check_kernel(const Fr *__restrict__ xs, const Fr *__restrict__ ws, uint32_t n, uint32_t *__restrict__ bad) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fr a = xs[i], b = ws[i];
    a.l[7] &= 0x0fffffffu;
    b.l[7] &= 0x0fffffffu;
    Fr29 x = f29_split<R29P>(a);
    const Fr29 W = f29_mul(f29_split<R29P>(b), Fr29::one());   // some R'-form constant
    if (i & 1) x = f29_add(f29_add(x, x), f29_add(x, x));       // a lazy operand: limbs up to 2^31, value up to 16 r
    Fr29 w, wq;
    shoup_pair(W, w, wq);
    const Fr29 z = f29_mul_shoup(x, w, wq);
    const Fr29 m = f29_mul_wide(x, canonical29(W));
    const Fr29 zc = canonical29(f29_mul(z, Fr29::one()));            // z * 1 (value * 2^261 * 2^-261): < 1.1 r, canonicalised
    const Fr29 mc = canonical29(f29_mul(m, Fr29::one()));
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) ok &= zc.l[k] == mc.l[k];
    if (!ok) atomicAdd(bad, 1u);
}

template <int FORM>
__global__ __launch_bounds__(256) void rate_kernel(Fr *__restrict__ io, uint32_t iters) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fr y0 = io[i];
    y0.l[7] &= 0x0fffffffu;
    const Fr29 W = f29_split<R29P>(y0);
    Fr29 x = W, w, wq;
    shoup_pair(W, w, wq);
    for (uint32_t it = 0; it < iters; ++it) x = FORM ? f29_mul_shoup(x, w, wq) : f29_mul(x, W);
    io[i] = f29_pack_canonical<FrP>(f29_mul(x, Fr29::one()));
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
int main() {
    const uint32_t n = 1u << 20;
    Fr *xs, *ws;
    uint32_t *bad, hbad = 0;
    CK(hipMalloc(&xs, sizeof(Fr) * n));
    CK(hipMalloc(&ws, sizeof(Fr) * n));
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    Fr *h = (Fr *)malloc(sizeof(Fr) * n);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t i = 0; i < n; ++i)
            for (int k = 0; k < 8; ++k) h[i].l[k] = (i < 64 && pass == 0) ? (i & 1 ? 0xffffffffu : 0u) : next();
        CK(hipMemcpy(pass ? ws : xs, h, sizeof(Fr) * n, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, 0, xs, ws, n, bad);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    printf("check: %u of %u products differ between the Shoup and the Montgomery form%s\n", hbad, n, hbad ? "  <-- WRONG" : " (all equal)");
    const uint32_t blocks = 256 * 12, iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int form = 0; form < 2; ++form)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (form) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, xs, iters);
            else hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, xs, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%s: %.4g products/s (%.3f ms)\n", form ? "shoup (143 multiplies)     " : "montgomery (171 multiplies)", (double)blocks * 256 * iters / (ms * 1e-3), ms);
        }
    return hbad ? 1 : 0;
}
