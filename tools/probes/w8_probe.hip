// r05 probe (tools/, not part of libh2hip): would an NTT pass whose waves own their sub-transforms — eight elements per lane in REGISTERS, radix-8
// rounds (three stages, five products by wave-uniform twiddles), explicit inter-step twiddle products, exchanges through a WAVE-PRIVATE LDS buffer
// with no block barrier — run closer to the 9 x 29-bit multiplier peak than the block-barrier radix-4 LDS rounds of ntt_tile_kernel (0.47-0.51 of
// the peak by algorithmic products, ~0.52 by executed products)?  The probe runs that inner structure back to back on synthetic data (no global
// I/O) and reports executed field products per second next to the plain product-chain peak measured in the same process.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I halo2-lib_amd/csrc tools/probes/w8_probe.hip -o tools/probes/w8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "field.cuh"
#include "fq29.cuh"
using namespace h2;

struct W8 {
    uint32_t w[4][9];   // w8^0 .. w8^3 in R' form (wave-uniform: kernel arguments -> scalar registers)
};
template <int K>
__device__ __forceinline__ Fr29 w8k(const W8 &t) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = t.w[K][i];
    return r;
}
// in-register radix-2 DIT DFT8: a[] bit-reversed in, natural out; inputs N and < 2r.  Trivial twiddles cost no product: stage 1 subtracts with K = 4,
// stage 2's trivial butterfly weak-reduces its b operand first.
template <int S, int I, int BLK>
__device__ __forceinline__ void bf8(Fr29 (&a)[8], const W8 &t) {
    constexpr int h = 1 << S, e0 = BLK * 2 * h + I, e1 = e0 + h, K = I * (4 >> S);
    const Fr29 x = a[e0];
    if constexpr (S == 0) {
        const Fr29 y = a[e1];
        a[e0] = f29_norm(f29_add(x, y));
        a[e1] = f29_sub<2>(x, y);
    } else if constexpr (K == 0) {
        const Fr29 y = S == 1 ? a[e1] : f29_weak_reduce(a[e1]);
        a[e0] = f29_norm(f29_add(x, y));
        if constexpr (S == 1) a[e1] = f29_sub<4>(x, y);
        else a[e1] = f29_sub<2>(x, y);
    } else {
        const Fr29 y = f29_mul(a[e1], w8k<K>(t));
        a[e0] = f29_norm(f29_add(x, y));
        a[e1] = f29_sub<2>(x, y);
    }
}
__device__ __forceinline__ void dft8(Fr29 (&a)[8], const W8 &t) {
    bf8<0, 0, 0>(a, t); bf8<0, 0, 1>(a, t); bf8<0, 0, 2>(a, t); bf8<0, 0, 3>(a, t);
    bf8<1, 0, 0>(a, t); bf8<1, 1, 0>(a, t); bf8<1, 0, 1>(a, t); bf8<1, 1, 1>(a, t);
    bf8<2, 0, 0>(a, t); bf8<2, 1, 0>(a, t); bf8<2, 2, 0>(a, t); bf8<2, 3, 0>(a, t);
}
struct P36 {
    uint32_t l[9];
};
// (template recursion instead of `#pragma unroll`: with the pragma the compiler re-rolled the loop and moved v[] to scratch memory)
template <int R>
__device__ __forceinline__ void twiddle_all(Fr29 (&v)[8], const P36 *twl, uint32_t lane, uint32_t it) {
    if constexpr (R < 8) {
        const P36 *tp = &twl[(lane + 64 * R + it) & 511];   // (consecutive lanes, 36-byte stride: conflict-free)
        Fr29 w;
#pragma unroll
        for (int l = 0; l < 9; ++l) w.l[l] = tp->l[l];
        v[R] = f29_mul(v[R], w);
        twiddle_all<R + 1>(v, twl, lane, it);
    }
}
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// MODE 0: registers only (DFT8 + 8 twiddle products, twiddles from LDS); MODE 1: + the wave-private LDS transposition (three planes); MODE 2: + a block barrier per round
// CODE (r05, second question — is the loop's SIZE what holds the waves back?): 0 = both blocks unrolled (13 products, ~31 KB of code per iteration);
// 1 = the DFT8 only (5 products, ~13 KB); 2 = the 8 twiddle products only (~14 KB); 3 = the DFT8 unrolled, the twiddle products as a ROLLED loop
// over elements parked in the wave's LDS buffer (one product's code: ~15 KB in all)
template <int MODE, int CODE = 0>
__global__ __launch_bounds__(256, 3) void w8_probe(uint32_t *out, int iters, W8 t) {
    __shared__ P36 twl[512];
    __shared__ v4u xbuf[4][8 * 68];   // per wave: 8 rows of 64 chunks (+4 pad) of 16 bytes
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < 512; i += 256) {
#pragma unroll
        for (int l = 0; l < 9; ++l) twl[i].l[l] = (((i & 1) ? t.w[2][l] : t.w[1][l]) + i) & MASK29;
    }
    __syncthreads();
    Fr29 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int l = 0; l < 9; ++l) v[r].l[l] = (t.w[r & 3][l] ^ (lane * 0x9E37u + r)) & (l == 8 ? 0x3FFFFFu : MASK29);
    v4u *xb = xbuf[wave];
    for (int it = 0; it < iters; ++it) {
        if (CODE != 2) dft8(v, t);
        if (CODE == 0 || CODE == 2) twiddle_all<0>(v, twl, lane, (uint32_t)it);
        if (CODE == 3) {   // park the eight elements in LDS (three planes of the wave's buffer would do; here: 9 words per element, lane-major), multiply them in a rolled loop
            P36 *park = reinterpret_cast<P36 *>(xb);   // 64 lanes x 8 x 36 B = 18 KB > the 8.5 KB buffer: the probe parks FOUR elements at a time
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int l = 0; l < 9; ++l) park[(r * 64 + lane)].l[l] = v[half * 4 + r].l[l];
#pragma unroll 1
                for (int r = 0; r < 4; ++r) {
                    const P36 *tp = &twl[(lane + 64 * (half * 4 + r) + it) & 511];
                    Fr29 w, x;
#pragma unroll
                    for (int l = 0; l < 9; ++l) {
                        w.l[l] = tp->l[l];
                        x.l[l] = park[r * 64 + lane].l[l];
                    }
                    x = f29_mul(x, w);
#pragma unroll
                    for (int l = 0; l < 9; ++l) park[r * 64 + lane].l[l] = x.l[l];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int l = 0; l < 9; ++l) v[half * 4 + r].l[l] = park[(r * 64 + lane)].l[l];
            }
        }
        if (MODE >= 1) {
            // transposition among the wave's 64 x 8 elements, plane by plane: write row r at chunk (lane), read 8 consecutive chunks of row (lane >> 3)...
            // (the index pattern of a 64 x 8 -> 8 x 64 exchange: writer (lane, r) -> slot r * 68 + lane; reader lane takes slots (j * 68 + perm(lane, j)))
#pragma unroll
            for (int plane = 0; plane < 3; ++plane) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    v4u c;
                    if (plane == 0) c = v4u{v[r].l[0], v[r].l[1], v[r].l[2], v[r].l[3]};
                    else if (plane == 1) c = v4u{v[r].l[4], v[r].l[5], v[r].l[6], v[r].l[7]};
                    else c = v4u{v[r].l[8], 0u, 0u, 0u};
                    xb[r * 68 + lane] = c;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint32_t src = (lane & 7u) * 8u + r;   // lane' = 8 * (lane & 7) + r, row = lane >> 3: an 8 x 8 x 8 index rotation
                    const v4u c = xb[(lane >> 3) * 68 + src];
                    if (plane == 0) { v[r].l[0] = c.x; v[r].l[1] = c.y; v[r].l[2] = c.z; v[r].l[3] = c.w; }
                    else if (plane == 1) { v[r].l[4] = c.x; v[r].l[5] = c.y; v[r].l[6] = c.z; v[r].l[7] = c.w; }
                    else v[r].l[8] = c.x;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int l = 0; l < 9; ++l) acc ^= v[r].l[l];
    out[blockIdx.x * 256 + tid] = acc;
}
// the plain peak: independent chains of products, as h2hip_bench_modmul29
__global__ __launch_bounds__(256) void peak_probe(uint32_t *out, int iters, W8 t) {
    Fr29 a = w8k<1>(t), b = w8k<2>(t), c = w8k<3>(t);
    a.l[0] ^= threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        a = f29_mul(a, b);
        c = f29_mul(c, b);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int l = 0; l < 9; ++l) acc ^= a.l[l] ^ c.l[l];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <class F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    W8 t;
    for (int k = 0; k < 4; ++k)
        for (int l = 0; l < 9; ++l) t.w[k][l] = (0x12345u * (k + 3) + 0x9E3779u * (l + 1)) & (l == 8 ? 0x1FFFFFu : MASK29);
    uint32_t *out = nullptr;
    const int blocks = 768 * 4, iters = 64;
    hipMalloc(&out, sizeof(uint32_t) * 256 * 16384);
    const double pk = time_ms([&] { hipLaunchKernelGGL(peak_probe, dim3(16384), dim3(256), 0, 0, out, 256, t); });
    const double peak = 16384.0 * 256 * 256 * 2 / (pk * 1e-3);
    printf("product-chain peak: %.4g products/s\n", peak);
    const double per_iter = 13.0 * 256;   // per workgroup and iteration: 5 (DFT8) + 8 (twiddles) products per lane
    double m0 = time_ms([&] { hipLaunchKernelGGL(w8_probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, t); });
    double m1 = time_ms([&] { hipLaunchKernelGGL(w8_probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, t); });
    double m2 = time_ms([&] { hipLaunchKernelGGL(w8_probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters, t); });
    const double work = per_iter * blocks * iters;
    printf("radix-8 rounds in registers (DFT8 + 8 twiddle products, twiddles from LDS): %.4g products/s = %.2f of the peak\n", work / (m0 * 1e-3), work / (m0 * 1e-3) / peak);
    printf("  + wave-private LDS transposition, three planes, no block barrier:          %.4g products/s = %.2f of the peak\n", work / (m1 * 1e-3), work / (m1 * 1e-3) / peak);
    printf("  + a block barrier per round:                                               %.4g products/s = %.2f of the peak\n", work / (m2 * 1e-3), work / (m2 * 1e-3) / peak);
    {
        const double a = time_ms([&] { hipLaunchKernelGGL((w8_probe<0, 1>), dim3(blocks), dim3(256), 0, 0, out, iters, t); });
        const double b = time_ms([&] { hipLaunchKernelGGL((w8_probe<0, 2>), dim3(blocks), dim3(256), 0, 0, out, iters, t); });
        const double c = time_ms([&] { hipLaunchKernelGGL((w8_probe<0, 3>), dim3(blocks), dim3(256), 0, 0, out, iters, t); });
        const double w5 = 5.0 * 256 * blocks * iters, w8 = 8.0 * 256 * blocks * iters;
        printf("code size: the DFT8 alone (5 products, ~13 KB per iteration):        %.4g products/s = %.2f of the peak\n", w5 / (a * 1e-3), w5 / (a * 1e-3) / peak);
        printf("code size: the 8 twiddle products alone (~14 KB):                    %.4g products/s = %.2f of the peak\n", w8 / (b * 1e-3), w8 / (b * 1e-3) / peak);
        printf("code size: DFT8 unrolled + twiddle products ROLLED through LDS (~15 KB): %.4g products/s = %.2f of the peak\n", work / (c * 1e-3), work / (c * 1e-3) / peak);
    }
    printf("(ntt_tile_kernel today: 0.47-0.51 of the peak by algorithmic products at 2^22 .. 2^24, x 11.5 / 11 executed)\n");
    hipFree(out);
    return 0;
}
