// r05 probe (tools/, not part of libh2hip): issue cost of the VALU instructions the 254-bit arithmetic is made of, RELATIVE to v_add_u32 — every kernel
// runs the same number of one instruction over eight independent registers per lane, full occupancy, so that the ratio of the kernels' times is the
// ratio of the instructions' issue costs at whatever clock the chip holds for that mix (s_memtime cycles per instruction are printed as well).
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/valu_rate.hip -o tools/probes/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define REP8(S) S S S S S S S S
enum { ADD32, AND32, ALIGNBIT, MUL_LO, MUL_HI, MAD_U32_U24, MAD64, LSHR64, LSHL_ADD64, ADD_CO_PAIR, MAD64_DEP, MIX_MAD_AND, MIX_MAD_3AND, NOPS };
static const char *NAMES[] = {"v_add_u32", "v_and_b32", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mad_u64_u32 (8 chains)", "v_lshrrev_b64",
                              "v_lshl_add_u64", "v_add_co_u32 + v_addc_co_u32", "v_mad_u64_u32 (1 chain)", "mix: mad64, and, mad64, and ...", "mix: mad64, and, and, and ..."};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t *out, uint32_t iters, uint32_t seed, unsigned long long *cycles) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
    const uint32_t k = seed | 1u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        if (OP == ADD32) {
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (OP == AND32) {
            REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (OP == ALIGNBIT) {
            REP8(asm volatile("v_alignbit_b32 %0, %0, %8, 29\n v_alignbit_b32 %1, %1, %8, 29\n v_alignbit_b32 %2, %2, %8, 29\n v_alignbit_b32 %3, %3, %8, 29\n v_alignbit_b32 %4, %4, %8, 29\n v_alignbit_b32 %5, %5, %8, 29\n v_alignbit_b32 %6, %6, %8, 29\n v_alignbit_b32 %7, %7, %8, 29"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (OP == MUL_LO) {
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (OP == MUL_HI) {
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (OP == MAD_U32_U24) {
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (OP == MAD64) {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
                              : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(k), "v"(a0) : "vcc");)
        } else if (OP == MAD64_DEP) {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0"
                              : "+v"(b0) : "v"(k), "v"(a0) : "vcc");)
        } else if (OP == MIX_MAD_AND) {   // 4 multiply-adds and 4 ands, alternating
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_and_b32 %4, %4, %8\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_and_b32 %5, %5, %8\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_and_b32 %6, %6, %8\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_and_b32 %7, %7, %8"
                              : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(a0) : "vcc");)
        } else if (OP == MIX_MAD_3AND) {  // 2 multiply-adds and 6 ands
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                              : "+v"(b0), "+v"(b1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(a0) : "vcc");)
        } else if (OP == LSHR64) {
            REP8(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n v_lshrrev_b64 %4, 1, %4\n v_lshrrev_b64 %5, 1, %5\n v_lshrrev_b64 %6, 1, %6\n v_lshrrev_b64 %7, 1, %7"
                              : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));)
        } else if (OP == LSHL_ADD64) {
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8"
                              : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(b0 | 1));)
        } else if (OP == ADD_CO_PAIR) {   // four 64-bit additions as add_co / addc pairs = 8 instructions
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
}

// operand-bank probe: the same multiply-add with its two 32-bit sources in the accumulator's register bank (v20, v24 with v[16:17]: all = 0 mod 4) or in three
// different banks (v21, v26): fixed registers (values are garbage: only the timing matters), eight accumulators
template <int SAME>
__global__ __launch_bounds__(256) void bank_kernel(uint64_t *out, uint32_t iters, unsigned long long *cycles) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        if (SAME) {
            REP8(asm volatile("v_mad_u64_u32 v[16:17], vcc, v20, v24, v[16:17]\n v_mad_u64_u32 v[32:33], vcc, v36, v40, v[32:33]\n v_mad_u64_u32 v[48:49], vcc, v52, v56, v[48:49]\n v_mad_u64_u32 v[64:65], vcc, v68, v72, v[64:65]\n"
                              "v_mad_u64_u32 v[80:81], vcc, v84, v88, v[80:81]\n v_mad_u64_u32 v[96:97], vcc, v100, v104, v[96:97]\n v_mad_u64_u32 v[112:113], vcc, v116, v120, v[112:113]\n v_mad_u64_u32 v[124:125], vcc, v20, v24, v[124:125]"
                              ::: "vcc", "v16", "v17", "v20", "v24", "v32", "v33", "v36", "v40", "v48", "v49", "v52", "v56", "v64", "v65", "v68", "v72", "v80", "v81", "v84", "v88", "v96", "v97", "v100", "v104",
                                  "v112", "v113", "v116", "v120", "v124", "v125");)
        } else {
            REP8(asm volatile("v_mad_u64_u32 v[16:17], vcc, v22, v27, v[16:17]\n v_mad_u64_u32 v[32:33], vcc, v38, v43, v[32:33]\n v_mad_u64_u32 v[48:49], vcc, v54, v59, v[48:49]\n v_mad_u64_u32 v[64:65], vcc, v70, v75, v[64:65]\n"
                              "v_mad_u64_u32 v[80:81], vcc, v86, v91, v[80:81]\n v_mad_u64_u32 v[96:97], vcc, v102, v107, v[96:97]\n v_mad_u64_u32 v[112:113], vcc, v118, v123, v[112:113]\n v_mad_u64_u32 v[124:125], vcc, v22, v27, v[124:125]"
                              ::: "vcc", "v16", "v17", "v22", "v27", "v32", "v33", "v38", "v43", "v48", "v49", "v54", "v59", "v64", "v65", "v70", "v75", "v80", "v81", "v86", "v91", "v96", "v97", "v102", "v107",
                                  "v112", "v113", "v118", "v123", "v124", "v125");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = t1;
}
template <int SAME>
static double run_bank(uint64_t *out, unsigned long long *cyc, int waves_per_simd) {
    const uint32_t iters = 4096, blocks = 256 * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(bank_kernel<SAME>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(bank_kernel<SAME>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

template <int OP>
static void run(uint64_t *out, unsigned long long *cyc, int waves_per_simd, double *ms_out, double *cyc_out) {
    const uint32_t iters = 4096, blocks = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs, 4 waves per workgroup: waves_per_simd workgroups per CU
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u, cyc);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u + r, cyc);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    unsigned long long c;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    *ms_out = best;
    *cyc_out = (double)c / (iters * 64.0);   // s_memtime ticks (100 MHz constant clock on this part?) per instruction of lane 0's wave
}

int main() {
    uint64_t *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(uint64_t) * 256 * 8 * 256);
    hipMalloc(&cyc, sizeof(*cyc));
    for (int wps : {1, 2, 4, 8}) {
        double ms[NOPS], cy[NOPS];
        run<ADD32>(out, cyc, wps, &ms[ADD32], &cy[ADD32]);
        run<AND32>(out, cyc, wps, &ms[AND32], &cy[AND32]);
        run<ALIGNBIT>(out, cyc, wps, &ms[ALIGNBIT], &cy[ALIGNBIT]);
        run<MUL_LO>(out, cyc, wps, &ms[MUL_LO], &cy[MUL_LO]);
        run<MUL_HI>(out, cyc, wps, &ms[MUL_HI], &cy[MUL_HI]);
        run<MAD_U32_U24>(out, cyc, wps, &ms[MAD_U32_U24], &cy[MAD_U32_U24]);
        run<MAD64>(out, cyc, wps, &ms[MAD64], &cy[MAD64]);
        run<LSHR64>(out, cyc, wps, &ms[LSHR64], &cy[LSHR64]);
        run<LSHL_ADD64>(out, cyc, wps, &ms[LSHL_ADD64], &cy[LSHL_ADD64]);
        run<ADD_CO_PAIR>(out, cyc, wps, &ms[ADD_CO_PAIR], &cy[ADD_CO_PAIR]);
        run<MAD64_DEP>(out, cyc, wps, &ms[MAD64_DEP], &cy[MAD64_DEP]);
        run<MIX_MAD_AND>(out, cyc, wps, &ms[MIX_MAD_AND], &cy[MIX_MAD_AND]);
        run<MIX_MAD_3AND>(out, cyc, wps, &ms[MIX_MAD_3AND], &cy[MIX_MAD_3AND]);
        printf("waves per SIMD = %d (4096 x 64 instructions per lane)\n", wps);
        printf("  v_mad_u64_u32, 32-bit sources in the accumulator's VGPR bank: %.3f ms; in three different banks: %.3f ms\n", run_bank<1>(out, cyc, wps), run_bank<0>(out, cyc, wps));
        for (int i = 0; i < NOPS; ++i)
            printf("  %-30s %8.3f ms   x%.2f of v_add_u32   (%.2f counter ticks per instruction in one wave)   %.3g wave-instr/s/SIMD\n", NAMES[i], ms[i], ms[i] / ms[ADD32], cy[i],
                   4096.0 * 64.0 * wps / (ms[i] * 1e-3));
    }
    return 0;
}
