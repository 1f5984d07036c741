// F_r on unsaturated 9 x 29-bit limbs for the prover's POINTWISE kernels (quotient identities, product factors): fq29.cuh's arithmetic (171
// multiplies + ~57 other instructions per product against ~500 for the saturated CIOS product, and one Montgomery reduction for a sum of two
// products) WITHOUT leaving the saturated Montgomery domain the prover's arrays are stored in — no conversion product per loaded element:
//
//   r29_load(x)     raw limb split of a stored element: the same integer X = x * 2^256 mod r  (N, value < r)
//   r29_load32(x)   limb split of 32 * X as a 261-bit integer (X < 2^254, so 32 X < 2^259 fits nine limbs): N, value < 32 r
//   r29_const(c)    a challenge / constant in R' = 2^261 form (C = c * 2^261 mod r; fr29_from_sat): N, value < 1.01 r
//
// f29_mul(A, B) = A * B * 2^-261, so
//   data x constant:  r29_load(x) * r29_const(c)              = (x c) * 2^256            — in the stored domain
//   data x data:      r29_load32(x) * r29_load(y)             = 32 X Y 2^-261 = (x y) * 2^256   — in the stored domain as well
// and products chain as long as ONE operand of every data x data product carries the factor 32 (taken at load time, or folded into the
// constants it is built from).  Bounds as in fq29.cuh: a product's output is N and below (1 + X_a X_b / 169.3) r for inputs below X_a r, X_b r
// with X_a X_b <= 169 — 32 x 3.3 fits —, r29_store needs a value below 2 r.  The emulated build asserts every limb bound.
#pragma once
#include "fq29.cuh"

namespace h2 {

H2_HD Fr29 r29_load(const Fr &x) { return f29_split<R29P>(x); }
H2_HD Fr29 r29_load32(const Fr &x) {
    Fr29 t;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        // bits [29 i - 5, 29 i + 24) of x (zeros below bit 0)
        const int bit = 29 * i - 5;
        if (bit < 0) {
            t.l[i] = (x.l[0] << 5) & MASK29;
        } else {
            const int w = bit >> 5, off = bit & 31;
            const uint64_t lo = w < 8 ? x.l[w] : 0, hi = (w + 1) < 8 ? x.l[w + 1] : 0;
            t.l[i] = (uint32_t)(((hi << 32) | lo) >> off) & MASK29;
        }
    }
    return t;
}
H2_HD Fr29 r29_const(const Fr &c) { return fr29_from_sat(c); }
H2_HD Fr fe_x32(Fr a) {   // 32 a (the factor a data x data product needs, folded into a constant)
#pragma unroll
    for (int t = 0; t < 5; ++t) a = fe_dbl(a);
    return a;
}
H2_HD Fr r29_store(const Fr29 &v) { return f29_pack_canonical<FrP>(v); }   // v N and < 2 r

// Table element of the power tables (NTT twiddles, ntt.hip): 9 limbs padded to 48 B so that it moves as 16-byte accesses
struct alignas(16) Fr29L {
    Fr29 v;
    uint32_t pad[3];
};
// omega^e from the two-level table of a (log_n, omega) pair (ntt.hip: get_twiddles / ntt_pow_table): omega^e = T2[e >> lo_bits] * T1[e & mask], both in
// R' = 2^261 form, so the result is too.  r06: the row kernels of the permutation argument start their per-lane chain omega^i0 here (one product, two
// table reads) instead of a ~28-product square-and-multiply in saturated arithmetic — which was a quarter to a half of those kernels' instructions.
struct OmegaTable {
    const Fr29L *t1, *t2;
    uint32_t lo_bits;
};
__device__ __forceinline__ Fr29 pow_lookup(const Fr29L *__restrict__ t1, const Fr29L *__restrict__ t2, uint32_t lo_bits, uint64_t e) {
    const uint32_t lo = (uint32_t)(e & ((1ull << lo_bits) - 1));
    const uint32_t hi = (uint32_t)(e >> lo_bits);
    return f29_mul(t2[hi].v, t1[lo].v);
}

}  // namespace h2
