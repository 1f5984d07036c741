// Quad-lane point arithmetic for the MSM's latency-bound tails (bucket reduction, window fold).
//
// A lone wave issues about one VALU instruction per 4 cycles, so a dependent chain of point additions (12M+2S each,
// all serial in one lane) costs ~9 us per addition no matter how idle the rest of the chip is.  Here one XYZZ point
// is spread over the 4 lanes of a quad — lane q = lane & 3 holds coordinate q of (X, Y, ZZ, ZZZ) — and the 14 field
// products of an addition are scheduled as 4 levels of 4 concurrent products, with the operands moved between the
// quad's lanes by ds_bpermute shuffles: ~3.3x shorter latency per addition (4 product latencies + ~120 shuffles).
// Same formulas and the same weak-reduction bounds as ec29.cuh.  All cross-lane calls are made wave-uniformly
// (divergent quads cost the same on SIMT hardware anyway); exceptional cases are resolved by selection, the rare
// doubling case by one wave vote.
#pragma once
#include "ec29.cuh"

namespace h2 {

// The quad-lane point operations are real functions (one shared copy per code object) on the GPU: the tail kernels
// that use them are latency-bound, and keeping their instruction footprint small matters when they share a CU's
// instruction cache with another MSM's accumulation kernel.
#ifdef H2_HIPEMU
#define H2_QUAD_FN __device__ __forceinline__
#else
#define H2_QUAD_FN __device__ __attribute__((noinline))
#endif

// Quad permutes are DPP moves (v_mov_b32_dpp quad_perm:[..], full-rate VALU, no LDS round trip).  PERM encodes the
// source lane (within the quad) of lanes 0..3: p0 | p1<<2 | p2<<4 | p3<<6.
constexpr int QP_BCAST0 = 0x00, QP_BCAST1 = 0x55, QP_BCAST2 = 0xAA, QP_BCAST3 = 0xFF;
constexpr int QP_PAIRS_LO = 0x50;   // lanes 0,1 <- lane 0 ; lanes 2,3 <- lane 1
constexpr int QP_PAIRS_HI = 0xFA;   // lanes 0,1 <- lane 2 ; lanes 2,3 <- lane 3
template <int PERM>
__device__ __forceinline__ uint32_t quad_perm_u32(uint32_t v, uint32_t lane) {
#ifdef H2_HIPEMU
    (void)lane;
    return __shfl(v, (int)((lane & ~3u) | ((PERM >> (2 * (lane & 3u))) & 3u)));
#else
    (void)lane;
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, PERM, 0xF, 0xF, false);
#endif
}
template <int PERM>
__device__ __forceinline__ Fq29 quad_get(const Fq29 &v, uint32_t lane) {
    Fq29 r;
#ifdef H2_HIPEMU
    hipemu_shfl_words<9>(r.l, v.l, (lane & ~3u) | ((PERM >> (2 * (lane & 3u))) & 3u));   // CPU emulation: one rendezvous for nine limbs
#else
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = quad_perm_u32<PERM>(v.l[i], lane);
#endif
    return r;
}
__device__ __forceinline__ Fq29 f29_select(bool c, const Fq29 &x, const Fq29 &y) {   // c ? x : y
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c ? x.l[i] : y.l[i];
    return r;
}
__device__ __forceinline__ uint32_t f29_or_limbs(const Fq29 &v) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= v.l[i];
    return o;
}
// is the quad's point the identity?  (ZZ, held by lane 2 of the quad, is exactly zero)
__device__ __forceinline__ bool quad_is_identity(const Fq29 &c, uint32_t lane) { return quad_perm_u32<QP_BCAST2>(f29_or_limbs(c), lane) == 0; }

// 2*P for the quad's point (3 product levels)
H2_QUAD_FN Fq29 quad_xyzz_double(Fq29 a, uint32_t lane) {
    const uint32_t q = lane & 3u;
    const bool a_id = quad_is_identity(a, lane);
    const Fq29 U = f29_add(a, a);   // lane 1: 2Y (lazy); other lanes: unused but within the limb bounds
    // L1: q0: X*X ; q1: U*U ; (q2, q3: same shape, unused)
    const Fq29 L1 = f29_mul(q == 1 ? U : a, q == 1 ? U : a);
    const Fq29 X2 = quad_get<QP_BCAST0>(L1, lane), V = quad_get<QP_BCAST1>(L1, lane);
    const Fq29 M = f29_norm(f29_add(f29_add(X2, X2), X2));                 // < 3.5 q
    // L2: q0: S = X*V ; q1: W = U*V ; q2: ZZ3 = ZZ*V ; q3: M*M
    const Fq29 L2 = f29_mul(q == 1 ? U : q == 3 ? M : a, q == 3 ? M : V);
    const Fq29 S = quad_get<QP_BCAST0>(L2, lane), W = quad_get<QP_BCAST1>(L2, lane), M2 = quad_get<QP_BCAST3>(L2, lane);
    const Fq29 X3 = f29_sub<3>(M2, f29_norm(f29_add(S, S)));
    // L3: q0: M*(S - X3 + 6q) ; q1: W*Y ; q3: ZZZ3 = W*ZZZ ; (q2 unused)
    const Fq29 L3 = f29_mul(q == 0 ? M : W, q == 0 ? f29_sub<6>(S, X3) : a);
    const Fq29 Y3a = quad_get<QP_BCAST0>(L3, lane);
    const Fq29 Y3 = f29_sub<2>(Y3a, L3);   // lane 1
    Fq29 res = q == 0 ? X3 : q == 1 ? Y3 : q == 2 ? L2 : L3;
    return f29_select(a_id, a, res);
}

// P + Q for two quad-distributed points (4 product levels)
H2_QUAD_FN Fq29 quad_xyzz_add(Fq29 a, Fq29 b, uint32_t lane) {
    const uint32_t q = lane & 3u;
    const bool a_id = quad_is_identity(a, lane), b_id = quad_is_identity(b, lane);
    // L1: q0: U1 = X1*ZZ2 ; q1: U2 = X2*ZZ1 ; q2: S1 = Y1*ZZZ2 ; q3: S2 = Y2*ZZZ1
    const Fq29 pa = quad_get<QP_PAIRS_LO>(a, lane), pb = quad_get<QP_PAIRS_LO>(b, lane);
    const Fq29 ra = quad_get<QP_PAIRS_HI>(a, lane), rb = quad_get<QP_PAIRS_HI>(b, lane);
    const bool odd = (q & 1u) != 0;
    const Fq29 L1 = f29_mul(f29_select(odd, pb, pa), f29_select(odd, ra, rb));
    const Fq29 U1 = quad_get<QP_BCAST0>(L1, lane), U2 = quad_get<QP_BCAST1>(L1, lane), S1 = quad_get<QP_BCAST2>(L1, lane), S2 = quad_get<QP_BCAST3>(L1, lane);
    const Fq29 Pd = f29_sub<2>(U2, U1), Rd = f29_sub<2>(S2, S1);   // in (0.95, 3.05) q
    const bool pz = f29_is_zero_mod_q<3>(Pd), rz = f29_is_zero_mod_q<3>(Rd);
    // L2: q0: PP = Pd^2 ; q1: R2 = Rd^2 ; q2: ZZ1*ZZ2 ; q3: ZZZ1*ZZZ2
    const Fq29 L2 = f29_mul(q == 0 ? Pd : q == 1 ? Rd : a, q == 0 ? Pd : q == 1 ? Rd : b);
    const Fq29 PP = quad_get<QP_BCAST0>(L2, lane), R2 = quad_get<QP_BCAST1>(L2, lane);
    // L3: q0: PPP = Pd*PP ; q1: Q = U1*PP ; q2: ZZ3 = (ZZ1*ZZ2)*PP ; q3: (ZZZ1*ZZZ2)*Pd
    const Fq29 L3 = f29_mul(q == 0 ? Pd : q == 1 ? U1 : L2, q == 3 ? Pd : PP);
    const Fq29 PPP = quad_get<QP_BCAST0>(L3, lane), Q = quad_get<QP_BCAST1>(L3, lane);
    const Fq29 X3 = f29_sub<4>(R2, f29_norm(f29_add(f29_add(PPP, Q), Q)));   // < 5.06 q
    // L4: q0: Rd*(Q - X3 + 6q) ; q1: S1*PPP ; q3: ZZZ3 = ((ZZZ1*ZZZ2)*Pd)*PP ; (q2 unused)
    const Fq29 L4 = f29_mul(q == 0 ? Rd : q == 1 ? S1 : L3, q == 0 ? f29_sub<6>(Q, X3) : q == 1 ? PPP : PP);
    const Fq29 Y3a = quad_get<QP_BCAST0>(L4, lane);
    const Fq29 Y3 = f29_sub<2>(Y3a, L4);   // lane 1
    Fq29 res = q == 0 ? X3 : q == 1 ? Y3 : q == 2 ? L3 : L4;
    // exceptional cases, all quad-uniform
    if (pz) res = Fq29::zero();                    // P = -Q (the P = Q case is patched below)
    res = f29_select(b_id, a, res);
    res = f29_select(a_id, b, res);
    const bool need_dbl = !a_id && !b_id && pz && rz;
    if (__any(need_dbl ? 1 : 0)) {
        const Fq29 d = quad_xyzz_double(a, lane);
        res = f29_select(need_dbl, d, res);
    }
    return res;
}

// k * P by double-and-add over a fixed number of bits (wave-uniform trip count; k may differ between quads)
__device__ __forceinline__ Fq29 quad_xyzz_small_mul(const Fq29 &p, uint32_t k, uint32_t nbits, uint32_t lane) {
    Fq29 r = Fq29::zero();   // identity in every lane
    for (int bit = (int)nbits - 1; bit >= 0; --bit) {
        if (__any(f29_or_limbs(r) ? 1 : 0)) r = quad_xyzz_double(r, lane);   // skipped while every quad still holds the identity
        const bool set = ((k >> bit) & 1u) != 0;
        if (__any(set ? 1 : 0)) {
            const Fq29 t = quad_xyzz_add(r, p, lane);
            r = f29_select(set, t, r);
        }
    }
    return r;
}

// loads / stores of one coordinate of an XYZZ29 stored as 4 consecutive Fq29
__device__ __forceinline__ Fq29 quad_load(const XYZZ29 *p, uint32_t q) { return reinterpret_cast<const Fq29 *>(p)[q]; }
__device__ __forceinline__ void quad_store(XYZZ29 *p, uint32_t q, const Fq29 &v) { reinterpret_cast<Fq29 *>(p)[q] = v; }

}  // namespace h2
