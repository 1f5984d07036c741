// BN254 F_q in an UNSATURATED 9 x 29-bit representation for the MSM's point arithmetic on CDNA4.
//
// Why: with saturated 8 x 32-bit limbs (field.cuh) a Montgomery product is 128+8 multiplies but ~370 further VALU
// slots of carry chains, zero-extension moves and VCC-hazard nops — on gfx950 the multiplies are only ~40 % of its
// issue time.  With 29-bit limbs the 81 partial products of a row-less schoolbook product (and the 81 of the
// interleaved Montgomery reduction) accumulate straight into 64-bit column registers with v_mad_u64_u32 and no
// carry handling at all (9*2^60 + 9*2^58 < 2^64): 171 multiplies + ~57 other instructions, measured 1.745e11
// products/s vs 1.27e11 saturated on MI355X (profiles/archive/r01_modmul_repr.md).
//
// Value semantics: an Fq29 holds an integer v = sum l[i]*2^(29 i) that represents the field element
// v * 2^-261 mod q (Montgomery radix R' = 2^261) and is only WEAKLY reduced: 0 <= v < X*q for a small bound X that
// every routine documents.  "Normalised" (N) means l[i] < 2^29 for i < 8.  Rules:
//   f29_mul / f29_sqr   inputs: limbs <= 2^30 (N, or one lazy add of two N values), X_a*X_b <= 169 ;
//                       output: N, value < (1 + X_a*X_b/169.3) q
//   f29_add             lazy limb-wise sum (no carries)          f29_norm  carry propagation -> N
//   f29_sub<K>          a - b + K*q, requires b N and b < K*q, a limbs <= 2^30; output N, value < a + K*q
// Zero tests compare against the multiples of q inside the documented range.  The bounds used by the point formulas
// (ec29.cuh) close with X < 5.25 q, Y < 3.3 q, ZZ,ZZZ < 1.3 q.  The emulated build (tests/emu) asserts every limb bound.
#pragma once
#include "field.cuh"

#ifdef H2_HIPEMU
#include <assert.h>
#define H2_ASSERT29(c) assert(c)
#else
#define H2_ASSERT29(c) ((void)0)
#endif

namespace h2 {

#include "fq29_constants.inc"   // struct Q29P (G1 coordinate field), struct R29P (scalar field)

constexpr uint32_t MASK29 = (1u << 29) - 1;

template <class P>
struct F29 {
    uint32_t l[9];
    H2_HD static F29 zero() {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = 0;
        return r;
    }
    H2_HD static F29 one() {   // 2^261 mod p
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = P::one(i);
        return r;
    }
    H2_HD bool is_zero_exact() const {   // the integer 0 (only ever assigned, never computed from non-zero elements)
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) o |= l[i];
        return o == 0;
    }
};

using Fq29 = F29<Q29P>;
using Fr29 = F29<R29P>;

template <class P>
H2_HD F29<P> f29_norm(const F29<P> &a) {
    F29<P> r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t v = a.l[i] + carry;
        r.l[i] = v & MASK29;
        carry = v >> 29;
    }
    r.l[8] = a.l[8] + carry;
    return r;
}
template <class P>
H2_HD F29<P> f29_add(const F29<P> &a, const F29<P> &b) {   // lazy
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

#define H2_SUBK_LIMB(K, i) ((K) == 2 ? P::sub2p(i) : (K) == 3 ? P::sub3p(i) : (K) == 4 ? P::sub4p(i) : (K) == 6 ? P::sub6p(i) : P::sub8p(i))
// a - b + K*q, normalised.  b must be N and < K*q.
template <int K, class P>
H2_HD F29<P> f29_sub(const F29<P> &a, const F29<P> &b) {
    static_assert(K == 2 || K == 3 || K == 4 || K == 6 || K == 8, "no constant for this K");
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        H2_ASSERT29(i == 8 || b.l[i] <= MASK29);
        H2_ASSERT29(H2_SUBK_LIMB(K, i) >= b.l[i]);
        r.l[i] = a.l[i] + (H2_SUBK_LIMB(K, i) - b.l[i]);
    }
    return f29_norm(r);
}
// a - b + K*q WITHOUT carry propagation: limbs < 2^29 + 2^30 when a is N (only valid as the wide operand of f29_mul_wide
// or as an operand of a normalising add / sub)
template <int K, class P>
H2_HD F29<P> f29_sub_lazy(const F29<P> &a, const F29<P> &b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        H2_ASSERT29(i == 8 || b.l[i] <= MASK29);
        H2_ASSERT29(H2_SUBK_LIMB(K, i) >= b.l[i]);
        r.l[i] = a.l[i] + (H2_SUBK_LIMB(K, i) - b.l[i]);
    }
    return r;
}
// (neg ? 2q - a : a) - b + 4q, normalised; a, b N with a < 2q and b < 4q.  Limbs stay below 2^31 before the carry pass.
template <class P>
H2_HD F29<P> f29_signed_sub4(const F29<P> &a, bool neg, const F29<P> &b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        H2_ASSERT29(i == 8 || (a.l[i] <= MASK29 && b.l[i] <= MASK29));
        H2_ASSERT29(P::sub2p(i) >= a.l[i] && P::sub4p(i) >= b.l[i]);
        const uint32_t u = neg ? P::sub2p(i) - a.l[i] : a.l[i];
        r.l[i] = u + (P::sub4p(i) - b.l[i]);
    }
    return f29_norm(r);
}
template <int K, class P>
H2_HD F29<P> f29_neg(const F29<P> &b) {   // K*p - b
    return f29_sub<K>(F29<P>::zero(), b);
}

// Montgomery reduction of 18 column sums (radix 2^29) -> normalised 9 limbs
template <class P>
H2_HD F29<P> f29_reduce_columns(uint64_t (&c)[18]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        uint32_t m = ((uint32_t)c[k] * P::INV) & MASK29;
#pragma unroll
        for (int j = 0; j < 9; ++j) c[k + j] += (uint64_t)m * P::p(j);
        c[k + 1] += c[k] >> 29;
    }
    F29<P> r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint64_t v = c[9 + k] + carry;
        r.l[k] = (uint32_t)v & MASK29;
        carry = v >> 29;
    }
    uint64_t top = c[17] + carry;
    H2_ASSERT29(top < (1ull << 29));
    r.l[8] = (uint32_t)top;
    return r;
}
template <class P>
H2_HD F29<P> f29_mul(const F29<P> &a, const F29<P> &b) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        H2_ASSERT29(a.l[i] <= (1u << 30) && b.l[i] <= (1u << 30));
#pragma unroll
        for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
    return f29_reduce_columns<P>(c);
}
// a*b with a "wide" first operand: limbs of a < 2^31 (a lazy sum or lazy difference of normalised values), b normalised
// (limbs < 2^29): a column holds at most 9 * 2^60 plus the reduction's 9 * 2^58 < 2^64.
template <class P>
H2_HD F29<P> f29_mul_wide(const F29<P> &a, const F29<P> &b) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        H2_ASSERT29(a.l[i] < (1u << 31) && b.l[i] <= (1u << 29));
#pragma unroll
        for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
    return f29_reduce_columns<P>(c);
}
// (a*b + c*d) * 2^-261 with ONE Montgomery reduction.  a and d must be normalised (limbs < 2^29); b and c may be LAZY
// differences of normalised values (f29_sub_lazy: limbs < 2^29 + 2^30): a column then holds at most
// 9 * 2^29 * 1.5 * 2^30 + 9 * 2^30 * 2^29 = 11.25 * 2^60 plus the reduction's 9 * 2^58 < 2^64.  X_a*X_b + X_c*X_d <= 169.
template <class P>
H2_HD F29<P> f29_mul2(const F29<P> &a, const F29<P> &b, const F29<P> &c2, const F29<P> &d) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        H2_ASSERT29(a.l[i] < (1u << 29) + (i == 8 ? (1u << 29) : 0) && b.l[i] < (3u << 29) && c2.l[i] <= (1u << 30) && d.l[i] <= (1u << 29));
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            c[i + j] += (uint64_t)a.l[i] * b.l[j];
            c[i + j] += (uint64_t)c2.l[i] * d.l[j];
        }
    }
    return f29_reduce_columns<P>(c);
}
// sum_b a[b]*b_[b] * 2^-261 with ONE Montgomery reduction (T <= 5): every input normalised (limbs < 2^29), so a column holds
// at most 9*T products < 2^58 plus the reduction's 9 * 2^58 (54 * 2^58 < 2^64); sum_b X_a*X_b <= 169.
template <int T, class P>
H2_HD F29<P> f29_dot(const F29<P> (&a)[T], const F29<P> (&b)[T]) {
    static_assert(T >= 1 && T <= 5, "column accumulators would overflow");
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            H2_ASSERT29(a[t].l[i] <= (1u << 29) && b[t].l[i] <= (1u << 29));
#pragma unroll
            for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a[t].l[i] * b[t].l[j];
        }
    }
    return f29_reduce_columns<P>(c);
}
template <class P>
H2_HD F29<P> f29_sqr(const F29<P> &a) {
    uint64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) c[k] = 0;
    uint32_t a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        H2_ASSERT29(a.l[i] <= (1u << 30));
        a2[i] = a.l[i] << 1;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < 9; ++j) c[i + j] += (uint64_t)a.l[i] * a2[j];
    }
    return f29_reduce_columns<P>(c);
}

// Weak reduction without a multiplication: N value v < 32 p  ->  N value < 2 p representing the same residue.
// q_est = floor(v8 * floor(2^43 / (p8 + 1)) / 2^43) never exceeds floor(v / p) and is at most one below it.
template <class P>
H2_HD F29<P> f29_weak_reduce(const F29<P> &a) {
    constexpr uint64_t D = (uint64_t)P::p(8) + 1;
    constexpr uint64_t M = ((uint64_t)1 << 43) / D;
    H2_ASSERT29(a.l[8] < (1u << 27));
    const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * M) >> 43);
    F29<P> r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int64_t v = (int64_t)a.l[i] - (int64_t)((uint64_t)q * P::p(i)) + carry;
        r.l[i] = i < 8 ? (uint32_t)(v & MASK29) : (uint32_t)v;
        carry = v >> 29;   // arithmetic shift: borrows propagate as negative carries
    }
    return r;
}

// a == k*q for some 0 <= k <= KMAX ?   (a must be N; use when a < (KMAX+1)*q)
#define H2_MULK_LIMB(k, i) ((k) == 0 ? 0u : (k) == 1 ? P::mul1p(i) : (k) == 2 ? P::mul2p(i) : (k) == 3 ? P::mul3p(i) : (k) == 4 ? P::mul4p(i) : (k) == 5 ? P::mul5p(i) : (k) == 6 ? P::mul6p(i) : (k) == 7 ? P::mul7p(i) : P::mul8p(i))
// The candidate multiple comes from the lowest limb alone (a = k*q  =>  k = a_0 * q_0^-1 mod 2^29: three instructions);
// the exact limb-by-limb comparison sits behind a WAVE-UNIFORM branch, so the common case really skips it (a per-lane
// `if` of this size is if-converted by the compiler and was ~150 always-executed instructions per point addition).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_HIPEMU)
#define H2_ANY_LANE(c) (__any((c) ? 1 : 0) != 0)
#else
#define H2_ANY_LANE(c) (c)
#endif
template <int KMAX, class P>
H2_HD bool f29_is_zero_mod_q(const F29<P> &a) {
    static_assert(KMAX <= 8, "range too large");
    const uint32_t kc = (0u - a.l[0] * P::INV) & MASK29;   // P::INV = -q^-1 mod 2^29
    const bool cand = kc <= (uint32_t)KMAX;
    if (!H2_ANY_LANE(cand)) return false;
    bool hit = false;
#pragma unroll
    for (int k = 0; k <= KMAX; ++k) {
        uint32_t d = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) d |= a.l[i] ^ H2_MULK_LIMB(k, i);
        hit |= (d == 0);
    }
    return cand && hit;
}

// saturated Montgomery (R = 2^256, canonical < q)  ->  unsaturated (R' = 2^261), N, value < 1.01 q
// raw limb split of a saturated element (no domain change: the integer is unchanged)
template <class P, class PS>
H2_HD F29<P> f29_split(const Fe<PS> &s) {
    F29<P> t;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 5, off = bit & 31;
        uint64_t lo = w < 8 ? s.l[w] : 0, hi = (w + 1) < 8 ? s.l[w + 1] : 0;
        t.l[i] = (uint32_t)(((hi << 32) | lo) >> off) & MASK29;
    }
    return t;
}
// N value < 2p (any integer below 2p) -> canonical (< p) saturated limbs
template <class PS, class P>
H2_HD Fe<PS> f29_pack_canonical(const F29<P> &t) {
    uint32_t d[9], borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t x = t.l[i] - P::p(i) - borrow;
        borrow = (x >> 31) & 1u;            // limbs are < 2^29, so a wrap sets the top bit
        d[i] = i < 8 ? (x & MASK29) : x;
    }
    uint32_t r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = borrow ? t.l[i] : d[i];
    Fe<PS> out;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        // word w covers bits [32w, 32w+32): limbs floor(32w/29) and the next one or two
        const int lo_limb = (32 * w) / 29, off = 32 * w - 29 * lo_limb;
        uint64_t v = (uint64_t)r[lo_limb] >> off;
        int have = 29 - off;
        if (lo_limb + 1 < 9) v |= (uint64_t)r[lo_limb + 1] << have;
        if (have + 29 < 32 && lo_limb + 2 < 9) v |= (uint64_t)r[lo_limb + 2] << (have + 29);
        out.l[w] = (uint32_t)v;
    }
    return out;
}

// N value below 2^256 (e.g. < 2 p) -> saturated limbs of the SAME integer, not reduced: for buffers only this library reads back (the NTT's
// inter-pass scratch: the next pass splits it again, bounds in ntt.hip) — f29_pack_canonical without the trial subtraction (~36 instructions)
template <class PS, class P>
H2_HD Fe<PS> f29_pack_weak(const F29<P> &t) {
    Fe<PS> out;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int lo_limb = (32 * w) / 29, off = 32 * w - 29 * lo_limb;
        H2_ASSERT29(lo_limb == 8 || t.l[lo_limb] <= MASK29);
        uint64_t v = (uint64_t)t.l[lo_limb] >> off;
        int have = 29 - off;
        if (lo_limb + 1 < 9) v |= (uint64_t)t.l[lo_limb + 1] << have;
        if (have + 29 < 32 && lo_limb + 2 < 9) v |= (uint64_t)t.l[lo_limb + 2] << (have + 29);
        out.l[w] = (uint32_t)v;
    }
    H2_ASSERT29(t.l[8] < (1u << 24));   // the value fits 256 bits
    return out;
}

H2_HD Fq29 f29_from_sat(const Fq &s) {
    Fq29 k;
#pragma unroll
    for (int i = 0; i < 9; ++i) k.l[i] = Q29P::conv_in(i);
    return f29_mul(f29_split<Q29P>(s), k);
}
// unsaturated (any documented bound) -> saturated Montgomery R = 2^256, canonical
H2_HD Fq f29_to_sat(const Fq29 &v) {
    Fq29 k;
#pragma unroll
    for (int i = 0; i < 9; ++i) k.l[i] = Q29P::conv_out(i);
    return f29_pack_canonical<FqP>(f29_mul(v, k));   // the product is N and < 1.04 q: at most one subtraction of q
}

// the same conversions for the scalar field (Poseidon batches, NTT tables)
H2_HD Fr29 fr29_from_sat(const Fr &s) {   // saturated Montgomery (x*2^256) -> R' = 2^261 form, N, < 1.01 r
    Fr29 k;
#pragma unroll
    for (int i = 0; i < 9; ++i) k.l[i] = R29P::conv_in(i);
    return f29_mul(f29_split<R29P>(s), k);
}
H2_HD Fr fr29_to_sat(const Fr29 &v) {
    Fr29 k;
#pragma unroll
    for (int i = 0; i < 9; ++i) k.l[i] = R29P::conv_out(i);
    return f29_pack_canonical<FrP>(f29_mul(v, k));
}

}  // namespace h2
