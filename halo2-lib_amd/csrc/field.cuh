// BN254 prime-field arithmetic for CDNA4 (gfx950) — F_r (scalars) and F_q (G1 coordinates).
//
// Memory format is bit-identical to halo2curves' in-memory Fr/Fq: 4 x u64 little-endian limbs in
// Montgomery form, R = 2^256 (reference: halo2-base/src/utils/mod.rs:28-38,342-377 rely on the [u64;4]
// LE view).  On the device an element is 8 x u32 limbs (same bytes).
//
// The multiplier is integer, not MFMA: one Montgomery product = 128 v_mad_u64_u32 + 8 v_mul_lo_u32
// (quarter-rate VALU) plus carry chains; rows are "lazy-carry" so the 8 MADs of a row are independent.
#pragma once
#include <utility>
#include <stdint.h>

#ifndef H2_HD
#define H2_HD __host__ __device__ __forceinline__
#endif

namespace h2 {

struct FrP {   // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    static constexpr uint32_t INV = 0xefffffffu;   // -r^-1 mod 2^32
    H2_HD static constexpr uint32_t m(int i) {
        constexpr uint32_t v[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    H2_HD static constexpr uint32_t r1(int i) {   // R mod r  (Montgomery one)
        constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    H2_HD static constexpr uint32_t r2(int i) {   // R^2 mod r
        constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return v[i];
    }
};
struct FqP {   // q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    static constexpr uint32_t INV = 0xe4866389u;
    H2_HD static constexpr uint32_t m(int i) {
        constexpr uint32_t v[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    H2_HD static constexpr uint32_t r1(int i) {
        constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    H2_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return v[i];
    }
};

H2_HD uint32_t addc32(uint32_t a, uint32_t b, unsigned &c) {
    unsigned co;
    uint32_t r = __builtin_addc(a, b, c, &co);
    c = co;
    return r;
}
H2_HD uint32_t subb32(uint32_t a, uint32_t b, unsigned &br) {
    unsigned bo;
    uint32_t r = __builtin_subc(a, b, br, &bo);
    br = bo;
    return r;
}

template <class P>
struct alignas(16) Fe {
    uint32_t l[8];

    H2_HD static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = 0;
        return r;
    }
    H2_HD static Fe one() {   // Montgomery 1
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = P::r1(i);
        return r;
    }
    H2_HD static Fe r2() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = P::r2(i);
        return r;
    }
    H2_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= l[i];
        return o == 0;
    }
    H2_HD bool operator==(const Fe &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    H2_HD bool operator!=(const Fe &b) const { return !(*this == b); }
};

// r = a + b mod p   (inputs < p; p < 2^254 so the raw sum never carries out of 256 bits)
template <class P>
H2_HD Fe<P> fe_add(const Fe<P> &a, const Fe<P> &b) {
    uint32_t t[8], s[8];
    unsigned c = 0, br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = addc32(a.l[j], b.l[j], c);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = subb32(t[j], P::m(j), br);
    Fe<P> r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.l[j] = br ? t[j] : s[j];
    return r;
}
template <class P>
H2_HD Fe<P> fe_sub(const Fe<P> &a, const Fe<P> &b) {
    uint32_t t[8];
    unsigned br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = subb32(a.l[j], b.l[j], br);
    uint32_t mask = 0u - br;
    Fe<P> r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.l[j] = addc32(t[j], P::m(j) & mask, c);
    return r;
}
template <class P>
H2_HD Fe<P> fe_neg(const Fe<P> &a) {
    return fe_sub(Fe<P>::zero(), a);
}
template <class P>
H2_HD Fe<P> fe_dbl(const Fe<P> &a) {
    return fe_add(a, a);
}

// Montgomery product a*b*R^-1 mod p.  CIOS by rows; inside a row the 8 partial products
// P_j = a_j*b_i + t_j are independent 64-bit MADs and one 9-word carry chain folds hi(P_{j-1}) into lo(P_j).
template <class P>
H2_HD Fe<P> fe_mul(const Fe<P> &a, const Fe<P> &b) {
    uint32_t t[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (uint64_t)a.l[j] * b.l[i] + t[j];
        unsigned c = 0;
        t[0] = (uint32_t)p[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) t[j] = addc32((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c);
        t[8] = addc32(t[8], (uint32_t)(p[7] >> 32), c);
        uint32_t t9 = c;
        uint32_t m = t[0] * P::INV;
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (uint64_t)m * P::m(j) + t[j];
        c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) t[j - 1] = addc32((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c);
        t[7] = addc32(t[8], (uint32_t)(p[7] >> 32), c);
        t[8] = t9 + c;
    }
    uint32_t s[8];
    unsigned br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = subb32(t[j], P::m(j), br);
    bool ge = (t[8] != 0) || (br == 0);
    Fe<P> r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.l[j] = ge ? s[j] : t[j];
    return r;
}
template <class P>
H2_HD Fe<P> fe_sqr(const Fe<P> &a) {
    return fe_mul(a, a);
}
// Montgomery -> canonical integer (a * R^-1): 8 reduction rows only.
template <class P>
H2_HD Fe<P> fe_from_mont(const Fe<P> &a) {
    uint32_t t[9];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = a.l[j];
    t[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t p[8];
        uint32_t m = t[0] * P::INV;
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (uint64_t)m * P::m(j) + t[j];
        unsigned c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) t[j - 1] = addc32((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c);
        t[7] = addc32(t[8], (uint32_t)(p[7] >> 32), c);
        t[8] = c;
    }
    uint32_t s[8];
    unsigned br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = subb32(t[j], P::m(j), br);
    bool ge = (t[8] != 0) || (br == 0);
    Fe<P> r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.l[j] = ge ? s[j] : t[j];
    return r;
}
template <class P>
H2_HD Fe<P> fe_to_mont(const Fe<P> &a) {
    return fe_mul(a, Fe<P>::r2());
}

// a^e for a 256-bit exponent given as 8 x u32 (little endian), plain square-and-multiply (MSB first).
template <class P>
H2_HD Fe<P> fe_pow(const Fe<P> &a, const uint32_t (&e)[8]) {
    Fe<P> acc = Fe<P>::one();
    bool started = false;
    for (int i = 255; i >= 0; --i) {
        if (started) acc = fe_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) {
            acc = started ? fe_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}
// a^(2^k-ish small exponent): used for twiddle generation
template <class P>
H2_HD Fe<P> fe_pow_u64(const Fe<P> &a, uint64_t e) {
    Fe<P> acc = Fe<P>::one();
    Fe<P> base = a;
    while (e) {
        if (e & 1) acc = fe_mul(acc, base);
        e >>= 1;
        if (e) base = fe_sqr(base);
    }
    return acc;
}
// a^-1 = a^(p-2) (Fermat); 0 -> 0.  Kept as the independent cross-check of fe_inv (tests) — the product path inverts
// with division steps (modinv.cuh), which needs 7x fewer instructions than this chain of ~380 dependent products.
template <class P>
H2_HD Fe<P> fe_inv_fermat(const Fe<P> &a) {
    uint32_t e[8];
    unsigned br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = subb32(P::m(j), j == 0 ? 2u : 0u, br);
    return fe_pow(a, e);
}

using Fr = Fe<FrP>;
using Fq = Fe<FqP>;

}  // namespace h2

#include "modinv.cuh"

namespace h2 {

// a^-1 in Montgomery form; 0 -> 0 (the halo2 `invert().unwrap_or(zero)` convention of BatchInvert).  The integer held
// in the limbs is a = X*R; its modular inverse is X^-1 * R^-1, and one Montgomery product with R^3 turns that into X^-1 * R.
template <class P>
H2_HD Fe<P> fe_inv(const Fe<P> &a) {
    Fe<P> r;
    modinv_limbs32<P>(a.l, r.l);
    return fe_mul(r, fe_mul(Fe<P>::r2(), Fe<P>::r2()));
}

// A compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}).  `#pragma unroll` is a request the
// compiler turns down when the unrolled body exceeds its size threshold (a few field multiplications per iteration do) — the loop then stays
// rolled and every array indexed by its counter moves to scratch memory; a pack expansion leaves it no choice.
template <class F, int... Js>
H2_HD void static_for_impl(F &&f, std::integer_sequence<int, Js...>) {
    (f(std::integral_constant<int, Js>{}), ...);
}
template <int N, class F>
H2_HD void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

}  // namespace h2
