// Modular inversion by Bernstein-Yang division steps ("safegcd", eprint 2019/266) for the two BN254 prime fields.
//
// Why: Fermat's a^(p-2) is a chain of ~380 DEPENDENT Montgomery products (~140k VALU instructions on one lane, 0.3 ms of
// pure latency) and sits on the critical path of every batch inversion (one inversion per lane / per run), of the affine
// conversion of a result and of the table normalisation.  Division steps need 25 batches of 30 branch-free steps on the
// low limbs plus four small-by-big products per batch: ~19k instructions, 7x less, with no field multiplications.
//
// Representation: nine signed 30-bit limbs (value = sum v[i] * 2^(30 i), limbs 0..7 in [0, 2^30), limb 8 signed).
// State (delta, f, g, d, e) with f = p, g = x, d = 0, e = 1; invariants d*x = f and e*x = g (mod p).  A batch computes
// the transition matrix t = [[u, v], [q, r]] of 30 division steps from the low 30 bits of f and g (|entries| <= 2^30,
// t * [f, g] = 2^30 * [f', g']), applies it exactly to (f, g) and modulo p to (d, e) (a multiple of p makes the
// numerator divisible by 2^30).  741 steps suffice for 256-bit inputs (the paper's bound floor((49*256+57)/17)); after
// 750, g = 0 and f = +-1, so x^-1 = +-d.  |d|, |e| grow by at most p per batch (< 27 p at the end: 259 bits of the 270).
// 0 maps to 0 (the `invert().unwrap_or(zero)` convention of halo2's BatchInvert).
#pragma once
#include "field.cuh"

namespace h2 {

struct S9 {
    int32_t v[9];
};
constexpr int32_t M30 = (1 << 30) - 1;

template <class P>
H2_HD S9 s9_from_limbs32(const uint32_t (&l)[8]) {
    S9 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 30 * i, w = bit >> 5, off = bit & 31;
        uint64_t lo = w < 8 ? l[w] : 0, hi = (w + 1) < 8 ? l[w + 1] : 0;
        r.v[i] = (int32_t)((((hi << 32) | lo) >> off) & (uint64_t)M30);
    }
    return r;
}
template <class P>
H2_HD S9 s9_modulus() {
    uint32_t m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = P::m(i);
    return s9_from_limbs32<P>(m);
}

// 30 division steps on the low bits; returns the new delta and the transition matrix
H2_HD int32_t divsteps30(int32_t delta, uint32_t f, uint32_t g, int32_t &u, int32_t &v, int32_t &q, int32_t &r) {
    u = 1;
    v = 0;
    q = 0;
    r = 1;
#pragma unroll 1
    for (int i = 0; i < 30; ++i) {
        const uint32_t godd = 0u - (g & 1u);                          // all ones when g is odd
        const uint32_t swap = godd & (0u - (uint32_t)(delta > 0));   // g odd and delta > 0
        // (delta, f, g, u, v, q, r) <- (-delta, g, -f, q, r, -u, -v) under `swap`
        const uint32_t nf = (f & ~swap) | (g & swap);
        const uint32_t ng = (g & ~swap) | ((0u - f) & swap);
        const int32_t nu = (int32_t)(((uint32_t)u & ~swap) | ((uint32_t)q & swap));
        const int32_t nv = (int32_t)(((uint32_t)v & ~swap) | ((uint32_t)r & swap));
        const int32_t nq = (int32_t)(((uint32_t)q & ~swap) | ((0u - (uint32_t)u) & swap));
        const int32_t nr = (int32_t)(((uint32_t)r & ~swap) | ((0u - (uint32_t)v) & swap));
        delta = (int32_t)((((uint32_t)delta) & ~swap) | ((0u - (uint32_t)delta) & swap));
        f = nf;
        // g is odd here exactly when it was odd before (f is always odd): add f to clear the low bit
        g = ng + (nf & godd);
        q = nq + (int32_t)((uint32_t)nu & godd);
        r = nr + (int32_t)((uint32_t)nv & godd);
        delta += 1;
        g >>= 1;
        u = (int32_t)((uint32_t)nu << 1);
        v = (int32_t)((uint32_t)nv << 1);
    }
    return delta;
}

// (f, g) <- t * (f, g) / 2^30, exact
H2_HD void s9_update_fg(S9 &f, S9 &g, int32_t u, int32_t v, int32_t q, int32_t r) {
    int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0];
    int64_t cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cf += (int64_t)u * f.v[i] + (int64_t)v * g.v[i];
        cg += (int64_t)q * f.v[i] + (int64_t)r * g.v[i];
        f.v[i - 1] = (int32_t)cf & M30;
        g.v[i - 1] = (int32_t)cg & M30;
        cf >>= 30;
        cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}
// (d, e) <- t * (d, e) / 2^30 mod p: the multiple of p that clears the low 30 bits is chosen with -p^-1 mod 2^30
template <class P>
H2_HD void s9_update_de(S9 &d, S9 &e, int32_t u, int32_t v, int32_t q, int32_t r, const S9 &p) {
    constexpr uint32_t NEG_PINV = P::INV & (uint32_t)M30;   // P::INV = -p^-1 mod 2^32
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    const int32_t md = (int32_t)(((uint32_t)cd * NEG_PINV) & (uint32_t)M30);
    const int32_t me = (int32_t)(((uint32_t)ce * NEG_PINV) & (uint32_t)M30);
    cd += (int64_t)md * p.v[0];
    ce += (int64_t)me * p.v[0];
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)md * p.v[i];
        ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)me * p.v[i];
        d.v[i - 1] = (int32_t)cd & M30;
        e.v[i - 1] = (int32_t)ce & M30;
        cd >>= 30;
        ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}

// x^-1 mod p for the INTEGER x < p given as 8 x 32-bit limbs (no Montgomery factor involved); 0 -> 0
template <class P>
H2_HD void modinv_limbs32(const uint32_t (&x)[8], uint32_t (&out)[8]) {
    const S9 p = s9_modulus<P>();
    S9 f = p, g = s9_from_limbs32<P>(x), d, e;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        d.v[i] = 0;
        e.v[i] = 0;
    }
    e.v[0] = 1;
    int32_t delta = 1;
#pragma unroll 1
    for (int batch = 0; batch < 25; ++batch) {
        int32_t u, v, q, r;
        delta = divsteps30(delta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), u, v, q, r);
        s9_update_fg(f, g, u, v, q, r);
        s9_update_de<P>(d, e, u, v, q, r, p);
    }
    // f = +-1 (or +-p when x = 0, where d = 0 anyway): x^-1 = sign(f) * d; bring it from (-27 p, 27 p) into [0, p)
    const int32_t fneg = f.v[8] >> 31;   // all ones when f < 0
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {        // d <- sign(f) * d + 32 p  (positive, < 59 p)
        c += (int64_t)((d.v[i] ^ fneg) - fneg) + ((int64_t)p.v[i] << 5);
        d.v[i] = i < 8 ? ((int32_t)c & M30) : (int32_t)c;
        if (i < 8) c >>= 30;
    }
#pragma unroll 1
    for (int k = 5; k >= 0; --k) {       // subtract 32 p, 16 p, ..., p while the result stays non-negative
        S9 t;
        int64_t b = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            b += (int64_t)d.v[i] - ((int64_t)p.v[i] << k);
            t.v[i] = i < 8 ? ((int32_t)b & M30) : (int32_t)b;
            if (i < 8) b >>= 30;
        }
        const int32_t keep = t.v[8] >> 31;   // negative: keep d
#pragma unroll
        for (int i = 0; i < 9; ++i) d.v[i] = (d.v[i] & keep) | (t.v[i] & ~keep);
    }
    // 9 x 30 -> 8 x 32
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int bit = 32 * w, i = bit / 30, off = bit % 30;
        uint64_t acc = (uint64_t)(uint32_t)d.v[i] >> off;
        acc |= (uint64_t)(uint32_t)d.v[i + 1] << (30 - off);
        if (i + 2 < 9) acc |= (uint64_t)(uint32_t)d.v[i + 2] << (60 - off);
        out[w] = (uint32_t)acc;
    }
}

}  // namespace h2
