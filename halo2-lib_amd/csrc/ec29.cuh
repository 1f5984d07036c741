// BN254 G1 point arithmetic over the unsaturated field representation (fq29.cuh) — what the MSM kernels run on.
// Same formulas as ec.cuh (XYZZ: madd-2008-s, add-2008-s, dbl-2008-s-1, all exceptional cases explicit), with the
// weak-reduction bounds carried in comments: accumulator coordinates always satisfy
//     X < 5.25 q,  Y < 3.3 q,  ZZ, ZZZ < 1.3 q      (all normalised)
// and table points (x, y) are < 1.05 q.  Every product's bound is (1 + Xa*Xb/169.3) q.
#pragma once
#include "ec.cuh"
#include "fq29.cuh"

namespace h2 {

struct alignas(16) G1Affine29 {   // 80 B: 18 limbs + pad, so one lane loads it with five 16-byte loads
    Fq29 x, y;
    uint32_t pad[2];
    H2_HD bool is_identity() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) o |= x.l[i] | y.l[i];
        return o == 0;
    }
};

struct alignas(16) XYZZ29 {   // 144 B
    Fq29 x, y, zz, zzz;
    H2_HD static XYZZ29 identity() {
        XYZZ29 r;
        r.x = Fq29::zero();
        r.y = Fq29::zero();
        r.zz = Fq29::zero();
        r.zzz = Fq29::zero();
        return r;
    }
    H2_HD bool is_identity() const { return zz.is_zero_exact(); }
};

// 2*(x, y), affine non-identity input with x, y < 2 q (N)
H2_HD XYZZ29 xyzz29_double_affine(const Fq29 &x, const Fq29 &y) {
    XYZZ29 r;
    Fq29 U = f29_add(y, y);                                   // lazy, < 4 q
    Fq29 V = f29_sqr(U);                                      // < 1.1
    Fq29 W = f29_mul(U, V);                                   // < 1.03
    Fq29 S = f29_mul(x, V);                                   // < 1.02
    Fq29 X2 = f29_sqr(x);                                     // < 1.03
    Fq29 M = f29_norm(f29_add(f29_add(X2, X2), X2));          // < 3.1
    Fq29 X3 = f29_sub<3>(f29_sqr(M), f29_norm(f29_add(S, S)));   // 1.06 + 3 = 4.06
    r.x = X3;
    r.y = f29_mul2(M, f29_sub<6>(S, X3), W, f29_neg<2>(y));   // y < 2 q
    r.zz = V;
    r.zzz = W;
    return r;
}

H2_HD XYZZ29 xyzz29_double(const XYZZ29 &p) {
    if (p.is_identity()) return p;
    XYZZ29 r;
    Fq29 U = f29_add(p.y, p.y);                               // lazy, < 6.6 q
    Fq29 V = f29_sqr(U);                                      // 43.6 -> < 1.26
    Fq29 W = f29_mul(U, V);                                   // < 1.05
    Fq29 S = f29_mul(p.x, V);                                 // < 1.04
    Fq29 X2 = f29_sqr(p.x);                                   // 27.6 -> < 1.17
    Fq29 M = f29_norm(f29_add(f29_add(X2, X2), X2));          // < 3.5
    Fq29 X3 = f29_sub<3>(f29_sqr(M), f29_norm(f29_add(S, S)));   // 1.08 + 3 = 4.08 (2S < 2.08 < 3)
    r.x = X3;
    r.y = f29_mul2(M, f29_sub<6>(S, X3), W, f29_neg<4>(p.y));   // M*(S - X3 + 6q) + W*(4q - Y): 24.6 + 4.2 -> < 1.18 q
    r.zz = f29_mul(V, p.zz);
    r.zzz = f29_mul(W, p.zzz);
    return r;
}

// acc += (+/-)(x2, y2), affine non-identity table point
H2_HD void xyzz29_add_affine(XYZZ29 &acc, const Fq29 &x2, const Fq29 &y2, bool neg) {
    if (acc.is_identity()) {
        acc.x = x2;
        acc.y = neg ? f29_neg<2>(y2) : y2;   // < 2 q
        acc.zz = Fq29::one();
        acc.zzz = Fq29::one();
        return;
    }
    Fq29 U2 = f29_mul(x2, acc.zz);                            // < 1.01
    Fq29 S2 = f29_mul(y2, acc.zzz);                           // < 1.01
    Fq29 Pd = f29_sub<6>(U2, acc.x);                          // in (0.75, 7.01) q
    // Rd = +-S2 - Y1 + K*q with the sign folded in limb-wise (one instruction stream for both signs of the digit):
    //   neg: (2q - S2) + (4q - Y1) in (1.69, 6] q ;  otherwise: S2 + (4q - Y1) in (0.7, 5.01) q
    Fq29 Rd = f29_signed_sub4(S2, neg, acc.y);
    if (f29_is_zero_mod_q<7>(Pd)) {
        if (f29_is_zero_mod_q<7>(Rd)) acc = xyzz29_double_affine(x2, neg ? f29_neg<2>(y2) : y2);
        else acc = XYZZ29::identity();
        return;
    }
    Fq29 PP = f29_sqr(Pd);                                    // 49 -> < 1.29
    Fq29 PPP = f29_mul(Pd, PP);                               // < 1.054
    Fq29 Q = f29_mul(acc.x, PP);                              // < 1.04
    Fq29 R2 = f29_sqr(Rd);                                    // 36 -> < 1.22
    Fq29 t = f29_norm(f29_add(f29_add(PPP, Q), Q));           // < 3.14
    Fq29 X3 = f29_sub<4>(R2, t);                              // < 5.22
    // Y3 = Rd*(Q - X3) - Y1*PPP as ONE reduction: Rd*(Q - X3 + 6q) + (4q - Y1)*PPP   (42.2 + 4.3 -> < 1.28 q)
    Fq29 Y3 = f29_mul2(Rd, f29_sub_lazy<6>(Q, X3), f29_sub_lazy<4>(Fq29::zero(), acc.y), PPP);   // lazy operands: no carry passes
    acc.x = X3;
    acc.y = Y3;
    acc.zz = f29_mul(acc.zz, PP);
    acc.zzz = f29_mul(acc.zzz, PPP);
}

// The same addition for an accumulator whose emptiness is carried in a flag instead of being read off ZZ (the MSM's inner loop: one
// instruction per step instead of nine ORs and a compare, and closing a run costs no 36-register reset).  `empty`: acc holds nothing
// (its registers are stale); set when a run cancels to the identity, cleared by the first addition.
H2_HD void xyzz29_add_affine_flag(XYZZ29 &acc, bool &empty, const Fq29 &x2, const Fq29 &y2, bool neg) {
    if (empty) {
        acc.x = x2;
        acc.y = neg ? f29_neg<2>(y2) : y2;   // < 2 q
        acc.zz = Fq29::one();
        acc.zzz = Fq29::one();
        empty = false;
        return;
    }
    Fq29 U2 = f29_mul(x2, acc.zz);                            // < 1.01
    Fq29 S2 = f29_mul(y2, acc.zzz);                           // < 1.01
    Fq29 Pd = f29_sub<6>(U2, acc.x);                          // in (0.75, 7.01) q
    Fq29 Rd = f29_signed_sub4(S2, neg, acc.y);
    if (f29_is_zero_mod_q<7>(Pd)) {
        if (f29_is_zero_mod_q<7>(Rd)) acc = xyzz29_double_affine(x2, neg ? f29_neg<2>(y2) : y2);
        else empty = true;
        return;
    }
    Fq29 PP = f29_sqr(Pd);                                    // 49 -> < 1.29
    Fq29 PPP = f29_mul(Pd, PP);                               // < 1.054
    Fq29 Q = f29_mul(acc.x, PP);                              // < 1.04
    Fq29 R2 = f29_sqr(Rd);                                    // 36 -> < 1.22
    Fq29 t = f29_norm(f29_add(f29_add(PPP, Q), Q));           // < 3.14
    Fq29 X3 = f29_sub<4>(R2, t);                              // < 5.22
    Fq29 Y3 = f29_mul2(Rd, f29_sub_lazy<6>(Q, X3), f29_sub_lazy<4>(Fq29::zero(), acc.y), PPP);   // lazy operands: no carry passes
    acc.x = X3;
    acc.y = Y3;
    acc.zz = f29_mul(acc.zz, PP);
    acc.zzz = f29_mul(acc.zzz, PPP);
}

// acc += b
H2_HD void xyzz29_add(XYZZ29 &acc, const XYZZ29 &b) {
    if (b.is_identity()) return;
    if (acc.is_identity()) {
        acc = b;
        return;
    }
    Fq29 U1 = f29_mul(acc.x, b.zz);                           // 6.8 -> < 1.05
    Fq29 U2 = f29_mul(b.x, acc.zz);
    Fq29 S1 = f29_mul(acc.y, b.zzz);                          // < 1.03
    Fq29 S2 = f29_mul(b.y, acc.zzz);
    Fq29 Pd = f29_sub<2>(U2, U1);                             // in (0.95, 3.05) q
    Fq29 Rd = f29_sub<2>(S2, S1);                             // in (0.97, 3.03) q
    if (f29_is_zero_mod_q<3>(Pd)) {
        if (f29_is_zero_mod_q<3>(Rd)) acc = xyzz29_double(acc);
        else acc = XYZZ29::identity();
        return;
    }
    Fq29 PP = f29_sqr(Pd);                                    // < 1.06
    Fq29 PPP = f29_mul(Pd, PP);                               // < 1.02
    Fq29 Q = f29_mul(U1, PP);                                 // < 1.01
    Fq29 R2 = f29_sqr(Rd);                                    // < 1.06
    Fq29 t = f29_norm(f29_add(f29_add(PPP, Q), Q));           // < 3.04
    Fq29 X3 = f29_sub<4>(R2, t);                              // < 5.06
    Fq29 Y3 = f29_mul2(Rd, f29_sub<6>(Q, X3), f29_neg<2>(S1), PPP);   // 21.3 + 2*1.02 -> < 1.14 q
    acc.x = X3;
    acc.y = Y3;
    acc.zz = f29_mul(f29_mul(acc.zz, b.zz), PP);
    acc.zzz = f29_mul(f29_mul(acc.zzz, b.zzz), PPP);
}

H2_HD G1Affine29 g1affine29_from_sat(const G1Affine &p) {
    G1Affine29 r;
    r.pad[0] = r.pad[1] = 0;
    if (p.is_identity()) {
        r.x = Fq29::zero();
        r.y = Fq29::zero();
    } else {
        r.x = f29_from_sat(p.x);
        r.y = f29_from_sat(p.y);
    }
    return r;
}
// back to the saturated representation (canonical coordinates), e.g. for the final Jacobian/affine conversion
H2_HD XYZZ xyzz29_to_sat(const XYZZ29 &p) {
    XYZZ r;
    if (p.is_identity()) return XYZZ::identity();
    r.x = f29_to_sat(p.x);
    r.y = f29_to_sat(p.y);
    r.zz = f29_to_sat(p.zz);
    r.zzz = f29_to_sat(p.zzz);
    return r;
}

}  // namespace h2
