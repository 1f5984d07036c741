// BN254 G1 (y^2 = x^3 + 3 over F_q) point arithmetic for the MSM kernels.
//
// Affine points use halo2curves' layout G1Affine{x,y} (64 B, Montgomery limbs, identity = (0,0);
// reference: halo2-ecc/benches/msm.rs:62, halo2-ecc/src/ecc/pippenger.rs:217).  Accumulators are
// extended-Jacobian XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity: ZZ = 0): mixed add 8M+2S,
// full add 12M+2S, double 6M+4S (a = 0).  Every exceptional case (either operand identity, P = Q, P = -Q)
// is handled explicitly — duplicate bases / sums to infinity are reference test cases
// (halo2-ecc/src/bn254/tests/msm_sum_infinity.rs:16-69).
#pragma once
#include "field.cuh"

namespace h2 {

struct alignas(16) G1Affine {
    Fq x, y;
    H2_HD bool is_identity() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= x.l[i] | y.l[i];
        return o == 0;
    }
};

struct alignas(16) G1Jac {   // what best_multiexp returns (C::Curve): x = X/Z^2, y = Y/Z^3, identity Z = 0
    Fq x, y, z;
};

struct alignas(16) XYZZ {
    Fq x, y, zz, zzz;
    H2_HD static XYZZ identity() {
        XYZZ r;
        r.x = Fq::zero();
        r.y = Fq::zero();
        r.zz = Fq::zero();
        r.zzz = Fq::zero();
        return r;
    }
    H2_HD bool is_identity() const { return zz.is_zero(); }
    H2_HD static XYZZ from_affine(const G1Affine &p) {
        XYZZ r;
        if (p.is_identity()) return identity();
        r.x = p.x;
        r.y = p.y;
        r.zz = Fq::one();
        r.zzz = Fq::one();
        return r;
    }
};

// 2*(x,y) for an affine, non-identity point (mdbl-2008-s-1)
H2_HD XYZZ xyzz_double_affine(const Fq &x, const Fq &y) {
    XYZZ r;
    Fq U = fe_dbl(y);
    Fq V = fe_sqr(U);
    Fq W = fe_mul(U, V);
    Fq S = fe_mul(x, V);
    Fq X2 = fe_sqr(x);
    Fq M = fe_add(fe_dbl(X2), X2);
    r.x = fe_sub(fe_sqr(M), fe_dbl(S));
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2*P (dbl-2008-s-1, a = 0).  y = 0 cannot occur on BN254 G1 (odd prime order), so U != 0 for P != identity.
H2_HD XYZZ xyzz_double(const XYZZ &p) {
    if (p.is_identity()) return p;
    XYZZ r;
    Fq U = fe_dbl(p.y);
    Fq V = fe_sqr(U);
    Fq W = fe_mul(U, V);
    Fq S = fe_mul(p.x, V);
    Fq X2 = fe_sqr(p.x);
    Fq M = fe_add(fe_dbl(X2), X2);
    r.x = fe_sub(fe_sqr(M), fe_dbl(S));
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = fe_mul(V, p.zz);
    r.zzz = fe_mul(W, p.zzz);
    return r;
}

// acc += (x2, y2) affine, non-identity (madd-2008-s)
H2_HD void xyzz_add_affine(XYZZ &acc, const Fq &x2, const Fq &y2) {
    if (acc.is_identity()) {
        acc.x = x2;
        acc.y = y2;
        acc.zz = Fq::one();
        acc.zzz = Fq::one();
        return;
    }
    Fq U2 = fe_mul(x2, acc.zz);
    Fq S2 = fe_mul(y2, acc.zzz);
    Fq Pd = fe_sub(U2, acc.x);
    Fq Rd = fe_sub(S2, acc.y);
    if (Pd.is_zero()) {
        if (Rd.is_zero()) acc = xyzz_double_affine(x2, y2);
        else acc = XYZZ::identity();
        return;
    }
    Fq PP = fe_sqr(Pd);
    Fq PPP = fe_mul(Pd, PP);
    Fq Q = fe_mul(acc.x, PP);
    Fq X3 = fe_sub(fe_sub(fe_sqr(Rd), PPP), fe_dbl(Q));
    Fq Y3 = fe_sub(fe_mul(Rd, fe_sub(Q, X3)), fe_mul(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fe_mul(acc.zz, PP);
    acc.zzz = fe_mul(acc.zzz, PPP);
}

// acc += b (add-2008-s)
H2_HD void xyzz_add(XYZZ &acc, const XYZZ &b) {
    if (b.is_identity()) return;
    if (acc.is_identity()) {
        acc = b;
        return;
    }
    Fq U1 = fe_mul(acc.x, b.zz);
    Fq U2 = fe_mul(b.x, acc.zz);
    Fq S1 = fe_mul(acc.y, b.zzz);
    Fq S2 = fe_mul(b.y, acc.zzz);
    Fq Pd = fe_sub(U2, U1);
    Fq Rd = fe_sub(S2, S1);
    if (Pd.is_zero()) {
        if (Rd.is_zero()) acc = xyzz_double(acc);
        else acc = XYZZ::identity();
        return;
    }
    Fq PP = fe_sqr(Pd);
    Fq PPP = fe_mul(Pd, PP);
    Fq Q = fe_mul(U1, PP);
    Fq X3 = fe_sub(fe_sub(fe_sqr(Rd), PPP), fe_dbl(Q));
    Fq Y3 = fe_sub(fe_mul(Rd, fe_sub(Q, X3)), fe_mul(S1, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fe_mul(fe_mul(acc.zz, b.zz), PP);
    acc.zzz = fe_mul(fe_mul(acc.zzz, b.zzz), PPP);
}

// XYZZ -> Jacobian without inversion: Z = ZZZ, X' = X*ZZ^2, Y' = Y*ZZZ^2  (Z^2 = ZZ^3, Z^3 = ZZZ^3).
// Identity -> (0, 1, 0)-style value with Z = 0.
H2_HD G1Jac xyzz_to_jacobian(const XYZZ &p) {
    G1Jac r;
    if (p.is_identity()) {
        r.x = Fq::zero();
        r.y = Fq::one();
        r.z = Fq::zero();
        return r;
    }
    r.x = fe_mul(p.x, fe_sqr(p.zz));
    r.y = fe_mul(p.y, fe_sqr(p.zzz));
    r.z = p.zzz;
    return r;
}

// XYZZ -> affine with one Fermat inversion (latency ~0.2 ms on one lane: used only when the caller asks
// for an affine result on the device).
H2_HD G1Affine xyzz_to_affine(const XYZZ &p) {
    G1Affine r;
    if (p.is_identity()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
        return r;
    }
    Fq i = fe_inv(fe_mul(p.zz, p.zzz));      // 1/(ZZ*ZZZ)
    r.x = fe_mul(p.x, fe_mul(i, p.zzz));     // X/ZZ
    r.y = fe_mul(p.y, fe_mul(i, p.zz));      // Y/ZZZ
    return r;
}

}  // namespace h2
