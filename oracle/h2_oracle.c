/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (CPU).  Never linked into or called by the product library
 * (halo2-lib_amd/csrc/libh2hip.so).  Users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg ("kind": "port").
 *
 * Plain-C restatement of the CPU algorithms on halo2-lib's proving hot path.  The reference
 * (/root/reference) reaches them only through create_proof (halo2-base/src/utils/testing.rs:40-47);
 * the code itself is in un-vendored crates halo2-axiom 0.5.3 (Cargo.lock:1063-1065) and
 * halo2curves-axiom 0.7.3 (Cargo.lock:1185-1188), so this restates their published algorithms:
 *   - Montgomery 4x64 F_r / F_q ([u64;4] LE limbs: halo2-base/src/utils/mod.rs:28-38,342-377)
 *   - G1 Jacobian arithmetic, identity (0,0) in affine form (halo2-ecc/src/ecc/pippenger.rs:217)
 *   - arithmetic::best_multiexp / multiexp_serial  (window ceil(ln n), 256/c+1 segments, bucket running sum,
 *     thread-chunked)                                               [UPSTREAM]
 *   - arithmetic::best_fft (bit reversal + radix-2 DIT layers)       [UPSTREAM]
 *   - EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff} [UPSTREAM]  
 *   - BatchInvert (0 -> 0), grand products, eval_polynomial, kate_division
 *
 * PARITY PINNING: MSM/NTT have no golden vectors in the reference ("parity unpinned");
 * this file is pinned against oracle/bn254.py (independent Python big-int) and the Poseidon KATs the
 * reference does hold (tests/test_oracle.py).
 *
 * Build: make -C oracle   (gcc -O3 -march=native -pthread -shared)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { fe m; uint64_t inv; fe r1; fe r2; } field_t;   /* modulus, -m^-1 mod 2^64, R mod m, R^2 mod m */

static const field_t FR = {
    {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    0xc2e1f593efffffffULL,
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};
static const field_t FQ = {
    {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    0x87d20782e4866389ULL,
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};

/* ------------------------------------------------------------------ field core */
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline void fe_sub_raw(fe *r, const fe *a, const fe *b) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a->l[i] - b->l[i] - br;
        r->l[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
static inline void fe_add(fe *r, const fe *a, const fe *b, const field_t *F) {
    u128 c = 0;
    fe t;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || fe_geq(&t, &F->m)) fe_sub_raw(&t, &t, &F->m);
    *r = t;
}
static inline void fe_sub(fe *r, const fe *a, const fe *b, const field_t *F) {
    fe t;
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - br;
        t.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t.l[i] + F->m.l[i];
            t.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    *r = t;
}
static inline void fe_neg(fe *r, const fe *a, const field_t *F) {
    if (fe_is_zero(a)) { *r = *a; return; }
    fe_sub_raw(r, &F->m, a);
}
/* CIOS Montgomery multiplication */
static inline void fe_mul(fe *r, const fe *a, const fe *b, const field_t *F) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->m.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * F->m.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&o, &F->m)) fe_sub_raw(&o, &o, &F->m);
    *r = o;
}
static inline void fe_sqr(fe *r, const fe *a, const field_t *F) { fe_mul(r, a, a, F); }
static inline void fe_from_mont(fe *r, const fe *a, const field_t *F) {
    fe one = {{1, 0, 0, 0}};
    fe_mul(r, a, &one, F);
}
static inline void fe_to_mont(fe *r, const fe *a, const field_t *F) { fe_mul(r, a, &F->r2, F); }
static void fe_pow(fe *r, const fe *a, const uint64_t e[4], const field_t *F) {
    fe acc = F->r1, base = *a;
    for (int i = 0; i < 256; ++i) {
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, &base, F);
        fe_sqr(&base, &base, F);
    }
    *r = acc;
}
/* inverse by Fermat; 0 -> 0 */
static void fe_inv(fe *r, const fe *a, const field_t *F) {
    fe two = {{2, 0, 0, 0}}, e;
    fe_sub_raw(&e, &F->m, &two);
    fe_pow(r, a, e.l, F);
}

/* ------------------------------------------------------------------ exported F_r helpers */
void orc_fr_mul_batch(fe *out, const fe *a, const fe *b, size_t n) { for (size_t i = 0; i < n; ++i) fe_mul(&out[i], &a[i], &b[i], &FR); }
void orc_fr_add_batch(fe *out, const fe *a, const fe *b, size_t n) { for (size_t i = 0; i < n; ++i) fe_add(&out[i], &a[i], &b[i], &FR); }
void orc_fr_sub_batch(fe *out, const fe *a, const fe *b, size_t n) { for (size_t i = 0; i < n; ++i) fe_sub(&out[i], &a[i], &b[i], &FR); }
void orc_fq_mul_batch(fe *out, const fe *a, const fe *b, size_t n) { for (size_t i = 0; i < n; ++i) fe_mul(&out[i], &a[i], &b[i], &FQ); }
void orc_fr_inv(fe *out, const fe *a) { fe_inv(out, a, &FR); }
void orc_fr_pow(fe *out, const fe *a, const uint64_t e[4]) { fe_pow(out, a, e, &FR); }
void orc_fr_from_mont(fe *out, const fe *a, size_t n) { for (size_t i = 0; i < n; ++i) fe_from_mont(&out[i], &a[i], &FR); }
void orc_fr_to_mont(fe *out, const fe *a, size_t n) { for (size_t i = 0; i < n; ++i) fe_to_mont(&out[i], &a[i], &FR); }

/* BatchInvert: Montgomery's trick, zeros skipped and left as zero [UPSTREAM ff::BatchInvert] */
void orc_fr_batch_invert(fe *a, size_t n) {
    fe *pre = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    fe acc = FR.r1;
    for (size_t i = 0; i < n; ++i) {
        pre[i] = acc;
        if (!fe_is_zero(&a[i])) fe_mul(&acc, &acc, &a[i], &FR);
    }
    fe_inv(&acc, &acc, &FR);
    for (size_t i = n; i-- > 0;) {
        if (fe_is_zero(&a[i])) continue;
        fe t;
        fe_mul(&t, &acc, &pre[i], &FR);
        fe_mul(&acc, &acc, &a[i], &FR);
        a[i] = t;
    }
    free(pre);
}

/* z[0]=1; z[i+1] = z[i]*num[i]/den[i]; writes n+1 values */
void orc_fr_grand_product(fe *z, const fe *num, const fe *den, size_t n) {
    fe *d = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    memcpy(d, den, sizeof(fe) * n);
    orc_fr_batch_invert(d, n);
    z[0] = FR.r1;
    for (size_t i = 0; i < n; ++i) {
        fe t;
        fe_mul(&t, &num[i], &d[i], &FR);
        fe_mul(&z[i + 1], &z[i], &t, &FR);
    }
    free(d);
}

/* Horner evaluation [UPSTREAM arithmetic::eval_polynomial] */
void orc_fr_eval_polynomial(fe *out, const fe *coeffs, size_t n, const fe *x) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t i = n; i-- > 0;) {
        fe_mul(&acc, &acc, x, &FR);
        fe_add(&acc, &acc, &coeffs[i], &FR);
    }
    *out = acc;
}

/* [UPSTREAM arithmetic::kate_division]: q = (f(X) - f(b)) / (X - b); q has n-1 coeffs */
void orc_fr_kate_division(fe *q, const fe *coeffs, size_t n, const fe *b) {
    fe tmp = {{0, 0, 0, 0}};
    for (size_t i = n - 1; i >= 1; --i) {
        fe t;
        fe_mul(&t, &tmp, b, &FR);
        fe_add(&tmp, &coeffs[i], &t, &FR);
        q[i - 1] = tmp;
    }
}

/* ------------------------------------------------------------------ best_fft restatement */
static size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
typedef struct { fe *a; const fe *tw; size_t n, chunk, tchunk, lo, hi; } fft_job;
static void *fft_layer_worker(void *p) {
    fft_job *j = (fft_job *)p;
    size_t half = j->chunk / 2;
    /* butterflies indexed b in [lo,hi): block = b / half, i = b % half */
    for (size_t b = j->lo; b < j->hi; ++b) {
        size_t blk = b / half, i = b % half;
        fe *u = &j->a[blk * j->chunk + i], *v = u + half, t;
        if (i == 0) t = *v; else fe_mul(&t, v, &j->tw[i * j->tchunk], &FR);
        fe x = *u;
        fe_add(u, &x, &t, &FR);
        fe_sub(v, &x, &t, &FR);
    }
    return NULL;
}
void orc_best_fft(fe *a, uint32_t log_n, const fe *omega, int threads) {
    size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; ++k) {
        size_t rk = bitrev(k, log_n);
        if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
    size_t nt = n / 2 ? n / 2 : 1;
    fe *tw = (fe *)malloc(sizeof(fe) * nt);
    tw[0] = FR.r1;
    for (size_t i = 1; i < n / 2; ++i) fe_mul(&tw[i], &tw[i - 1], omega, &FR);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    size_t chunk = 2, tchunk = n / 2;
    for (uint32_t layer = 0; layer < log_n; ++layer) {
        size_t total = n / 2;
        int T = (total < 4096) ? 1 : threads;
        pthread_t th[256];
        fft_job jobs[256];
        for (int t = 0; t < T; ++t) {
            jobs[t] = (fft_job){a, tw, n, chunk, tchunk, total * t / T, total * (t + 1) / T};
            if (T > 1) pthread_create(&th[t], NULL, fft_layer_worker, &jobs[t]); else fft_layer_worker(&jobs[t]);
        }
        if (T > 1) for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
        chunk *= 2;
        tchunk /= 2;
    }
    free(tw);
}
/* EvaluationDomain::ifft = best_fft(omega_inv) then * n^-1 */
void orc_ifft(fe *a, uint32_t log_n, const fe *omega, int threads) {
    fe winv, ninv, nn = {{(uint64_t)1 << log_n, 0, 0, 0}};
    fe_inv(&winv, omega, &FR);
    fe_to_mont(&nn, &nn, &FR);
    fe_inv(&ninv, &nn, &FR);
    orc_best_fft(a, log_n, &winv, threads);
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) fe_mul(&a[i], &a[i], &ninv, &FR);
}
/* EvaluationDomain::coeff_to_extended: scale by zeta^(i mod 3), zero pad, fft with extended omega */
void orc_coeff_to_extended(fe *out, const fe *in, uint32_t k, uint32_t ext_k, const fe *ext_omega, const fe *zeta, int threads) {
    size_t n = (size_t)1 << k, ne = (size_t)1 << ext_k;
    fe z[3];
    z[0] = FR.r1; z[1] = *zeta; fe_mul(&z[2], zeta, zeta, &FR);
    for (size_t i = 0; i < n; ++i) fe_mul(&out[i], &in[i], &z[i % 3], &FR);
    memset(out + n, 0, sizeof(fe) * (ne - n));
    orc_best_fft(out, ext_k, ext_omega, threads);
}
/* EvaluationDomain::extended_to_coeff (no truncation): ifft, then * [1, zeta^2, zeta][i mod 3] */
void orc_extended_to_coeff(fe *a, uint32_t ext_k, const fe *ext_omega, const fe *zeta, int threads) {
    size_t ne = (size_t)1 << ext_k;
    orc_ifft(a, ext_k, ext_omega, threads);
    fe z[3];
    z[0] = FR.r1; fe_mul(&z[1], zeta, zeta, &FR); z[2] = *zeta;
    for (size_t i = 0; i < ne; ++i) fe_mul(&a[i], &a[i], &z[i % 3], &FR);
}

/* ------------------------------------------------------------------ G1 Jacobian */
typedef struct { fe x, y; } g1a;          /* affine, identity = (0,0) */
typedef struct { fe x, y, z; } g1j;       /* Jacobian, identity z=0 */
static inline int g1a_is_id(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void g1j_set_id(g1j *p) { memset(p, 0, sizeof(*p)); p->x = FQ.r1; p->y = FQ.r1; }
static inline void g1j_from_affine(g1j *r, const g1a *p) {
    if (g1a_is_id(p)) { g1j_set_id(r); return; }
    r->x = p->x; r->y = p->y; r->z = FQ.r1;
}
static void g1j_double(g1j *r, const g1j *p) {
    if (fe_is_zero(&p->z)) { *r = *p; return; }
    fe A, B, C, D, E, F, t, X3, Y3, Z3;
    fe_sqr(&A, &p->x, &FQ);
    fe_sqr(&B, &p->y, &FQ);
    fe_sqr(&C, &B, &FQ);
    fe_add(&t, &p->x, &B, &FQ); fe_sqr(&t, &t, &FQ); fe_sub(&t, &t, &A, &FQ); fe_sub(&t, &t, &C, &FQ);
    fe_add(&D, &t, &t, &FQ);
    fe_add(&E, &A, &A, &FQ); fe_add(&E, &E, &A, &FQ);
    fe_sqr(&F, &E, &FQ);
    fe_sub(&X3, &F, &D, &FQ); fe_sub(&X3, &X3, &D, &FQ);
    fe_sub(&t, &D, &X3, &FQ); fe_mul(&Y3, &E, &t, &FQ);
    fe_add(&t, &C, &C, &FQ); fe_add(&t, &t, &t, &FQ); fe_add(&t, &t, &t, &FQ);
    fe_sub(&Y3, &Y3, &t, &FQ);
    fe_mul(&Z3, &p->y, &p->z, &FQ); fe_add(&Z3, &Z3, &Z3, &FQ);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void g1j_add(g1j *r, const g1j *p, const g1j *q) {
    if (fe_is_zero(&p->z)) { *r = *q; return; }
    if (fe_is_zero(&q->z)) { *r = *p; return; }
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    fe_sqr(&Z1Z1, &p->z, &FQ); fe_sqr(&Z2Z2, &q->z, &FQ);
    fe_mul(&U1, &p->x, &Z2Z2, &FQ); fe_mul(&U2, &q->x, &Z1Z1, &FQ);
    fe_mul(&S1, &p->y, &q->z, &FQ); fe_mul(&S1, &S1, &Z2Z2, &FQ);
    fe_mul(&S2, &q->y, &p->z, &FQ); fe_mul(&S2, &S2, &Z1Z1, &FQ);
    if (fe_eq(&U1, &U2)) {
        if (fe_eq(&S1, &S2)) { g1j_double(r, p); return; }
        g1j_set_id(r); return;
    }
    fe_sub(&H, &U2, &U1, &FQ);
    fe_add(&I, &H, &H, &FQ); fe_sqr(&I, &I, &FQ);
    fe_mul(&J, &H, &I, &FQ);
    fe_sub(&rr, &S2, &S1, &FQ); fe_add(&rr, &rr, &rr, &FQ);
    fe_mul(&V, &U1, &I, &FQ);
    fe_sqr(&X3, &rr, &FQ); fe_sub(&X3, &X3, &J, &FQ); fe_sub(&X3, &X3, &V, &FQ); fe_sub(&X3, &X3, &V, &FQ);
    fe_sub(&t, &V, &X3, &FQ); fe_mul(&Y3, &rr, &t, &FQ);
    fe_mul(&t, &S1, &J, &FQ); fe_add(&t, &t, &t, &FQ); fe_sub(&Y3, &Y3, &t, &FQ);
    fe_add(&Z3, &p->z, &q->z, &FQ); fe_sqr(&Z3, &Z3, &FQ); fe_sub(&Z3, &Z3, &Z1Z1, &FQ); fe_sub(&Z3, &Z3, &Z2Z2, &FQ);
    fe_mul(&Z3, &Z3, &H, &FQ);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void g1j_add_affine(g1j *r, const g1j *p, const g1a *q) {
    if (g1a_is_id(q)) { *r = *p; return; }
    if (fe_is_zero(&p->z)) { g1j_from_affine(r, q); return; }
    fe Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;
    fe_sqr(&Z1Z1, &p->z, &FQ);
    fe_mul(&U2, &q->x, &Z1Z1, &FQ);
    fe_mul(&S2, &q->y, &p->z, &FQ); fe_mul(&S2, &S2, &Z1Z1, &FQ);
    if (fe_eq(&p->x, &U2)) {
        if (fe_eq(&p->y, &S2)) { g1j_double(r, p); return; }
        g1j_set_id(r); return;
    }
    fe_sub(&H, &U2, &p->x, &FQ);
    fe_sqr(&HH, &H, &FQ);
    fe_add(&I, &HH, &HH, &FQ); fe_add(&I, &I, &I, &FQ);
    fe_mul(&J, &H, &I, &FQ);
    fe_sub(&rr, &S2, &p->y, &FQ); fe_add(&rr, &rr, &rr, &FQ);
    fe_mul(&V, &p->x, &I, &FQ);
    fe_sqr(&X3, &rr, &FQ); fe_sub(&X3, &X3, &J, &FQ); fe_sub(&X3, &X3, &V, &FQ); fe_sub(&X3, &X3, &V, &FQ);
    fe_sub(&t, &V, &X3, &FQ); fe_mul(&Y3, &rr, &t, &FQ);
    fe_mul(&t, &p->y, &J, &FQ); fe_add(&t, &t, &t, &FQ); fe_sub(&Y3, &Y3, &t, &FQ);
    fe_add(&Z3, &p->z, &H, &FQ); fe_sqr(&Z3, &Z3, &FQ); fe_sub(&Z3, &Z3, &Z1Z1, &FQ); fe_sub(&Z3, &Z3, &HH, &FQ);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void g1j_to_affine(g1a *r, const g1j *p) {
    if (fe_is_zero(&p->z)) { memset(r, 0, sizeof(*r)); return; }
    fe zi, zi2, zi3;
    fe_inv(&zi, &p->z, &FQ);
    fe_sqr(&zi2, &zi, &FQ);
    fe_mul(&zi3, &zi2, &zi, &FQ);
    fe_mul(&r->x, &p->x, &zi2, &FQ);
    fe_mul(&r->y, &p->y, &zi3, &FQ);
}

void orc_g1_add(g1a *out, const g1a *a, const g1a *b) {
    g1j t; g1j_from_affine(&t, a); g1j_add_affine(&t, &t, b); g1j_to_affine(out, &t);
}
/* scalar: Montgomery F_r element */
void orc_g1_mul(g1a *out, const g1a *p, const fe *scalar_mont) {
    fe s; fe_from_mont(&s, scalar_mont, &FR);
    g1j acc, base; g1j_set_id(&acc); g1j_from_affine(&base, p);
    for (int i = 0; i < 256; ++i) {
        if ((s.l[i >> 6] >> (i & 63)) & 1) g1j_add(&acc, &acc, &base);
        g1j_double(&base, &base);
    }
    g1j_to_affine(out, &acc);
}
int orc_g1_is_on_curve(const g1a *p) {
    if (g1a_is_id(p)) return 1;
    fe y2, x3, three = {{3, 0, 0, 0}};
    fe_to_mont(&three, &three, &FQ);
    fe_sqr(&y2, &p->y, &FQ);
    fe_sqr(&x3, &p->x, &FQ); fe_mul(&x3, &x3, &p->x, &FQ); fe_add(&x3, &x3, &three, &FQ);
    return fe_eq(&y2, &x3);
}
/* P_i = (k0 + i*d)*G, G=(1,2): P_0 = k0*G, then repeated add of D = d*G, batch-normalised */
void orc_known_dlog_bases(g1a *out, size_t n, const fe *k0_mont, const fe *d_mont) {
    g1a G, P0, D;
    fe one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    fe_to_mont(&G.x, &one, &FQ); fe_to_mont(&G.y, &two, &FQ);
    orc_g1_mul(&P0, &G, k0_mont);
    orc_g1_mul(&D, &G, d_mont);
    g1j *J = (g1j *)malloc(sizeof(g1j) * (n ? n : 1));
    g1j cur; g1j_from_affine(&cur, &P0);
    for (size_t i = 0; i < n; ++i) { J[i] = cur; g1j_add_affine(&cur, &cur, &D); }
    /* batch normalise */
    fe *pre = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    fe acc = FQ.r1;
    for (size_t i = 0; i < n; ++i) { pre[i] = acc; if (!fe_is_zero(&J[i].z)) fe_mul(&acc, &acc, &J[i].z, &FQ); }
    fe_inv(&acc, &acc, &FQ);
    for (size_t i = n; i-- > 0;) {
        if (fe_is_zero(&J[i].z)) { memset(&out[i], 0, sizeof(g1a)); continue; }
        fe zi, zi2, zi3;
        fe_mul(&zi, &acc, &pre[i], &FQ);
        fe_mul(&acc, &acc, &J[i].z, &FQ);
        fe_sqr(&zi2, &zi, &FQ); fe_mul(&zi3, &zi2, &zi, &FQ);
        fe_mul(&out[i].x, &J[i].x, &zi2, &FQ);
        fe_mul(&out[i].y, &J[i].y, &zi3, &FQ);
    }
    free(pre); free(J);
}

/* out[i] = scalars[i] * base for n Montgomery scalars (the SRS definition g[i] = s^i G, g_lagrange[i] = L_i(s) G evaluated point by point:
   ParamsKZG::setup).  Fixed-base windows of 8 bits: T[w][d] = d * 2^(8w) * base (32 x 255 affine points), 32 mixed additions
   per scalar, Montgomery's trick for the final normalisation of every thread's range.  Independent of best_multiexp. */
typedef struct { g1a *out; const fe *scalars; const g1a *table; size_t lo, hi; } fb_job;
static void *fb_worker(void *p) {
    fb_job *j = (fb_job *)p;
    size_t n = j->hi - j->lo;
    if (!n) return NULL;
    g1j *J = (g1j *)malloc(sizeof(g1j) * n);
    fe *pre = (fe *)malloc(sizeof(fe) * n);
    for (size_t i = 0; i < n; ++i) {
        fe s; fe_from_mont(&s, &j->scalars[j->lo + i], &FR);
        g1j acc; g1j_set_id(&acc);
        for (int w = 0; w < 32; ++w) {
            unsigned d = (unsigned)((s.l[w >> 3] >> ((w & 7) * 8)) & 0xff);
            if (d) g1j_add_affine(&acc, &acc, &j->table[(size_t)w * 256 + d]);
        }
        J[i] = acc;
    }
    fe acc = FQ.r1;
    for (size_t i = 0; i < n; ++i) { pre[i] = acc; if (!fe_is_zero(&J[i].z)) fe_mul(&acc, &acc, &J[i].z, &FQ); }
    fe_inv(&acc, &acc, &FQ);
    for (size_t i = n; i-- > 0;) {
        g1a *o = &j->out[j->lo + i];
        if (fe_is_zero(&J[i].z)) { memset(o, 0, sizeof(g1a)); continue; }
        fe zi, zi2, zi3;
        fe_mul(&zi, &acc, &pre[i], &FQ);
        fe_mul(&acc, &acc, &J[i].z, &FQ);
        fe_sqr(&zi2, &zi, &FQ); fe_mul(&zi3, &zi2, &zi, &FQ);
        fe_mul(&o->x, &J[i].x, &zi2, &FQ);
        fe_mul(&o->y, &J[i].y, &zi3, &FQ);
    }
    free(pre); free(J);
    return NULL;
}
void orc_g1_fixed_base_batch(g1a *out, const g1a *base, const fe *scalars_mont, size_t n, int threads) {
    g1a *table = (g1a *)calloc((size_t)32 * 256, sizeof(g1a));
    g1j cur; g1j_from_affine(&cur, base);                 /* 2^(8w) * base */
    for (int w = 0; w < 32; ++w) {
        g1j acc; g1j_set_id(&acc);
        g1a step; g1j_to_affine(&step, &cur);
        for (int d = 1; d < 256; ++d) { g1j_add_affine(&acc, &acc, &step); g1j_to_affine(&table[(size_t)w * 256 + d], &acc); }
        for (int b = 0; b < 8; ++b) g1j_double(&cur, &cur);
    }
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64]; fb_job jobs[64];
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo > n) lo = n; if (hi > n) hi = n;
        jobs[t] = (fb_job){out, scalars_mont, table, lo, hi};
        pthread_create(&th[t], NULL, fb_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(table);
}

/* ------------------------------------------------------------------ best_multiexp restatement */
static inline size_t get_at(size_t segment, size_t c, const fe *canon) {
    size_t skip_bits = segment * c;
    if (skip_bits >= 256) return 0;
    size_t limb = skip_bits / 64, off = skip_bits % 64;
    uint64_t v = canon->l[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= canon->l[limb + 1] << (64 - off);
    return (size_t)(v & (((uint64_t)1 << c) - 1));
}
static void multiexp_serial(const fe *canon, const g1a *bases, size_t n, g1j *acc) {
    size_t c;
    if (n < 4) c = 1; else if (n < 32) c = 3; else c = (size_t)ceil(log((double)n));
    size_t segments = 256 / c + 1, nb = ((size_t)1 << c) - 1;
    g1j *buckets = (g1j *)malloc(sizeof(g1j) * nb);
    unsigned char *state = (unsigned char *)malloc(nb);   /* 0 none, 1 affine (stored z=1), 2 projective */
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t i = 0; i < c; ++i) g1j_double(acc, acc);
        memset(state, 0, nb);
        for (size_t i = 0; i < n; ++i) {
            size_t d = get_at(seg, c, &canon[i]);
            if (!d) continue;
            g1j *b = &buckets[d - 1];
            if (!state[d - 1]) { g1j_from_affine(b, &bases[i]); state[d - 1] = 1; }
            else { g1j_add_affine(b, b, &bases[i]); state[d - 1] = 2; }
        }
        g1j running; g1j_set_id(&running);
        for (size_t k = nb; k-- > 0;) {
            if (state[k]) g1j_add(&running, &running, &buckets[k]);
            g1j_add(acc, acc, &running);
        }
    }
    free(buckets); free(state);
}
typedef struct { const fe *canon; const g1a *bases; size_t n; g1j acc; } msm_job;
static void *msm_worker(void *p) {
    msm_job *j = (msm_job *)p;
    g1j_set_id(&j->acc);
    multiexp_serial(j->canon, j->bases, j->n, &j->acc);
    return NULL;
}
/* scalars: Montgomery F_r; bases: affine Montgomery F_q; out: affine. threads ~ rayon current_num_threads */
void orc_best_multiexp(g1a *out, const fe *scalars_mont, const g1a *bases, size_t n, int threads) {
    fe *canon = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) fe_from_mont(&canon[i], &scalars_mont[i], &FR);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    g1j total; g1j_set_id(&total);
    if (n > (size_t)threads) {
        size_t chunk = n / threads, nchunks = (n + chunk - 1) / chunk;
        msm_job *jobs = (msm_job *)malloc(sizeof(msm_job) * nchunks);
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nchunks);
        for (size_t t = 0; t < nchunks; ++t) {
            size_t lo = t * chunk, hi = lo + chunk > n ? n : lo + chunk;
            jobs[t].canon = canon + lo; jobs[t].bases = bases + lo; jobs[t].n = hi - lo;
            pthread_create(&th[t], NULL, msm_worker, &jobs[t]);
        }
        for (size_t t = 0; t < nchunks; ++t) { pthread_join(th[t], NULL); g1j_add(&total, &total, &jobs[t].acc); }
        free(jobs); free(th);
    } else {
        multiexp_serial(canon, bases, n, &total);
    }
    g1j_to_affine(out, &total);
    free(canon);
}

/* ================================================================== create_proof vector steps
 * CPU restatement of the data-parallel steps of halo2's PLONK/KZG prover around the MSMs and FFTs
 * [UPSTREAM halo2-axiom 0.5.3: plonk/{prover,permutation/prover,lookup/prover,vanishing/prover,evaluation}.rs,
 * poly/domain.rs — SURVEY.md §3.2, A.3-A.6], reached from the reference only through
 * halo2-base/src/utils/testing.rs:40-47.  oracle/plonk.py orchestrates them (transcript, challenges, ordering);
 * bench.py's create_proof cpu_baseline times exactly that composition.  `threads` mirrors upstream's rayon
 * `parallelize` chunking; results do not depend on it. */
typedef void (*range_fn)(size_t lo, size_t hi, void *arg);
typedef struct { range_fn fn; void *arg; size_t lo, hi; } pf_job;
static void *pf_worker(void *p) { pf_job *j = (pf_job *)p; j->fn(j->lo, j->hi, j->arg); return NULL; }
static void parallel_for(size_t n, int threads, range_fn fn, void *arg) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (n < 2048 || threads == 1) { fn(0, n, arg); return; }
    pthread_t th[256];
    pf_job jobs[256];
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (pf_job){fn, arg, n * (size_t)t / threads, n * (size_t)(t + 1) / threads};
        pthread_create(&th[t], NULL, pf_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
}
static void fe_pow_u64(fe *r, const fe *a, uint64_t e) { uint64_t ee[4] = {e, 0, 0, 0}; fe_pow(r, a, ee, &FR); }

/* out[i] = a[i]*sa (+ b[i]*sb) (+ c);  sa / sb / c may be NULL (= 1 / 1 / 0), b may be NULL */
typedef struct { fe *out; const fe *a, *sa, *b, *sb, *c; } lin_arg;
static void lin_range(size_t lo, size_t hi, void *p) {
    lin_arg *g = (lin_arg *)p;
    for (size_t i = lo; i < hi; ++i) {
        fe v = g->a[i], t;
        if (g->sa) fe_mul(&v, &v, g->sa, &FR);
        if (g->b) { t = g->b[i]; if (g->sb) fe_mul(&t, &t, g->sb, &FR); fe_add(&v, &v, &t, &FR); }
        if (g->c) fe_add(&v, &v, g->c, &FR);
        g->out[i] = v;
    }
}
void orc_fr_lincomb(fe *out, const fe *a, const fe *sa, const fe *b, const fe *sb, const fe *c, size_t n, int threads) {
    lin_arg g = {out, a, sa, b, sb, c};
    parallel_for(n, threads, lin_range, &g);
}
typedef struct { fe *out; const fe *a, *b; } mul_arg;
static void mul_range(size_t lo, size_t hi, void *p) { mul_arg *g = (mul_arg *)p; for (size_t i = lo; i < hi; ++i) fe_mul(&g->out[i], &g->a[i], &g->b[i], &FR); }
void orc_fr_mul_batch_mt(fe *out, const fe *a, const fe *b, size_t n, int threads) { mul_arg g = {out, a, b}; parallel_for(n, threads, mul_range, &g); }
/* out[i] = start * ratio^i */
typedef struct { fe *out; const fe *start, *ratio; } geom_arg;
static void geom_range(size_t lo, size_t hi, void *p) {
    geom_arg *g = (geom_arg *)p;
    fe cur; fe_pow_u64(&cur, g->ratio, lo); fe_mul(&cur, &cur, g->start, &FR);
    for (size_t i = lo; i < hi; ++i) { g->out[i] = cur; fe_mul(&cur, &cur, g->ratio, &FR); }
}
void orc_fr_geom(fe *out, const fe *start, const fe *ratio, size_t n, int threads) { geom_arg g = {out, start, ratio}; parallel_for(n, threads, geom_range, &g); }
static void inv_range(size_t lo, size_t hi, void *p) { orc_fr_batch_invert((fe *)p + lo, hi - lo); }
void orc_fr_batch_invert_mt(fe *a, size_t n, int threads) { parallel_for(n, threads, inv_range, a); }
/* z[0] = first; z[i+1] = z[i]*vals[i] for i < n  (n+1 values) */
void orc_fr_running_product(fe *z, const fe *first, const fe *vals, size_t n) {
    z[0] = *first;
    for (size_t i = 0; i < n; ++i) fe_mul(&z[i + 1], &z[i], &vals[i], &FR);
}
typedef struct { fe *y; const fe *a, *x; } axpy_arg;
static void axpy_range(size_t lo, size_t hi, void *p) {
    axpy_arg *g = (axpy_arg *)p;
    for (size_t i = lo; i < hi; ++i) { fe t; fe_mul(&t, &g->x[i], g->a, &FR); fe_add(&g->y[i], &g->y[i], &t, &FR); }
}
void orc_fr_axpy(fe *y, const fe *a, const fe *x, size_t n, int threads) { axpy_arg g = {y, a, x}; parallel_for(n, threads, axpy_range, &g); }
static void scale_range(size_t lo, size_t hi, void *p) { axpy_arg *g = (axpy_arg *)p; for (size_t i = lo; i < hi; ++i) fe_mul(&g->y[i], &g->y[i], g->a, &FR); }
void orc_fr_scale(fe *y, const fe *s, size_t n, int threads) { axpy_arg g = {y, s, NULL}; parallel_for(n, threads, scale_range, &g); }

/* lookup::prover::permute_expression_pair over the usable rows [UPSTREAM]: a' = sorted input (Ord on the canonical
 * representation); s'[row] = a'[row] where a run of equal inputs starts (consuming one table element of that value); the
 * leftover table elements, in ascending order, fill the repeated rows popped from the END.  Returns 0, or -1 when an
 * input value is missing from the table (upstream: Error::ConstraintSystemFailure). */
static int canon_cmp(const void *x, const void *y) {
    const fe *a = (const fe *)x, *b = (const fe *)y;
    for (int i = 3; i >= 0; --i) if (a->l[i] != b->l[i]) return a->l[i] < b->l[i] ? -1 : 1;
    return 0;
}
int orc_permute_expression_pair(fe *ap, fe *sp, const fe *a, const fe *s, size_t usable) {
    fe *ca = (fe *)malloc(sizeof(fe) * (usable ? usable : 1)), *cs = (fe *)malloc(sizeof(fe) * (usable ? usable : 1));
    size_t *rep = (size_t *)malloc(sizeof(size_t) * (usable ? usable : 1));
    unsigned char *used = (unsigned char *)calloc(usable ? usable : 1, 1);
    for (size_t i = 0; i < usable; ++i) { fe_from_mont(&ca[i], &a[i], &FR); fe_from_mont(&cs[i], &s[i], &FR); }
    qsort(ca, usable, sizeof(fe), canon_cmp);
    qsort(cs, usable, sizeof(fe), canon_cmp);   /* sorted table = the BTreeMap's iteration order, with multiplicity */
    size_t nrep = 0, t = 0;
    int rc = 0;
    for (size_t row = 0; row < usable && !rc; ++row) {
        if (row == 0 || canon_cmp(&ca[row], &ca[row - 1]) != 0) {
            while (t < usable && canon_cmp(&cs[t], &ca[row]) < 0) ++t;          /* first table element >= the input value */
            if (t >= usable || canon_cmp(&cs[t], &ca[row]) != 0) { rc = -1; break; }
            used[t++] = 1;
            sp[row] = ca[row];
        } else rep[nrep++] = row;
    }
    if (!rc) {
        for (size_t j = 0; j < usable; ++j) if (!used[j]) sp[rep[--nrep]] = cs[j];
        for (size_t i = 0; i < usable; ++i) { fe_to_mont(&ap[i], &ca[i], &FR); fe_to_mont(&sp[i], &sp[i], &FR); }
    }
    free(ca); free(cs); free(rep); free(used);
    return rc;
}

/* evaluate_h, custom gate of halo2-base (reference halo2-base/src/gates/flex_gate/mod.rs:80-91):
 * acc[i] = acc[i]*y + q[i]*(a[i] + a[i+s]*a[i+2s] - a[i+3s]) on the extended domain, s = rot_scale */
typedef struct { fe *acc; const fe *q, *a, *y; size_t ne, step; } gate_arg;
static void gate_range(size_t lo, size_t hi, void *p) {
    gate_arg *g = (gate_arg *)p;
    size_t m = g->ne - 1;
    for (size_t i = lo; i < hi; ++i) {
        fe t, v;
        fe_mul(&t, &g->a[(i + g->step) & m], &g->a[(i + 2 * g->step) & m], &FR);
        fe_add(&t, &t, &g->a[i], &FR);
        fe_sub(&t, &t, &g->a[(i + 3 * g->step) & m], &FR);
        fe_mul(&t, &t, &g->q[i], &FR);
        fe_mul(&v, &g->acc[i], g->y, &FR);
        fe_add(&g->acc[i], &v, &t, &FR);
    }
}
void orc_quotient_gate(fe *acc, const fe *q, const fe *a, const fe *y, size_t ne, size_t step, int threads) {
    gate_arg g = {acc, q, a, y, ne, step};
    parallel_for(ne, threads, gate_range, &g);
}
/* evaluate_h, permutation argument [UPSTREAM plonk/evaluation.rs]: per extended-domain point, in this order:
 * l0*(1-z_0); l_last*(z_last^2-z_last); for sets 1..: l0*(z_i - z_{i-1}(w^last X)); for every set:
 * l_active*(z_i(wX)*prod(v + beta*sigma + gamma) - z_i*prod(v + delta^j*beta*X + gamma)), each folded with y. */
typedef struct { fe *acc; const fe *const *z; uint32_t nsets; const fe *const *cols, *const *sig; uint32_t ncols, chunk; const fe *l0, *l_last, *l_active;
                 size_t ne, step; int64_t last_rot; const fe *beta, *gamma, *y, *delta, *zeta, *ext_omega; } perm_arg;
static void perm_range(size_t lo, size_t hi, void *p) {
    perm_arg *g = (perm_arg *)p;
    size_t m = g->ne - 1;
    fe one = FR.r1, beta_term, delta_start;
    fe_pow_u64(&beta_term, g->ext_omega, lo);
    fe_mul(&delta_start, g->beta, g->zeta, &FR);
    for (size_t i = lo; i < hi; ++i) {
        size_t r_next = (i + g->step) & m, r_last = (size_t)((int64_t)i + g->last_rot * (int64_t)g->step) & m;
        fe v = g->acc[i], t, u;
        const fe *zf = g->z[0], *zl = g->z[g->nsets - 1];
        fe_sub(&t, &one, &zf[i], &FR); fe_mul(&t, &t, &g->l0[i], &FR);
        fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        fe_mul(&t, &zl[i], &zl[i], &FR); fe_sub(&t, &t, &zl[i], &FR); fe_mul(&t, &t, &g->l_last[i], &FR);
        fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        for (uint32_t s = 1; s < g->nsets; ++s) {
            fe_sub(&t, &g->z[s][i], &g->z[s - 1][r_last], &FR); fe_mul(&t, &t, &g->l0[i], &FR);
            fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        }
        fe cur_delta; fe_mul(&cur_delta, &delta_start, &beta_term, &FR);
        for (uint32_t s = 0, c0 = 0; s < g->nsets; ++s, c0 += g->chunk) {
            uint32_t c1 = c0 + g->chunk > g->ncols ? g->ncols : c0 + g->chunk;
            fe left = g->z[s][r_next], right = g->z[s][i];
            for (uint32_t c = c0; c < c1; ++c) {
                fe_mul(&t, g->beta, &g->sig[c][i], &FR); fe_add(&t, &t, &g->cols[c][i], &FR); fe_add(&t, &t, g->gamma, &FR);
                fe_mul(&left, &left, &t, &FR);
            }
            for (uint32_t c = c0; c < c1; ++c) {
                fe_add(&u, &g->cols[c][i], &cur_delta, &FR); fe_add(&u, &u, g->gamma, &FR);
                fe_mul(&right, &right, &u, &FR);
                fe_mul(&cur_delta, &cur_delta, g->delta, &FR);
            }
            fe_sub(&t, &left, &right, &FR); fe_mul(&t, &t, &g->l_active[i], &FR);
            fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        }
        g->acc[i] = v;
        fe_mul(&beta_term, &beta_term, g->ext_omega, &FR);
    }
}
void orc_quotient_permutation(fe *acc, const fe *const *z_sets, uint32_t nsets, const fe *const *cols, const fe *const *sigmas, uint32_t ncols,
                              uint32_t chunk_len, const fe *l0, const fe *l_last, const fe *l_active, size_t ne, size_t step, int32_t last_rotation,
                              const fe *beta, const fe *gamma, const fe *y, const fe *delta, const fe *zeta, const fe *ext_omega, int threads) {
    perm_arg g = {acc, z_sets, nsets, cols, sigmas, ncols, chunk_len, l0, l_last, l_active, ne, step, last_rotation, beta, gamma, y, delta, zeta, ext_omega};
    parallel_for(ne, threads, perm_range, &g);
}
/* evaluate_h, one lookup argument [UPSTREAM plonk/evaluation.rs]; input / table: the theta-compressed expressions evaluated on
 * the extended domain (for halo2-base: q_lookup*a or a, and the table column; halo2-base/src/gates/range/mod.rs:131-150) */
typedef struct { fe *acc; const fe *z, *in, *tab, *ap, *sp, *l0, *l_last, *l_active; size_t ne, step; const fe *beta, *gamma, *y; } lk_arg;
static void lk_range(size_t lo, size_t hi, void *p) {
    lk_arg *g = (lk_arg *)p;
    size_t m = g->ne - 1;
    fe one = FR.r1;
    for (size_t i = lo; i < hi; ++i) {
        size_t r_next = (i + g->step) & m, r_prev = (i - g->step) & m;
        fe v = g->acc[i], t, u, table_value, a_minus_s;
        fe_add(&t, &g->in[i], g->beta, &FR); fe_add(&u, &g->tab[i], g->gamma, &FR); fe_mul(&table_value, &t, &u, &FR);
        fe_sub(&a_minus_s, &g->ap[i], &g->sp[i], &FR);
        fe_sub(&t, &one, &g->z[i], &FR); fe_mul(&t, &t, &g->l0[i], &FR);
        fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        fe_mul(&t, &g->z[i], &g->z[i], &FR); fe_sub(&t, &t, &g->z[i], &FR); fe_mul(&t, &t, &g->l_last[i], &FR);
        fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        fe_add(&t, &g->ap[i], g->beta, &FR); fe_add(&u, &g->sp[i], g->gamma, &FR); fe_mul(&t, &t, &u, &FR); fe_mul(&t, &t, &g->z[r_next], &FR);
        fe_mul(&u, &g->z[i], &table_value, &FR); fe_sub(&t, &t, &u, &FR); fe_mul(&t, &t, &g->l_active[i], &FR);
        fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        fe_mul(&t, &a_minus_s, &g->l0[i], &FR);
        fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        fe_sub(&t, &g->ap[i], &g->ap[r_prev], &FR); fe_mul(&t, &t, &a_minus_s, &FR); fe_mul(&t, &t, &g->l_active[i], &FR);
        fe_mul(&v, &v, g->y, &FR); fe_add(&v, &v, &t, &FR);
        g->acc[i] = v;
    }
}
void orc_quotient_lookup(fe *acc, const fe *z, const fe *input, const fe *table, const fe *ap, const fe *sp, const fe *l0, const fe *l_last,
                         const fe *l_active, size_t ne, size_t step, const fe *beta, const fe *gamma, const fe *y, int threads) {
    lk_arg g = {acc, z, input, table, ap, sp, l0, l_last, l_active, ne, step, beta, gamma, y};
    parallel_for(ne, threads, lk_range, &g);
}
/* EvaluationDomain::divide_by_vanishing_poly: a[i] *= t_evaluations[i mod 2^(ext_k-k)], t_evaluations[j] = 1/((zeta*w_ext^j)^n - 1) */
typedef struct { fe *a; const fe *t; size_t mask; } van_arg;
static void van_range(size_t lo, size_t hi, void *p) { van_arg *g = (van_arg *)p; for (size_t i = lo; i < hi; ++i) fe_mul(&g->a[i], &g->a[i], &g->t[i & g->mask], &FR); }
void orc_divide_by_vanishing(fe *a, uint32_t ext_k, uint32_t k, const fe *ext_omega, const fe *zeta, int threads) {
    size_t cnt = (size_t)1 << (ext_k - k);
    fe *t = (fe *)malloc(sizeof(fe) * cnt), zn, wn, cur;
    fe_pow_u64(&zn, zeta, (uint64_t)1 << k);
    fe_pow_u64(&wn, ext_omega, (uint64_t)1 << k);
    cur = zn;
    for (size_t j = 0; j < cnt; ++j) { fe_sub(&t[j], &cur, &FR.r1, &FR); fe_inv(&t[j], &t[j], &FR); fe_mul(&cur, &cur, &wn, &FR); }
    van_arg g = {a, t, cnt - 1};
    parallel_for((size_t)1 << ext_k, threads, van_range, &g);
    free(t);
}
