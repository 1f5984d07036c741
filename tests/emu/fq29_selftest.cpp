// TEST INFRASTRUCTURE: cross-checks the unsaturated 9x29-bit arithmetic (fq29.cuh / ec29.cuh) against the saturated
// 8x32-bit arithmetic (field.cuh / ec.cuh) on the host, over random and adversarial inputs.  The saturated path is
// itself pinned against the oracle by the parity tests, so agreement here pins the unsaturated path too.
#include <hip/hip_runtime.h>   // tests/emu stand-in: makes H2_HD functions plain host functions, enables limb-bound asserts
#include <stdio.h>
#include <stdlib.h>

#include "../../halo2-lib_amd/csrc/ec29.cuh"

using namespace h2;

static uint64_t seed = 0x1234567;
static uint64_t sm() {
    seed += 0x9E3779B97F4A7C15ULL;
    uint64_t z = seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
template <class P>
static Fe<P> rnd() {
    Fe<P> r;
    for (int i = 0; i < 8; ++i) r.l[i] = (uint32_t)sm();
    r.l[7] &= 0x0fffffffu;   // < 2^252 < p: a valid canonical element
    return r;
}
template <class P>
static Fe<P> from_u64(uint64_t v) {
    Fe<P> r = Fe<P>::zero();
    r.l[0] = (uint32_t)v;
    r.l[1] = (uint32_t)(v >> 32);
    return fe_to_mont(r);
}
static int fails = 0;
#define CHECK(c)                                              \
    do {                                                      \
        if (!(c)) {                                           \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                          \
        }                                                     \
    } while (0)

template <class P29, class PS>
static void field_checks(const char *name) {
    F29<P29> cin, cout;
    for (int i = 0; i < 9; ++i) {
        cin.l[i] = P29::conv_in(i);
        cout.l[i] = P29::conv_out(i);
    }
    auto to29 = [&](const Fe<PS> &s) { return f29_mul(f29_split<P29>(s), cin); };
    auto to32 = [&](const F29<P29> &v) { return f29_pack_canonical<PS>(f29_mul(v, cout)); };
    Fe<PS> specials[6] = {Fe<PS>::zero(), Fe<PS>::one(), fe_neg(Fe<PS>::one()), from_u64<PS>(2), fe_neg(from_u64<PS>(2)), from_u64<PS>(0xffffffffffffffffULL)};
    for (int it = 0; it < 4000; ++it) {
        Fe<PS> a = it < 36 ? specials[it / 6] : rnd<PS>(), b = it < 36 ? specials[it % 6] : rnd<PS>();
        F29<P29> A = to29(a), B = to29(b);
        CHECK(to32(A) == a);
        CHECK(to32(f29_mul(A, B)) == fe_mul(a, b));
        CHECK(to32(f29_sqr(A)) == fe_sqr(a));
        CHECK(to32(f29_norm(f29_add(A, B))) == fe_add(a, b));
        CHECK(to32(f29_sub<2>(A, B)) == fe_sub(a, b));
        CHECK(to32(f29_sub<8>(f29_sub<6>(A, B), f29_sub<4>(B, A))) == fe_sub(fe_sub(a, b), fe_sub(b, a)));
        CHECK(to32(f29_mul(f29_add(A, A), f29_add(B, B))) == fe_mul(fe_dbl(a), fe_dbl(b)));   // lazy-add inputs
        CHECK(to32(f29_sqr(f29_add(A, A))) == fe_sqr(fe_dbl(a)));
        {   // weak reduction of a value grown by repeated additions (up to ~20 p)
            F29<P29> g = A;
            Fe<PS> gs = a;
            for (int r = 0; r < (int)(it % 19); ++r) {
                g = f29_norm(f29_add(g, B));
                gs = fe_add(gs, b);
            }
            F29<P29> w = f29_weak_reduce(g);
            CHECK(to32(w) == gs);
            CHECK(f29_pack_canonical<PS>(w) == f29_pack_canonical<PS>(f29_mul(g, F29<P29>::one())));
        }
        // zero tests over the documented ranges
        F29<P29> d = f29_sub<6>(A, B);
        CHECK(f29_is_zero_mod_q<7>(d) == (a == b));
        F29<P29> z = f29_sub<6>(A, A);
        CHECK(f29_is_zero_mod_q<7>(z));
        F29<P29> z3 = f29_sub<2>(A, A);
        CHECK(f29_is_zero_mod_q<3>(z3));
    }
    printf("%s field checks done\n", name);
}

static bool same_point(const XYZZ &p, const XYZZ &q) {
    G1Affine a = xyzz_to_affine(p), b = xyzz_to_affine(q);
    return a.x == b.x && a.y == b.y;
}

int main() {
    field_checks<Q29P, FqP>("Fq");
    field_checks<R29P, FrP>("Fr");
    // points: G = (1, 2) and multiples built with the saturated formulas
    G1Affine G;
    G.x = from_u64<FqP>(1);
    G.y = from_u64<FqP>(2);
    XYZZ acc = XYZZ::from_affine(G);
    XYZZ29 acc29 = XYZZ29::identity();
    G1Affine29 G29 = g1affine29_from_sat(G);
    xyzz29_add_affine(acc29, G29.x, G29.y, false);
    CHECK(same_point(xyzz29_to_sat(acc29), acc));
    G1Affine pts[64];
    for (int i = 0; i < 64; ++i) {
        pts[i] = xyzz_to_affine(acc);
        acc = xyzz_double(acc);
        xyzz_add_affine(acc, G.x, G.y);
    }
    // random walks of mixed adds (both signs), full adds and doublings, including P+P, P-P and identity operands
    XYZZ s = XYZZ::identity();
    XYZZ29 s29 = XYZZ29::identity();
    for (int it = 0; it < 3000; ++it) {
        uint64_t r = sm();
        int idx = (int)(r & 63), op = (int)((r >> 8) % 6);
        G1Affine p = pts[idx];
        G1Affine29 p29 = g1affine29_from_sat(p);
        if (op <= 1) {   // mixed add, +/-
            bool neg = op == 1;
            G1Affine q = p;
            if (neg) q.y = fe_neg(q.y);
            xyzz_add_affine(s, q.x, q.y);
            xyzz29_add_affine(s29, p29.x, p29.y, neg);
        } else if (op == 2) {
            s = xyzz_double(s);
            s29 = xyzz29_double(s29);
        } else if (op == 3) {   // full add with a fresh XYZZ value
            XYZZ t = XYZZ::from_affine(p);
            t = xyzz_double(t);
            XYZZ29 t29 = XYZZ29::identity();
            xyzz29_add_affine(t29, p29.x, p29.y, false);
            t29 = xyzz29_double(t29);
            xyzz_add(s, t);
            xyzz29_add(s29, t29);
        } else if (op == 4) {   // s + s through the generic add (doubling branch) or s - s
            XYZZ c = s;
            xyzz_add(s, c);
            XYZZ29 c29 = s29;
            xyzz29_add(s29, c29);
        } else {   // add the current point's own affine form (mixed-add doubling branch), then subtract it twice
            if (!s.is_identity()) {
                G1Affine self = xyzz_to_affine(s);
                G1Affine29 self29 = g1affine29_from_sat(self);
                xyzz_add_affine(s, self.x, self.y);
                xyzz29_add_affine(s29, self29.x, self29.y, false);
                G1Affine m = self;
                m.y = fe_neg(m.y);
                xyzz_add_affine(s, m.x, m.y);
                xyzz29_add_affine(s29, self29.x, self29.y, true);
                xyzz_add_affine(s, m.x, m.y);   // back to identity + ... exercises P + (-P)
                xyzz29_add_affine(s29, self29.x, self29.y, true);
            }
        }
        CHECK(s.is_identity() == s29.is_identity());
        if (!s.is_identity()) CHECK(same_point(xyzz29_to_sat(s29), s));
        if (fails > 5) break;
    }
    printf(fails ? "fq29 selftest FAILED (%d)\n" : "fq29 selftest OK\n", fails);
    return fails ? 1 : 0;
}
