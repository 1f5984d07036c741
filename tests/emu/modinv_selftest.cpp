// Host check of the division-step inversion (modinv.cuh) against Fermat's a^(p-2) for both BN254 fields:
// edge values and a few thousand pseudo-random elements.  Prints "modinv selftest OK".
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../halo2-lib_amd/csrc/field.cuh"

using namespace h2;

static uint64_t sm(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
template <class P>
static bool eq(const Fe<P> &a, const Fe<P> &b) {
    for (int i = 0; i < 8; ++i)
        if (a.l[i] != b.l[i]) return false;
    return true;
}
template <class P>
static int run(const char *name) {
    uint64_t seed = 7;
    int bad = 0;
    auto check = [&](const Fe<P> &a, const char *what) {
        Fe<P> f = fe_inv_fermat(a), d = fe_inv(a);
        if (!eq(f, d)) {
            ++bad;
            fprintf(stderr, "%s: mismatch (%s)\n", name, what);
        }
        if (!a.is_zero() && !eq(fe_mul(a, d), Fe<P>::one())) {
            ++bad;
            fprintf(stderr, "%s: a * inv(a) != 1 (%s)\n", name, what);
        }
    };
    Fe<P> z = Fe<P>::zero(), one = Fe<P>::one();
    check(z, "0");
    check(one, "1");
    check(fe_sub(z, one), "-1");
    Fe<P> two = fe_add(one, one);
    check(two, "2");
    check(fe_sub(z, two), "-2");
    Fe<P> raw1 = Fe<P>::zero();   // the integer 1 in the limbs (= R^-1 as a field element)
    raw1.l[0] = 1;
    check(raw1, "raw 1");
    Fe<P> pm1;                    // the integer p - 1
    for (int i = 0; i < 8; ++i) pm1.l[i] = P::m(i);
    pm1.l[0] -= 1;
    check(pm1, "raw p-1");
    for (int t = 0; t < 3000; ++t) {
        Fe<P> a;
        for (int i = 0; i < 8; i += 2) {
            uint64_t w = sm(seed);
            a.l[i] = (uint32_t)w;
            a.l[i + 1] = (uint32_t)(w >> 32);
        }
        a.l[7] &= 0x1fffffffu;   // < 2^253 < p
        if (t % 7 == 0) {        // small and sparse values too
            for (int i = 1 + t % 5; i < 8; ++i) a.l[i] = 0;
        }
        check(a, "random");
    }
    return bad;
}
int main() {
    int bad = run<FrP>("Fr") + run<FqP>("Fq");
    if (bad) {
        fprintf(stderr, "modinv selftest FAILED: %d mismatches\n", bad);
        return 1;
    }
    printf("modinv selftest OK\n");
    return 0;
}
