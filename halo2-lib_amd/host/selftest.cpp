// Exercises the C++ host mirror end to end through the C ABI: domain round trips and the SRS identity
// commit_lagrange(values) == commit(lagrange_to_coeff(values)).  Prints "selftest OK" and exits 0.
// usage: selftest [k]    (links against libh2hip.so — or, in CPU tests, the emulated build)
#include <stdio.h>
#include <stdlib.h>

#include "halo2_proofs.hpp"

using namespace halo2_proofs;

static uint64_t sm(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static bool jac_equal(Backend &b, const G1 &p, const G1 &q) {
    // compare via the library: p + (-q) must be the identity
    G1 pts[2] = {p, q};
    // negate q.y on the host: y -> q - y is not available without F_q here, so compare affine forms instead
    void *d = nullptr;
    G1Affine a[2];
    for (int i = 0; i < 2; ++i) {
        check(h2hip_malloc(b.raw(), sizeof(G1), &d));
        check(h2hip_upload(b.raw(), d, &pts[i], sizeof(G1)));
        check(h2hip_g1_sum_jacobian_dev(b.raw(), d, 1, H2HIP_POINT_AFFINE, &a[i]));
        check(h2hip_free(b.raw(), d));
    }
    return a[0] == a[1];
}

// A small BaseConfig circuit built on the host (1 advice column with the lookup behind q_lookup, 1 constants column), proven twice:
// the same RNG stream must give the same bytes, another stream different ones.  With --dump-proof the proof is printed in hex so that
// tests/test_host_cpp.py can compare it with the oracle prover's bytes for the same (deterministic) circuit and RNG stream.
static Fr draw_fr(uint64_t &s) {
    uint64_t c[4] = {sm(s), sm(s), sm(s), sm(s) >> 4};
    return host_fr::from_canonical(c);
}
struct StreamRng {
    uint64_t state;
    void operator()(Fr *out, size_t n) {
        for (size_t i = 0; i < n; ++i) out[i] = draw_fr(state);
    }
};
static std::vector<uint8_t> prove_small_circuit(Backend &be, uint32_t k, uint64_t rng_seed, uint64_t circuit_seed) {
    const uint32_t lb = k - 2;
    h2hip_base_circuit_params bp = {k, 1, 1, 1, 0, (int32_t)lb};
    h2hip_plonk_shape sh;
    check(h2hip_plonk_shape_of(&bp, &sh));
    const size_t n = (size_t)1 << k, m = sh.usable_rows / 4;
    const Fr zero = {{0, 0, 0, 0}}, one = host_fr::R1;
    std::vector<std::vector<Fr>> fixed(sh.num_fixed_total, std::vector<Fr>(n, zero)), advice(1, std::vector<Fr>(n, zero));
    for (size_t i = 0; i < ((size_t)1 << lb); ++i) fixed[sh.table_col][i] = host_fr::from_u64(i);
    uint64_t s = circuit_seed;
    for (size_t j = 0; j < m; ++j) {
        Fr a = (j % 3 == 0) ? host_fr::from_u64(sm(s) & (((uint64_t)1 << lb) - 1)) : draw_fr(s);
        Fr b = draw_fr(s), c = draw_fr(s);
        if (j == 1) b = advice[0][1];   // gates 0 and 1 share their `b` input (a copy constraint between advice cells)
        advice[0][4 * j] = a;
        advice[0][4 * j + 1] = b;
        advice[0][4 * j + 2] = c;
        advice[0][4 * j + 3] = host_fr::add(a, host_fr::mul(b, c));
        fixed[sh.first_q_enable_col][4 * j] = one;
        if (j % 3 == 0) fixed[sh.q_lookup_col][4 * j] = one;
    }
    std::vector<uint32_t> copies = {1, 1, 1, 5};   // permutation columns: 0 = constants, 1 = advice
    for (uint32_t t = 0; t < 8 && t < m; ++t) {    // constants exposed in the fixed column, tied to the cells that use them
        fixed[sh.first_constant_col][t] = advice[0][4 * t + 2];
        copies.insert(copies.end(), {0u, t, 1u, 4 * t + 2});
    }
    poly::kzg::ParamsKZG params = poly::kzg::ParamsKZG::setup(be, k, host_fr::from_u64(0x5eed5eed5eedULL), k >= 10);
    plonk::ProvingKey pk(be, bp, params, fixed, copies);
    pk.set_transcript_repr(host_fr::from_u64(0x1234567890abcdefULL));   // the Rust side's VerifyingKey::transcript_repr stand-in
    StreamRng rng{rng_seed};
    return plonk::create_proof(be, pk, advice, {}, rng);
}

// SURVEY.md §8 row a5 in C++: threads of chained gates (running-sum cells shared by consecutive gates) laid out over two gate columns by
// virtual_region::assign_with_constraints, proven from the columns virtual_region::assign_witnesses rebuilds out of the break points, and
// checked by libh2hip's verifier; a witness whose duplicated break cell differs from the original must be rejected.
static void layout_selftest(Backend &be) {
    const uint32_t k = 8;
    h2hip_base_circuit_params bp = {k, 2, 0, 1, 0, -1};   // two gate columns, no range chip, one constants column
    h2hip_plonk_shape sh;
    check(h2hip_plonk_shape_of(&bp, &sh));
    const size_t n = (size_t)1 << k;
    const Fr zero = {{0, 0, 0, 0}}, one = host_fr::R1;
    uint64_t s = 2024;
    std::vector<virtual_region::Thread> threads(2);
    size_t budget = sh.usable_rows + sh.usable_rows / 2;   // more cells than one column holds
    for (size_t t = 0; t < 2; ++t) {
        virtual_region::Thread &th = threads[t];
        Fr sum = zero;
        th.advice.push_back(sum);                      // | 0 | a0 | b0 | s1 | a1 | b1 | s2 | ...  (inner_product_simple's layout)
        th.selector.push_back(1);
        for (size_t j = 0; th.advice.size() + 3 <= budget / 2; ++j) {
            Fr a = draw_fr(s), b = draw_fr(s);
            sum = host_fr::add(sum, host_fr::mul(a, b));
            th.advice.insert(th.advice.end(), {a, b, sum});
            th.selector.insert(th.selector.end(), {0, 0, 1});
        }
        th.selector.back() = 0;                        // the last running sum starts no gate
    }
    virtual_region::Layout lay = virtual_region::assign_with_constraints(threads, 2, k, sh.usable_rows);
    if (lay.break_points.size() != 1 || lay.break_copies.size() != 1) throw Error(-1, "layout: expected exactly one break");
    if (!(lay.break_copies[0].first == virtual_region::RawCell{1, 0})) throw Error(-1, "layout: the break cell's duplicate is not at row 0 of the next column");
    if (!(lay.columns[1][0] == lay.columns[0][lay.break_points[0]])) throw Error(-1, "layout: duplicate value");
    std::vector<std::vector<Fr>> advice = virtual_region::assign_witnesses(threads, 2, k, lay.break_points);
    if (!(advice == lay.columns)) throw Error(-1, "layout: assign_witnesses differs from assign_with_constraints");
    std::vector<std::vector<Fr>> fixed(sh.num_fixed_total, std::vector<Fr>(n, zero));
    for (size_t c = 0; c < 2; ++c)
        for (uint32_t r : lay.q_enable_rows[c]) {
            fixed[sh.first_q_enable_col + c][r] = one;
            if (!(host_fr::add(advice[c][r], host_fr::mul(advice[c][r + 1], advice[c][r + 2])) == advice[c][r + 3])) throw Error(-1, "layout: a gate row does not hold");
        }
    // permutation columns: 0 = constants, 1.. = advice (gate columns first); the break copy, and each thread's leading 0 tied to the constant 0
    std::vector<uint32_t> copies;
    for (auto &bc : lay.break_copies) copies.insert(copies.end(), {1 + bc.first.column, bc.first.row, 1 + bc.second.column, bc.second.row});
    for (size_t t = 0; t < 2; ++t) copies.insert(copies.end(), {0u, 0u, 1 + lay.cell_of[t][0].column, lay.cell_of[t][0].row});
    Fr toxic = host_fr::from_u64(0xabcdef12345ULL);
    poly::kzg::ParamsKZG params = poly::kzg::ParamsKZG::setup(be, k, toxic, false);
    plonk::ProvingKey pk(be, bp, params, fixed, copies);
    const Fr repr = host_fr::from_u64(77);
    pk.set_transcript_repr(repr);
    StreamRng rng{5};
    std::vector<uint8_t> proof = plonk::create_proof(be, pk, advice, {}, rng);
    std::vector<G1Affine> g = params.get_g().download();
    uint8_t g2[128], s_g2[128];
    poly::kzg::ParamsKZG::g2_pair(be, toxic, g2, s_g2);
    if (!plonk::verify_proof(pk, bp, repr, g[0], g2, s_g2, {}, proof)) throw Error(-1, "layout: the proof of the laid-out circuit does not verify");
    advice[1][0] = host_fr::add(advice[1][0], one);   // the duplicate no longer equals the break cell (its own gate row is re-satisfied below)
    advice[1][3] = host_fr::add(advice[1][0], host_fr::mul(advice[1][1], advice[1][2]));
    StreamRng rng2{6};
    std::vector<uint8_t> forged = plonk::create_proof(be, pk, advice, {}, rng2);
    if (plonk::verify_proof(pk, bp, repr, g[0], g2, s_g2, {}, forged)) throw Error(-1, "layout: a broken break-cell copy was accepted");
}

int main(int argc, char **argv) {
    uint32_t k = argc > 1 ? (uint32_t)atoi(argv[1]) : 10;
    try {
        Backend be(0);
        if (argc > 2 && std::string(argv[2]) == "--dump-proof") {
            std::vector<uint8_t> proof = prove_small_circuit(be, k, 7, 99);
            for (uint8_t c : proof) printf("%02x", c);
            printf("\n");
            return 0;
        }
        {
            std::vector<uint8_t> p1 = prove_small_circuit(be, k, 7, 99), p2 = prove_small_circuit(be, k, 7, 99), p3 = prove_small_circuit(be, k, 8, 99);
            if (p1.empty() || p1 != p2) throw Error(-1, "create_proof is not deterministic for a fixed RNG stream");
            if (p1 == p3) throw Error(-1, "create_proof ignores the RNG");
        }
        poly::EvaluationDomain dom(be, 5, k);
        if (dom.extended_k() != k + 2) throw Error(-1, "extended_k");
        uint64_t seed = 42;
        std::vector<Fr> vals((size_t)1 << k);
        for (auto &v : vals) {
            uint64_t c[4] = {sm(seed), sm(seed), sm(seed), sm(seed) >> 4};
            v = host_fr::from_canonical(c);
        }
        std::vector<Fr> coeffs = vals;
        dom.lagrange_to_coeff(coeffs);
        std::vector<Fr> back = coeffs;
        arithmetic::best_fft(be, back, dom.get_omega(), k);
        if (!(back == vals)) throw Error(-1, "fft(ifft(x)) != x");
        std::vector<Fr> ext = dom.coeff_to_extended(coeffs);
        dom.extended_to_coeff(ext);
        for (size_t i = 0; i < ext.size(); ++i) {
            Fr want = i < coeffs.size() ? coeffs[i] : Fr{{0, 0, 0, 0}};
            if (!(ext[i] == want)) throw Error(-1, "extended round trip");
        }
        Fr s = host_fr::from_u64(0x123456789abcdefULL);
        poly::kzg::ParamsKZG params = poly::kzg::ParamsKZG::setup(be, k, s, true);
        G1 c1 = params.commit_lagrange(vals), c2 = params.commit(coeffs);
        if (!jac_equal(be, c1, c2)) throw Error(-1, "commit_lagrange(values) != commit(coeffs)");
        // eval_polynomial vs host Horner; kate_division: q(y)*(y - b) + f(b) == f(y)
        Fr x = host_fr::from_u64(0xfeedfacecafeULL), y = host_fr::from_u64(0x1234567ULL);
        Fr hx = Fr{{0, 0, 0, 0}}, hy = hx;
        for (size_t i = coeffs.size(); i-- > 0;) {
            hx = host_fr::add(host_fr::mul(hx, x), coeffs[i]);
            hy = host_fr::add(host_fr::mul(hy, y), coeffs[i]);
        }
        if (!(arithmetic::eval_polynomial(be, coeffs, x) == hx)) throw Error(-1, "eval_polynomial");
        std::vector<Fr> q = arithmetic::kate_division(be, coeffs, x);
        Fr qy = arithmetic::eval_polynomial(be, q, y);
        if (!(host_fr::add(host_fr::mul(qy, host_fr::add(y, host_fr::neg(x))), hx) == hy)) throw Error(-1, "kate_division");
        // divide_by_vanishing_poly: (f * t) / t == f on the extended coset, t(x_i) = zeta^n * (w_ext^n)^i - 1 from the host
        {
            std::vector<Fr> e = dom.coeff_to_extended(coeffs), num(e.size());
            uint64_t nexp[4] = {(uint64_t)1 << k, 0, 0, 0};
            Fr zn = host_fr::pow(host_fr::from_canonical(host_fr::ZETA), nexp), step = host_fr::pow(dom.get_extended_omega(), nexp);
            Fr xn = zn, minus_one = host_fr::neg(host_fr::R1);
            for (size_t i = 0; i < e.size(); ++i) {
                num[i] = host_fr::mul(e[i], host_fr::add(xn, minus_one));
                xn = host_fr::mul(xn, step);
            }
            dom.divide_by_vanishing_poly(num);
            if (!(num == e)) throw Error(-1, "divide_by_vanishing_poly");
        }
        // lookup permutation: inputs drawn from a small table
        {
            size_t rows = (size_t)1 << k, usable = rows - 6;
            std::vector<Fr> table(rows), input(rows);
            for (size_t i = 0; i < rows; ++i) table[i] = host_fr::from_u64(i < 64 ? i : 0);
            for (size_t i = 0; i < rows; ++i) input[i] = host_fr::from_u64(sm(seed) % 64);
            for (size_t i = usable; i < rows; ++i) table[i] = input[i] = host_fr::from_u64(sm(seed));   // blinding rows: ignored
            auto perm = plonk::lookup::permute_expression_pair(be, input, table, usable);
            if (perm.first.size() != usable || perm.second.size() != usable) throw Error(-1, "permute size");
            if (!(perm.first[0] == perm.second[0])) throw Error(-1, "permute: first row");
            for (size_t i = 1; i < usable; ++i) {
                bool same_as_prev = perm.first[i] == perm.first[i - 1];
                if (!same_as_prev && !(perm.first[i] == perm.second[i])) throw Error(-1, "permute: A'[i] != S'[i] at a run start");
            }
            bool threw2 = false;
            try {
                input[0] = host_fr::from_u64(1000);   // not in the table
                plonk::lookup::permute_expression_pair(be, input, table, usable);
            } catch (const Error &) {
                threw2 = true;
            }
            if (!threw2) throw Error(-1, "permute: missing table value not reported");
        }
        if (k <= 12) {   // g_to_lagrange(g) reproduces the setup's Lagrange basis
            poly::kzg::ParamsKZG derived = poly::kzg::ParamsKZG::from_parts(be, k, Bases(be, params.get_g().download(), false), false);
            if (!(derived.get_g_lagrange().download() == params.get_g_lagrange().download())) throw Error(-1, "g_to_lagrange");
        }
        bool threw = false;
        try {
            std::vector<Fr> shorter(coeffs.begin(), coeffs.end() - 1);
            arithmetic::best_multiexp(be, shorter, params.get_g());
        } catch (const Error &) {
            threw = true;
        }
        if (!threw) throw Error(-1, "length assertion missing");
        layout_selftest(be);
        printf("selftest OK (k=%u)\n", k);
        return 0;
    } catch (const Error &e) {
        fprintf(stderr, "selftest FAILED: %s (code %d)\n", e.what(), e.code);
        return 1;
    }
}
