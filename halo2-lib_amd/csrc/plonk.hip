// plonk::create_proof for halo2-base circuits with every polynomial resident in HBM.
//
// What the reference runs at halo2-base/src/utils/testing.rs:32-50
//     create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<_>, _, Blake2bWrite<Vec<u8>, G1Affine, _>, _>
// lives in the un-vendored halo2-axiom 0.5.3 (Cargo.lock:1063-1065); the protocol below restates it [UPSTREAM-RECALL, SURVEY.md §3.2
// and Appendix A] for the one constraint system halo2-lib builds, BaseConfig::configure (halo2-base/src/gates/circuit/mod.rs:70-96):
//     FlexGateConfig   one selector + the gate q*(a + b*c - d) at rotations 0..3 per advice column, constants columns with equality
//                      (halo2-base/src/gates/flex_gate/mod.rs:61-91,120-146)
//     RangeConfig      table column first; with a single advice column the lookup is (q_lookup * a, table) on that column, else
//                      dedicated lookup-advice columns (a, table)  (halo2-base/src/gates/range/mod.rs:71-150)
//     instance columns with equality (gates/circuit/mod.rs:89-91)
// This file is host code: the Blake2b transcript, the order of operations, and SHPLONK's O(#openings) bookkeeping.  All vector work
// goes through the kernels behind the C ABI (include/h2hip.h).  First phase only.
#include <algorithm>
#include <chrono>
#include <map>
#include <vector>

#include "blake2b.h"
#include "internal.h"

namespace h2 {
namespace plonk {

// ---------------------------------------------------------------------------------------------- host field helpers
static Fr fr_from_canonical_u64x4(const uint64_t v[4]) {
    Fr a;
    memcpy(a.l, v, 32);
    return fe_to_mont(a);
}
static Fr fr_from_u64(uint64_t v) {
    uint64_t w[4] = {v, 0, 0, 0};
    return fr_from_canonical_u64x4(w);
}
// canonical little-endian bytes (to_repr)
static void fr_repr(const Fr &a, uint8_t out[32]) {
    Fr c = fe_from_mont(a);
    memcpy(out, c.l, 32);
}
static void fq_repr(const Fq &a, uint8_t out[32]) {
    Fq c = fe_from_mont(a);
    memcpy(out, c.l, 32);
}
// numeric order of the canonical values (Ord for Fr compares to_repr from the most significant byte)
static int fr_cmp(const Fr &a, const Fr &b) {
    Fr x = fe_from_mont(a), y = fe_from_mont(b);
    for (int i = 7; i >= 0; --i)
        if (x.l[i] != y.l[i]) return x.l[i] < y.l[i] ? -1 : 1;
    return 0;
}
struct FrLess {
    bool operator()(const Fr &a, const Fr &b) const { return fr_cmp(a, b) < 0; }
};
// Fr::from_uniform_bytes: 512-bit little-endian integer mod r = d0 + d1 * 2^256
static Fr fr_from_uniform_bytes(const uint8_t b[64]) {
    Fr d0, d1;
    memcpy(d0.l, b, 32);
    memcpy(d1.l, b + 32, 32);
    const Fr r2 = Fr::r2(), r3 = fe_mul(r2, r2);
    return fe_add(fe_mul(d0, r2), fe_mul(d1, r3));
}
static const uint64_t ROOT_OF_UNITY[4] = {0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL};   // 7^((r-1)/2^28)
static const uint64_t ZETA[4] = {0xb8ca0b2d36636f23ULL, 0xcc37a73fec2bc5e9ULL, 0x048b6e193fd84104ULL, 0x30644e72e131a029ULL};            // 7^(2(r-1)/3)
static const uint64_t DELTA[4] = {0x870e56bbe533e9a2ULL, 0x5b5f898e5e963f25ULL, 0x64ec26aad4c86e71ULL, 0x09226b6e22c6f0caULL};           // 7^(2^28)

// ---------------------------------------------------------------------------------------------- constraint-system shape
struct Lookup {
    int q_col;   // fixed column of the complex selector, or -1
    int advice_col, table_col;
};
struct ColumnRef {
    int kind;   // 0 = fixed, 1 = advice, 2 = instance
    int index;
};
struct Shape {
    h2hip_base_circuit_params p;
    uint32_t k, n;
    bool with_range, single;
    int table_col = -1, q_lookup_col = -1, first_constant_col = -1, first_q_enable_col = -1;
    uint32_t num_advice_total, num_fixed_total;
    std::vector<Lookup> lookups;
    std::vector<ColumnRef> perm_columns;
    std::vector<std::pair<int, int>> advice_queries, fixed_queries;   // (column, rotation) in first-query order
    uint32_t degree, blinding_factors, usable_rows, chunk_len, quotient_pieces, extended_k, num_perm_sets;

    int init(const h2hip_base_circuit_params &bp) {
        p = bp;
        k = bp.k;
        H2_REQUIRE(k >= 4 && k <= 26, "k out of range (4..26)");
        // the reference's configurations go up to 291 gate + 53 lookup advice columns (halo2-ecc/configs/secp256k1/bench_ecdsa.config:9)
        H2_REQUIRE(bp.num_advice >= 1 && bp.num_advice <= 1024 && bp.num_lookup_advice <= 256 && bp.num_fixed <= 16 && bp.num_instance <= 8,
                   "column counts out of range");
        H2_REQUIRE(bp.lookup_bits < (int32_t)k, "lookup_bits must be less than k");
        n = 1u << k;
        with_range = bp.lookup_bits >= 0 && bp.num_lookup_advice != 0;
        single = with_range && bp.num_advice == 1;   // range/mod.rs:93-95: the lookup sits on the gate column behind a complex selector
        int nf = 0;
        if (with_range) table_col = nf++;            // meta.lookup_table_column() is created first (range/mod.rs:82)
        first_constant_col = bp.num_fixed ? nf : -1;
        nf += (int)bp.num_fixed;                     // flex_gate/mod.rs:123-129
        // selectors are compressed into fixed columns after the circuit's own ones: the complex selector keeps a column of its own, and
        // the per-column gate selectors are enabled on common rows, so none of them can share a column either [UPSTREAM compress_selectors]
        if (single) q_lookup_col = nf++;
        first_q_enable_col = nf;
        nf += (int)bp.num_advice;
        num_fixed_total = (uint32_t)nf;
        const uint32_t nla = (single || !with_range) ? 0 : bp.num_lookup_advice;
        num_advice_total = bp.num_advice + nla;
        if (single) lookups.push_back({q_lookup_col, 0, table_col});
        for (uint32_t i = 0; i < nla; ++i) lookups.push_back({-1, (int)(bp.num_advice + i), table_col});
        // enable_equality order: constants, gate advice, lookup advice, instance
        for (uint32_t i = 0; i < bp.num_fixed; ++i) perm_columns.push_back({0, first_constant_col + (int)i});
        for (uint32_t i = 0; i < num_advice_total; ++i) perm_columns.push_back({1, (int)i});
        for (uint32_t i = 0; i < bp.num_instance; ++i) perm_columns.push_back({2, (int)i});
        for (uint32_t a = 0; a < bp.num_advice; ++a)
            for (int r = 0; r < 4; ++r) advice_queries.push_back({(int)a, r});
        for (uint32_t i = 0; i < nla; ++i) advice_queries.push_back({(int)(bp.num_advice + i), 0});
        for (uint32_t i = 0; i < bp.num_fixed; ++i) fixed_queries.push_back({first_constant_col + (int)i, 0});
        if (with_range) fixed_queries.push_back({table_col, 0});
        if (single) fixed_queries.push_back({q_lookup_col, 0});
        for (uint32_t i = 0; i < bp.num_advice; ++i) fixed_queries.push_back({first_q_enable_col + (int)i, 0});
        degree = 3;   // gate and permutation argument
        for (const Lookup &l : lookups) degree = std::max<uint32_t>(degree, std::max<uint32_t>(4, 2 + (l.q_col >= 0 ? 2 : 1) + 1));
        blinding_factors = std::max<uint32_t>(3, 4) + 2;   // a gate column is queried at four rotations
        H2_REQUIRE(n > blinding_factors + 8, "k too small for the blinding rows");
        usable_rows = n - (blinding_factors + 1);
        chunk_len = degree - 2;
        quotient_pieces = degree - 1;
        extended_k = k;
        while (((uint64_t)1 << extended_k) < (uint64_t)n * quotient_pieces) ++extended_k;
        H2_REQUIRE(extended_k <= 28, "extended domain exceeds the 2-adicity of F_r");
        num_perm_sets = (uint32_t)((perm_columns.size() + chunk_len - 1) / chunk_len);
        // range/mod.rs:117-121: the table must fit gate.max_rows = 2^k - meta.minimum_rows() = n - (blinding_factors + 3)
        if (bp.lookup_bits >= 0 && with_range) H2_REQUIRE(((uint64_t)1 << bp.lookup_bits) <= n - (blinding_factors + 3), "lookup table is too large for the circuit degree plus blinding factors");
        return H2HIP_OK;
    }
    uint32_t num_commitments() const {
        return num_advice_total + 3 * (uint32_t)lookups.size() + num_perm_sets + 1 + quotient_pieces + 2;
    }
    uint32_t num_evals() const {
        return (uint32_t)advice_queries.size() + (uint32_t)fixed_queries.size() + 1 + (uint32_t)perm_columns.size() +
               (num_perm_sets ? 3 * num_perm_sets - 1 : 0) + 5 * (uint32_t)lookups.size();
    }
};

struct Domain {
    Fr omega, omega_inv, ext_omega, ext_omega_inv, zeta, zeta_inv, ifft_divisor, ext_ifft_divisor, delta;
    void init(uint32_t k, uint32_t ek) {
        ext_omega = fr_from_canonical_u64x4(ROOT_OF_UNITY);
        for (uint32_t i = ek; i < 28; ++i) ext_omega = fe_sqr(ext_omega);
        omega = ext_omega;
        for (uint32_t i = k; i < ek; ++i) omega = fe_sqr(omega);
        omega_inv = fe_inv(omega);
        ext_omega_inv = fe_inv(ext_omega);
        zeta = fr_from_canonical_u64x4(ZETA);
        zeta_inv = fe_sqr(zeta);
        ifft_divisor = fe_inv(fr_from_u64((uint64_t)1 << k));
        ext_ifft_divisor = fe_inv(fr_from_u64((uint64_t)1 << ek));
        delta = fr_from_canonical_u64x4(DELTA);
    }
};

// ---------------------------------------------------------------------------------------------- device buffers
// size-keyed pool: a proof's buffers go back to the key's pool when it is done, so a second proof allocates nothing
struct BufPool {
    std::multimap<size_t, void *> free_;
    std::vector<void *> all_;
    int take(size_t bytes, void **out) {
        auto it = free_.find(bytes);
        if (it != free_.end()) {
            *out = it->second;
            free_.erase(it);
            return H2HIP_OK;
        }
        hipError_t e = hipMalloc(out, bytes ? bytes : 256);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return H2HIP_ERR_NOMEM;
        }
        all_.push_back(*out);
        return H2HIP_OK;
    }
    void give(size_t bytes, void *p) { free_.insert({bytes, p}); }
    void destroy() {
        for (void *p : all_) hipFree(p);
        all_.clear();
        free_.clear();
    }
};
// buffers taken during one call; returned to the pool on scope exit (also on the error paths)
struct Scope {
    BufPool *pool;
    std::vector<std::pair<size_t, void *>> held;
    explicit Scope(BufPool *p) : pool(p) {}
    ~Scope() {
        for (auto &h : held) pool->give(h.first, h.second);
    }
    int take(size_t elems, Fr **out) {
        void *p = nullptr;
        H2_CHK(pool->take(sizeof(Fr) * elems, &p));
        held.push_back({sizeof(Fr) * elems, p});
        *out = (Fr *)p;
        return H2HIP_OK;
    }
    void release(Fr *p) {   // early return of one buffer
        for (size_t i = 0; i < held.size(); ++i)
            if (held[i].second == (void *)p) {
                pool->give(held[i].first, held[i].second);
                held.erase(held.begin() + (long)i);
                return;
            }
    }
};

}  // namespace plonk
}  // namespace h2

using namespace h2;
using namespace h2::plonk;

struct h2hip_plonk_pk {
    Shape sh;
    Domain dom;
    h2hip_ctx *ctx = nullptr;
    const h2hip_bases *g = nullptr, *g_lagrange = nullptr;
    std::vector<Fr *> fixed_values, fixed_polys, fixed_cosets, sigma_values, sigma_polys, sigma_cosets;
    Fr *l0 = nullptr, *l_last = nullptr, *l_blind = nullptr;   // extended-domain evaluations
    void *table_sorted = nullptr;                               // sorted keys of the lookup table column (prepared once: the table is fixed)
    std::vector<G1Affine> fixed_commitments, permutation_commitments;
    Fr transcript_repr;
    bool have_repr = false;
    BufPool pool;
    std::vector<void *> owned;
    // multi-GPU (h2hip_plonk_pk_set_sharding): point-range sharding of every commitment, coset sharding of h(X)'s numerator
    const h2hip_bases *g_shard = nullptr, *g_lagrange_shard = nullptr;
    size_t shard_offset = 0, shard_len = 0;
    uint32_t shard_world = 1, shard_rank = 0;
    h2hip_comm *comm = nullptr;
    bool shard_quotient = false, shard_products = false, shard_ntt = false;
    std::vector<uint32_t> my_cosets;          // cosets of the extended domain (rows = coset mod 2^(ek-k)) this rank evaluates h(X) on
    uint32_t max_cosets = 1;                  // cosets of the busiest rank (the all-gather's uniform slot count)
    std::vector<Fr *> fixed_cosets_sh, sigma_cosets_sh;   // [my_cosets][n] slices of the key's extended-domain arrays
    Fr *l0_sh = nullptr, *l_last_sh = nullptr, *l_blind_sh = nullptr;
    std::vector<void *> shard_owned;
    // exchanges of the running sharded proof: every host exchange carries a status word, so that a rank that fails between two
    // exchanges can tell its peers (it takes part in the NEXT exchange with an error status and a zero payload of the scheduled size)
    std::vector<size_t> exch_sizes;
    size_t exch_next = 0;
    hipStream_t copy_stream = nullptr;   // the RNG-drawn random polynomial is uploaded on its own stream, next to the NTTs
    hipEvent_t copy_ev = nullptr;
    h2hip_ctx *side = nullptr;           // child context (own stream, NTT scratch and twiddle cache): the transforms that run next to an MSM's tail
    hipEvent_t side_ev = nullptr, side_ev1 = nullptr;   // side_ev: everything queued on the side stream so far; side_ev1: the first-round columns' transforms
    Fr *host_stage = nullptr;   // pinned staging for the RNG-drawn scalars (the n coefficients of the random polynomial, the blinding rows)
    size_t host_stage_elems = 0;
};

namespace h2 {
namespace plonk {

// ---------------------------------------------------------------------------------------------- transcript
static const unsigned SIGN_BIT = 6, INF_BIT = 7;   // compressed G1: sign(y) and identity flags in the top byte (halo2curves new_curve_impl!; the
                                                   // positions are UNVERIFIED for halo2curves-axiom 0.7.3, see oracle/transcript.py)
struct Transcript {   // Blake2bWrite<Vec<u8>, G1Affine, Challenge255<_>>  
    Blake2b st;
    std::vector<uint8_t> proof;
    Transcript() : st(64, "Halo2-Transcript") {}
    void common_scalar(const Fr &s) {
        uint8_t b[33];
        b[0] = 0x02;
        fr_repr(s, b + 1);
        st.update(b, 33);
    }
    void write_scalar(const Fr &s) {
        common_scalar(s);
        uint8_t b[32];
        fr_repr(s, b);
        proof.insert(proof.end(), b, b + 32);
    }
    int write_point(const G1Affine &p) {
        if (p.x.is_zero() && p.y.is_zero()) {   // upstream: io::Error "cannot write points at infinity to the transcript"
            set_error("create_proof: a commitment is the point at infinity and cannot be written to the transcript");
            return H2HIP_ERR_INVALID;
        }
        uint8_t b[65];
        b[0] = 0x01;
        fq_repr(p.x, b + 1);
        fq_repr(p.y, b + 33);
        st.update(b, 65);
        uint8_t c[32];
        memcpy(c, b + 1, 32);
        c[31] |= (uint8_t)((b[33] & 1) << SIGN_BIT);
        proof.insert(proof.end(), c, c + 32);
        return H2HIP_OK;
    }
    Fr squeeze_challenge() {
        uint8_t z = 0x00, d[64];
        st.update(&z, 1);
        st.digest(d);
        return fr_from_uniform_bytes(d);
    }
};

static G1Affine jacobian_to_affine(const G1Jac &p) {
    G1Affine r;
    if (p.z.is_zero()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
        return r;
    }
    const Fq zi = fe_inv(p.z), zi2 = fe_sqr(zi);
    r.x = fe_mul(p.x, zi2);
    r.y = fe_mul(p.y, fe_mul(zi2, zi));
    return r;
}

// ---------------------------------------------------------------------------------------------- permutation keygen
// permutation::keygen::Assembly [UPSTREAM-RECALL]: cycles merged by `copy`; sigma_i(omega^j) = delta^i' * omega^j' for mapping[i][j] = (i', j')
struct Assembly {
    uint32_t m, n;
    std::vector<uint32_t> map_c, map_r, aux_c, aux_r, sizes;
    Assembly(uint32_t m_, uint32_t n_) : m(m_), n(n_), map_c((size_t)m_ * n_), map_r((size_t)m_ * n_), aux_c((size_t)m_ * n_), aux_r((size_t)m_ * n_), sizes((size_t)m_ * n_, 1) {
        for (uint32_t c = 0; c < m; ++c)
            for (uint32_t r = 0; r < n; ++r) {
                size_t i = (size_t)c * n + r;
                map_c[i] = aux_c[i] = c;
                map_r[i] = aux_r[i] = r;
            }
    }
    void copy(uint32_t lc, uint32_t lr, uint32_t rc, uint32_t rr) {
        size_t li = (size_t)lc * n + lr, ri = (size_t)rc * n + rr;
        uint32_t lcy_c = aux_c[li], lcy_r = aux_r[li], rcy_c = aux_c[ri], rcy_r = aux_r[ri];
        if (lcy_c == rcy_c && lcy_r == rcy_r) return;
        size_t lcy = (size_t)lcy_c * n + lcy_r, rcy = (size_t)rcy_c * n + rcy_r;
        if (sizes[lcy] < sizes[rcy]) {
            std::swap(lcy, rcy);
            std::swap(lcy_c, rcy_c);
            std::swap(lcy_r, rcy_r);
        }
        sizes[lcy] += sizes[rcy];
        size_t i = rcy;
        do {
            aux_c[i] = lcy_c;
            aux_r[i] = lcy_r;
            i = (size_t)map_c[i] * n + map_r[i];
        } while (i != rcy);
        std::swap(map_c[li], map_c[ri]);
        std::swap(map_r[li], map_r[ri]);
    }
};

static int dev_alloc(h2hip_plonk_pk *pk, size_t elems, Fr **out) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, sizeof(Fr) * (elems ? elems : 1));
    if (e != hipSuccess) {
        set_error("hipMalloc for the proving key failed: %s", hipGetErrorString(e));
        return H2HIP_ERR_NOMEM;
    }
    pk->owned.push_back(p);
    *out = (Fr *)p;
    return H2HIP_OK;
}

// values (n, resident) -> (poly, coset): lagrange_to_coeff then coeff_to_extended
static int to_poly_and_coset(h2hip_plonk_pk *pk, const Fr *values, Fr **poly, Fr **coset) {
    h2hip_ctx *ctx = pk->ctx;
    const Shape &sh = pk->sh;
    const Domain &d = pk->dom;
    H2_CHK(dev_alloc(pk, sh.n, poly));
    H2_CHK(dev_alloc(pk, (size_t)1 << sh.extended_k, coset));
    H2_HIPCHK(hipMemcpyAsync(*poly, values, sizeof(Fr) * sh.n, hipMemcpyDeviceToDevice, ctx->stream));
    H2_CHK(h2hip_ifft_dev(ctx, *poly, &d.omega_inv, sh.k, &d.ifft_divisor));
    return h2hip_coeff_to_extended_dev(ctx, *poly, sh.k, *coset, sh.extended_k, &d.ext_omega, &d.zeta);
}

static int keygen_impl(h2hip_ctx *ctx, h2hip_plonk_pk *pk, const void *const *fixed_host, const uint32_t *copies, size_t ncopies) {
    const Shape &sh = pk->sh;
    const uint32_t n = sh.n, m = (uint32_t)sh.perm_columns.size();
    // ---- fixed columns
    for (uint32_t c = 0; c < sh.num_fixed_total; ++c) {
        H2_REQUIRE(fixed_host[c], "NULL fixed column");
        Fr *v = nullptr, *p = nullptr, *e = nullptr;
        H2_CHK(dev_alloc(pk, n, &v));
        H2_HIPCHK(hipMemcpyAsync(v, fixed_host[c], sizeof(Fr) * n, hipMemcpyHostToDevice, ctx->stream));
        H2_CHK(to_poly_and_coset(pk, v, &p, &e));
        pk->fixed_values.push_back(v);
        pk->fixed_polys.push_back(p);
        pk->fixed_cosets.push_back(e);
    }
    // ---- permutation: cycles on the host, sigma values delta^c' * omega^r'
    Assembly as(m, n);
    for (size_t i = 0; i < ncopies; ++i) {
        const uint32_t *c = copies + 4 * i;
        H2_REQUIRE(c[0] < m && c[2] < m, "copy constraint names a column outside the permutation");
        H2_REQUIRE(c[1] < sh.usable_rows && c[3] < sh.usable_rows, "copy constraint outside the usable rows (NotEnoughRowsAvailable)");
        as.copy(c[0], c[1], c[2], c[3]);
    }
    std::vector<Fr> wpow(n), dpow(m ? m : 1), col(n);
    wpow[0] = Fr::one();
    for (uint32_t j = 1; j < n; ++j) wpow[j] = fe_mul(wpow[j - 1], pk->dom.omega);
    dpow[0] = Fr::one();
    for (uint32_t i = 1; i < m; ++i) dpow[i] = fe_mul(dpow[i - 1], pk->dom.delta);
    for (uint32_t c = 0; c < m; ++c) {
        for (uint32_t r = 0; r < n; ++r) {
            size_t i = (size_t)c * n + r;
            col[r] = as.map_c[i] == 0 ? wpow[as.map_r[i]] : fe_mul(dpow[as.map_c[i]], wpow[as.map_r[i]]);
        }
        Fr *v = nullptr, *p = nullptr, *e = nullptr;
        H2_CHK(dev_alloc(pk, n, &v));
        H2_HIPCHK(hipMemcpyAsync(v, col.data(), sizeof(Fr) * n, hipMemcpyHostToDevice, ctx->stream));
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));   // `col` is reused
        H2_CHK(to_poly_and_coset(pk, v, &p, &e));
        pk->sigma_values.push_back(v);
        pk->sigma_polys.push_back(p);
        pk->sigma_cosets.push_back(e);
    }
    if (!sh.lookups.empty()) {   // every lookup of a BaseConfig circuit uses the one range table
        void *ts = nullptr;
        hipError_t e = hipMalloc(&ts, h2hip_lookup_sorted_table_bytes(sh.usable_rows));
        if (e != hipSuccess) {
            set_error("hipMalloc for the sorted lookup table failed: %s", hipGetErrorString(e));
            return H2HIP_ERR_NOMEM;
        }
        pk->owned.push_back(ts);
        pk->table_sorted = ts;
        H2_CHK(h2hip_lookup_table_sort_dev(ctx, pk->fixed_values[sh.table_col], sh.usable_rows, ts));
    }
    // ---- commitments of the verifying key
    std::vector<const void *> cols;
    for (Fr *v : pk->fixed_values) cols.push_back(v);
    for (Fr *v : pk->sigma_values) cols.push_back(v);
    std::vector<G1Affine> comm(cols.size());
    if (!cols.empty()) H2_CHK(h2hip_msm_g1_batch_dev(ctx, pk->g_lagrange, cols.data(), n, cols.size(), H2HIP_POINT_AFFINE, comm.data()));
    pk->fixed_commitments.assign(comm.begin(), comm.begin() + sh.num_fixed_total);
    pk->permutation_commitments.assign(comm.begin() + sh.num_fixed_total, comm.end());
    // ---- l_0, l_last, l_blind on the extended domain
    Fr *tmp = nullptr, *tp = nullptr;
    H2_CHK(dev_alloc(pk, n, &tmp));
    std::vector<Fr> lv(n);
    const Fr one = Fr::one(), zero = Fr::zero();
    const uint32_t bf = sh.blinding_factors;
    Fr **dst[3] = {&pk->l0, &pk->l_last, &pk->l_blind};
    for (int which = 0; which < 3; ++which) {
        for (uint32_t r = 0; r < n; ++r) lv[r] = zero;
        if (which == 0) lv[0] = one;
        if (which == 1) lv[n - bf - 1] = one;
        if (which == 2)
            for (uint32_t r = n - bf; r < n; ++r) lv[r] = one;
        H2_HIPCHK(hipMemcpyAsync(tmp, lv.data(), sizeof(Fr) * n, hipMemcpyHostToDevice, ctx->stream));
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));
        H2_CHK(to_poly_and_coset(pk, tmp, &tp, dst[which]));
    }
    H2_HIPCHK(hipStreamSynchronize(ctx->stream));
    return H2HIP_OK;
}

// ---------------------------------------------------------------------------------------------- SHPLONK bookkeeping
constexpr size_t SHPLONK_MAX_OPENINGS = 64;   // (rotation set, point) pairs of one proof: the slots of the sharded prover's carry exchange
struct Query {
    int poly;   // index into the prover's polynomial list
    Fr point, eval;
};
struct RotationSet {
    std::vector<Fr> points;                          // ascending
    std::vector<int> polys;                          // in order of first appearance
    std::vector<std::vector<Fr>> evals;              // [poly][point]
};
// poly/kzg/multiopen/shplonk.rs::construct_intermediate_sets [UPSTREAM-RECALL]
static void construct_intermediate_sets(const std::vector<Query> &queries, std::vector<RotationSet> &sets, std::vector<Fr> &super_points) {
    // A proof asks ~1000 queries at a handful of points (x and its rotations): every point gets a small index first, every polynomial a bit
    // mask of the points it is opened at — sets are then found by comparing masks, without a container per polynomial.
    std::vector<Fr> distinct;                  // in order of first appearance
    std::vector<uint32_t> point_of(queries.size());
    int max_poly = -1;
    for (size_t qi = 0; qi < queries.size(); ++qi) {
        uint32_t pi = 0;
        while (pi < distinct.size() && fr_cmp(distinct[pi], queries[qi].point) != 0) ++pi;
        if (pi == distinct.size()) distinct.push_back(queries[qi].point);
        point_of[qi] = pi;
        if (queries[qi].poly > max_poly) max_poly = queries[qi].poly;
    }
    const size_t np = distinct.size();
    std::vector<uint32_t> by_value(np);        // point indices in ascending order of the field element
    for (uint32_t i = 0; i < np; ++i) by_value[i] = i;
    std::sort(by_value.begin(), by_value.end(), [&](uint32_t a, uint32_t b) { return fr_cmp(distinct[a], distinct[b]) < 0; });
    super_points.clear();
    for (uint32_t i : by_value) super_points.push_back(distinct[i]);
    if (np > 64) {   // cannot happen for halo2-base's constraint system (rotations -1..3 and the last row); keep the masks honest
        sets.clear();
        return;
    }
    const size_t P = (size_t)max_poly + 1;
    std::vector<uint64_t> mask(P, 0);
    std::vector<int> first_query(P, -1), order;
    std::vector<size_t> eval_at(P * np, (size_t)-1);   // [poly][point] -> first query asking it
    for (size_t qi = 0; qi < queries.size(); ++qi) {
        const int poly = queries[qi].poly;
        if (first_query[poly] < 0) {
            first_query[poly] = (int)qi;
            order.push_back(poly);
        }
        mask[poly] |= 1ull << point_of[qi];
        size_t &slot = eval_at[(size_t)poly * np + point_of[qi]];
        if (slot == (size_t)-1) slot = qi;
    }
    sets.clear();
    std::vector<uint64_t> set_mask;
    for (int poly : order) {
        size_t si = 0;
        while (si < set_mask.size() && set_mask[si] != mask[poly]) ++si;
        if (si == set_mask.size()) {
            set_mask.push_back(mask[poly]);
            sets.push_back(RotationSet());
            for (uint32_t i : by_value)
                if (mask[poly] >> i & 1) sets.back().points.push_back(distinct[i]);
        }
        RotationSet &rs = sets[si];
        rs.polys.push_back(poly);
        rs.evals.emplace_back();
        std::vector<Fr> &ev = rs.evals.back();
        ev.reserve(rs.points.size());
        for (uint32_t i : by_value)
            if (mask[poly] >> i & 1) ev.push_back(queries[eval_at[(size_t)poly * np + i]].eval);
    }
}
// Lagrange basis of a point set: basis[j] = coefficients (low to high) of L_j(X) = prod_{i != j} (X - x_i) / (x_j - x_i).  Computed once
// per rotation set (its m field inversions are the expensive part on the host), then shared by all polynomials opened on that set.
static std::vector<std::vector<Fr>> lagrange_basis(const std::vector<Fr> &points, std::vector<Fr> &weights) {
    const size_t m = points.size();
    std::vector<std::vector<Fr>> basis(m);
    weights.assign(m, Fr::one());   // weights[j] = 1 / prod_{i != j} (x_j - x_i)
    for (size_t j = 0; j < m; ++j) {
        std::vector<Fr> num(1, Fr::one());
        Fr den = Fr::one();
        for (size_t i = 0; i < m; ++i) {
            if (i == j) continue;
            std::vector<Fr> nxt(num.size() + 1, Fr::zero());
            for (size_t t = 0; t < num.size(); ++t) {
                nxt[t] = fe_sub(nxt[t], fe_mul(num[t], points[i]));
                nxt[t + 1] = fe_add(nxt[t + 1], num[t]);
            }
            num.swap(nxt);
            den = fe_mul(den, fe_sub(points[j], points[i]));
        }
        const Fr scale = m == 1 ? Fr::one() : fe_inv(den);
        weights[j] = scale;
        for (Fr &c : num) c = fe_mul(c, scale);
        basis[j] = num;
    }
    return basis;
}
// coefficients (low to high) of the polynomial of degree < m through (points[j], evals[j])
static std::vector<Fr> lagrange_interpolate(const std::vector<std::vector<Fr>> &basis, const std::vector<Fr> &evals) {
    const size_t m = basis.size();
    std::vector<Fr> out(m, Fr::zero());
    for (size_t j = 0; j < m; ++j)
        for (size_t t = 0; t < m; ++t) out[t] = fe_add(out[t], fe_mul(basis[j][t], evals[j]));
    return out;
}
static Fr eval_small(const std::vector<Fr> &c, const Fr &x) {
    Fr acc = Fr::zero();
    for (size_t i = c.size(); i-- > 0;) acc = fe_add(fe_mul(acc, x), c[i]);
    return acc;
}

// ---------------------------------------------------------------------------------------------- create_proof
static const char *STAGE_NAMES[H2HIP_PLONK_STAGES] = {
    "advice_upload_blinding", "lookup_permute", "commit_advice_lookup_permuted", "grand_products", "ntt_round1_columns_and_commit_products_random",
    "lagrange_to_coeff", "coeff_to_extended", "quotient_terms", "quotient_to_coeff", "commit_h_pieces", "evaluations", "multiopen_shplonk"};
enum { ST_UPLOAD = 0, ST_LOOKUP_PERMUTE, ST_COMMIT_ROUND1, ST_PRODUCTS, ST_COMMIT_PRODUCTS, ST_TO_COEFF, ST_TO_EXT, ST_QUOTIENT, ST_H_COEFF, ST_COMMIT_H,
       ST_EVALS, ST_MULTIOPEN };

struct Laps {
    h2hip_ctx *ctx;
    double *ms;
    std::chrono::steady_clock::time_point t;
    Laps(h2hip_ctx *c, double *m) : ctx(c), ms(m) {
        if (ms) {
            for (int i = 0; i < H2HIP_PLONK_STAGES; ++i) ms[i] = 0;
            t = std::chrono::steady_clock::now();
        }
    }
    void lap(int stage) {
        if (!ms) return;
        hipStreamSynchronize(ctx->stream);
        auto now = std::chrono::steady_clock::now();
        ms[stage] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
    }
};

static int create_proof_impl(h2hip_ctx *ctx, h2hip_plonk_pk *pk, const void *const *advice, bool advice_on_device, const void *const *instances,
                             const size_t *instance_lens, h2hip_rng_fill_fn rng, void *rng_user, std::vector<uint8_t> &proof_out, double *stage_ms) {
    const Shape &sh = pk->sh;
    const Domain &dom = pk->dom;
    const uint32_t n = sh.n, k = sh.k, ek = sh.extended_k, bf = sh.blinding_factors, u = sh.usable_rows;
    const size_t ne = (size_t)1 << ek;
    hipStream_t st = ctx->stream;
    Scope sc(&pk->pool);
    Laps laps(ctx, stage_ms);
    Transcript tr;
    // the RNG writes straight into pinned host memory, from where the DMA engine takes it (n scalars for the random polynomial)
    if (!pk->copy_stream) {
        H2_HIPCHK(hipStreamCreateWithFlags(&pk->copy_stream, hipStreamNonBlocking));
        H2_HIPCHK(hipEventCreateWithFlags(&pk->copy_ev, hipEventDisableTiming));
    }
    const size_t stage_cap = (size_t)n + ((size_t)1 << 14);   // the blinding tails of a proof and the n random-polynomial scalars without wrapping
    if (pk->host_stage_elems < stage_cap) {
        if (pk->host_stage) hipHostFree(pk->host_stage);
        pk->host_stage = nullptr;
        pk->host_stage_elems = 0;
        H2_HIPCHK(hipHostMalloc((void **)&pk->host_stage, sizeof(Fr) * stage_cap, 0));
        pk->host_stage_elems = stage_cap;
    }
    // bump allocation over the staging buffer: a draw's values stay where they are until the buffer wraps, so the (asynchronous) uploads
    // that read them need no synchronisation per draw — wide circuits draw hundreds of small tails
    // the vanishing argument's random polynomial: with libh2hip's counter-mode generator its n scalars are a function of (seed, position),
    // and the position of that draw follows from the shape — generated NOW, on the copy stream, next to the first rounds
    Fr *random_poly = nullptr;
    bool rng_ahead = false;
    uint64_t rng_ahead_pos = 0;
    if (rng == h2hip_chacha_rng_fill && rng_user) {
        const h2hip_chacha_rng *cr = (const h2hip_chacha_rng *)rng_user;
        const uint64_t L = sh.lookups.size(), S = sh.num_perm_sets, A = sh.num_advice_total;
        rng_ahead_pos = cr->pos + A * (uint64_t)(n - u) + A + L * (2 * ((uint64_t)bf + 1) + 2) + S * ((uint64_t)bf + 1) + L * ((uint64_t)bf + 1);
        H2_CHK(sc.take(n, &random_poly));
        H2_CHK(rng_chacha_fill_dev(ctx, random_poly, n, cr->seed, cr->rounds, rng_ahead_pos, pk->copy_stream));
        H2_HIPCHK(hipEventRecord(pk->copy_ev, pk->copy_stream));
        rng_ahead = true;
    }
    size_t stage_off = 0;
    auto draw = [&](size_t cnt) -> const Fr * {
        if (stage_off + cnt > pk->host_stage_elems) {
            hipStreamSynchronize(st);
            stage_off = 0;
        }
        Fr *p = pk->host_stage + stage_off;
        if (cnt) rng(rng_user, p, cnt);   // cnt <= n <= capacity by construction
        stage_off += cnt;
        return p;
    };
    auto put = [&](Fr *dst, const Fr *src, size_t cnt) -> int {   // pageable host -> device; the copy returns once `src` is consumed
        if (cnt) H2_HIPCHK(hipMemcpyAsync(dst, src, sizeof(Fr) * cnt, hipMemcpyHostToDevice, st));
        return H2HIP_OK;
    };
    // Blinding rows of MANY columns: drawn back to back they sit `stride` apart in the staging buffer, so one upload and one scatter launch per
    // 32 columns replace a small copy per column (a wide shape has hundreds).  reserve() makes room for the whole run first (it may
    // synchronise and rewind the buffer); runs longer than the buffer fall back to one copy per column.
    struct TailRun {
        std::vector<Fr *> dst;
        const Fr *first = nullptr;
        size_t stride = 0, len = 0;
        bool batched = false;
    };
    auto tails_reserve = [&](TailRun &t, size_t columns, size_t stride, size_t len) {
        t.stride = stride;
        t.len = len;
        t.batched = columns >= 4 && columns * stride <= pk->host_stage_elems;
        if (t.batched && stage_off + columns * stride > pk->host_stage_elems) {
            hipStreamSynchronize(st);
            stage_off = 0;
        }
    };
    auto tails_add = [&](TailRun &t, Fr *dst, const Fr *src) -> int {   // src: the draw holding this column's `len` rows
        if (!t.batched) return put(dst, src, t.len);
        if (t.dst.empty()) t.first = src;
        t.dst.push_back(dst);
        return H2HIP_OK;
    };
    auto tails_flush = [&](TailRun &t) -> int {
        if (!t.batched || t.dst.empty() || !t.len) return H2HIP_OK;
        const size_t span = (t.dst.size() - 1) * t.stride + t.len;
        Fr *tmp = nullptr;
        H2_CHK(sc.take(span, &tmp));
        H2_HIPCHK(hipMemcpyAsync(tmp, t.first, sizeof(Fr) * span, hipMemcpyHostToDevice, st));
        H2_CHK(fr_scatter_rows(ctx, t.dst.data(), t.dst.size(), tmp, t.stride, t.len));
        // (the few KiB stay with this proof's scope: a buffer handed back now could be picked up by the copy stream's upload while the scatter still reads it)
        t.dst.clear();
        return H2HIP_OK;
    };
    // ---- multi-GPU exchanges.  Host payloads travel as [8-byte status][payload] per rank; a non-zero status of any rank aborts the proof on
    // every rank at the same exchange (H2HIP_ERR_PEER), so nobody is left waiting in the next collective.
    const bool sharded_any = pk->comm != nullptr;   // (set only for world > 1, or for a forced single-rank run of the sharded path)
    auto exchange_host = [&](const void *payload, size_t bytes, std::vector<uint8_t> &all) -> int {
        H2_REQUIRE(pk->exch_next < pk->exch_sizes.size() && pk->exch_sizes[pk->exch_next] == bytes, "internal: exchange out of schedule");
        std::vector<uint8_t> send(8 + bytes, 0);
        if (bytes) memcpy(send.data() + 8, payload, bytes);
        all.assign((8 + bytes) * (size_t)pk->shard_world, 0);
        H2_CHK(h2hip_comm_allgather_host(pk->comm, ctx, send.data(), send.size(), all.data()));
        pk->exch_next++;
        for (uint32_t r = 0; r < pk->shard_world; ++r) {
            uint64_t status;
            memcpy(&status, all.data() + (size_t)r * (8 + bytes), 8);
            if (status) {
                set_error("create_proof: rank %u of the sharded proof reported an error; all ranks abort", r);
                return H2HIP_ERR_PEER;
            }
        }
        return H2HIP_OK;
    };
    // `bases_per_col`: empty = all columns over `bases`
    auto commit_points_multi = [&](const h2hip_bases *bases, const std::vector<const h2hip_bases *> &bases_per_col, const std::vector<const void *> &cols,
                                   size_t len, std::vector<G1Affine> &pts) -> int {
        std::vector<G1Jac> jac(cols.size());
        const bool sharded = sharded_any;
        std::vector<const h2hip_bases *> bpc(cols.size());
        for (size_t i = 0; i < cols.size(); ++i) {
            const h2hip_bases *b = bases_per_col.empty() ? bases : bases_per_col[i];
            bpc[i] = sharded ? (b == pk->g ? pk->g_shard : pk->g_lagrange_shard) : b;
        }
        // sharded: point-range sharding: this rank's partial MSM over its resident slice of the SRS, one all-gather of the
        // 96-byte partials per round (RCCL has no group-law reduction), the N-term sums on the host.  Every rank ends up with every
        // commitment, so the replicated transcripts stay in lock step and all ranks emit the same proof bytes.
        const size_t lo = sharded ? std::min(pk->shard_offset, len) : 0, hi = sharded ? std::min(pk->shard_offset + pk->shard_len, len) : len;
        std::vector<const void *> local(cols.size());
        for (size_t i = 0; i < cols.size(); ++i) local[i] = (const Fr *)cols[i] + lo;
        // (r05 ran a lone commitment — SHPLONK's W and W' — as 2 / 4 point-range parts on the batch API's lanes, co-running accumulations and shorter
        // sorts against one more join: the k = 19 proof 13.2-13.4 / 13.5-13.7 vs 13.0-13.1 ms, k = 21 51.6-52.5 vs 49.8-51.8: slower, removed —
        // profiles/r05_split_single_msm_ab.log)
        if (cols.size() == 1)
            H2_CHK(h2hip_msm_g1_dev(ctx, bpc[0], local[0], hi - lo, H2HIP_POINT_JACOBIAN, jac.data()));
        else if (!cols.empty())
            H2_CHK(h2hip_msm_g1_multi_dev(ctx, bpc.data(), local.data(), hi - lo, cols.size(), H2HIP_POINT_JACOBIAN, jac.data()));
        pts.resize(cols.size());
        if (sharded) {
            std::vector<uint8_t> all;
            H2_CHK(exchange_host(jac.data(), sizeof(G1Jac) * cols.size(), all));
            const size_t slot = 8 + sizeof(G1Jac) * cols.size();
            for (size_t i = 0; i < cols.size(); ++i) {
                XYZZ acc = XYZZ::identity();
                for (uint32_t r = 0; r < pk->shard_world; ++r) {
                    G1Jac p;
                    memcpy(&p, all.data() + (size_t)r * slot + 8 + sizeof(G1Jac) * i, sizeof(G1Jac));
                    if (p.z.is_zero()) continue;
                    XYZZ q;
                    q.x = p.x;
                    q.y = p.y;
                    q.zz = fe_sqr(p.z);
                    q.zzz = fe_mul(q.zz, p.z);
                    xyzz_add(acc, q);
                }
                pts[i] = xyzz_to_affine(acc);
            }
            return H2HIP_OK;
        }
        // commitments come back as Jacobian points (C::Curve, like best_multiexp) and are normalised here: one field inversion on the
        // host costs microseconds, on a single GPU lane ~0.1 ms of an otherwise idle chip
        for (size_t i = 0; i < jac.size(); ++i) pts[i] = jacobian_to_affine(jac[i]);
        return H2HIP_OK;
    };
    auto commit_points = [&](const h2hip_bases *bases, const std::vector<const void *> &cols, size_t len, std::vector<G1Affine> &pts) -> int {
        return commit_points_multi(bases, std::vector<const h2hip_bases *>(), cols, len, pts);
    };
    auto commit_batch = [&](const h2hip_bases *bases, const std::vector<const void *> &cols, size_t len) -> int {
        std::vector<G1Affine> pts;
        H2_CHK(commit_points(bases, cols, len, pts));
        for (const G1Affine &p : pts) H2_CHK(tr.write_point(p));
        return H2HIP_OK;
    };

    const bool qshard = sharded_any && pk->shard_quotient;
    // ---- tail overlap (r04).  A commitment round ends with the MSMs' bucket reduction: ~0.25 ms of dependent point additions on a few waves
    // while the rest of the chip idles.  Rounds 1 and 3 are followed by transforms that do not depend on the round's challenge (the
    // coefficient and extended forms of the columns just committed): they are queued on a side stream (a child context: own NTT scratch and
    // twiddle cache) behind the event the batch MSM records once its accumulations are done, read the Lagrange values the MSM also reads and
    // write NEW buffers, so they run next to the reduction and to the pointwise kernels that follow.  Single GPU only.
    const bool overlap = !sharded_any && ctx->plonk_tail_overlap != 0;
    // r05, sharded proofs: lagrange_to_coeff (with its all-gather of the coefficient forms when the columns are dealt to the ranks) and the cosets'
    // coeff_to_extended of the FIRST-ROUND columns run on the side context next to round 2's commitments instead of in front of them
    // with RCCL and more than one rank the side stream's all-gather and the main stream's host exchanges would use ONE communicator from
    // two streams at once — a pattern no run has exercised (no box shows more than one device: profiles/r06_rccl_cpx_refusal.log); until one does, the
    // side stream carries collectives only over the callback transport (which blocks the host anyway) or when plonk_shard_side = 2 forces it
    int comm_world = 1, comm_rccl = 0;
    if (sharded_any) h2hip_comm_info(pk->comm, &comm_world, nullptr, &comm_rccl);
    const bool side_sharded = sharded_any && (ctx->plonk_shard_side == 2 || (ctx->plonk_shard_side == 1 && !(comm_rccl && comm_world > 1)));
    if ((overlap || side_sharded) && !pk->side_ev) H2_HIPCHK(hipEventCreateWithFlags(&pk->side_ev, hipEventDisableTiming));
    if (overlap && !pk->side_ev1) H2_HIPCHK(hipEventCreateWithFlags(&pk->side_ev1, hipEventDisableTiming));
    // (r04 also computed the random polynomial's COMMITMENT ahead — its scalars depend on nothing once the generator is counter-mode — as one
    // MSM on a second side context behind round 1's accumulations: 14.72-14.99 vs 14.71-14.98 ms, profiles/archive/r04_tail_overlap_ab.log.  The
    // chip is busy with something ~97 % of the time; only work moved into LOW-occupancy stretches gains, and that MSM is not such work.  Removed.)
    // Which contexts run the side work.  The device serves its streams through a handful of hardware queues (4 by default: with more,
    // measured, everything gets slower), and the batch MSM's lanes already hold as many: a further stream shares a queue with one of them.
    // After a round's accumulations the lanes idle until the next round, so by default the side work runs ON the last lane's context
    // (a full context: own stream, NTT scratch, twiddle cache); a separate context only where that lane does not exist (one lane from 2^20
    // points on) or with plonk_side_on_lanes = 0.
    h2hip_ctx *side_c = nullptr;
    auto pick_side = [&](h2hip_ctx *lane, h2hip_ctx **own) -> h2hip_ctx * {
        h2hip_ctx *c = (ctx->plonk_side_on_lanes && lane) ? lane : *own;
        if (!c) {
            if (h2hip_init(ctx->device, nullptr, own) != H2HIP_OK) return nullptr;
            c = *own;
        }
        inherit_knobs(c, ctx);
        return c;
    };
    bool side_busy = false;   // work is queued on the side stream that the main stream has not waited for yet
    // queues `fn` on the side stream behind the batch MSM's accumulations (or, if the commitment took a path without lanes, behind what
    // the main stream holds after it) — `arm` before the commitment, `fire_if_pending` after it
    bool side_r1_recorded = false;   // side_ev1 marks the end of the first-round columns' transforms on the side stream
    auto side_arm = [&](std::function<int()> fn, bool first_round = false) {
        ctx->msm_tail_hook = [&, fn, first_round](hipEvent_t ev) -> int {
            if (!side_c) side_c = pick_side(ctx->lane[2], &pk->side);   // chosen once per proof: side_join() waits for ONE stream's event
            H2_REQUIRE(side_c, "create_proof: no side context");
            H2_HIPCHK(hipStreamWaitEvent(side_c->stream, ev, 0));
            H2_CHK(fn());
            H2_HIPCHK(hipEventRecord(pk->side_ev, side_c->stream));
            if (first_round) {
                H2_HIPCHK(hipEventRecord(pk->side_ev1, side_c->stream));
                side_r1_recorded = true;
            }
            side_busy = true;
            return H2HIP_OK;
        };
    };
    auto side_fire_if_pending = [&]() -> int {
        if (!ctx->msm_tail_hook) return H2HIP_OK;
        std::function<int(hipEvent_t)> hook;
        hook.swap(ctx->msm_tail_hook);
        if (!ctx->tail_ev) H2_HIPCHK(hipEventCreateWithFlags(&ctx->tail_ev, hipEventDisableTiming));
        H2_HIPCHK(hipEventRecord(ctx->tail_ev, st));
        return hook(ctx->tail_ev);
    };
    auto side_join = [&]() -> int {   // the main stream continues after everything queued on the side stream
        if (side_busy) H2_HIPCHK(hipStreamWaitEvent(st, pk->side_ev, 0));
        side_busy = false;
        return H2HIP_OK;
    };
    // out-of-place lagrange_to_coeff + coeff_to_extended of `src` columns on the side context: coef[i] / cos[i] are taken here
    // (two halves: plonk_early_intt queues the first in front of a round's commitments and only the second behind their accumulations)
    auto side_to_coeff = [&](const std::vector<Fr *> &src, std::vector<Fr *> &coef) -> int {
        coef.assign(src.size(), nullptr);
        for (size_t i = 0; i < src.size(); ++i) H2_CHK(sc.take(n, &coef[i]));
        const Fr out3[3] = {dom.ifft_divisor, dom.ifft_divisor, dom.ifft_divisor};
        return ntt_run_batch(side_c, coef.data(), (const Fr *const *)src.data(), src.size(), k, dom.omega_inv, n, nullptr, out3);
    };
    auto side_to_ext = [&](const std::vector<Fr *> &coef, std::vector<Fr *> &cos) -> int {
        cos.assign(coef.size(), nullptr);
        for (size_t i = 0; i < coef.size(); ++i) H2_CHK(sc.take(ne, &cos[i]));
        return h2hip_coeff_to_extended_batch_dev(side_c, (const void *const *)coef.data(), k, (void *const *)cos.data(), ek, coef.size(), &dom.ext_omega, &dom.zeta);
    };
    auto side_transforms = [&](const std::vector<Fr *> &src, std::vector<Fr *> &coef, std::vector<Fr *> &cos) -> int {
        H2_CHK(side_to_coeff(src, coef));
        return side_to_ext(coef, cos);
    };
    std::vector<Fr *> r1_src, r1_coef, r1_cos, r3_src, r3_coef, r3_cos;
    const size_t ncm = qshard ? pk->my_cosets.size() : 0;                 // cosets of the extended domain evaluated here
    const size_t ne_loc = qshard ? std::max<size_t>(ncm * (size_t)n, 1) : ne;   // extended-domain evaluations of a column held by this rank
    pk->exch_sizes.clear();
    pk->exch_next = 0;
    if (sharded_any) {   // the proof's host exchanges in order (payload bytes): hello, the five commitment rounds, the quotient's go-ahead
        const size_t P = sizeof(G1Jac);
        pk->exch_sizes = {9 * sizeof(uint64_t), P * (sh.num_advice_total + 2 * sh.lookups.size())};
        if (pk->shard_products) pk->exch_sizes.push_back(sizeof(Fr) * (sh.num_perm_sets + sh.lookups.size()));   // the row ranges' total products
        if (pk->shard_ntt) pk->exch_sizes.push_back(0);   // go-ahead of the column-dealt lagrange_to_coeff of the first-round columns (queued before round 2's commitments) ...
        pk->exch_sizes.push_back(P * (sh.num_perm_sets + sh.lookups.size() + 1));
        if (pk->shard_ntt) pk->exch_sizes.push_back(0);   // ... and of the product columns
        if (qshard) pk->exch_sizes.push_back(0);
        pk->exch_sizes.push_back(P * sh.quotient_pieces);
        pk->exch_sizes.push_back(sizeof(Fr) * (sh.num_evals() + 1));   // the evaluations: partial sums over this rank's coefficient range
        pk->exch_sizes.push_back(sizeof(Fr) * SHPLONK_MAX_OPENINGS);   // SHPLONK by coefficient range: the rotation sets' partial evaluations (carries)
        pk->exch_sizes.push_back(P);
        pk->exch_sizes.push_back(sizeof(Fr));                          // ... and the linearisation's
        pk->exch_sizes.push_back(P);
    }
    tr.common_scalar(pk->transcript_repr);   // vk.hash_into(transcript)
    // ---- instance columns: values are hashed, not committed (KZG: QUERY_INSTANCE = false)
    std::vector<Fr *> inst_values(sh.p.num_instance), adv(sh.num_advice_total);
    for (uint32_t i = 0; i < sh.p.num_instance; ++i) {
        const size_t len = instance_lens ? instance_lens[i] : 0;
        H2_REQUIRE(len <= u, "InstanceTooLarge");
        H2_REQUIRE(len == 0 || (instances && instances[i]), "NULL instance column");
        std::vector<Fr> vals(len ? len : 1);   // the caller's buffer need not be aligned like Fr
        if (len) memcpy(vals.data(), instances[i], sizeof(Fr) * len);
        for (size_t j = 0; j < len; ++j) tr.common_scalar(vals[j]);
        H2_CHK(sc.take(n, &inst_values[i]));
        H2_HIPCHK(hipMemsetAsync(inst_values[i], 0, sizeof(Fr) * n, st));
        H2_CHK(put(inst_values[i], vals.data(), len));
        H2_HIPCHK(hipStreamSynchronize(st));
    }
    // ---- advice columns: witness rows from the caller, blinding rows from the RNG
    const bool lazy_upload = !advice_on_device && overlap && ctx->plonk_lazy_upload != 0 && sh.num_advice_total >= 2 &&
                             (sh.lookups.empty() || ctx->plonk_permute_in_commit != 0);
    {
        TailRun tails;
        tails_reserve(tails, sh.num_advice_total, n - u, n - u);
        for (uint32_t c = 0; c < sh.num_advice_total; ++c) {
            H2_REQUIRE(advice[c], "NULL advice column");
            H2_CHK(sc.take(n, &adv[c]));
            // r06: host-resident columns behind the first are uploaded INSIDE round 1's commitment batch (msm_col_hook below), each right before its
            // MSM is queued: the pageable copy blocks the host, not the GPU, which is working on the columns before it by then
            if (!(lazy_upload && c >= 1))
                H2_HIPCHK(hipMemcpyAsync(adv[c], advice[c], sizeof(Fr) * u, advice_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
            const Fr *tail = draw(n - u);
            H2_CHK(tails_add(tails, adv[c] + u, tail));
        }
        H2_CHK(tails_flush(tails));
    }
    draw(sh.num_advice_total);   // Blind(Fr::random) per column: drawn, unused by KZG
    std::vector<std::pair<uint64_t, uint64_t>> rank_ranges;   // sharded: (offset, len) of every rank's coefficient / point range
    if (sharded_any) {
        // every rank must run the same proof: same shape, same RNG stream (a digest of the first column's blinding rows), and point ranges
        // that tile [0, n) — otherwise the ranks would emit a proof that fails to verify without any of them noticing
        struct Hello {
            uint64_t offset, len, rank, k, ncols, digest[4];
        } me;
        me.offset = pk->shard_offset;
        me.len = pk->shard_len;
        me.rank = pk->shard_rank;
        me.k = k;
        me.ncols = (uint64_t)sh.num_advice_total | (uint64_t)(pk->shard_quotient ? 1 : 0) << 32 | (uint64_t)(pk->shard_products ? 1 : 0) << 33 |
                   (uint64_t)(pk->shard_ntt ? 1 : 0) << 34;   // + the stages sharded: the exchange schedule depends on them
        {
            Blake2b h(32, "h2hip-shard-rng");
            h.update(pk->host_stage, sizeof(Fr) * (size_t)(n - u));   // the first draw of this proof sits at the start of the staging buffer
            uint8_t d[32];
            h.digest(d);
            memcpy(me.digest, d, 32);
        }
        std::vector<uint8_t> all;
        H2_CHK(exchange_host(&me, sizeof(me), all));
        std::vector<std::pair<uint64_t, uint64_t>> ranges;
        rank_ranges.clear();
        for (uint32_t r = 0; r < pk->shard_world; ++r) {
            Hello o;
            memcpy(&o, all.data() + (size_t)r * (8 + sizeof(Hello)) + 8, sizeof(Hello));
            if (o.k != me.k || o.ncols != me.ncols || memcmp(o.digest, me.digest, 32) != 0 || o.rank != r) {
                set_error("create_proof: rank %u of the sharded proof runs a different proof (shape, rank order or RNG stream differ)", r);
                return H2HIP_ERR_INVALID;
            }
            ranges.push_back({o.offset, o.len});
        }
        rank_ranges = ranges;   // by rank, for the carries of the range divisions
        std::sort(ranges.begin(), ranges.end());
        uint64_t pos = 0;
        for (auto &rg : ranges) {
            if (rg.first != pos) {
                set_error("create_proof: the ranks' point ranges do not tile [0, 2^k)");
                return H2HIP_ERR_INVALID;
            }
            pos += rg.second;
        }
        if (pos != n) {
            set_error("create_proof: the ranks' point ranges do not tile [0, 2^k)");
            return H2HIP_ERR_INVALID;
        }
    }
    laps.lap(ST_UPLOAD);
    // ---- lookups: permuted input / table columns.  Every lookup of halo2-base compresses a single (input, table) expression pair, so
    // theta does not enter the values: the permuted columns are computed before the advice commitments are out, and all of this round's
    // and the next round's commitments go through ONE batched MSM call (the transcript still sees them in upstream's order)
    struct LookupState {
        Fr *inp, *ap, *sp, *z;
        bool own_inp;
    };
    std::vector<LookupState> lks(sh.lookups.size());
    {
        std::vector<const void *> cols(adv.begin(), adv.end());
        for (size_t li = 0; li < sh.lookups.size(); ++li) {
            const Lookup &l = sh.lookups[li];
            LookupState &s = lks[li];
            s.own_inp = l.q_col >= 0;
            if (s.own_inp) H2_CHK(sc.take(n, &s.inp));
            else s.inp = adv[l.advice_col];
            H2_CHK(sc.take(n, &s.ap));
            H2_CHK(sc.take(n, &s.sp));
            cols.push_back(s.ap);
            cols.push_back(s.sp);
        }
        // the permutation itself (a dozen small, latency-bound launches and two host synchronisations: ~0.3 ms at k = 19) and the blinding rows
        auto permute_lookups = [&]() -> int {
            for (size_t li = 0; li < sh.lookups.size(); ++li) {
                const Lookup &l = sh.lookups[li];
                if (lks[li].own_inp) H2_CHK(h2hip_fr_mul_batch_dev(ctx, lks[li].inp, pk->fixed_values[l.q_col], adv[l.advice_col], n));
            }
            {   // all lookups read the one table column: their multiset checks come back in one host synchronisation
                std::vector<const void *> ins(lks.size());
                std::vector<void *> aps(lks.size()), sps(lks.size());
                for (size_t li = 0; li < lks.size(); ++li) {
                    ins[li] = lks[li].inp;
                    aps[li] = lks[li].ap;
                    sps[li] = lks[li].sp;
                }
                if (!lks.empty()) H2_CHK(h2hip_lookup_permute_presorted_batch_dev(ctx, ins.data(), pk->table_sorted, u, aps.data(), sps.data(), lks.size()));
            }
            TailRun ta, ts;   // per lookup the staging holds [a' rows][s' rows][2 blinds]
            tails_reserve(ta, lks.size(), 2 * ((size_t)bf + 1) + 2, bf + 1);
            ts = ta;
            for (size_t li = 0; li < sh.lookups.size(); ++li) {
                LookupState &s = lks[li];
                const Fr *t1 = draw(bf + 1);
                H2_CHK(tails_add(ta, s.ap + u, t1));
                const Fr *t2 = draw(bf + 1);
                H2_CHK(tails_add(ts, s.sp + u, t2));
                draw(2);   // the two commitment blinds
            }
            H2_CHK(tails_flush(ta));
            return tails_flush(ts);
        };
        // r04: the ADVICE columns' commitments do not wait for it — the batch MSM queues them on its lanes first, then calls the permutation (on
        // this stream), then queues the permuted columns behind it (msm_mid_hook): the first accumulation starts ~0.3 ms earlier
        const bool permute_inside = overlap && ctx->plonk_permute_in_commit && !lks.empty() && !adv.empty();
        if (permute_inside) {
            ctx->msm_mid_hook = permute_lookups;
            ctx->msm_mid_after = adv.size();
        } else {
            H2_CHK(permute_lookups());
        }
        laps.lap(ST_LOOKUP_PERMUTE);
        std::vector<G1Affine> pts;
        if (overlap) {   // this round's columns -> coefficient and extended form, next to the commitments' reduction and the grand products
            r1_src.assign(adv.begin(), adv.end());
            for (Fr *p : inst_values) r1_src.push_back(p);
            for (LookupState &s : lks) {
                r1_src.push_back(s.ap);
                r1_src.push_back(s.sp);
            }
            side_arm([&]() -> int { return side_transforms(r1_src, r1_coef, r1_cos); }, true);
        }
        if (lazy_upload) {
            const uint32_t A = sh.num_advice_total;
            ctx->msm_col_hook = [&, A](size_t j) -> int {
                if (j >= 1 && j < A) H2_HIPCHK(hipMemcpyAsync(adv[j], advice[j], sizeof(Fr) * u, hipMemcpyHostToDevice, st));
                return H2HIP_OK;
            };
        }
        H2_CHK(commit_points(pk->g_lagrange, cols, n, pts));
        H2_REQUIRE(!ctx->msm_mid_hook && !ctx->msm_col_hook, "internal: the commitment round did not run the lookup permutation / the advice uploads");
        H2_CHK(side_fire_if_pending());
        for (size_t i = 0; i < adv.size(); ++i) H2_CHK(tr.write_point(pts[i]));
        const Fr theta = tr.squeeze_challenge();
        (void)theta;
        for (size_t i = adv.size(); i < pts.size(); ++i) H2_CHK(tr.write_point(pts[i]));
        laps.lap(ST_COMMIT_ROUND1);
    }
    const Fr beta = tr.squeeze_challenge();
    const Fr gamma = tr.squeeze_challenge();
    // ---- grand products: permutation sets (chained through the last usable row), then the lookups.  The factors of all sets (all lookups)
    // are laid out back to back and go through one batched inversion and one prefix product: over the concatenated sets that prefix product
    // IS the chain z_i(0) = z_{i-1}(last usable row), so neither a host round trip nor a rescaling pass is needed.
    auto column_values = [&](const ColumnRef &c) -> const Fr * {
        return c.kind == 0 ? pk->fixed_values[c.index] : c.kind == 1 ? adv[c.index] : inst_values[c.index];
    };
    std::vector<Fr *> perm_z(sh.num_perm_sets);
    const bool pshard = sharded_any && pk->shard_products;
    if (pshard) {
        // Sharded by ROW RANGE (r04): a product column is a prefix product over the rows, so rank r forms the factors of its rows [a, b) — the
        // rows of its point range below the blinding rows — inverts and multiplies them up locally (1/N of the single-GPU work), the ranks
        // exchange their ranges' total products (32 bytes per product), every rank scales its rows by the product of everything before them (for
        // the permutation argument that includes the earlier sets: z_i(0) = z_{i-1}(last usable row)), and ONE device-to-device all-gather
        // completes the columns on every rank (their coefficient forms are needed everywhere).  This rank's commitment partials only read its
        // own rows.  The blinding rows come from the (replicated) RNG stream.
        const size_t S = sh.num_perm_sets, nseg = S + lks.size();
        const size_t a = std::min<size_t>(pk->shard_offset, u), b = std::min<size_t>(pk->shard_offset + pk->shard_len, u), m = b - a;
        size_t max_m = 0;
        for (auto &rg : rank_ranges) max_m = std::max<size_t>(max_m, std::min<size_t>(rg.first + rg.second, u) - std::min<size_t>(rg.first, u));
        // r06 (built): when lagrange_to_coeff is dealt by column, a product column's rows are only ever needed in full by the rank that
        // transforms it (commitments read a rank's own rows, everything later the coefficient forms) — so the rows go STRAIGHT to the column's owner
        // (column j -> rank j mod N, the order to_coeff deals in) by an all-to-all (grouped ncclSend / ncclRecv: h2hip_comm_alltoall_dev) instead
        // of to everybody by an all-gather: a rank receives 1 / N of the bytes, each block over its own xGMI link.
        const uint32_t NW = pk->shard_world;
        const bool route = pk->shard_ntt && ctx->plonk_route_rows != 0;
        const size_t row_stride = max_m + 1, slot_cols = (nseg + NW - 1) / NW, blk_elems = slot_cols * row_stride;
        const size_t slot_elems = route ? blk_elems * NW : nseg * row_stride;
        Fr *num = nullptr, *den = nullptr, *sendb = nullptr, *recvb = nullptr;
        std::vector<Fr *> zcol(nseg);
        if (nseg) {
            H2_CHK(sc.take(nseg * std::max<size_t>(m, 1), &num));
            H2_CHK(sc.take(nseg * std::max<size_t>(m, 1), &den));
            H2_CHK(sc.take(slot_elems, &sendb));
            H2_CHK(sc.take(route ? slot_elems : slot_elems * pk->shard_world, &recvb));
            if (route) H2_CHK(comm_reserve_alltoall_dev(pk->comm, sizeof(Fr) * blk_elems));
            else H2_CHK(comm_reserve_allgather_dev(pk->comm, sizeof(Fr) * slot_elems));
        }
        for (uint32_t si = 0; si < S; ++si) {
            H2_CHK(sc.take(n, &perm_z[si]));
            zcol[si] = perm_z[si];
        }
        for (size_t li = 0; li < lks.size(); ++li) {
            H2_CHK(sc.take(n, &lks[li].z));
            zcol[S + li] = lks[li].z;
        }
        if (S && m) {
            std::vector<const void *> pcols(sh.perm_columns.size()), psig(sh.perm_columns.size());
            for (size_t c = 0; c < sh.perm_columns.size(); ++c) {
                pcols[c] = column_values(sh.perm_columns[c]);
                psig[c] = pk->sigma_values[c];
            }
            H2_CHK(h2hip_permutation_product_terms_rows_dev(ctx, num, den, pcols.data(), psig.data(), (uint32_t)pcols.size(), sh.chunk_len, a, m, &beta, &gamma,
                                                            &dom.delta, &dom.omega));
        }
        for (size_t li = 0; li < lks.size() && m; ++li) {
            LookupState &s = lks[li];
            H2_CHK(h2hip_lookup_product_terms_dev(ctx, num + (S + li) * m, den + (S + li) * m, s.inp + a, pk->fixed_values[sh.lookups[li].table_col] + a, s.ap + a,
                                                  s.sp + a, m, &beta, &gamma));
        }
        std::vector<Fr> totals(nseg, Fr::one());
        if (nseg) {
            // local prefix products, each starting at 1: column s gets rows [a, b] (m + 1 values; the last one = this range's total)
            std::vector<void *> zloc(nseg);
            for (size_t j = 0; j < nseg; ++j) zloc[j] = zcol[j] + a;
            H2_CHK(h2hip_fr_grand_products_dev(ctx, zloc.data(), num, den, nseg, m, 0));
            for (size_t j = 0; j < nseg; ++j) H2_HIPCHK(hipMemcpyAsync(&totals[j], zcol[j] + b, sizeof(Fr), hipMemcpyDeviceToHost, st));
            H2_HIPCHK(hipStreamSynchronize(st));
        }
        std::vector<uint8_t> all;
        H2_CHK(exchange_host(totals.data(), sizeof(Fr) * nseg, all));   // (also the go-ahead of the all-gather below: nothing after it can fail on one rank alone)
        if (nseg) {
            const size_t slot = 8 + sizeof(Fr) * nseg;
            Fr chain = Fr::one();   // the permutation sets run on from one another
            for (size_t j = 0; j < nseg; ++j) {
                Fr before = Fr::one(), whole = Fr::one();
                for (uint32_t r = 0; r < pk->shard_world; ++r) {
                    Fr t;
                    memcpy(&t, all.data() + (size_t)r * slot + 8 + sizeof(Fr) * j, sizeof(Fr));
                    whole = fe_mul(whole, t);
                    if (rank_ranges[r].first < pk->shard_offset) before = fe_mul(before, t);
                }
                const Fr start = j < S ? fe_mul(chain, before) : before;
                if (j < S) chain = fe_mul(chain, whole);
                H2_CHK(h2hip_fr_scale_dev(ctx, zcol[j] + a, &start, m + 1));
                Fr *slot = route ? sendb + (j % NW) * blk_elems + (j / NW) * row_stride : sendb + j * row_stride;   // routed: block of the column's owner
                H2_HIPCHK(hipMemcpyAsync(slot, zcol[j] + a, sizeof(Fr) * (m + 1), hipMemcpyDeviceToDevice, st));
            }
            if (route) H2_CHK(h2hip_comm_alltoall_dev(pk->comm, ctx, sendb, sizeof(Fr) * blk_elems, recvb));
            else H2_CHK(h2hip_comm_allgather_dev(pk->comm, ctx, sendb, sizeof(Fr) * slot_elems, recvb));
            for (uint32_t r = 0; r < pk->shard_world; ++r) {
                if (r == pk->shard_rank) continue;
                const size_t ar = std::min<size_t>(rank_ranges[r].first, u), br = std::min<size_t>(rank_ranges[r].first + rank_ranges[r].second, u);
                if (rank_ranges[r].second == 0) continue;
                std::vector<Fr *> dst;
                if (route) {   // rank r's rows of the columns THIS rank transforms
                    for (size_t j = pk->shard_rank; j < nseg; j += NW) dst.push_back(zcol[j] + ar);
                    if (!dst.empty()) H2_CHK(fr_scatter_rows(ctx, dst.data(), dst.size(), recvb + (size_t)r * blk_elems, row_stride, br - ar + 1));
                    continue;
                }
                dst.resize(nseg);
                for (size_t j = 0; j < nseg; ++j) dst[j] = zcol[j] + ar;
                H2_CHK(fr_scatter_rows(ctx, dst.data(), nseg, recvb + (size_t)r * slot_elems, row_stride, br - ar + 1));
            }
        }
        {
            TailRun tz;
            tails_reserve(tz, nseg, (size_t)bf + 1, bf);
            for (size_t j = 0; j < nseg; ++j) {
                const Fr *tail = draw(bf);
                H2_CHK(tails_add(tz, zcol[j] + (n - bf), tail));
                draw(1);   // blind
            }
            H2_CHK(tails_flush(tz));
        }
        if (recvb) sc.release(recvb);
        if (sendb) sc.release(sendb);
        if (den) sc.release(den);
        if (num) sc.release(num);
        if (stage_ms) laps.lap(ST_PRODUCTS);
    } else {
        // r05: with ONE permutation set (nothing to chain) the permutation's and the lookups' factors go through ONE batched inversion and ONE
        // prefix product — segments [set | lookup 0 | lookup 1 ...] — instead of two passes of seven launches each (the k = 19 ECDSA shape: the second
        // pass was ~0.25 ms of small launches on the critical path).  Same values: every segment is its own product either way.
        const bool merged = ctx->plonk_merge_products != 0 && sh.num_perm_sets == 1 && !lks.empty();
        const size_t segs = merged ? 1 + lks.size() : std::max<size_t>(sh.num_perm_sets, lks.size());
        Fr *num = nullptr, *den = nullptr;
        if (segs) {
            H2_CHK(sc.take(segs * (size_t)u, &num));
            H2_CHK(sc.take(segs * (size_t)u, &den));
        }
        if (sh.num_perm_sets) {
            std::vector<const void *> pcols(sh.perm_columns.size()), psig(sh.perm_columns.size());
            for (size_t c = 0; c < sh.perm_columns.size(); ++c) {
                pcols[c] = column_values(sh.perm_columns[c]);
                psig[c] = pk->sigma_values[c];
            }
            H2_CHK(h2hip_permutation_product_terms_sets_dev(ctx, num, den, pcols.data(), psig.data(), (uint32_t)pcols.size(), sh.chunk_len, u, &beta, &gamma,
                                                            &dom.delta, &dom.omega));
        }
        for (uint32_t si = 0; si < sh.num_perm_sets; ++si) H2_CHK(sc.take(n, &perm_z[si]));
        if (sh.num_perm_sets && !merged) H2_CHK(h2hip_fr_grand_products_dev(ctx, (void *const *)perm_z.data(), num, den, sh.num_perm_sets, u, 1));
        {
            TailRun tz;
            tails_reserve(tz, sh.num_perm_sets, (size_t)bf + 1, bf);
            for (uint32_t si = 0; si < sh.num_perm_sets; ++si) {
                const Fr *tail = draw(bf);
                H2_CHK(tails_add(tz, perm_z[si] + (n - bf), tail));
                draw(1);   // blind
            }
            H2_CHK(tails_flush(tz));
        }
        std::vector<void *> lk_z(lks.size());
        const size_t lk0 = merged ? 1 : 0;   // the lookups' first segment
        for (size_t li = 0; li < lks.size(); ++li) {
            LookupState &s = lks[li];
            H2_CHK(h2hip_lookup_product_terms_dev(ctx, num + (lk0 + li) * (size_t)u, den + (lk0 + li) * (size_t)u, s.inp, pk->fixed_values[sh.lookups[li].table_col],
                                                  s.ap, s.sp, u, &beta, &gamma));
            H2_CHK(sc.take(n, &s.z));
            lk_z[li] = s.z;
        }
        if (merged) {
            std::vector<void *> all_z(1, perm_z[0]);
            all_z.insert(all_z.end(), lk_z.begin(), lk_z.end());
            H2_CHK(h2hip_fr_grand_products_dev(ctx, all_z.data(), num, den, all_z.size(), u, 0));
        } else if (!lks.empty()) {
            H2_CHK(h2hip_fr_grand_products_dev(ctx, lk_z.data(), num, den, lks.size(), u, 0));
        }
        {
            TailRun tz;
            tails_reserve(tz, lks.size(), (size_t)bf + 1, bf);
            for (size_t li = 0; li < lks.size(); ++li) {
                const Fr *tail = draw(bf);
                H2_CHK(tails_add(tz, lks[li].z + (n - bf), tail));
                draw(1);   // blind
            }
            H2_CHK(tails_flush(tz));
        }
        if (num) sc.release(num);
        if (den) sc.release(den);
        if (stage_ms) laps.lap(ST_PRODUCTS);
    }
    // ---- this round's commitments are the grand products AND the vanishing argument's random polynomial (nothing is squeezed between
    // them): one batched call over two base sets.  Before the host draws the 2^k random scalars, the GPU is given everything that no longer
    // needs the Lagrange values of the first-round columns: their coefficient and extended forms — RNG, upload (own stream) and NTTs overlap.
    if (!random_poly) H2_CHK(sc.take(n, &random_poly));
    // all columns of a round go through the transforms together (32 per launch): a wide shape's 2^14-row columns are far too small to fill the chip alone
    // `xc`: the context (stream, NTT scratch) the transforms are queued on — this one, or (r05, sharded) the side context, so that the coefficient
    // forms' all-gather and the cosets' transforms of the first-round columns run NEXT TO the round's commitments; `deferred`: buffers that go back
    // to the pool only after the side stream has been joined (the pool hands buffers out in the order of THIS stream)
    std::vector<Fr *> side_deferred;
    auto to_coeff = [&](const std::vector<Fr *> &cols, h2hip_ctx *xc) -> int {
        hipStream_t st = xc->stream;
        h2hip_ctx *ctx = xc;
        const bool defer = xc != pk->ctx;
        if (!(sharded_any && pk->shard_ntt))
            return h2hip_ifft_batch_dev(ctx, (void *const *)cols.data(), cols.size(), &dom.omega_inv, k, &dom.ifft_divisor);
        // Sharded by COLUMN (H2HIP_SHARD_NTT_COLUMNS; north_star: "independent NTT columns shard across the GPUs"): column j is transformed by rank
        // j mod N — in the all-gather's send buffer — and ONE device-to-device all-gather hands every rank every coefficient form (each needs
        // them all: its cosets read whole polynomials).  Pays when a transform takes longer than moving one column between GPUs (k = 21: 0.3 ms
        // against 64 MiB over the links a rank receives on: eight fully connected GPUs, not two).  Everything that can fail on this rank alone
        // comes before the go-ahead exchange.
        const uint32_t N = pk->shard_world, me_ = pk->shard_rank;
        const size_t slot_cols = (cols.size() + N - 1) / N, slot_elems = slot_cols * (size_t)n;
        if (cols.empty()) {
            std::vector<uint8_t> all;
            return exchange_host(nullptr, 0, all);
        }
        (void)defer;
        Fr *sendb = nullptr, *recvb = nullptr;
        H2_CHK(sc.take(slot_elems, &sendb));
        H2_CHK(sc.take(slot_elems * N, &recvb));
        std::vector<Fr *> mine;
        for (size_t j = me_; j < cols.size(); j += N) {
            Fr *dst = sendb + (j / N) * (size_t)n;
            H2_HIPCHK(hipMemcpyAsync(dst, cols[j], sizeof(Fr) * n, hipMemcpyDeviceToDevice, st));
            mine.push_back(dst);
        }
        if (!mine.empty()) H2_CHK(h2hip_ifft_batch_dev(ctx, (void *const *)mine.data(), mine.size(), &dom.omega_inv, k, &dom.ifft_divisor));
        H2_CHK(comm_reserve_allgather_dev(pk->comm, sizeof(Fr) * slot_elems));
        std::vector<uint8_t> all;
        H2_CHK(exchange_host(nullptr, 0, all));
        H2_CHK(h2hip_comm_allgather_dev(pk->comm, ctx, sendb, sizeof(Fr) * slot_elems, recvb));
        for (size_t j = 0; j < cols.size(); ++j)
            H2_HIPCHK(hipMemcpyAsync(cols[j], recvb + (j % N) * slot_elems + (j / N) * (size_t)n, sizeof(Fr) * n, hipMemcpyDeviceToDevice, st));
        if (defer) {
            side_deferred.push_back(recvb);
            side_deferred.push_back(sendb);
        } else {
            sc.release(recvb);
            sc.release(sendb);
        }
        return H2HIP_OK;
    };
    // the shift of coset c of the extended domain: s_c = zeta * omega_e^c (rows i = (j << (ek - k)) + c are the points s_c * omega^j)
    auto coset_shift = [&](uint32_t c) -> Fr { return fe_mul(dom.zeta, fe_pow_u64(dom.ext_omega, c)); };
    auto to_ext = [&](const std::vector<Fr *> &polys, const std::vector<Fr **> &outs, h2hip_ctx *xc) -> int {
        h2hip_ctx *ctx = xc;
        std::vector<void *> o(outs.size());
        for (size_t i = 0; i < outs.size(); ++i) {
            H2_CHK(sc.take(ne_loc, outs[i]));
            o[i] = *outs[i];
        }
        if (!qshard) return h2hip_coeff_to_extended_batch_dev(ctx, (const void *const *)polys.data(), k, o.data(), ek, polys.size(), &dom.ext_omega, &dom.zeta);
        // sharded: only this rank's cosets, each a 2^k-point transform of f(s_c X) — [coset][n] per column
        for (size_t m = 0; m < ncm; ++m) {
            std::vector<void *> om(outs.size());
            for (size_t i = 0; i < outs.size(); ++i) om[i] = (Fr *)o[i] + m * (size_t)n;
            const Fr s_c = coset_shift(pk->my_cosets[m]);
            H2_CHK(h2hip_fr_coset_scale_batch_dev(ctx, om.data(), (const void *const *)polys.data(), polys.size(), n, &s_c));
            H2_CHK(ntt_run_batch(ctx, (Fr *const *)om.data(), nullptr, om.size(), k, dom.omega, 0, nullptr, nullptr));
        }
        return H2HIP_OK;
    };
    // the proving key's extended-domain arrays as this rank holds them
    auto fixed_cos = [&](int col) -> const Fr * { return qshard ? pk->fixed_cosets_sh[col] : pk->fixed_cosets[col]; };
    auto sigma_cos = [&](size_t col) -> const Fr * { return qshard ? pk->sigma_cosets_sh[col] : pk->sigma_cosets[col]; };
    struct LookupCosets {
        Fr *z, *ap, *sp, *inp;
    };
    std::vector<Fr *> adv_cos(adv.size()), inst_cos(inst_values.size()), perm_cos(sh.num_perm_sets);
    std::vector<LookupCosets> lk_cos(lks.size());
    std::vector<Fr *> lagrange_done;   // Lagrange-form buffers the side stream may still be reading: released after side_join()
    if (overlap) {
        // the side stream holds (or is still producing) the coefficient and extended forms of the first-round columns; the grand products
        // above were the last readers of their Lagrange values on THIS stream: the names move on to the new buffers
        size_t i = 0;
        for (size_t c = 0; c < adv.size(); ++c, ++i) {
            lagrange_done.push_back(adv[c]);
            adv[c] = r1_coef[i];
            adv_cos[c] = r1_cos[i];
        }
        for (size_t c = 0; c < inst_values.size(); ++c, ++i) {
            lagrange_done.push_back(inst_values[c]);
            inst_values[c] = r1_coef[i];
            inst_cos[c] = r1_cos[i];
        }
        for (size_t li = 0; li < lks.size(); ++li) {
            LookupState &s = lks[li];
            if (s.own_inp) {
                sc.release(s.inp);   // (a product of this stream, never read by the side stream)
                s.inp = nullptr;
            }
            lagrange_done.push_back(s.ap);
            s.ap = r1_coef[i];
            lk_cos[li].ap = r1_cos[i++];
            lagrange_done.push_back(s.sp);
            s.sp = r1_coef[i];
            lk_cos[li].sp = r1_cos[i++];
        }
    } else
    {
        std::vector<Fr *> round1(adv.begin(), adv.end());
        std::vector<Fr **> round1_cos;
        for (size_t i = 0; i < adv.size(); ++i) round1_cos.push_back(&adv_cos[i]);
        for (size_t i = 0; i < inst_values.size(); ++i) {
            round1.push_back(inst_values[i]);
            round1_cos.push_back(&inst_cos[i]);
        }
        for (size_t li = 0; li < lks.size(); ++li) {
            LookupState &s = lks[li];
            if (s.own_inp) {
                sc.release(s.inp);
                s.inp = nullptr;
            }
            round1.push_back(s.ap);
            round1_cos.push_back(&lk_cos[li].ap);
            round1.push_back(s.sp);
            round1_cos.push_back(&lk_cos[li].sp);
        }
        if (side_sharded) {
            // everything the main stream holds so far (the grand products: the last readers of these columns' Lagrange values) precedes the side
            // work; the main stream goes on with this round's commitments and joins the side stream before the first reader of the cosets
            side_c = pick_side(nullptr, &pk->side);
            H2_REQUIRE(side_c, "create_proof: no side context");
            if (!ctx->tail_ev) H2_HIPCHK(hipEventCreateWithFlags(&ctx->tail_ev, hipEventDisableTiming));
            H2_HIPCHK(hipEventRecord(ctx->tail_ev, st));
            H2_HIPCHK(hipStreamWaitEvent(side_c->stream, ctx->tail_ev, 0));
            H2_CHK(to_coeff(round1, side_c));
            H2_CHK(to_ext(round1, round1_cos, side_c));
            H2_HIPCHK(hipEventRecord(pk->side_ev, side_c->stream));
            side_busy = true;
        } else {
            H2_CHK(to_coeff(round1, ctx));
            H2_CHK(to_ext(round1, round1_cos, ctx));
        }
    }
    auto lookup_input_cosets = [&]() -> int {
        for (size_t li = 0; li < lks.size(); ++li) {
            const Lookup &l = sh.lookups[li];
            LookupCosets &c = lk_cos[li];
            c.inp = nullptr;
            if (l.q_col >= 0) {   // the product of the cosets is the coset of the product polynomial q_lookup(X) * a(X)
                H2_CHK(sc.take(ne_loc, &c.inp));
                H2_CHK(h2hip_fr_mul_batch_dev(ctx, c.inp, fixed_cos(l.q_col), adv_cos[l.advice_col], qshard ? ncm * (size_t)n : ne));
            }
        }
        return H2HIP_OK;
    };
    for (LookupCosets &c : lk_cos) {
        c.z = nullptr;
        c.inp = nullptr;
    }
    if (!overlap && !side_sharded) H2_CHK(lookup_input_cosets());   // (overlapping: after the side stream's extended forms have been joined, below)
    G1Affine random_commitment;
    {
        if (rng == h2hip_chacha_rng_fill) {
            // libh2hip's own seeded ChaCha generator: counter mode, so the n elements are generated where they are needed — on the
            // device, straight into the polynomial — and the host only advances the stream position (same values as the host callback).
            // Normally they were generated at the start of the proof already (position known from the shape), next to everything else.
            h2hip_chacha_rng *cr = (h2hip_chacha_rng *)rng_user;
            if (rng_ahead && cr->pos == rng_ahead_pos) {
                H2_HIPCHK(hipStreamWaitEvent(st, pk->copy_ev, 0));
            } else {
                if (rng_ahead) H2_HIPCHK(hipStreamWaitEvent(st, pk->copy_ev, 0));   // (the buffer is overwritten in stream order)
                H2_CHK(rng_chacha_fill_dev(ctx, random_poly, n, cr->seed, cr->rounds, cr->pos, st));
            }
            cr->pos += n;
        } else {
            const Fr *vals = draw(n);
            H2_HIPCHK(hipMemcpyAsync(random_poly, vals, sizeof(Fr) * n, hipMemcpyHostToDevice, pk->copy_stream));
            H2_HIPCHK(hipEventRecord(pk->copy_ev, pk->copy_stream));
            H2_HIPCHK(hipStreamWaitEvent(st, pk->copy_ev, 0));
        }
        draw(1);   // random_blind
        std::vector<const void *> cols(perm_z.begin(), perm_z.end());
        std::vector<const h2hip_bases *> bases(cols.size(), pk->g_lagrange);
        for (LookupState &s : lks) {
            cols.push_back(s.z);
            bases.push_back(pk->g_lagrange);
        }
        cols.push_back(random_poly);
        bases.push_back(pk->g);
        std::vector<G1Affine> pts;
        if (overlap) {   // the grand products' coefficient and extended forms do not depend on y: next to this round's reduction
            r3_src.assign(perm_z.begin(), perm_z.end());
            for (LookupState &s : lks) r3_src.push_back(s.z);
            if (ctx->plonk_early_intt) {
                // the products are complete on this stream: their lagrange_to_coeff goes to the side context NOW — it runs next to the round's sorts
                // (and, on the last lane's context, in front of that lane's column: the third accumulation of the round, a millisecond away)
                if (!side_c) side_c = pick_side(ctx->lane[2], &pk->side);
                H2_REQUIRE(side_c, "create_proof: no side context");
                if (!ctx->tail_ev) H2_HIPCHK(hipEventCreateWithFlags(&ctx->tail_ev, hipEventDisableTiming));
                H2_HIPCHK(hipEventRecord(ctx->tail_ev, st));
                H2_HIPCHK(hipStreamWaitEvent(side_c->stream, ctx->tail_ev, 0));
                H2_CHK(side_to_coeff(r3_src, r3_coef));
                H2_HIPCHK(hipEventRecord(pk->side_ev, side_c->stream));
                side_busy = true;
                side_arm([&]() -> int { return side_to_ext(r3_coef, r3_cos); });
            } else
                side_arm([&]() -> int { return side_transforms(r3_src, r3_coef, r3_cos); });
        }
        H2_CHK(commit_points_multi(nullptr, bases, cols, n, pts));
        H2_CHK(side_fire_if_pending());
        for (size_t i = 0; i + 1 < pts.size(); ++i) H2_CHK(tr.write_point(pts[i]));
        random_commitment = pts.back();
        laps.lap(ST_COMMIT_PRODUCTS);
    }
    H2_CHK(tr.write_point(random_commitment));
    const Fr y = tr.squeeze_challenge();
    std::function<int()> late_join;   // (plonk_gate_before_join) the grand products' transforms are joined inside the quotient pass, behind the gate identities
    if (overlap) {
        auto join_products = [&]() -> int {   // both rounds' transforms are complete before anything behind this reads their results or reuses their inputs
            H2_CHK(side_join());
            size_t i = 0;
            for (uint32_t si = 0; si < sh.num_perm_sets; ++si, ++i) {
                sc.release(perm_z[si]);
                perm_z[si] = r3_coef[i];
                perm_cos[si] = r3_cos[i];
            }
            for (size_t li = 0; li < lks.size(); ++li, ++i) {
                sc.release(lks[li].z);
                lks[li].z = r3_coef[i];
                lk_cos[li].z = r3_cos[i];
            }
            return H2HIP_OK;
        };
        if (ctx->plonk_gate_before_join && side_r1_recorded && !stage_ms) {
            // the gate identities (and q_lookup * a's cosets) read first-round columns only: they start when THOSE transforms are done — the grand
            // products' coset transforms, queued behind this round's accumulations, are usually still running when y arrives
            H2_HIPCHK(hipStreamWaitEvent(st, pk->side_ev1, 0));
            late_join = join_products;
        } else
            H2_CHK(join_products());
        for (Fr *p : lagrange_done) sc.release(p);
        H2_CHK(lookup_input_cosets());
        if (stage_ms) laps.lap(ST_TO_COEFF);
    } else {
        if (side_sharded) {   // the first-round columns' coefficient forms and cosets are complete; their helpers go back to the pool
            H2_CHK(side_join());
            for (Fr *p : side_deferred) sc.release(p);
            side_deferred.clear();
            H2_CHK(lookup_input_cosets());
        }
        std::vector<Fr *> zs(perm_z.begin(), perm_z.end());
        std::vector<Fr **> zs_cos;
        for (uint32_t si = 0; si < sh.num_perm_sets; ++si) zs_cos.push_back(&perm_cos[si]);
        for (size_t li = 0; li < lks.size(); ++li) {
            zs.push_back(lks[li].z);
            zs_cos.push_back(&lk_cos[li].z);
        }
        H2_CHK(to_coeff(zs, ctx));
        if (stage_ms) laps.lap(ST_TO_COEFF);
        H2_CHK(to_ext(zs, zs_cos, ctx));
    }
    Fr *acc = nullptr;
    H2_CHK(sc.take(ne, &acc));
    Fr *acc_loc = acc;   // sharded: [coset][n] numerators of this rank's cosets (max_cosets slots: the all-gather's send buffer)
    if (qshard) H2_CHK(sc.take((size_t)pk->max_cosets * n, &acc_loc));
    H2_HIPCHK(hipMemsetAsync(acc_loc, 0, sizeof(Fr) * (qshard ? (size_t)pk->max_cosets * n : ne), st));
    if (stage_ms) laps.lap(ST_TO_EXT);
    // ---- h(X) numerator on the extended domain: the pointwise identities, folded by y in evaluate_h's order.  All gate columns, the whole
    // permutation argument and all lookups go through batched launches (every launch reads and writes the accumulator once).
    // One pass covers the whole extended domain (single GPU) or one coset of it (sharded: the same kernels on a 2^k-point domain with
    // shift s_c and generator omega; `off` = the coset's offset in the [coset][n] arrays), followed by the division by X^n - 1.
    auto quotient_pass = [&](size_t off, uint32_t ek_, const Fr &zeta_, const Fr &w_) -> int {
        {
            std::vector<const void *> gq(sh.p.num_advice), ga(sh.p.num_advice);
            for (uint32_t a = 0; a < sh.p.num_advice; ++a) {
                gq[a] = fixed_cos(sh.first_q_enable_col + (int)a) + off;
                ga[a] = adv_cos[a] + off;
            }
            H2_CHK(h2hip_quotient_flex_gate_batch_dev(ctx, acc_loc + off, gq.data(), ga.data(), gq.size(), ek_, k, &y));
        }
        if (late_join) {
            std::function<int()> f;
            f.swap(late_join);
            H2_CHK(f());
        }
        const Fr *l0 = (qshard ? pk->l0_sh : pk->l0) + off, *l_last = (qshard ? pk->l_last_sh : pk->l_last) + off,
                 *l_blind = (qshard ? pk->l_blind_sh : pk->l_blind) + off;
        if (sh.num_perm_sets) {
            std::vector<const void *> pcols(sh.perm_columns.size()), psig(sh.perm_columns.size()), pz(perm_cos.size());
            for (size_t i = 0; i < perm_cos.size(); ++i) pz[i] = perm_cos[i] + off;
            for (size_t c = 0; c < sh.perm_columns.size(); ++c) {
                const ColumnRef &r = sh.perm_columns[c];
                pcols[c] = (r.kind == 0 ? fixed_cos(r.index) : r.kind == 1 ? adv_cos[r.index] : inst_cos[r.index]) + off;
                psig[c] = sigma_cos(c) + off;
            }
            H2_CHK(h2hip_quotient_permutation_sets_dev(ctx, acc_loc + off, pz.data(), sh.num_perm_sets, pcols.data(), psig.data(), (uint32_t)pcols.size(),
                                                       sh.chunk_len, l0, l_last, l_blind, ek_, k, -(int32_t)(bf + 1), &beta, &gamma, &dom.delta, &zeta_, &w_, &y));
        }
        if (!lks.empty()) {
            std::vector<const void *> lz(lks.size()), la(lks.size()), ls(lks.size()), lap(lks.size()), lsp(lks.size());
            for (size_t li = 0; li < lks.size(); ++li) {
                const Lookup &l = sh.lookups[li];
                const LookupCosets &c = lk_cos[li];
                lz[li] = c.z + off;
                la[li] = (c.inp ? c.inp : adv_cos[l.advice_col]) + off;
                ls[li] = fixed_cos(l.table_col) + off;
                lap[li] = c.ap + off;
                lsp[li] = c.sp + off;
            }
            H2_CHK(h2hip_quotient_lookups_dev(ctx, acc_loc + off, lz.data(), la.data(), ls.data(), lap.data(), lsp.data(), lks.size(), l0, l_last, l_blind, ek_,
                                              k, &beta, &gamma, &y));
        }
        return h2hip_divide_by_vanishing_poly_dev(ctx, acc_loc + off, ek_, k, &w_, &zeta_);   // vanishing.construct: numerator / (X^n - 1)
    };
    if (!qshard) {
        H2_CHK(quotient_pass(0, ek, dom.zeta, dom.ext_omega));
    } else {
        for (size_t m = 0; m < ncm; ++m) {
            const Fr s_c = coset_shift(pk->my_cosets[m]);
            H2_CHK(quotient_pass(m * (size_t)n, k, s_c, dom.omega));
        }
        // every rank needs all of h's coefficients (its slices of the pieces are committed next, its evaluations opened later): one go-ahead
        // exchange (status only), then ONE all-gather of the cosets, device to device, and the interleave into extended-domain order
        // Everything that can fail on this rank alone — the gathered buffer (256 MiB at k = 21), the transport's staging area — is done BEFORE
        // the go-ahead: a rank that fails here reports it through that exchange; after it only the collective itself is left
        Fr *gathered = nullptr;
        const size_t slot_elems = (size_t)pk->max_cosets * n;
        H2_CHK(sc.take(slot_elems * pk->shard_world, &gathered));
        H2_CHK(comm_reserve_allgather_dev(pk->comm, sizeof(Fr) * slot_elems));
        std::vector<uint8_t> all;
        H2_CHK(exchange_host(nullptr, 0, all));
        // r04: extended_to_coeff by cosets as well.  The inverse transform of the whole extended domain factors into the size-n inverse coset
        // transform of every coset (iNTT, then s_c^-t: done HERE, where the coset's values are — 1 / 2^(ek-k) of the whole transform's work per
        // coset) and a 2^(ek-k)-point inverse DFT across the cosets for every t, which is pointwise: it runs on every rank after the all-gather
        // (h2hip_fr_coset_combine_dev) and leaves h(X)'s coefficients where extended_to_coeff would
        for (size_t m = 0; m < ncm; ++m) {
            const Fr s_c_inv = fe_inv(coset_shift(pk->my_cosets[m]));
            Fr *col = acc_loc + m * (size_t)n;
            H2_CHK(h2hip_ifft_dev(ctx, col, &dom.omega_inv, k, &dom.ifft_divisor));
            H2_CHK(h2hip_fr_coset_scale_batch_dev(ctx, (void *const *)&col, (const void *const *)&col, 1, n, &s_c_inv));
        }
        H2_CHK(h2hip_comm_allgather_dev(pk->comm, ctx, acc_loc, sizeof(Fr) * slot_elems, gathered));
        const uint32_t log_c = ek - k;
        uint32_t slots[16] = {0};
        for (uint32_t c = 0; c < (1u << log_c); ++c) slots[c] = (c % pk->shard_world) * pk->max_cosets + c / pk->shard_world;
        const Fr rho_inv = fe_inv(fe_pow_u64(dom.ext_omega, (uint64_t)n)), zeta_n_inv = fe_inv(fe_pow_u64(dom.zeta, (uint64_t)n));
        H2_CHK(h2hip_fr_coset_combine_dev(ctx, acc, gathered, slots, log_c, n, &rho_inv, &zeta_n_inv));
        sc.release(gathered);
        sc.release(acc_loc);
    }
    laps.lap(ST_QUOTIENT);
    for (Fr *p : perm_cos) sc.release(p);
    for (LookupCosets &c : lk_cos) {
        sc.release(c.z);
        sc.release(c.ap);
        sc.release(c.sp);
        if (c.inp) sc.release(c.inp);
    }
    for (Fr *p : adv_cos) sc.release(p);
    for (Fr *p : inst_cos) sc.release(p);
    // ---- back to coefficients, split into pieces, commit
    if (!qshard) H2_CHK(h2hip_extended_to_coeff_dev(ctx, acc, ek, &dom.ext_omega_inv, &dom.ext_ifft_divisor, &dom.zeta_inv));   // (sharded: done by cosets above)
    laps.lap(ST_H_COEFF);
    draw(sh.quotient_pieces);   // h_blinds
    {
        std::vector<const void *> cols;
        for (uint32_t i = 0; i < sh.quotient_pieces; ++i) cols.push_back(acc + (size_t)i * n);
        H2_CHK(commit_batch(pk->g, cols, n));
    }
    laps.lap(ST_COMMIT_H);
    const Fr x = tr.squeeze_challenge();
    const Fr xn = fe_pow_u64(x, n);
    // ---- evaluations: every (polynomial, point) pair of the round goes through ONE batched launch; the transcript then takes the
    // values in upstream's order
    auto rotate = [&](int rot) -> Fr {
        int64_t r = ((int64_t)rot % (int64_t)n + (int64_t)n) % (int64_t)n;
        return fe_mul(x, fe_pow_u64(dom.omega, (uint64_t)r));
    };
    std::vector<const Fr *> polys;     // SHPLONK's polynomial list; Query.poly indexes it
    std::map<const Fr *, int> poly_index;
    auto poly_id = [&](const Fr *p) -> int {
        auto it = poly_index.find(p);
        if (it != poly_index.end()) return it->second;
        polys.push_back(p);
        poly_index[p] = (int)polys.size() - 1;
        return (int)polys.size() - 1;
    };
    // vanishing.evaluate: h(X) = sum_i x^(n i) h_i(X) (its evaluation is not written to the proof; the multiopen needs it)
    Fr *h_poly = nullptr;
    H2_CHK(sc.take(n, &h_poly));
    {
        std::vector<const void *> pieces(sh.quotient_pieces);
        std::vector<Fr> xn_pows(sh.quotient_pieces);
        Fr xp = Fr::one();
        for (uint32_t i = 0; i < sh.quotient_pieces; ++i) {
            pieces[i] = acc + (size_t)i * n;
            xn_pows[i] = xp;
            xp = fe_mul(xp, xn);
        }
        H2_CHK(h2hip_fr_linear_combination_dev(ctx, h_poly, pieces.data(), xn_pows.data(), pieces.size(), n));
    }
    const Fr x_next = rotate(1), x_last = rotate(-(int)(bf + 1)), x_inv = rotate(-1);
    std::vector<Query> evq;            // in the order the values are written: advice, fixed, random, sigma, permutation sets, lookups; then h
    std::vector<const void *> ev_polys;
    auto want = [&](const Fr *poly, const Fr &point) -> size_t {
        evq.push_back(Query{-1, point, Fr::zero()});
        ev_polys.push_back(poly);
        return evq.size() - 1;
    };
    const size_t i_adv = evq.size();
    for (auto &aq : sh.advice_queries) want(adv[aq.first], rotate(aq.second));
    const size_t i_fixed = evq.size();
    for (auto &fq : sh.fixed_queries) want(pk->fixed_polys[fq.first], rotate(fq.second));
    const size_t i_random = want(random_poly, x);
    const size_t i_sigma = evq.size();
    for (Fr *sp_ : pk->sigma_polys) want(sp_, x);
    const size_t i_perm = evq.size();
    for (uint32_t si = 0; si < sh.num_perm_sets; ++si) {
        want(perm_z[si], x);
        want(perm_z[si], x_next);
        if (si + 1 != sh.num_perm_sets) want(perm_z[si], x_last);
    }
    const size_t i_lookup = evq.size();
    for (LookupState &s : lks) {   // product_eval, product_next_eval, permuted_input_eval, permuted_input_inv_eval, permuted_table_eval
        want(s.z, x);
        want(s.z, x_next);
        want(s.ap, x);
        want(s.ap, x_inv);
        want(s.sp, x);
    }
    const size_t n_written = evq.size();
    const size_t i_h = want(h_poly, x);
    {
        std::vector<Fr> points(evq.size()), vals(evq.size());
        for (size_t i = 0; i < evq.size(); ++i) points[i] = evq[i].point;
        if (!sharded_any) {
            std::vector<size_t> lens(evq.size(), n);
            H2_CHK(h2hip_fr_eval_polynomial_batch_dev(ctx, ev_polys.data(), lens.data(), points.data(), evq.size(), vals.data()));
        } else {
            // sharded (r04): every polynomial is evaluated over this rank's COEFFICIENT range [lo, hi) — the range of its SRS slice —
            // p(x) = sum_ranks x^lo * sum_j c[lo + j] x^j: the same batched launch on 1/N of the coefficients, one exchange of 32 bytes per
            // query, N-term sums on the host.  Every rank ends up with every evaluation (the transcripts stay in lock step).
            H2_REQUIRE(evq.size() == (size_t)sh.num_evals() + 1, "internal: evaluation count differs from the exchange schedule");
            const size_t lo = std::min<size_t>(pk->shard_offset, n), hi = std::min<size_t>(pk->shard_offset + pk->shard_len, n);
            std::vector<const void *> part(evq.size());
            std::vector<size_t> lens(evq.size(), hi - lo);
            for (size_t i = 0; i < evq.size(); ++i) part[i] = (const Fr *)ev_polys[i] + lo;
            if (hi > lo) H2_CHK(h2hip_fr_eval_polynomial_batch_dev(ctx, part.data(), lens.data(), points.data(), evq.size(), vals.data()));
            for (size_t i = 0; i < evq.size(); ++i) vals[i] = hi > lo ? fe_mul(vals[i], fe_pow_u64(points[i], lo)) : Fr::zero();
            std::vector<uint8_t> all;
            H2_CHK(exchange_host(vals.data(), sizeof(Fr) * vals.size(), all));
            const size_t slot = 8 + sizeof(Fr) * vals.size();
            for (size_t i = 0; i < vals.size(); ++i) {
                Fr acc = Fr::zero();
                for (uint32_t r = 0; r < pk->shard_world; ++r) {
                    Fr v;
                    memcpy(&v, all.data() + (size_t)r * slot + 8 + sizeof(Fr) * i, sizeof(Fr));
                    acc = fe_add(acc, v);
                }
                vals[i] = acc;
            }
        }
        for (size_t i = 0; i < evq.size(); ++i) evq[i].eval = vals[i];
    }
    for (size_t i = 0; i < n_written; ++i) tr.write_scalar(evq[i].eval);
    // the multiopen's query list in upstream's order: advice, permutation (sets at x / x_next, then sets.rev().skip(1) at x_last), lookups
    // (product, permuted input, permuted table at x; permuted input at x_inv; product at x_next), fixed, permutation polynomials, h, random
    std::vector<Query> queries;
    auto ask = [&](size_t i) {
        Query q = evq[i];
        q.poly = poly_id((const Fr *)ev_polys[i]);
        queries.push_back(q);
    };
    for (size_t i = i_adv; i < i_fixed; ++i) ask(i);
    {
        std::vector<size_t> tail;
        size_t i = i_perm;
        for (uint32_t si = 0; si < sh.num_perm_sets; ++si) {
            ask(i);
            ask(i + 1);
            i += 2;
            if (si + 1 != sh.num_perm_sets) tail.push_back(i++);
        }
        for (size_t t = tail.size(); t-- > 0;) ask(tail[t]);
    }
    for (size_t li = 0; li < lks.size(); ++li) {
        const size_t i = i_lookup + 5 * li;
        ask(i);       // product at x
        ask(i + 2);   // permuted input at x
        ask(i + 4);   // permuted table at x
        ask(i + 3);   // permuted input at x_inv
        ask(i + 1);   // product at x_next
    }
    for (size_t i = i_fixed; i < i_random; ++i) ask(i);
    for (size_t i = i_sigma; i < i_perm; ++i) ask(i);
    ask(i_h);
    ask(i_random);
    laps.lap(ST_EVALS);
    // ---- ProverSHPLONK::create_proof [UPSTREAM-RECALL poly/kzg/multiopen/shplonk/prover.rs]
    // Sharded (r04): every step between the evaluations and the two commitments is pointwise in the coefficient index — S_i = sum_j y^j P_ij,
    // the v-weighted sum of the sets' quotients, the linearisation — except the divisions by (X - root), whose suffix Horner needs ONE value per
    // (range boundary, root): what the coefficients above the range evaluate to.  With the commitments sharded by point range = coefficient
    // range a rank only ever needs ITS range [lo, hi) of these polynomials: it forms S_i on its range, the ranks exchange the ranges' partial
    // evaluations at the roots (32 bytes per (set, root) pair), and each divides its range with the carry assembled from the ranges above
    // (h2hip_fr_kate_division_range_dev).  The same for the final division of the linearisation by (X - u).
    {
        const Fr yq = tr.squeeze_challenge();
        std::vector<RotationSet> sets;
        std::vector<Fr> super_points;
        construct_intermediate_sets(queries, sets, super_points);
        H2_REQUIRE(!sets.empty(), "more than 64 distinct opening points");
        const Fr v = tr.squeeze_challenge();
        const size_t lo = sharded_any ? std::min<size_t>(pk->shard_offset, n) : 0, hi = sharded_any ? std::min<size_t>(pk->shard_offset + pk->shard_len, n) : n;
        const size_t L = hi - lo;   // coefficients held here (everything on one GPU)
        // carry of this rank's range for root p: sum over the ranges above of E_s(p) * p^(lo_s - hi), E_s = rank s's partial evaluation relative to lo_s
        auto carry_from = [&](const std::vector<uint8_t> &all, size_t slot_bytes, size_t idx, const Fr &p) -> Fr {
            Fr c = Fr::zero();
            for (uint32_t r = 0; r < pk->shard_world; ++r) {
                if (rank_ranges[r].second == 0 || rank_ranges[r].first < hi) continue;
                Fr e;
                memcpy(&e, all.data() + (size_t)r * slot_bytes + 8 + sizeof(Fr) * idx, sizeof(Fr));
                c = fe_add(c, fe_mul(e, fe_pow_u64(p, rank_ranges[r].first - hi)));
            }
            return c;
        };
        // S_i(X) = sum_j y^j P_ij(X) (kept for the linearisation), r_i(X) = sum_j y^j * interpolant of P_ij on the set's points
        std::vector<Fr *> S(sets.size());
        std::vector<std::vector<Fr>> low(sets.size()), pf_weights(sets.size());
        Fr *buf_a = nullptr, *buf_b = nullptr, *h_x = nullptr;
        H2_CHK(sc.take(n, &buf_a));
        H2_CHK(sc.take(n, &buf_b));
        H2_CHK(sc.take(n, &h_x));
        H2_HIPCHK(hipMemsetAsync(h_x, 0, sizeof(Fr) * n, st));
        size_t n_open = 0;
        for (size_t i = 0; i < sets.size(); ++i) {
            const RotationSet &rs = sets[i];
            H2_REQUIRE(rs.points.size() <= 8, "more than 8 rotations in one opening set");
            n_open += rs.points.size();
            H2_CHK(sc.take(n, &S[i]));
            low[i].assign(rs.points.size(), Fr::zero());
            Fr ypow = Fr::one();
            const std::vector<std::vector<Fr>> basis = lagrange_basis(rs.points, pf_weights[i]);
            std::vector<const void *> terms(rs.polys.size());
            std::vector<Fr> ypows(rs.polys.size());
            for (size_t j = 0; j < rs.polys.size(); ++j) {
                terms[j] = polys[rs.polys[j]] + lo;
                ypows[j] = ypow;
                std::vector<Fr> r = lagrange_interpolate(basis, rs.evals[j]);
                for (size_t t = 0; t < r.size(); ++t) low[i][t] = fe_add(low[i][t], fe_mul(ypow, r[t]));
                ypow = fe_mul(ypow, yq);
            }
            if (L) H2_CHK(h2hip_fr_linear_combination_dev(ctx, S[i] + lo, terms.data(), ypows.data(), terms.size(), L));   // every P_j read once
        }
        std::vector<std::vector<Fr>> carries(sets.size());
        if (sharded_any) {
            H2_REQUIRE(n_open <= SHPLONK_MAX_OPENINGS, "sharded create_proof: more (rotation set, point) pairs than the exchange holds");
            std::vector<const void *> part;
            std::vector<size_t> lens;
            std::vector<Fr> at, vals(SHPLONK_MAX_OPENINGS, Fr::zero());
            for (size_t i = 0; i < sets.size(); ++i)
                for (const Fr &pt : sets[i].points) {
                    part.push_back(S[i] + lo);
                    lens.push_back(L);
                    at.push_back(pt);
                }
            if (L) H2_CHK(h2hip_fr_eval_polynomial_batch_dev(ctx, part.data(), lens.data(), at.data(), part.size(), vals.data()));
            std::vector<uint8_t> all;
            H2_CHK(exchange_host(vals.data(), sizeof(Fr) * SHPLONK_MAX_OPENINGS, all));
            size_t idx = 0;
            for (size_t i = 0; i < sets.size(); ++i)
                for (const Fr &pt : sets[i].points) carries[i].push_back(carry_from(all, 8 + sizeof(Fr) * SHPLONK_MAX_OPENINGS, idx++, pt));
        }
        Fr vpow = Fr::one();
        // (S_i - r_i) / prod_j (X - point_j) = sum_j w_j (S_i(X) - S_i(point_j)) / (X - point_j) (partial fractions; r_i interpolates S_i
        // on the points by construction): all roots in ONE pass over S_i, no copy and no explicit subtraction of r_i; v^i folded into the
        // weights, the sets' quotients add up in h_x — single GPU: ALL sets in one call (one job table, one carry launch)
        if (!sharded_any) {
            std::vector<const void *> polys_s(sets.size());
            std::vector<Fr> pts, wts;
            std::vector<uint32_t> sizes(sets.size());
            for (size_t i = 0; i < sets.size(); ++i) {
                polys_s[i] = S[i];
                sizes[i] = (uint32_t)sets[i].points.size();
                for (size_t j = 0; j < sets[i].points.size(); ++j) {
                    pts.push_back(sets[i].points[j]);
                    wts.push_back(fe_mul(pf_weights[i][j], vpow));
                }
                vpow = fe_mul(vpow, v);
            }
            H2_CHK(h2hip_fr_kate_division_sets_dev(ctx, h_x, polys_s.data(), n, pts.data(), wts.data(), sizes.data(), sets.size(), 1));   // (h_x starts as zeros)
        } else {
            for (size_t i = 0; i < sets.size(); ++i) {
                const RotationSet &rs = sets[i];
                if (L) {
                    H2_CHK(h2hip_fr_kate_division_range_dev(ctx, buf_b + lo, S[i] + lo, L, rs.points.data(), pf_weights[i].data(), carries[i].data(),
                                                            (uint32_t)rs.points.size()));
                    H2_CHK(h2hip_fr_axpy_dev(ctx, h_x + lo, &vpow, buf_b + lo, L));
                }
                vpow = fe_mul(vpow, v);
            }
        }
        {
            std::vector<const void *> cols(1, h_x);
            H2_CHK(commit_batch(pk->g, cols, n));
        }
        const Fr uq = tr.squeeze_challenge();
        // linearisation L(X) = sum_i v^i Z_{T\S_i}(u) (S_i(X) - r_i(u)) - Z_T(u) h(X), which vanishes at u
        Fr *l_x = buf_a;
        Fr const0 = Fr::zero(), z_diff_0 = Fr::one();
        vpow = Fr::one();
        std::vector<const void *> lin_terms;
        std::vector<Fr> lin_coeffs;
        for (size_t i = 0; i < sets.size(); ++i) {
            Fr z_i = Fr::one();
            for (const Fr &p : super_points) {
                bool in_set = false;
                for (const Fr &sp : sets[i].points) in_set |= fr_cmp(sp, p) == 0;
                if (!in_set) z_i = fe_mul(z_i, fe_sub(uq, p));
            }
            if (i == 0) z_diff_0 = z_i;
            const Fr c_i = fe_mul(vpow, z_i);
            lin_terms.push_back(S[i] + lo);
            lin_coeffs.push_back(c_i);
            const0 = fe_add(const0, fe_mul(c_i, eval_small(low[i], uq)));
            vpow = fe_mul(vpow, v);
        }
        Fr zt = Fr::one();
        for (const Fr &p : super_points) zt = fe_mul(zt, fe_sub(uq, p));
        lin_terms.push_back(h_x + lo);
        lin_coeffs.push_back(fe_neg(zt));
        if (L) H2_CHK(h2hip_fr_linear_combination_dev(ctx, l_x + lo, lin_terms.data(), lin_coeffs.data(), lin_terms.size(), L));
        if (L && lo == 0) H2_CHK(h2hip_fr_sub_low_dev(ctx, l_x, &const0, 1));
        const Fr inv0 = fe_inv(z_diff_0);
        if (!sharded_any) {
            H2_CHK(h2hip_fr_kate_division_dev(ctx, buf_b, l_x, n, &uq));
            H2_CHK(h2hip_fr_scale_dev(ctx, buf_b, &inv0, (size_t)n - 1));
        } else {
            Fr e = Fr::zero();
            const void *part = l_x + lo;
            if (L) H2_CHK(h2hip_fr_eval_polynomial_batch_dev(ctx, &part, &L, &uq, 1, &e));
            std::vector<uint8_t> all;
            H2_CHK(exchange_host(&e, sizeof(Fr), all));
            const Fr carry = carry_from(all, 8 + sizeof(Fr), 0, uq), one = Fr::one();
            if (L) {
                H2_CHK(h2hip_fr_kate_division_range_dev(ctx, buf_b + lo, l_x + lo, L, &uq, &one, &carry, 1));
                H2_CHK(h2hip_fr_scale_dev(ctx, buf_b + lo, &inv0, L));
            }
        }
        std::vector<const void *> cols(1, buf_b);
        H2_CHK(commit_batch(pk->g, cols, (size_t)n - 1));
    }
    laps.lap(ST_MULTIOPEN);
    proof_out.swap(tr.proof);
    return H2HIP_OK;
}

}  // namespace plonk
}  // namespace h2

extern "C" {

int h2hip_plonk_shape_of(const h2hip_base_circuit_params *params, h2hip_plonk_shape *out) {
    H2_REQUIRE(params && out, "NULL argument");
    Shape sh;
    H2_CHK(sh.init(*params));
    out->num_advice_total = sh.num_advice_total;
    out->num_fixed_total = sh.num_fixed_total;
    out->table_col = sh.table_col;
    out->first_constant_col = sh.first_constant_col;
    out->q_lookup_col = sh.q_lookup_col;
    out->first_q_enable_col = sh.first_q_enable_col;
    out->num_lookups = (uint32_t)sh.lookups.size();
    out->num_perm_columns = (uint32_t)sh.perm_columns.size();
    out->num_perm_sets = sh.num_perm_sets;
    out->degree = sh.degree;
    out->extended_k = sh.extended_k;
    out->blinding_factors = sh.blinding_factors;
    out->usable_rows = sh.usable_rows;
    out->quotient_pieces = sh.quotient_pieces;
    out->num_commitments = sh.num_commitments();
    out->num_evals = sh.num_evals();
    return H2HIP_OK;
}

// Shape::init gives every gate column's q_enable selector a fixed column of its own.  Upstream's compress_selectors [UPSTREAM-RECALL] would
// instead COMBINE simple selectors that are never enabled on a common row (as far as the degree bound allows) — e.g. a gate column that the
// circuit left empty — which changes the number of fixed columns, the verifying key and the proof size.  Such circuits are rejected here
// rather than proven against a key halo2 would not derive: every pair of q_enable columns must share an enabled row.
static int check_selectors_stay_apart(const Shape &sh, const void *const *fixed_host) {
    const uint32_t na = sh.p.num_advice;
    if (na < 2) return H2HIP_OK;
    const size_t words = (sh.n + 63) / 64;
    std::vector<uint64_t> bits((size_t)na * words, 0);
    for (uint32_t a = 0; a < na; ++a) {
        const Fr *col = (const Fr *)fixed_host[sh.first_q_enable_col + (int)a];
        H2_REQUIRE(col, "NULL fixed column");
        for (size_t r = 0; r < sh.usable_rows; ++r)
            if (!col[r].is_zero()) bits[a * words + (r >> 6)] |= 1ull << (r & 63);
    }
    // a row enabled in EVERY column settles all pairs at once (the usual case: gates start at row 0 of every column)
    for (size_t w = 0; w < words; ++w) {
        uint64_t all = ~0ull;
        for (uint32_t a = 0; a < na && all; ++a) all &= bits[a * words + w];
        if (all) return H2HIP_OK;
    }
    for (uint32_t a = 0; a < na; ++a)
        for (uint32_t b = a + 1; b < na; ++b) {
            bool share = false;
            for (size_t w = 0; w < words && !share; ++w) share = (bits[a * words + w] & bits[b * words + w]) != 0;
            if (!share) {
                set_error("h2hip_plonk_keygen: the selectors of gate columns %u and %u are never enabled on a common row (an empty gate column?): "
                          "halo2's selector compression would merge them into one fixed column, which this backend's fixed-column layout does not model", a, b);
                return H2HIP_ERR_INVALID;
            }
        }
    return H2HIP_OK;
}

int h2hip_plonk_keygen(h2hip_ctx *ctx, const h2hip_base_circuit_params *params, const h2hip_bases *g, const h2hip_bases *g_lagrange,
                       const void *const *fixed_host, const uint32_t *copies, size_t ncopies, h2hip_plonk_pk **out) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && params && g && g_lagrange && fixed_host && out && (ncopies == 0 || copies), "NULL argument");
    h2hip_plonk_pk *pk = new h2hip_plonk_pk();
    int rc = pk->sh.init(*params);
    if (rc == H2HIP_OK && (g->n < pk->sh.n || g_lagrange->n < pk->sh.n)) {
        set_error("h2hip_plonk_keygen: the SRS holds fewer than 2^k bases");
        rc = H2HIP_ERR_INVALID;
    }
    if (rc == H2HIP_OK) rc = check_selectors_stay_apart(pk->sh, fixed_host);
    if (rc == H2HIP_OK) {
        pk->ctx = ctx;
        pk->g = g;
        pk->g_lagrange = g_lagrange;
        pk->dom.init(pk->sh.k, pk->sh.extended_k);
        rc = keygen_impl(ctx, pk, fixed_host, copies, ncopies);
    }
    if (rc != H2HIP_OK) {
        h2hip_plonk_pk_free(ctx, pk);
        return rc;
    }
    // Warm-up (r04): one throw-away proof of the all-zero witness, so that what a FIRST create_proof used to pay for — the key's buffer pool,
    // the twiddle tables of both domains, the MSM lanes and their scratch, the pinned staging buffer, the LDS attributes — exists when keygen
    // returns (the first proof after keygen took 22-27 ms against 15 ms warm).  Best effort: a failure here only leaves the first proof cold.
    if (ctx->plonk_warm_keygen) {
        std::vector<Fr *> zero_cols(pk->sh.num_advice_total, nullptr);
        bool ok = true;
        for (auto &c : zero_cols) {
            if (hipMalloc((void **)&c, sizeof(Fr) * pk->sh.n) != hipSuccess) {
                c = nullptr;
                ok = false;
                break;
            }
            hipMemsetAsync(c, 0, sizeof(Fr) * pk->sh.n, ctx->stream);
        }
        if (ok) {
            std::vector<const void *> adv(zero_cols.begin(), zero_cols.end());
            std::vector<Fr> inst_zero(1, Fr::zero());
            std::vector<const void *> inst(pk->sh.p.num_instance, inst_zero.data());
            std::vector<size_t> inst_len(pk->sh.p.num_instance, 0);
            h2hip_chacha_rng wr;
            uint8_t seed[32] = {0};
            h2hip_chacha_rng_init(&wr, seed, 8);
            std::vector<uint8_t> throwaway;
            const Fr saved = pk->transcript_repr;
            pk->transcript_repr = Fr::zero();
            const int wrc = create_proof_impl(ctx, pk, adv.data(), true, inst.data(), inst_len.data(), h2hip_chacha_rng_fill, &wr, throwaway, nullptr);
            pk->transcript_repr = saved;
            ctx->msm_tail_hook = nullptr;   // both hooks capture create_proof_impl's frame: never leave one on the context
            ctx->msm_mid_hook = nullptr;
            ctx->msm_col_hook = nullptr;
            if (pk->side) hipStreamSynchronize(pk->side->stream);
            for (h2hip_ctx *l : ctx->lane)
                if (l) hipStreamSynchronize(l->stream);
            if (pk->copy_stream) hipStreamSynchronize(pk->copy_stream);
            hipStreamSynchronize(ctx->stream);
            if (wrc != H2HIP_OK) set_error("");   // not the caller's error
        }
        hipStreamSynchronize(ctx->stream);
        for (Fr *c : zero_cols)
            if (c) hipFree(c);
    }
    *out = pk;
    return H2HIP_OK;
}

void h2hip_plonk_pk_free(h2hip_ctx *ctx, h2hip_plonk_pk *pk) {
    H2_DEVICE_GUARD(ctx);
    if (!pk) return;
    if (ctx) hipStreamSynchronize(ctx->stream);
    for (void *p : pk->owned) hipFree(p);
    for (void *p : pk->shard_owned) hipFree(p);
    if (pk->host_stage) hipHostFree(pk->host_stage);
    if (pk->copy_ev) hipEventDestroy(pk->copy_ev);
    if (pk->copy_stream) hipStreamDestroy(pk->copy_stream);
    if (pk->side_ev) hipEventDestroy(pk->side_ev);
    if (pk->side_ev1) hipEventDestroy(pk->side_ev1);
    if (pk->side) h2hip_destroy(pk->side);
    pk->pool.destroy();
    delete pk;
}

int h2hip_plonk_pk_commitments(const h2hip_plonk_pk *pk, void *fixed_out, void *permutation_out) {
    H2_REQUIRE(pk, "NULL argument");
    if (fixed_out && !pk->fixed_commitments.empty()) memcpy(fixed_out, pk->fixed_commitments.data(), sizeof(G1Affine) * pk->fixed_commitments.size());
    if (permutation_out && !pk->permutation_commitments.empty())
        memcpy(permutation_out, pk->permutation_commitments.data(), sizeof(G1Affine) * pk->permutation_commitments.size());
    return H2HIP_OK;
}

int h2hip_plonk_pk_set_transcript_repr(h2hip_plonk_pk *pk, const void *fr) {
    H2_REQUIRE(pk && fr, "NULL argument");
    memcpy(&pk->transcript_repr, fr, sizeof(Fr));
    pk->have_repr = true;
    return H2HIP_OK;
}

static void shard_release(h2hip_plonk_pk *pk) {
    if (pk->ctx) hipStreamSynchronize(pk->ctx->stream);
    for (void *q : pk->shard_owned) hipFree(q);
    pk->shard_owned.clear();
    pk->fixed_cosets_sh.clear();
    pk->sigma_cosets_sh.clear();
    pk->l0_sh = pk->l_last_sh = pk->l_blind_sh = nullptr;
    pk->my_cosets.clear();
    pk->shard_world = 1;
    pk->shard_rank = 0;
    pk->g_shard = pk->g_lagrange_shard = nullptr;
    pk->comm = nullptr;
    pk->shard_quotient = false;
    pk->shard_products = false;
    pk->shard_ntt = false;
}

// the host exchanges of the key's last sharded proof, in order: payload bytes per rank (each travels with an 8-byte status word); what a
// multi-GPU run reports about itself (bench.py --gpus N).  count: the number of exchanges; sizes: the first min(count, cap) of them
int h2hip_plonk_pk_last_exchanges(const h2hip_plonk_pk *pk, size_t *sizes, size_t cap, size_t *count) {
    H2_REQUIRE(pk && count && (cap == 0 || sizes), "NULL argument");
    *count = pk->exch_sizes.size();
    for (size_t i = 0; i < pk->exch_sizes.size() && i < cap; ++i) sizes[i] = pk->exch_sizes[i];
    return H2HIP_OK;
}

int h2hip_plonk_pk_set_sharding(h2hip_plonk_pk *pk, h2hip_comm *comm, const h2hip_bases *g_shard, const h2hip_bases *g_lagrange_shard, size_t offset,
                                size_t len, uint32_t flags) {
    H2_REQUIRE(pk, "NULL argument");
    H2_DEVICE_GUARD(pk->ctx);
    shard_release(pk);
    int world = 1, rank = 0;
    if (comm) H2_CHK(h2hip_comm_info(comm, &world, &rank, nullptr));
    if (!comm || (world <= 1 && !(flags & H2HIP_SHARD_FORCE))) return H2HIP_OK;
    H2_REQUIRE(g_shard && g_lagrange_shard, "NULL argument");
    H2_REQUIRE(offset <= pk->sh.n && len <= pk->sh.n - offset && g_shard->n >= len && g_lagrange_shard->n >= len, "shard range outside the SRS / shard base sets too small");
    pk->g_shard = g_shard;
    pk->g_lagrange_shard = g_lagrange_shard;
    pk->shard_offset = offset;
    pk->shard_len = len;
    pk->shard_world = (uint32_t)world;
    pk->shard_rank = (uint32_t)rank;
    pk->comm = comm;
    pk->shard_quotient = (flags & H2HIP_SHARD_QUOTIENT) != 0;
    pk->shard_products = (flags & H2HIP_SHARD_PRODUCTS) != 0;
    pk->shard_ntt = (flags & H2HIP_SHARD_NTT_COLUMNS) != 0;
    if (!pk->shard_quotient) return H2HIP_OK;
    // this rank's cosets of the extended domain: c = rank, rank + world, ... < 2^(ek - k), and the [coset][n] slices of the key's arrays
    const uint32_t log_c = pk->sh.extended_k - pk->sh.k, ncos = 1u << log_c, n = pk->sh.n;
    H2_REQUIRE(log_c <= 4, "more than 16 cosets");
    for (uint32_t c = (uint32_t)rank; c < ncos; c += (uint32_t)world) pk->my_cosets.push_back(c);
    pk->max_cosets = (ncos + (uint32_t)world - 1) / (uint32_t)world;
    const size_t cnt = pk->my_cosets.size(), elems = std::max<size_t>(cnt * (size_t)n, 1);
    auto slice = [&](const Fr *full, Fr **out) -> int {
        void *d = nullptr;
        if (hipMalloc(&d, sizeof(Fr) * elems) != hipSuccess) {
            set_error("hipMalloc for the sharded proving key failed");
            return H2HIP_ERR_NOMEM;
        }
        pk->shard_owned.push_back(d);
        *out = (Fr *)d;
        return h2hip_fr_coset_gather_dev(pk->ctx, d, full, pk->my_cosets.data(), (uint32_t)cnt, log_c, n);
    };
    int rc = H2HIP_OK;
    pk->fixed_cosets_sh.assign(pk->fixed_cosets.size(), nullptr);
    pk->sigma_cosets_sh.assign(pk->sigma_cosets.size(), nullptr);
    for (size_t i = 0; rc == H2HIP_OK && i < pk->fixed_cosets.size(); ++i) rc = slice(pk->fixed_cosets[i], &pk->fixed_cosets_sh[i]);
    for (size_t i = 0; rc == H2HIP_OK && i < pk->sigma_cosets.size(); ++i) rc = slice(pk->sigma_cosets[i], &pk->sigma_cosets_sh[i]);
    if (rc == H2HIP_OK) rc = slice(pk->l0, &pk->l0_sh);
    if (rc == H2HIP_OK) rc = slice(pk->l_last, &pk->l_last_sh);
    if (rc == H2HIP_OK) rc = slice(pk->l_blind, &pk->l_blind_sh);
    if (rc == H2HIP_OK && hipStreamSynchronize(pk->ctx->stream) != hipSuccess) rc = H2HIP_ERR_HIP;
    if (rc != H2HIP_OK) shard_release(pk);
    return rc;
}

void h2hip_array_rng_fill(void *user, void *out_fr, size_t n) {
    h2hip_array_rng *r = (h2hip_array_rng *)user;
    if (!r || !out_fr) return;
    const size_t have = r->pos < r->count ? r->count - r->pos : 0, take = n < have ? n : have;
    if (take) memcpy(out_fr, (const char *)r->values + sizeof(Fr) * r->pos, sizeof(Fr) * take);
    if (take < n) {
        memset((char *)out_fr + sizeof(Fr) * take, 0, sizeof(Fr) * (n - take));
        r->exhausted = 1;
    }
    r->pos += take;
}

const char *h2hip_plonk_stage_name(int stage) { return stage >= 0 && stage < H2HIP_PLONK_STAGES ? STAGE_NAMES[stage] : ""; }

int h2hip_plonk_create_proof(h2hip_ctx *ctx, h2hip_plonk_pk *pk, const void *const *advice, int advice_on_device, const void *const *instances_host,
                             const size_t *instance_lens, h2hip_rng_fill_fn rng, void *rng_user, uint8_t *proof_out, size_t proof_cap,
                             size_t *proof_len, double *stage_ms) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && pk && advice && rng && proof_out && proof_len, "NULL argument");
    H2_REQUIRE(pk->ctx == ctx, "the proving key belongs to another context");
    H2_REQUIRE(pk->have_repr, "h2hip_plonk_pk_set_transcript_repr has not been called");
    H2_REQUIRE(pk->sh.p.num_instance == 0 || instance_lens, "instance_lens is required");
    const size_t need = 32 * (size_t)(pk->sh.num_commitments() + pk->sh.num_evals());
    H2_REQUIRE(proof_cap >= need, "proof buffer too small");
    if (rng == h2hip_chacha_rng_fill) {   // the library's own generator: the device fill reads its state, so check it once, here
        const h2hip_chacha_rng *cr = (const h2hip_chacha_rng *)rng_user;
        H2_REQUIRE(cr, "h2hip_chacha_rng_fill needs its h2hip_chacha_rng as rng_user");
        H2_REQUIRE(cr->rounds == 8 || cr->rounds == 12 || cr->rounds == 20, "h2hip_chacha_rng: rounds must be 8, 12 or 20");
    }
    std::vector<uint8_t> proof;
    int rc = create_proof_impl(ctx, pk, advice, advice_on_device != 0, instances_host, instance_lens, rng, rng_user, proof, stage_ms);
    ctx->msm_tail_hook = nullptr;   // (never leave a hook of this proof behind: it captures the proof's frame)
    ctx->msm_mid_hook = nullptr;
    ctx->msm_col_hook = nullptr;
    if (pk->side && ctx->profiling && ctx->prof_filter.empty()) prof_fold_child(ctx, pk->side);   // (the lanes' timers are folded when the table is read)
    if (rc != H2HIP_OK) {
        if (pk->copy_stream) hipStreamSynchronize(pk->copy_stream);
        if (pk->side) hipStreamSynchronize(pk->side->stream);
        for (h2hip_ctx *l : ctx->lane)
            if (l) hipStreamSynchronize(l->stream);
        hipStreamSynchronize(ctx->stream);   // nothing of the failed proof may still run on buffers that go back to the pool
        if (rc != H2HIP_ERR_PEER && pk->comm && pk->exch_next < pk->exch_sizes.size()) {
            // a sharded proof failed HERE (a lookup value missing from the table, an identity commitment, an allocation ...): the other
            // ranks are on their way into the next exchange — take part in it with an error status so that all of them return too
            const std::string msg = h2hip_last_error();
            const size_t bytes = 8 + pk->exch_sizes[pk->exch_next];
            std::vector<uint8_t> send(bytes, 0), all(bytes * (size_t)pk->shard_world);
            const uint64_t status = 1;
            memcpy(send.data(), &status, 8);
            h2hip_comm_allgather_host(pk->comm, ctx, send.data(), bytes, all.data());
            pk->exch_next = pk->exch_sizes.size();
            set_error("%s", msg.c_str());
        }
        return rc;
    }
    if (proof.size() != need) {
        set_error("create_proof: internal error: proof has %zu bytes, expected %zu", proof.size(), need);
        return H2HIP_ERR_INVALID;
    }
    memcpy(proof_out, proof.data(), need);
    *proof_len = need;
    return H2HIP_OK;
}

}  // extern "C"
