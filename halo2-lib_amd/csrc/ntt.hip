// Radix-2^m Stockham NTT over BN254 F_r for gfx950 — the device replacement for
// halo2_proofs::arithmetic::best_fft and EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff}
// [UPSTREAM halo2-axiom 0.5.3; the reference reaches them only through create_proof,
// /root/reference/halo2-base/src/utils/testing.rs:40-47; semantics restated in SURVEY.md A.2].
//
// Natural order in, natural order out, no bit-reversal pass: log_n is split into P passes of m_i bits.
// Pass i (stride s = 2^(m_1+..+m_{i-1})) lets one 256-thread workgroup own a tile of C = 2^cb adjacent
// "columns" j: it reads rows x[j + t*N/R] (each row a contiguous C*32 B segment), runs the R-point
// decimation-in-time butterflies in LDS on unsaturated 9x29-bit limbs (fq29.cuh), multiplies by the inter-pass
// twiddle omega^((j-q)u) and writes y[(j-q)R + q + u*s], q = j mod s (again contiguous in q).  All passes but the last are
// out-of-place (ping-pong with a context-owned scratch buffer); the last pass touches the same index set
// it reads, so it runs in place.  Input scaling by zeta^(i mod 3) with implicit zero padding
// (coeff_to_extended) is fused into the first pass, output scaling (ifft divisor, zeta^-(i mod 3)) into
// the last.  Twiddles come from a two-level table omega^e = T2[e >> lo] * T1[e & mask] built once per
// (log_n, omega) and cached in the context.
#include "internal.h"
#include "fr29.cuh"

namespace h2 {

// ordering point between LDS writes and reads of the lanes of ONE wave: the wave issues its LDS operations in order, so only the
// compiler has to be kept from moving them across (the emulated build, where lanes are fibers, needs a real barrier)
#ifdef H2_HIPEMU
#define H2_WAVE_SYNC() hipemu_wave_sync()
#else
#define H2_WAVE_SYNC()                                         \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif

struct NttScale {
    Fr in3[3];
    Fr out3[3];
};


__global__ void ntt_twiddle_kernel(Fr29L *t1, Fr29L *t2, Fr omega, uint32_t lo_bits, uint32_t hi_count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo_count = 1u << lo_bits;
    if (i < lo_count) t1[i].v = fr29_from_sat(fe_pow_u64(omega, (uint64_t)i));
    if (i < hi_count) t2[i].v = fr29_from_sat(fe_pow_u64(omega, (uint64_t)i << lo_bits));
}

__device__ __forceinline__ Fr29 tw_lookup(const Fr29L *__restrict__ t1, const Fr29L *__restrict__ t2, uint32_t lo_bits, uint64_t e) { return pow_lookup(t1, t2, lo_bits, e); }

__device__ __forceinline__ uint32_t bitrev_m(uint32_t x, uint32_t m) { return m ? (__brev(x) >> (32 - m)) : 0; }

// One pass; persistent 256-thread workgroups loop over tiles (grid = min(#tiles, 3 per CU)).  Data moves through HBM as
// saturated canonical Montgomery limbs (the caller's format); inside the workgroup it lives in LDS as unsaturated
// 9 x 29-bit limbs (36 B: a 9-word stride is bank-conflict free), where the integer value*2^256 is kept, only weakly
// reduced.  Stage twiddles and inter-pass twiddles are stored in the R' = 2^261 Montgomery form, so mont29(x, w)
// keeps the integer's 2^256 scaling and no conversion multiply is ever needed.  Butterflies are decimation-in-time
// (t = w*b; a + t, a - t + 2r), two stages per LDS round trip with four elements per lane in registers: a value's
// bound grows by at most 2r per stage (<= 21 r after 10 stages) instead of doubling.  While a tile is being
// transformed, the lane's share of the NEXT tile is already in flight from HBM into registers.
constexpr uint32_t NTT_EPT = 4;   // elements per lane per tile (tile <= 1024 elements)
// A launch transforms up to NTT_BATCH equal-size columns (blockIdx.y = column): a circuit with hundreds of 2^14-row columns gets its
// transforms in a few full-chip launches instead of one 16-workgroup launch per column and pass.
constexpr uint32_t NTT_BATCH = 32;
struct NttCols {
    const Fr *x[NTT_BATCH];
    Fr *y[NTT_BATCH];
};
__global__ __launch_bounds__(256) void ntt_pass_kernel(NttCols cols, uint32_t log_n, uint32_t m,
                                                       uint32_t log_s, uint32_t cb, const Fr29L *__restrict__ t1,
                                                       const Fr29L *__restrict__ t2, uint32_t lo_bits, const Fr29L *__restrict__ tdirect,
                                                       uint64_t in_len, int in_mul, int out_mul, NttScale sc, int debug_skip) {
    HIP_DYNAMIC_SHARED(Fr29L, lds)
    const Fr *__restrict__ x = cols.x[blockIdx.y];
    Fr *__restrict__ y = cols.y[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    const uint32_t R = 1u << m, C = 1u << cb;
    const uint32_t elems = R << cb;
    Fr29L *tw_s = lds + elems;   // omega_R^k, k < R/2
    Fr29L *scale_s = tw_s + (R >> 1) + 1;               // [0..3) input scales, [3..6) output scales (R' form)
    const uint64_t rows_stride = 1ull << (log_n - m);   // N/R
    const uint32_t ntiles = 1u << (log_n - m - cb);
    const bool has_tw = (log_s + m) < log_n;            // the last pass has j - q == 0 everywhere
    const uint64_t smask = (1ull << log_s) - 1;

    for (uint32_t k = tid; k < (R >> 1); k += 256) tw_s[k].v = tw_lookup(t1, t2, lo_bits, (uint64_t)k << (log_n - m));
    if (tid < 3 && in_mul) scale_s[tid].v = fr29_from_sat(sc.in3[tid]);
    if (tid >= 3 && tid < 6 && out_mul) scale_s[tid].v = fr29_from_sat(sc.out3[tid - 3]);

    Fr pre[NTT_EPT];
    auto fetch = [&](uint32_t tile) {
        const uint64_t j0 = (uint64_t)tile << cb;
#pragma unroll
        for (uint32_t k = 0; k < NTT_EPT; ++k) {
            uint32_t e = tid + 256 * k;
            uint64_t idx = j0 + (e & (C - 1)) + (uint64_t)(e >> cb) * rows_stride;
            pre[k] = (e < elems && idx < in_len) ? x[idx] : Fr::zero();
        }
    };
    uint32_t tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    __syncthreads();   // tw_s / scale_s visible

    for (; tile < ntiles; tile += gridDim.x) {
        const uint64_t j0 = (uint64_t)tile << cb;
#pragma unroll
        for (uint32_t k = 0; k < NTT_EPT; ++k) {
            uint32_t e = tid + 256 * k;
            if (e < elems) {
                uint32_t t = e >> cb, c = e & (C - 1);
                Fr29 v = f29_split<R29P>(pre[k]);
                if (in_mul) {
                    uint64_t idx = j0 + c + (uint64_t)t * rows_stride;
                    v = f29_mul(v, scale_s[idx % 3].v);   // zero padding stays zero
                }
                lds[(bitrev_m(t, m) << cb) + c].v = v;   // DIT: bit-reversed rows in, natural rows out
            }
        }
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);   // next tile's loads overlap this tile's arithmetic

        uint32_t st = (debug_skip & 1) ? m : 0;
        if ((m & 1) && !(debug_skip & 1)) {   // odd number of stages: one radix-2 stage (twiddle 1), then radix-4 rounds
            __syncthreads();
            for (uint32_t b = tid; b < (elems >> 1); b += 256) {
                uint32_t c = b & (C - 1), p = b >> cb;
                uint32_t e0 = ((p << 1) << cb) + c, e1 = e0 + C;
                Fr29 a = lds[e0].v, t = lds[e1].v;
                lds[e0].v = f29_norm(f29_add(a, t));
                lds[e1].v = f29_sub<2>(a, t);
            }
            st = 1;
        }
        for (; st < m; st += 2) {
            const uint32_t h = 1u << st;
            __syncthreads();
            for (uint32_t g = tid; g < (elems >> 2); g += 256) {
                uint32_t c = g & (C - 1), p = g >> cb;
                uint32_t i = p & (h - 1), blk = p >> st;
                uint32_t e0 = (((blk << (st + 2)) + i) << cb) + c, stride = h << cb;
                Fr29 x0 = lds[e0].v, x1 = lds[e0 + stride].v, x2 = lds[e0 + 2 * stride].v, x3 = lds[e0 + 3 * stride].v;
                if (st) {   // stage st: omega_{2h}^i (for st == 0 it is 1)
                    Fr29 w1 = tw_s[i << (m - 1 - st)].v;
                    x1 = f29_mul(x1, w1);
                    x3 = f29_mul(x3, w1);
                }
                // the stage-st outputs stay lazy (no carry propagation): y0, y1 only feed normalising adds / subs and
                // y2, y3 are the wide operands of the twiddle products — four of the round's eight normalisations go away
                Fr29 y0 = f29_add(x0, x1), y1 = f29_sub_lazy<2>(x0, x1);
                // stage st+1 (half = 2h): omega_{4h}^i and omega_{4h}^(i+h)
                Fr29 y2 = f29_mul_wide(f29_add(x2, x3), tw_s[i << (m - 2 - st)].v);
                Fr29 y3 = f29_mul_wide(f29_sub_lazy<2>(x2, x3), tw_s[(i + h) << (m - 2 - st)].v);
                lds[e0].v = f29_norm(f29_add(y0, y2));
                lds[e0 + 2 * stride].v = f29_sub<2>(y0, y2);
                lds[e0 + stride].v = f29_norm(f29_add(y1, y3));
                lds[e0 + 3 * stride].v = f29_sub<2>(y1, y3);
            }
        }
        __syncthreads();

        for (uint32_t e = tid; e < elems; e += 256) {
            uint32_t u, c;
            if (log_s == 0) {   // first pass: output (j0+c)*R + u is contiguous in u
                c = e >> m;
                u = e & (R - 1);
            } else {            // later passes: contiguous in q (i.e. in c)
                u = e >> cb;
                c = e & (C - 1);
            }
            uint64_t j = j0 + c, q = j & smask, jq = j - q;
            Fr29 v = lds[(u << cb) + c].v;
            uint64_t oidx = (jq << m) + q + ((uint64_t)u << log_s);
            if (has_tw && !(debug_skip & 2)) {
                // omega^(jq*u): jq is a multiple of s, so a direct table of omega^(s*t), t < N/s, serves later passes
                Fr29 w = tdirect ? tdirect[(jq >> log_s) * u].v : tw_lookup(t1, t2, lo_bits, jq * u);   // log_s = 0: jq = j
                v = f29_mul(v, w);
            }
            if (out_mul) v = f29_mul(v, scale_s[3 + oidx % 3].v);
            if (!has_tw && !out_mul) v = f29_weak_reduce(v);   // weak bound (<= 21 r) -> < 2 r before packing, no multiply
            y[oidx] = f29_pack_canonical<FrP>(v);
        }
        __syncthreads();   // LDS is overwritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------- full-tile pass (r04)
// The same pass for the case every large transform is made of — a FULL 1024-element tile (m + cb = 10), the inter-pass twiddles from a
// direct table — specialised so that no global-memory latency is left exposed.  What the generic kernel's ISA showed (r04, hipcc -S):
//   * its "prefetch" of the next tile was per-lane conditional (zero padding), the compiler merged the zero / loaded values with moves
//     and put `s_waitcnt vmcnt(1)` right behind the loads: every tile paid a full HBM round trip with all four waves of the workgroup
//     waiting at the same place;
//   * its read-out was a runtime loop, one element per trip: LDS read -> twiddle load -> wait -> product -> pack -> store, i.e. four
//     more serialised L2 / HBM latencies per tile.  (rocprofv3: pass time = VALU time + streaming time, no overlap at all.)
// Here the next tile's loads are unconditional (clamped address, the padding zeros are selected at use), issued before the butterflies and
// first waited for at the next tile's fill; the four read-out twiddles of a lane are requested together, before the barrier that ends the
// last butterfly round, so that one latency (hidden behind that barrier and the LDS reads) is paid instead of four; every per-lane loop is
// four compile-time iterations.  Same LDS layout, same arithmetic, bit-identical results.
// KIND: 0 = first pass (log_s = 0; IN_MUL: fused coset scaling with implicit zero padding), 1 = middle pass, 2 = last pass (no
// inter-pass twiddles; OUT_MUL: fused ifft divisor / zeta^-i scaling).
// LDS: the 1024-element tile as 48-byte elements (three 16-byte accesses per element), the stage twiddles and the three scales behind it as
// bare 36-byte elements, so that a pass with m = 8 (128 twiddles) still fits three workgroups into a CU's 160 KiB (with 48-byte twiddles it
// took 55.7 KiB and only two fitted).  Measured and left behind in r04 (profiles/archive/r04_ntt_experiments.log): a 2048-element build on 36-byte
// elements with 512 lanes, at most 128 registers and no register prefetch — four waves per SIMD instead of three — ran at the same speed
// (2^22 0.60 vs 0.61 ms), as did start delays that put a CU's workgroups out of phase.
struct Fr29P {   // packed LDS element
    uint32_t l[9];
};
__device__ __forceinline__ Fr29 ld29(const Fr29P *p) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p->l[i];
    return r;
}
__device__ __forceinline__ void st29(Fr29P *p, const Fr29 &v) {
#pragma unroll
    for (int i = 0; i < 9; ++i) p->l[i] = v.l[i];
}
// ---- r06: the tile's LDS layout.  PMC had the tile kernel at SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.42; tools/ntt_lds_model.py (the banking
// rules of MI355X_MICROARCH.md §LDS applied to this kernel's index math, access class by access class) reproduces 0.40 for the 2^22 transform
// and names the accesses:
//   * the 9th limb of a 48-byte element moves as a ds_read/write_b32 at a 12-dword lane stride: 4-way, in EVERY access (a third of all conflicts);
//   * rounds, fills and read-outs whose lanes step through elements at a stride of 4 / 8 (cb + st < 4, the bit-reversed fill, the first pass's
//     u-contiguous read-out with cb = 2): 2- to 4-way on the 16-byte accesses;
//   * the packed 36-byte stage twiddles read at power-of-two index strides: up to 4-way on nine 4-byte reads per twiddle.
// MEASURED (profiles/r06_ntt_lds_layout_pmc.log, _times.log, _proof_ab.log): SQ_LDS_BANK_CONFLICT 8.78e6 -> 3.5e5 per pass launch (ratio 0.42 ->
// 0.03), SQ_LDS_IDX_ACTIVE 2.10e7 -> 1.11e7 — and the pass time moves by -2 ... -4 % at 2^22 .. 2^24 and +1 ... +3 % at 2^19 .. 2^21, whole
// proofs unchanged: the LDS was never what this kernel waited for (its LDS pipe is ~5 % busy; it is VALU-issue bound at 313 executed lane
// instructions per algorithmic product).  The 48-byte layout and two other variants (planes without a swizzle, a two-shift swizzle: within 1 % of
// this one everywhere) were A/B'd in the same process and removed.
// The layout stores an element as three PLANES — limbs 0-3 (16 B), limbs 4-7 (16 B), limb 8 (4 B), 36 KiB per tile instead of 48 — at
// the swizzled index e ^ X[e >> 5], X a fixed GF(2)-linear map of the upper five index bits into the lower five (found by hill climbing over the
// model on all (m, cb, pass kind) the planner produces: 0.37 -> 0.06 conflict cycles per LDS cycle on the data), and the stage twiddles in the
// same three planes at the skewed index k + (k >> 4) (0 conflicts).  Modelled LDS-array cycles per tile: 59.8 k -> 29.9 k.  Linear: swz(a ^ b)
// = swz(a) ^ swz(b), so a round's four rows e0 + j * stride cost one swizzle and three wave-uniform XORs.
struct TileLayoutPlanes {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    static __host__ __device__ constexpr uint32_t tw_slots(uint32_t m) { return (1u << (m - 1)) + (1u << (m - 1) >> 4) + 1; }
    static __host__ __device__ constexpr uint32_t tw_stride(uint32_t m) { return (tw_slots(m) + 3u) & ~3u; }   // keeps every plane 16-byte aligned
    static __host__ __device__ constexpr size_t bytes(uint32_t m) { return (size_t)36 * 1024 + (size_t)36 * tw_stride(m) + sizeof(Fr29P) * 3 + 16; }
    v4u *pa, *pb, *ta, *tb;
    uint32_t *pc, *tc;
    Fr29P *scale_s;
    __device__ __forceinline__ void init(void *raw, uint32_t m) {
        const uint32_t ts = tw_stride(m);   // (bytes() sizes the planes with the same stride: the emulated build under AddressSanitizer caught a version that did not)
        pa = reinterpret_cast<v4u *>(raw);
        pb = pa + 1024;
        ta = pb + 1024;
        tb = ta + ts;
        pc = reinterpret_cast<uint32_t *>(tb + ts);
        tc = pc + 1024;
        scale_s = reinterpret_cast<Fr29P *>(tc + ts);
    }
    // X[u] bit b = parity(u & ROW_b), ROW = {1, 6, 18, 15, 27}: bit u of the truth-table word TTb, one v_bfe_u32 per output bit
    static constexpr uint32_t tt(uint32_t row) {
        uint32_t w = 0;
        for (uint32_t u = 0; u < 32; ++u) {
            uint32_t x = u & row, par = 0;
            for (; x; x >>= 1) par ^= x & 1;
            w |= par << u;
        }
        return w;
    }
    __device__ __forceinline__ uint32_t swz(uint32_t e) const {
        constexpr uint32_t T0 = tt(1), T1 = tt(6), T2 = tt(18), T3 = tt(15), T4 = tt(27);
        const uint32_t u = e >> 5;
        const uint32_t x = ((T0 >> u) & 1u) | (((T1 >> u) & 1u) << 1) | (((T2 >> u) & 1u) << 2) | (((T3 >> u) & 1u) << 3) | (((T4 >> u) & 1u) << 4);
        return e ^ x;
    }
    __device__ __forceinline__ Fr29 ldp(uint32_t p) const {
        const v4u a = pa[p], b = pb[p];
        Fr29 r;
        r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
        r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
        r.l[8] = pc[p];
        return r;
    }
    __device__ __forceinline__ void stp(uint32_t p, const Fr29 &v) const {
        v4u a, b;
        a.x = v.l[0]; a.y = v.l[1]; a.z = v.l[2]; a.w = v.l[3];
        b.x = v.l[4]; b.y = v.l[5]; b.z = v.l[6]; b.w = v.l[7];
        pa[p] = a;
        pb[p] = b;
        pc[p] = v.l[8];
    }
    __device__ __forceinline__ Fr29 ldtw(uint32_t k) const {
        const uint32_t p = k + (k >> 4);
        const v4u a = ta[p], b = tb[p];
        Fr29 r;
        r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
        r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
        r.l[8] = tc[p];
        return r;
    }
    __device__ __forceinline__ void sttw(uint32_t k, const Fr29 &v) const {
        const uint32_t p = k + (k >> 4);
        v4u a, b;
        a.x = v.l[0]; a.y = v.l[1]; a.z = v.l[2]; a.w = v.l[3];
        b.x = v.l[4]; b.y = v.l[5]; b.z = v.l[6]; b.w = v.l[7];
        ta[p] = a;
        tb[p] = b;
        tc[p] = v.l[8];
    }
};
template <int KIND, bool MUL, typename L>
__global__ __launch_bounds__(256, 3) void ntt_tile_kernel(NttCols cols, uint32_t log_n, uint32_t m, uint32_t log_s, uint32_t cb,
                                                          const Fr29L *__restrict__ t1, const Fr29L *__restrict__ t2, uint32_t lo_bits,
                                                          const Fr29L *__restrict__ tdirect, uint32_t in_len, NttScale sc) {
    constexpr bool FIRST = KIND == 0, LAST = KIND == 2, PREFETCH = true;
    constexpr uint32_t TB = 10, TILE = 1u << TB, T = TILE / 4;
    HIP_DYNAMIC_SHARED(uint4, lds_tile_raw)
    L lay;
    lay.init(lds_tile_raw, m);
    const Fr *__restrict__ x = cols.x[blockIdx.y];
    Fr *__restrict__ y = cols.y[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    const uint32_t R = 1u << m, C = 1u << cb;   // R * C = TILE
    Fr29P *scale_s = lay.scale_s;                          // [0..3) the three scales (R' form); the stage twiddles omega_R^k, k < R/2: lay.ldtw(k)
    const uint32_t rows_stride = 1u << (log_n - m);
    const uint32_t ntiles = 1u << (log_n - TB);
    const uint32_t smask = (1u << log_s) - 1;
    for (uint32_t k = tid; k < (R >> 1); k += T) lay.sttw(k, tw_lookup(t1, t2, lo_bits, (uint64_t)k << (log_n - m)));
    if (MUL && tid < 3) st29(&scale_s[tid], fr29_from_sat(FIRST ? sc.in3[tid] : sc.out3[tid]));

    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u pre[4][2];
    // r05: coeff_to_extended by a factor of >= 4 (every coset transform of the prover: n coefficients on a 4n- or 8n-point domain) pads with zeros
    // from row R/4 on, and a lane's element k sits in rows [k R/4, (k+1) R/4): elements 1..3 of EVERY lane are zero.  Their loads, splits and
    // coset-scaling products are skipped (3 of the 4 scale products), and the first butterfly round sees (x0, 0, 0, 0) per group: for an even m
    // it is a copy of x0 to the group's four rows (no product), for an odd m the radix-2 stage is a copy and the first radix-4 round drops the
    // two products by zero (`quarter`, wave-uniform).  Same residues, hence the same canonical output bits.
    const bool quarter = FIRST && in_len <= ((1u << log_n) >> 2) && log_n >= 2;
    auto fetch = [&](uint32_t tile) {
        const uint32_t j0 = tile << cb;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            if (FIRST && k && quarter) continue;
            const uint32_t e = tid + T * k;
            uint32_t idx = j0 + (e & (C - 1)) + (e >> cb) * rows_stride;
            if (FIRST) idx = idx < in_len ? idx : 0u;   // implicit zero padding: a harmless in-range address, the zero is selected at the fill
            const v4u *q = reinterpret_cast<const v4u *>(x + idx);
            pre[k][0] = q[0];
            pre[k][1] = q[1];
        }
    };
    uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    if (PREFETCH) fetch(tile);
    __syncthreads();   // tw_s / scale_s visible

    // one radix-4 group per lane and round
    const uint32_t gc = tid & (C - 1), gp = tid >> cb;
    // zeros: 0 = a general round; 1 = (st == 0, quarter) x1 = x2 = x3 = 0: the round is a copy; 2 = (st == 1, quarter, odd m) x1 = x3 = 0
    auto round4 = [&](uint32_t st, int zeros) {
        const uint32_t h = 1u << st;
        const uint32_t i = gp & (h - 1), blk = gp >> st;
        const uint32_t e0 = (((blk << (st + 2)) + i) << cb) + gc, stride = h << cb;
        // the four rows e0 + j * stride (bits st + cb, st + cb + 1 of e0 are clear): one swizzle per lane, the rows' offsets are wave-uniform XORs
        const uint32_t p0 = lay.swz(e0), p1 = p0 ^ lay.swz(stride), p2 = p0 ^ lay.swz(2 * stride), p3 = p0 ^ lay.swz(3 * stride);
        if (zeros == 1) {
            const Fr29 x0 = lay.ldp(p0);
            lay.stp(p1, x0);
            lay.stp(p2, x0);
            lay.stp(p3, x0);
            return;
        }
        if (zeros == 2) {   // y0 = y1 = x0 (mod r), y2 = x2 w, y3 = x2 w'
            const Fr29 x0 = lay.ldp(p0), x2 = lay.ldp(p2);
            const Fr29 y2 = f29_mul(x2, lay.ldtw(i << (m - 2 - st)));
            const Fr29 y3 = f29_mul(x2, lay.ldtw((i + h) << (m - 2 - st)));
            lay.stp(p0, f29_norm(f29_add(x0, y2)));
            lay.stp(p2, f29_sub<2>(x0, y2));
            lay.stp(p1, f29_norm(f29_add(x0, y3)));
            lay.stp(p3, f29_sub<2>(x0, y3));
            return;
        }
        Fr29 x0 = lay.ldp(p0), x1 = lay.ldp(p1), x2 = lay.ldp(p2), x3 = lay.ldp(p3);
        if (st) {   // stage st: omega_{2h}^i (for st == 0 it is 1)
            const Fr29 w1 = lay.ldtw(i << (m - 1 - st));
            x1 = f29_mul(x1, w1);
            x3 = f29_mul(x3, w1);
        }
        const Fr29 y0 = f29_add(x0, x1), y1 = f29_sub_lazy<2>(x0, x1);
        // stage st+1 (half = 2h): omega_{4h}^i and omega_{4h}^(i+h)
        const Fr29 y2 = f29_mul_wide(f29_add(x2, x3), lay.ldtw(i << (m - 2 - st)));
        const Fr29 y3 = f29_mul_wide(f29_sub_lazy<2>(x2, x3), lay.ldtw((i + h) << (m - 2 - st)));
        lay.stp(p0, f29_norm(f29_add(y0, y2)));
        lay.stp(p2, f29_sub<2>(y0, y2));
        lay.stp(p1, f29_norm(f29_add(y1, y3)));
        lay.stp(p3, f29_sub<2>(y1, y3));
    };

    for (;;) {
        const uint32_t j0 = tile << cb;
        if (!PREFETCH) fetch(tile);
        // ---- fill: the rows go to LDS bit-reversed (DIT: bit-reversed rows in, natural rows out)
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t e = tid + T * k;
            const uint32_t t = e >> cb, c = e & (C - 1);
            if (FIRST && k && quarter) {   // a zero row; with an even m the first round (a copy of x0) overwrites it anyway
                if (m & 1) lay.stp(lay.swz((bitrev_m(t, m) << cb) + c), Fr29::zero());
                continue;
            }
            Fr s;
            s.l[0] = pre[k][0].x; s.l[1] = pre[k][0].y; s.l[2] = pre[k][0].z; s.l[3] = pre[k][0].w;
            s.l[4] = pre[k][1].x; s.l[5] = pre[k][1].y; s.l[6] = pre[k][1].z; s.l[7] = pre[k][1].w;
            Fr29 v = f29_split<R29P>(s);
            if (FIRST) {
                const uint32_t idx = j0 + c + t * rows_stride;
                if (MUL) v = f29_mul(v, ld29(&scale_s[idx % 3u]));
                if (idx >= in_len) v = Fr29::zero();
            }
            lay.stp(lay.swz((bitrev_m(t, m) << cb) + c), v);
        }
        const uint32_t next = tile + gridDim.x;
        const bool has_next = next < ntiles;       // wave-uniform
        if (PREFETCH && has_next) fetch(next);     // in flight during the whole transform of this tile
        __syncthreads();
        uint32_t st = 0;
        if (m & 1) {   // odd number of stages: one radix-2 stage (twiddle 1) first
#pragma unroll
            for (uint32_t k = 0; k < 2; ++k) {
                const uint32_t b = tid + T * k;
                const uint32_t c = b & (C - 1), p = b >> cb;
                const uint32_t e0 = lay.swz(((p << 1) << cb) + c), e1 = e0 ^ lay.swz(C);   // bit cb of the row index is clear
                const Fr29 a = lay.ldp(e0), t = lay.ldp(e1);
                lay.stp(e0, f29_norm(f29_add(a, t)));
                lay.stp(e1, f29_sub<2>(a, t));
            }
            st = 1;
            __syncthreads();
        }
        for (; st + 2 < m; st += 2) {
            if (FIRST && quarter && st < 2) round4(st, st == 0 ? 1 : 2);
            else round4(st, 0);
            __syncthreads();
        }
        if (FIRST && quarter && st < 2) round4(st, st == 0 ? 1 : 2);   // (m = 2 or 3: the only round)
        else round4(st, 0);   // st == m - 2
        // ---- the lane's four read-out positions and their inter-pass twiddles, requested between the last round's LDS writes and the barrier that ends it (the round's
        // temporaries are dead by then: requesting them before the round costs 40 more registers per lane, i.e. a wave per SIMD)
        uint32_t oidx[4], lidx[4];
        Fr29 twr[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t e = tid + T * k;
            uint32_t u, c;
            if (FIRST) {   // first pass: output (j0 + c) * R + u is contiguous in u
                c = e >> m;
                u = e & (R - 1);
            } else {       // later passes: contiguous in q (i.e. in c)
                u = e >> cb;
                c = e & (C - 1);
            }
            const uint32_t j = j0 + c, q = j & smask, jq = j - q;
            lidx[k] = lay.swz((u << cb) + c);
            oidx[k] = (jq << m) + q + (u << log_s);
            if (!LAST) {   // omega^(jq*u): jq is a multiple of s, the table holds omega^(s*t)
                const v4u *tp = reinterpret_cast<const v4u *>(tdirect + (size_t)(jq >> log_s) * u);
                const v4u a = tp[0], b = tp[1];
                const uint32_t c8 = reinterpret_cast<const uint32_t *>(tp)[8];
                twr[k].l[0] = a.x; twr[k].l[1] = a.y; twr[k].l[2] = a.z; twr[k].l[3] = a.w;
                twr[k].l[4] = b.x; twr[k].l[5] = b.y; twr[k].l[6] = b.z; twr[k].l[7] = b.w;
                twr[k].l[8] = c8;
            }
        }
        __syncthreads();
        // ---- read-out
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            Fr29 v = lay.ldp(lidx[k]);
            if (!LAST) v = f29_mul(v, twr[k]);
            if (LAST && MUL) v = f29_mul(v, ld29(&scale_s[oidx[k] % 3u]));
            if (LAST && !MUL) v = f29_weak_reduce(v);   // weak bound (<= 21 r) -> < 2 r before packing, no multiply
            // a pass that another pass of this transform reads back leaves the product (N, < 1.2 r) unreduced in the scratch buffer: the next fill splits
            // it again and its butterflies' bound moves from 21 r to 21.2 r (< 32 r: f29_weak_reduce); only the LAST pass writes canonical elements
            const Fr o = LAST ? f29_pack_canonical<FrP>(v) : f29_pack_weak<FrP>(v);
            v4u *yp = reinterpret_cast<v4u *>(y + oidx[k]);
            v4u lo, hi;
            lo.x = o.l[0]; lo.y = o.l[1]; lo.z = o.l[2]; lo.w = o.l[3];
            hi.x = o.l[4]; hi.y = o.l[5]; hi.z = o.l[6]; hi.w = o.l[7];
            yp[0] = lo;
            yp[1] = hi;
        }
        if (!has_next) break;
        tile = next;
        __syncthreads();   // LDS is overwritten by the next tile
    }
}

__global__ void ntt_direct_twiddle_kernel(Fr29L *out, Fr omega, uint32_t log_s, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i].v = fr29_from_sat(fe_pow_u64(omega, (uint64_t)i << log_s));
}

static int get_twiddles(h2hip_ctx *ctx, uint32_t log_n, const Fr &omega, TwiddleSet **out) {
    for (auto &t : ctx->twiddles)
        if (t.log_n == log_n && t.omega == omega) {
            *out = &t;
            return H2HIP_OK;
        }
    TwiddleSet t;
    t.log_n = log_n;
    t.omega = omega;
    t.lo_bits = (log_n + 1) / 2;
    uint32_t lo_count = 1u << t.lo_bits, hi_count = 1u << (log_n - t.lo_bits);
    H2_HIPCHK(hipMalloc((void **)&t.t1, sizeof(Fr29L) * lo_count));
    H2_HIPCHK(hipMalloc((void **)&t.t2, sizeof(Fr29L) * hi_count));
    uint32_t cnt = lo_count > hi_count ? lo_count : hi_count;
    prof_begin(ctx, "ntt_twiddle_kernel");
    hipLaunchKernelGGL(ntt_twiddle_kernel, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, (Fr29L *)t.t1, (Fr29L *)t.t2, omega, t.lo_bits, hi_count);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    if (ctx->twiddles.size() >= 16) {   // bounded cache: drop the oldest table
        H2_HIPCHK(hipStreamSynchronize(ctx->stream));
        hipFree(ctx->twiddles.front().t1);
        hipFree(ctx->twiddles.front().t2);
        for (int k = 0; k < 4; ++k) {
            if (ctx->twiddles.front().direct[k]) hipFree(ctx->twiddles.front().direct[k]);
        }
        ctx->twiddles.erase(ctx->twiddles.begin());
    }
    ctx->twiddles.push_back(t);
    *out = &ctx->twiddles.back();
    return H2HIP_OK;
}

// the (log_n, omega) power table for kernels outside this file (fr29.cuh: pow_lookup); cached like every twiddle set of the context
int ntt_pow_table(h2hip_ctx *ctx, uint32_t log_n, const Fr &omega, OmegaTable *out) {
    TwiddleSet *tw = nullptr;
    H2_CHK(get_twiddles(ctx, log_n, omega, &tw));
    out->t1 = (const Fr29L *)tw->t1;
    out->t2 = (const Fr29L *)tw->t2;
    out->lo_bits = tw->lo_bits;
    return H2HIP_OK;
}

// direct table omega^(t << log_s), t < 2^(log_n - log_s), for a non-first pass (only when it is small: <= 2^16 entries)
static int get_direct_table(h2hip_ctx *ctx, TwiddleSet *tw, uint32_t log_s, const Fr29L **out) {
    *out = nullptr;
    const uint32_t bits = tw->log_n - log_s;
    // later passes: small tables (<= 2^16 entries); first pass (log_s = 0): the full omega^e table, e < N, replaces the
    // composed two-level lookup (one multiply per element) at 48 B of extra HBM read per element — the NTT is
    // multiplier-bound, not bandwidth-bound.  Capped at 2^23 entries (384 MiB).
    if ((log_s != 0 && bits > 16) || (log_s == 0 && (bits > 23 || !ctx->ntt_full_table))) return H2HIP_OK;
    for (int k = 0; k < 4; ++k)
        if (tw->direct[k] && tw->direct_log_s[k] == log_s) {
            *out = (const Fr29L *)tw->direct[k];
            return H2HIP_OK;
        }
    for (int k = 0; k < 4; ++k)
        if (!tw->direct[k]) {
            const uint32_t count = 1u << bits;
            H2_HIPCHK(hipMalloc(&tw->direct[k], sizeof(Fr29L) * count));
            tw->direct_log_s[k] = log_s;
            prof_begin(ctx, "ntt_twiddle_kernel");
            hipLaunchKernelGGL(ntt_direct_twiddle_kernel, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, (Fr29L *)tw->direct[k], tw->omega, log_s, count);
            prof_end(ctx);
            H2_HIPCHK(hipGetLastError());
            *out = (const Fr29L *)tw->direct[k];
            return H2HIP_OK;
        }
    return H2HIP_OK;   // all slots taken: fall back to the composed lookup
}

// a[j]: N = 2^log_n device elements each (results land there), ncols equal-size columns transformed together.  in_override (optional):
// column j reads its input from in_override[j] (in_len valid elements, implicit zeros beyond).  in_scale3 / out_scale3 (optional, host
// pointers to 3 Fr): multiply input / output element i by scale[i mod 3].
int ntt_run_batch(h2hip_ctx *ctx, Fr *const *a, const Fr *const *in_override, size_t ncols, uint32_t log_n, const Fr &omega, uint64_t in_len,
                  const Fr *in_scale3, const Fr *out_scale3) {
    H2_REQUIRE(log_n <= 28, "log_n exceeds the 2-adicity of F_r (28)");
    const uint64_t N = 1ull << log_n;
    if (!in_override) in_len = N;
    H2_REQUIRE(in_len <= N, "input longer than the transform");
    if (!ncols) return H2HIP_OK;
    NttScale sc;
    for (int i = 0; i < 3; ++i) {
        sc.in3[i] = in_scale3 ? in_scale3[i] : Fr::one();
        sc.out3[i] = out_scale3 ? out_scale3[i] : Fr::one();
    }
    uint32_t LT = (uint32_t)ctx->ntt_tile_bits;
    uint32_t mlist[8], P;
    auto plan = [&](uint32_t lt) {
        if (log_n <= lt) {
            P = 1;
            mlist[0] = log_n;
        } else {
            uint32_t maxm = lt - (uint32_t)ctx->ntt_min_col_bits;   // at least 2^min_col_bits adjacent columns per tile (row segments of 32 B each)
            P = (log_n + maxm - 1) / maxm;
            for (uint32_t i = 0; i < P; ++i) mlist[i] = log_n / P + (i < log_n % P ? 1 : 0);
        }
    };
    plan(LT);
    TwiddleSet *tw = nullptr;
    H2_CHK(get_twiddles(ctx, log_n, omega, &tw));
    const size_t group = ncols < NTT_BATCH ? ncols : NTT_BATCH;
    Fr *scratch = nullptr;
    if (P > 1) H2_CHK(ws_reserve(ctx, h2hip_ctx::WS_NTT, sizeof(Fr) * N * group, (void **)&scratch));

    for (size_t c0 = 0; c0 < ncols; c0 += NTT_BATCH) {
        const uint32_t gc = (uint32_t)(ncols - c0 < NTT_BATCH ? ncols - c0 : NTT_BATCH);
        bool in_scratch = false;   // where the group's current data lives (same for every column: the pass structure is shared)
        uint32_t log_s = 0;
        for (uint32_t i = 0; i < P; ++i) {
            const uint32_t m = mlist[i];
            const bool first = (i == 0), last = (i == P - 1);
            const bool to_scratch = !last && !in_scratch;
            NttCols cols;
            for (uint32_t j = 0; j < NTT_BATCH; ++j) {
                const size_t col = c0 + (j < gc ? j : 0);
                H2_REQUIRE(a[col] && (!in_override || in_override[col]), "NULL column");
                cols.x[j] = first ? (in_override ? in_override[col] : a[col]) : (in_scratch ? scratch + N * j : a[col]);
                cols.y[j] = to_scratch ? scratch + N * j : a[col];
            }
            uint32_t cb = LT - m;
            if (cb > log_n - m) cb = log_n - m;
            if (i > 0 && cb > log_s) cb = log_s;
            const uint32_t tiles = 1u << (log_n - m - cb);
            const uint32_t grid = tiles < (uint32_t)ctx->num_cus * 3 ? tiles : (uint32_t)ctx->num_cus * 3;
            const Fr29L *tdirect = nullptr;
            if (i + 1 < P) H2_CHK(get_direct_table(ctx, tw, log_s, &tdirect));
            // the specialised full-tile kernel: every pass of a transform larger than the tile
            if (ctx->ntt_tile_kernel && P >= 2 && LT == 10 && m + cb == 10 && m >= 2 && (last || tdirect) && N <= (1ull << 28)) {
                const size_t shmem_t = TileLayoutPlanes::bytes(m);
                // one persistent workgroup per slot (measured against equal shares — ceil(tiles / rounds) workgroups, every one walking the same number
                // of tiles: 2^22 0.62 vs 0.575 ms, profiles/archive/r04_ntt_experiments.log: fewer workgroups than slots leave a third of the CUs one short)
                const uint32_t slots = (uint32_t)ctx->num_cus * 3;
                const uint32_t grid_t = tiles < slots ? tiles : slots;
                const bool mul = first ? in_scale3 != nullptr : (last && out_scale3 != nullptr);
                const uint32_t in_len32 = (uint32_t)(first ? in_len : N);
                prof_begin(ctx, "ntt_pass_kernel");
#define H2_NTT_TILE(KIND, MUL)                                                                                                                                    \
    hipLaunchKernelGGL((ntt_tile_kernel<KIND, MUL, TileLayoutPlanes>), dim3(grid_t, gc), dim3(256), shmem_t, ctx->stream, cols, log_n, m, log_s, cb, (const Fr29L *)tw->t1, \
                       (const Fr29L *)tw->t2, tw->lo_bits, tdirect, in_len32, sc)
                if (first) {
                    if (mul) H2_NTT_TILE(0, true);
                    else H2_NTT_TILE(0, false);
                } else if (!last) {
                    H2_NTT_TILE(1, false);
                } else {
                    if (mul) H2_NTT_TILE(2, true);
                    else H2_NTT_TILE(2, false);
                }
#undef H2_NTT_TILE
                prof_end(ctx);
                H2_HIPCHK(hipGetLastError());
                in_scratch = to_scratch;
                log_s += m;
                continue;
            }
            const size_t shmem = sizeof(Fr29L) * (((size_t)1 << (m + cb)) + ((size_t)1 << (m ? m - 1 : 0)) + 8);
            prof_begin(ctx, "ntt_pass_kernel");
            hipLaunchKernelGGL(ntt_pass_kernel, dim3(grid, gc), dim3(256), shmem, ctx->stream, cols, log_n, m, log_s, cb, (const Fr29L *)tw->t1,
                               (const Fr29L *)tw->t2, tw->lo_bits, tdirect, first ? in_len : N, (first && in_scale3) ? 1 : 0,
                               (last && out_scale3) ? 1 : 0, sc, ctx->ntt_debug_skip);
            prof_end(ctx);
            H2_HIPCHK(hipGetLastError());
            in_scratch = to_scratch;
            log_s += m;
        }
    }
    return H2HIP_OK;
}

int ntt_run(h2hip_ctx *ctx, Fr *a, uint32_t log_n, const Fr &omega, const Fr *in_override, uint64_t in_len, const Fr *in_scale3,
            const Fr *out_scale3) {
    return ntt_run_batch(ctx, &a, in_override ? &in_override : nullptr, 1, log_n, omega, in_len, in_scale3, out_scale3);
}

}  // namespace h2
