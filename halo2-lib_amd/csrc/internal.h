// libh2hip internals shared by the translation units (context, workspace, error plumbing, kernel timers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/h2hip.h"
#include "ec.cuh"
#include "ec29.cuh"

namespace h2 {

void set_error(const char *fmt, ...);

#define H2_HIPCHK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            h2::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));     \
            return H2HIP_ERR_HIP;                                                                    \
        }                                                                                            \
    } while (0)

#define H2_CHK(expr)                      \
    do {                                  \
        int r__ = (expr);                 \
        if (r__ != H2HIP_OK) return r__;  \
    } while (0)

#define H2_REQUIRE(cond, msg)                                        \
    do {                                                             \
        if (!(cond)) {                                               \
            h2::set_error("%s: invalid argument: %s", __func__, msg); \
            return H2HIP_ERR_INVALID;                                \
        }                                                            \
    } while (0)

// growable device scratch buffer owned by the context
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct TwiddleSet {   // omega^i tables for one (log_n, omega)
    uint32_t log_n = 0, lo_bits = 0;
    Fr omega;
    void *t1 = nullptr;   // omega^i,            i < 2^lo_bits          (Fr29, R' = 2^261 Montgomery form)
    void *t2 = nullptr;   // omega^(i<<lo_bits), i < 2^(log_n-lo_bits)
    void *direct[4] = {nullptr, nullptr, nullptr, nullptr};   // per-stride direct tables omega^(t << log_s) for later passes
    uint32_t direct_log_s[4] = {0, 0, 0, 0};
};

struct KernelStat {
    double total_ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<float, float>> spans;   // (start, end) of every launch in ms since the profile's reference event
};

}  // namespace h2

struct h2hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    // scratch
    enum { WS_NTT = 0, WS_DIGITS, WS_COUNTS, WS_OFFSETS, WS_CURSOR, WS_SKEY, WS_SVAL, WS_BUCKETS, WS_PKEY0, WS_PVAL0, WS_PKEY1,
           WS_PVAL1, WS_SEG, WS_WIN, WS_OUT, WS_SCAN, WS_TMP0, WS_TMP1, WS_TMP2, WS_STAGE, WS_POSEIDON, WS_FBTABLE, WS_BATCH, WS_LK0, WS_LK1, WS_LK2, WS_LK3, WS_LK4, WS_LK5, WS_VANISH, WS_BATCH_BUCKETS, WS_COUNT };
    h2::DevBuf ws[WS_COUNT];
    std::vector<h2::TwiddleSet> twiddles;
    // tuning knobs (h2hip_set_param)
    int msm_window_bits = 0;   // 0 = auto
    int msm_chunk = 0;         // level-1 entries per lane (0 = auto: 8..64, keeping >= 4 waves per SIMD)
    int msm_seg = 4;           // buckets per running-sum segment (4 / 2 measured 1-2 % faster than 8 on whole proofs at k = 15..19: the chain per segment is 2 additions per bucket + a fixed multiplication by the segment offset)
    int ntt_tile_bits = 10;
    int ntt_min_col_bits = 2;    // log2 of the minimum number of adjacent columns per tile (coalescing vs number of passes)
    int ntt_full_table = 1;      // first pass reads a full omega^e table instead of composing two table entries
    int ntt_tile_kernel = 1;     // 1 (default): full 1024-element tiles go through ntt_tile_kernel (r04: no exposed global-memory latency); 0: the generic pass kernel
    int ntt_debug_skip = 0;      // diagnostics only: 1 = skip butterflies, 2 = skip inter-pass twiddles (wrong results)
    int msm_quad_tails = 1;      // 1: bucket reduction / fold on quad-lane point arithmetic (quad29.cuh)
    int msm_sort_threads = 1024; // workgroup size of the LDS histogram / scatter kernels (256, 512 or 1024)
    int msm_scatter_split = 0;   // bucket sub-ranges per window in the scatter (0 = auto, power of two)
    int msm_scatter_full_lds = 0;   // 1: the scatter declares the full 128 KiB of LDS (one workgroup per CU: one segment per XCD at a time; the default until r06); 0: only its cursors — a scatter workgroup then fits into the slot a retiring accumulation workgroup leaves: whole proofs -1 ... -2 % at k = 17 / 19, -0.5 % at k = 21 (profiles/r06_scatter_lds_ab.log)
    bool is_lane = false;           // a batch lane (child context of h2hip_msm_g1_batch_dev): its MSMs run beside other lanes' kernels
    int msm_chunk_lone = -1;        // entries per lane of a LONE MSM (not on a batch lane): -1 = r05's rule (emax / 2^18 in 8 .. 64: nothing waits for its workgroups' slots, and alone the longer lanes are 1 - 3 % faster), 0 = the batch rule, else the value
    int msm_hist_split = 0;         // bucket sub-ranges per window in the histogram kernel (0 / 1 = the whole window; r06 measured 2 and 4: no gain)
    int msm_sort_groups = 0;        // chunks per window of the counting sort (0 = 32; r06 measured 15 .. 30 and a whole-rounds rule: no gain)
    int msm_hist_packed = 1;        // r06: the LDS histogram as 16-bit counter pairs when a chunk holds < 2^16 scalars (half the LDS: fits beside running accumulations)
    uint32_t pos_t = 0, pos_rf = 0, pos_rp = 0;   // Poseidon spec resident in ws[WS_POSEIDON]
    // per-kernel timing (h2hip_profile_*): HIP events on `stream` around each launch when enabled
    bool profiling = false;
    std::string prof_filter;             // non-empty: only kernels whose name starts with it are bracketed (h2hip_profile_filter)
    std::map<std::string, h2::KernelStat> stats;
    struct PendingLaunch {
        std::string name;
        hipEvent_t begin, end;
        bool ended;   // prof_end ran (a bracket left open by an early return is dropped, not timed)
    };
    std::vector<PendingLaunch> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t prof_ref = nullptr;       // reference event (recorded at h2hip_profile_reset) the launch spans are measured from
    bool own_prof_ref = false;           // lanes borrow the parent's reference
    // batch lanes (h2hip_msm_g1_batch_dev): child contexts with their own stream + scratch
    h2hip_ctx *lane[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // pinned ring for small job tables (kernel argument tables too large for the kernarg segment): staged there, they are uploaded
    // asynchronously without a stream synchronisation per call (upload_jobs in capi.hip)
    char *job_ring = nullptr;
    size_t job_ring_off = 0;
    // One-shot hook of the batch MSM (set by the prover before a commitment round): called on the host right after the lanes' accumulations and
    // merges have been joined into `stream` — `ev` is recorded there at that point — and BEFORE the latency-bound bucket reduction is queued, so
    // that work the caller queues on another stream behind `ev` runs next to the reduction's few waves instead of after them (plonk.hip: the
    // round's challenge-independent transforms).  Cleared before it is called; left set if the call took a path without lanes.
    // a batch MSM whose LAST columns are still being produced when it is called (round 1 of create_proof: the permuted lookup columns): the
    // batch queues its first msm_mid_after columns on the lanes, calls the hook — the caller queues the producing work on this context's stream,
    // host synchronisations allowed — makes the lanes wait for that stream, and goes on with the remaining columns.  Consumed by the batch call.
    std::function<int()> msm_mid_hook;
    size_t msm_mid_after = 0;
    // (r06) a batch MSM whose columns ARRIVE one by one: called with j right before column j's group is queued on its lane; the caller queues what
    // produces column j on this context's stream (create_proof: the host-to-device copy of advice column j — pageable, so it blocks the host while
    // the GPU already works on the columns before it), the lane then waits for that stream.  Consumed by the batch call.
    std::function<int(size_t)> msm_col_hook;
    hipEvent_t fork_ev3 = nullptr;
    int plonk_lazy_upload = 1;       // create_proof with host-resident advice and >= 2 advice columns: column j >= 1 is uploaded inside round 1's commitment batch, right before its MSM is queued
    hipEvent_t fork_ev2 = nullptr;
    std::function<int(hipEvent_t ev)> msm_tail_hook;
    hipEvent_t tail_ev = nullptr;
    int msm_lanes = 0;   // lanes used by h2hip_msm_g1_batch_dev: 0 = auto by size, 1..4
    int plonk_tail_overlap = 1;      // create_proof: the challenge-independent transforms of rounds 1 and 3 run on a side stream next to the commitment MSMs' bucket reduction
    int quotient_29 = 1;             // the quotient identities' kernels on unsaturated 9 x 29-bit limbs (fr29.cuh); 0: the saturated kernels
    int plonk_permute_in_commit = 1; // round 1: the lookup permutation runs inside the commitment batch, behind the advice columns' MSMs (msm_mid_hook)
    int clean_on_lane = 1;           // the batch MSM's bucket zero-fill on its first lane's stream (0: the context's clean stream)
    int kate_29 = 1;                 // the kate division and batched evaluation kernels on unsaturated 9 x 29-bit limbs; 0: the saturated kernels
    int kate_coeffs_per_lane = 0;    // multi-point kate division: coefficients per lane (1, 2, 4, 8); 0 = by length
    int plonk_merge_products = 1;    // one permutation set: its factors and the lookups' go through ONE batched inversion / prefix product
    int plonk_shard_side = 1;        // sharded create_proof: the first-round columns' lagrange_to_coeff (+ all-gather) and coset transforms on a side stream next to round 2's commitments
    int plonk_route_rows = 1;        // r06, sharded create_proof with column-dealt lagrange_to_coeff: the grand products' rows go to the columns' owners by an all-to-all (ncclSend / ncclRecv) instead of to every rank by an all-gather
    int plonk_early_intt = 1;        // round 3: the grand products' lagrange_to_coeff is queued on the side context BEFORE the round's commitments (next to their sorts), only the coset transforms behind the accumulations
    int plonk_gate_before_join = 0;  // the quotient's gate identities start when the FIRST-round columns' cosets are done; the grand products' transforms are joined behind them
    int msm_stagger_sorts = -1;      // batch MSM: lane l's first sort starts when lane l-1's sort is done (the first accumulation starts after ONE sort, not next to NL of them); -1 = auto: with two lanes (from 2^20 points), where it measured -1 % per k = 20 proof; with three lanes it costs 1 - 2.5 % (profiles/r05_early_intt_stagger_ab.log)
    hipEvent_t sorted_ev = nullptr;  // (a lane context) recorded behind the scatter of the next msm_run_cols when sorted_arm is set
    bool sorted_arm = false;
    int plonk_side_on_lanes = 1;     // the side work of plonk_tail_overlap runs on the batch MSM's last (idle) lane context instead of a context of its own
#ifdef H2_HIPEMU
    int plonk_warm_keygen = 0;       // (the CPU-emulated test build does not pay for a second proof per key)
#else
    int plonk_warm_keygen = 1;       // h2hip_plonk_keygen ends with one throw-away proof (pool, twiddles, lanes warm when it returns)
#endif
    int fr_invert_run = 0;           // elements per lane (= per inversion) in h2hip_fr_batch_invert_dev; 0 = auto (n / 2^16 in 4..32)
    int lookup_big_tile_bits = 19;   // lookup sort: 4096-key LDS tiles from 2^bits padded keys (12..28), 1024-key tiles below
    int msm_quad_seg_max = 32768;   // bucket reduction: quad-lane kernels up to this many segments (latency-bound), one-lane kernels above
    int msm_defer_reduce = 1;   // batch API, precomputed bases, > 2^17 points: one bucket reduction for all columns after the lanes join
    int msm_fuse_cols = 0;   // columns fused into one multi-column MSM by h2hip_msm_g1_batch_dev (precomputed bases): 0 = auto (4 up to 2^17 points, else 1)
    // Bucket arrays are zero-filled AFTER their reduction has read them, on a side stream, instead of before the next MSM's sort (a 40 MB
    // fill per 2^19-point MSM on the lane's critical path): clean_bytes[i] leading bytes of clean_ptr[i] are zero once clean_ev has fired.
    // [0] = WS_BUCKETS (an MSM reduced by its own context), [1] = WS_BATCH_BUCKETS (a batch's deferred reduction).
    hipStream_t clean_stream = nullptr;
    hipEvent_t clean_ev = nullptr, clean_ev1 = nullptr, used_ev = nullptr;   // clean_ev: slot 0 (a lone MSM's buckets), clean_ev1: slot 1 (the batch's shared array): the fills may run on different streams
    void *clean_ptr[2] = {nullptr, nullptr};
    size_t clean_bytes[2] = {0, 0};
    hipEvent_t fork_ev = nullptr;
    hipEvent_t timer_ev[2] = {nullptr, nullptr};   // h2hip_timer_start / _stop
    // Host round trips without the runtime's wait (r05): a one-workgroup kernel copies a small result into HOST-MAPPED memory and raises a sequence
    // flag there with a system-scope release; the host spins on the flag (sync_results / sync_stream in capi.hip).  Replaces hipMemcpyAsync(D2H) +
    // hipStreamSynchronize on the prover's ~12 round trips per proof (commitments out, challenges in).  0: the runtime's memcpy + wait.
    int msm_table_split = 1;   // base sets are prepared with 128-byte table entries pre-split into 9 x 29-bit limbs (read when a base set is uploaded / generated)
    int host_poll = 1;
    char *poll_host = nullptr;                 // hipHostMalloc'ed (mapped, coherent): [0, 8) the flag, [64, 64 + POLL_BYTES) the payload
    char *poll_dev = nullptr;                  // the same memory as the device sees it
    unsigned long long poll_seq = 0;
    bool msm_lds_attr_set = false, lookup_lds_attr_set = false, ntt_lds_attr_set = false;   // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for this context's device
};

namespace h2 {
// every tuning knob a child context (a batch lane, the prover's side context) takes from its parent — ADVICE r05: host_poll and the r05
// scheduling knobs did not reach the children, so an A/B through h2hip_set_param switched only part of a proof
inline void inherit_knobs(h2hip_ctx *c, const h2hip_ctx *p) {
    c->msm_window_bits = p->msm_window_bits;
    c->msm_chunk = p->msm_chunk;
    c->msm_seg = p->msm_seg;
    c->msm_scatter_split = p->msm_scatter_split;
    c->msm_scatter_full_lds = p->msm_scatter_full_lds;
    c->msm_hist_packed = p->msm_hist_packed;
    c->msm_sort_groups = p->msm_sort_groups;
    c->msm_hist_split = p->msm_hist_split;
    c->msm_sort_threads = p->msm_sort_threads;
    c->msm_quad_tails = p->msm_quad_tails;
    c->msm_quad_seg_max = p->msm_quad_seg_max;
    c->msm_seg = p->msm_seg;
    c->ntt_tile_bits = p->ntt_tile_bits;
    c->ntt_min_col_bits = p->ntt_min_col_bits;
    c->ntt_full_table = p->ntt_full_table;
    c->ntt_tile_kernel = p->ntt_tile_kernel;
    c->quotient_29 = p->quotient_29;
    c->kate_29 = p->kate_29;
    c->kate_coeffs_per_lane = p->kate_coeffs_per_lane;
    c->fr_invert_run = p->fr_invert_run;
    c->lookup_big_tile_bits = p->lookup_big_tile_bits;
    c->host_poll = p->host_poll;
    c->plonk_route_rows = p->plonk_route_rows;
    c->profiling = p->profiling;
    c->prof_filter = p->prof_filter;
    c->prof_ref = p->prof_ref;   // launch spans of all children share the parent's time origin
}
// HIP's current device is per-thread: every extern "C" entry that takes a context makes the context's device current for
// its duration (allocations, events and attribute calls would otherwise target whatever device the calling thread used
// last — another context's, or torch's) and restores the caller's device on exit.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const h2hip_ctx *c) {
        if (c && hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
}  // namespace h2
#define H2_DEVICE_GUARD(ctx) h2::DeviceGuard device_guard__(ctx)

struct h2hip_bases {
    h2::G1Affine *pts = nullptr;      // [n] affine, saturated Montgomery limbs (as uploaded; h2hip_bases_download)
    h2::G1Affine *pts29 = nullptr;    // [tables][n] the same points in the unsaturated domain (x*2^261, y*2^261): packed 8 x 32-bit limbs, 64 B per entry — or,
                                      // with `split`, 128-byte entries that hold the 9 x 29-bit limbs the accumulation consumes (h2::TableEntry29)
    bool split = false;               // r05 (msm_table_split): pts29 holds TableEntry29, not G1Affine
    size_t n = 0;
    uint32_t window_bits = 0;      // precomputed mode: window the table was built for
    uint32_t tables = 1;           // 1 = plain; W = precomputed 2^(c*w) multiples
};

namespace h2 {
// A table entry as msm_accum_kernel consumes it (r05): x and y already split into 9 x 29-bit limbs, 72 of the 128 bytes used.  A 64-byte gather
// costs the memory system a 128-byte line anyway (profiles/archive/r02_hbm_counter_calibration.md), so the wider entry moves the same HBM bytes and saves the
// two f29_split per addition (54 of ~2440 VALU instructions per step).  Identity = all-zero.
struct alignas(128) TableEntry29 {
    uint32_t x[9], y[9];
    uint32_t pad[14];
};
static_assert(sizeof(TableEntry29) == 128, "one entry per 128-byte line");
int ws_reserve(h2hip_ctx *ctx, int slot, size_t bytes, void **out);

// RAII-less kernel timer: prof_begin/prof_end bracket one launch with events when ctx->profiling.
void prof_begin(h2hip_ctx *ctx, const char *name);
void prof_end(h2hip_ctx *ctx);
bool prof_launch_events(h2hip_ctx *ctx, const char *name, hipEvent_t *start, hipEvent_t *stop);   // events for hipExtLaunchKernelGGL to record

// implemented in ntt.hip / msm.hip / fr_ops.hip, all on device pointers
int ntt_run(h2hip_ctx *ctx, Fr *a, uint32_t log_n, const Fr &omega, const Fr *in_override, uint64_t in_len, const Fr *in_scale3,
            const Fr *out_scale3);
int ntt_run_batch(h2hip_ctx *ctx, Fr *const *a, const Fr *const *in_override, size_t ncols, uint32_t log_n, const Fr &omega, uint64_t in_len,
                  const Fr *in_scale3, const Fr *out_scale3);   // the same transform over ncols equal-size columns, 32 columns per launch
int fr_scatter_rows(h2hip_ctx *ctx, Fr *const *dst, size_t count, const Fr *src, size_t src_stride, size_t len);   // dst[j][i] = src[j*src_stride + i]
int upload_jobs(h2hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   // async H2D of a small table through the context's pinned ring
int exclusive_scan_u32_segments(h2hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t segments, size_t in_stride,
                                size_t out_stride);   // `segments` independent scans of n elements, in_stride / out_stride elements apart
int exclusive_scan_u32(h2hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n);   // out[i] = sum_{j<i} in[j]; in != out
int batch_normalize_jac(h2hip_ctx *ctx, const G1Jac *tmp, G1Affine *out, uint32_t n);   // msm_tables.hip
uint32_t pick_window(size_t n, bool precomp = false);   // Pippenger window width for n points (msm_tables.hip)
int msm_prepare_bases(h2hip_ctx *ctx, h2hip_bases *bases, bool precompute);
int msm_run(h2hip_ctx *ctx, const h2hip_bases *bases, const Fr *scalars_dev, size_t n, XYZZ *out_dev);
constexpr uint32_t MSM_MAX_COLS = 32;   // columns one fused multi-column MSM handles
// ext_buckets != nullptr: stop after the merge and leave the columns' buckets ([col][windows][B]; zeroed here unless ext_buckets_zeroed) there for msm_reduce_cols
int msm_run_cols(h2hip_ctx *ctx, const h2hip_bases *bases, const Fr *const *scalars_dev, uint32_t ncols, size_t n, XYZZ *out_dev,
                 XYZZ29 *ext_buckets, bool ext_buckets_zeroed = false);
// zero-fill-after-use of the bucket arrays (see h2hip_ctx::clean_*): is the buffer's head already (scheduled to be) zero?  (a true answer
// CONSUMES the state: the caller dirties the array) / schedule the fill
bool buckets_prezeroed(h2hip_ctx *ctx, int which, const void *buf, size_t bytes);
int buckets_clean_after_use(h2hip_ctx *ctx, int which, void *buf, size_t bytes, hipStream_t on = nullptr);
// capi.hip: the host waits for everything queued on ctx->stream (sync_stream), optionally fetching `bytes` (<= 16 KiB with host_poll, any size without)
// of results from device memory first (sync_results) — through the host-mapped flag when ctx->host_poll, else hipMemcpyAsync + hipStreamSynchronize
int sync_stream(h2hip_ctx *ctx);
int sync_results(h2hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
// capi.hip: a child context's (MSM lane, the prover's side stream) kernel timers folded into the parent's table
void prof_fold_child(h2hip_ctx *parent, h2hip_ctx *child);
// rng.hip: n elements of the ChaCha Fr::random stream from element first_block on, on `stream`
int rng_chacha_fill_dev(h2hip_ctx *ctx, Fr *out_dev, size_t n, const uint8_t seed[32], int rounds, uint64_t first_block, hipStream_t stream);
// comm.hip: the fallible preparations of a later h2hip_comm_allgather_dev of `bytes` per rank, done ahead of time
int comm_reserve_allgather_dev(h2hip_comm *comm, size_t bytes);
int comm_reserve_alltoall_dev(h2hip_comm *c, size_t bytes);
int msm_reduce_cols(h2hip_ctx *ctx, const h2hip_bases *bases, uint32_t window_bits, const XYZZ29 *buckets, uint32_t ncols, XYZZ *out_dev);
}  // namespace h2
