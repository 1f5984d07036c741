/*
 * libh2hip — C ABI of the MI355X (gfx950) prover backend for halo2-lib's proving hot path.
 *
 * This is the drop-in boundary: a `halo2_proofs`-compatible Rust crate selected through
 * halo2-lib's existing backend seam (reference halo2-base/src/lib.rs:25-28, the `halo2_base::halo2_proofs`
 * re-export toggled by a cargo feature exactly like `cuda`) binds these symbols where upstream
 * halo2-axiom 0.5.3 calls its CPU kernels.  INTEGRATION.md shows the `extern "C"` block and the call sites.
 *
 * Data formats (bit-identical to halo2curves' in-memory types; reference halo2-base/src/utils/mod.rs:28-38,
 * 342-377 and halo2-ecc/benches/msm.rs:62):
 *   Fr / Fq      4 x u64 little-endian limbs, MONTGOMERY form (R = 2^256)          32 B
 *   G1Affine     { Fq x; Fq y }, identity = all-zero bytes                         64 B
 *   G1 (curve)   { Fq x; Fq y; Fq z } Jacobian, identity has z = 0                 96 B
 *
 * Conventions: every function returns H2HIP_OK (0) or a negative H2HIP_ERR_* code and never throws or
 * aborts; h2hip_last_error() returns a thread-local message.  Host buffers stay caller-owned.  A context
 * serialises its work on one HIP stream; use it from one thread at a time (the reference calls
 * create_proof from a single thread, halo2-base/src/utils/testing.rs:32-50).  DIFFERENT contexts (each with its own
 * base sets and proving keys) may be used from different threads at the same time, also on one device: the library holds no
 * mutable global state (tests/test_plonk_prover.py::test_create_proof_gpu_three_contexts_in_flight).  There is NO CPU fallback:
 * without a usable HIP device h2hip_init fails with H2HIP_ERR_NO_DEVICE.
 *
 * Functions ending in `_dev` take device pointers (from h2hip_malloc, or any HIP allocation on the
 * context's device, e.g. a torch tensor's data_ptr) so polynomials can stay resident in HBM between calls;
 * the un-suffixed forms stage host buffers through the context's stream.
 */
#ifndef H2HIP_H
#define H2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2HIP_OK 0
#define H2HIP_ERR_INVALID (-1)   /* bad argument (message says which) */
#define H2HIP_ERR_HIP (-2)       /* HIP runtime / kernel launch failure */
#define H2HIP_ERR_NOMEM (-3)
#define H2HIP_ERR_NO_DEVICE (-4)
#define H2HIP_ERR_PEER (-5)      /* another rank of a sharded (multi-GPU) call reported an error: every rank returns */

#define H2HIP_POINT_JACOBIAN 0   /* 96 B, what best_multiexp returns (C::Curve) */
#define H2HIP_POINT_AFFINE 1     /* 64 B, normalised on the device (one field inversion) */

#define H2HIP_BASES_PLAIN 0u
#define H2HIP_BASES_PRECOMPUTE 1u /* also store 2^(c*w)*P_i for every window: fixed-base (SRS) fast path */

typedef struct h2hip_ctx h2hip_ctx;
typedef struct h2hip_bases h2hip_bases;

/* ---- context ------------------------------------------------------------------------------------- */
const char *h2hip_last_error(void);
int h2hip_version(void);
int h2hip_device_count(int *count);
/* hip_stream: an existing hipStream_t to run on (NULL: the context creates its own). */
int h2hip_init(int device, void *hip_stream, h2hip_ctx **out);
void h2hip_destroy(h2hip_ctx *ctx);
int h2hip_sync(h2hip_ctx *ctx);
/* tuning knobs (defaults are the tuned values): "msm_window_bits" (0 = auto), "msm_chunk" (0 = auto), "msm_seg",
 * "msm_scatter_split" (0 = auto), "msm_lanes", "msm_quad_tails", "msm_fuse_cols", "msm_defer_reduce", "ntt_tile_bits" (10),
 * "ntt_tile_kernel" (1 = the specialised full-tile pass kernel, 0 = the generic one), "ntt_min_col_bits", "ntt_full_table",
 * "msm_scatter_full_lds", "msm_sort_threads", "msm_quad_seg_max", "fr_invert_run" (0 = auto), "lookup_big_tile_bits",
 * "kate_coeffs_per_lane" (0 = by length); arithmetic form of the pointwise kernels: "quotient_29", "kate_29" (1 = unsaturated 9 x 29-bit
 * limbs, 0 = the saturated kernels: same results); the prover's scheduling: "plonk_tail_overlap", "plonk_side_on_lanes",
 * "plonk_permute_in_commit", "plonk_warm_keygen", "clean_on_lane" (1 everywhere: 0 switches the overlap off, same proof bytes);
 * r05: "host_poll" (1: results of a round come back through a host-mapped flag instead of hipMemcpyAsync + hipStreamSynchronize),
 * "msm_table_split" (1: base sets uploaded / generated AFTER the call get 128-byte table entries pre-split into 9 x 29-bit limbs),
 * "plonk_merge_products" (1: one batched inversion / prefix product for the permutation set and the lookups when nothing chains),
 * "plonk_shard_side" (1: sharded proofs run the first-round columns' transforms and all-gather on a side stream),
 * "plonk_early_intt" (1: the grand products' lagrange_to_coeff is queued on the side context in front of their commitment round),
 * "plonk_gate_before_join" (0; 1: the quotient's gate identities start before the grand products' transforms are joined — measured neutral),
 * "msm_stagger_sorts" (-1 = auto: two-lane batches; 1 / 0: a batch's lanes start their first sorts one behind the other / together),
 * r06: "msm_hist_packed" (1: the sort's LDS histogram as 16-bit counter pairs whenever a chunk holds < 2^16 scalars), "msm_scatter_full_lds"
 * (0: the scatter declares its cursors only, so that it fits beside running accumulations), "msm_hist_split" (0 / 1 = the whole window per
 * histogram workgroup; n = n bucket sub-ranges), "msm_sort_groups" (0 = auto), "msm_chunk_lone" (-1 = auto: a lone MSM keeps the longer
 * entries-per-lane rule, batch lanes the shorter one), "plonk_lazy_upload" (1: host-resident advice columns 1.. are uploaded inside round 1's
 * commitment batch, each right before its MSM is queued), "plonk_route_rows" (1: sharded proofs route the grand products' rows to the column
 * owners by h2hip_comm_alltoall_dev instead of all-gathering them; same proof bytes);
 * profiling aid: "ntt_debug_skip" (produces wrong results).  The variants r01-r03 measured slower (two-level sort, bucket-major sort,
 * split streams, split windows, accumulation builds 2/5/6/7, radix-8 and wave-local NTT passes) were removed in r04, r05's (two-wave
 * accumulation, sort-first batches, split lone commitments) in r05, r05's wave-owned radix-8 NTT pass ("ntt_w8") and the 48-byte NTT tile
 * layout in r06 (tools/probes/); their A/B logs stay under profiles/. */
int h2hip_set_param(h2hip_ctx *ctx, const char *name, int value);
int h2hip_get_param(h2hip_ctx *ctx, const char *name, int *value);

/* ---- device memory (for callers without a HIP runtime of their own, e.g. the Rust shim) ----------- */
int h2hip_malloc(h2hip_ctx *ctx, size_t bytes, void **dptr);
int h2hip_free(h2hip_ctx *ctx, void *dptr);
int h2hip_upload(h2hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* synchronous */
int h2hip_download(h2hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* synchronous */
/* page-lock / release a caller-owned host buffer (a prover's long-lived Vec<Fr> columns): transfers from registered memory are true
 * asynchronous DMA; unregistered (pageable) buffers work everywhere too, through the runtime's staging copies */
int h2hip_host_register(h2hip_ctx *ctx, void *host_ptr, size_t bytes);
int h2hip_host_unregister(h2hip_ctx *ctx, void *host_ptr);

/* ---- per-kernel timing (HIP events on the context's stream around every launch) ------------------- */
int h2hip_profile_enable(h2hip_ctx *ctx, int on);
int h2hip_profile_reset(h2hip_ctx *ctx);
/* restrict the event brackets to kernels whose name starts with `prefix` (NULL or "" = all): a timed region can then carry the dominant
 * kernel's events only (two events per launch cost ~1 us each; a k = 19 proof has ~200 launches) */
int h2hip_profile_filter(h2hip_ctx *ctx, const char *prefix);
/* total milliseconds and launch count accumulated for kernels whose name starts with `prefix` */
int h2hip_profile_get(h2hip_ctx *ctx, const char *prefix, double *total_ms, uint64_t *launches);
/* milliseconds (since the last h2hip_profile_reset) during which at least one launch of the matching kernels was executing:
 * the union of the launch spans — with pipelined MSMs several launches of one kernel overlap */
int h2hip_profile_get_busy(h2hip_ctx *ctx, const char *prefix, double *busy_ms);
/* the whole account since the last reset as text, one "name total_ms launches busy_ms" line per kernel name (NUL-terminated, truncated at
 * cap; *needed = bytes for all of it) — what bench.py turns into the per-kernel table of a create_proof */
int h2hip_profile_dump(h2hip_ctx *ctx, char *out, size_t cap, size_t *needed);
/* bracket an arbitrary region with events on the context's stream (bench.py's timed region) */
int h2hip_timer_start(h2hip_ctx *ctx);
int h2hip_timer_stop(h2hip_ctx *ctx, double *elapsed_ms);

/* ---- K1: multi-scalar multiplication  (replaces arithmetic::best_multiexp [UPSTREAM], reached from the
 *      reference via ParamsKZG::commit / commit_lagrange inside create_proof,
 *      halo2-base/src/utils/testing.rs:40-47) ------------------------------------------------------- */
/* Upload n G1Affine bases once (the SRS `g` or `g_lagrange` of ParamsKZG, reference
 * halo2-base/src/utils/mod.rs:401-443); they stay resident in HBM. */
int h2hip_bases_upload(h2hip_ctx *ctx, const void *g1_affine_host, size_t n, uint32_t flags, h2hip_bases **out);
int h2hip_bases_from_device(h2hip_ctx *ctx, const void *g1_affine_dev, size_t n, uint32_t flags, h2hip_bases **out);
void h2hip_bases_free(h2hip_ctx *ctx, h2hip_bases *bases);
size_t h2hip_bases_len(const h2hip_bases *bases);
/* out = sum_{i<n} scalars[i] * bases[i];  n <= h2hip_bases_len.  point_format selects 96 B Jacobian or 64 B affine. */
int h2hip_msm_g1(h2hip_ctx *ctx, const h2hip_bases *bases, const void *scalars_host, size_t n, int point_format, void *out_host);
int h2hip_msm_g1_dev(h2hip_ctx *ctx, const h2hip_bases *bases, const void *scalars_dev, size_t n, int point_format, void *out_host);

/* `count` independent MSMs over the same bases (all advice columns of a phase, the pieces of h(X), ...).
 * scalars_dev: host array of `count` device pointers, n scalars each; out_host: `count` points.  The MSMs are
 * pipelined over the context's lanes ("msm_lanes" internal streams; default: chosen by size) so that one MSM's sort and latency-bound tail
 * overlap another one's accumulation.  With precomputed tables, columns of up to 2^17 scalars are FUSED into multi-column MSMs of about
 * 2^19 scalars (up to 16 columns share every launch), and the latency-bound bucket reduction is deferred: it runs once per 64 columns of
 * the batch after the lanes have joined. */
int h2hip_msm_g1_batch_dev(h2hip_ctx *ctx, const h2hip_bases *bases, const void *const *scalars_dev, size_t n, size_t count,
                           int point_format, void *out_host);
/* the same with a base set PER COLUMN: the commitments of one prover round that live over different SRS columns (Lagrange-basis and
 * monomial-basis polynomials) in one pipelined call.  The sets must share their table layout (all plain, or all precomputed with the same
 * window — true for the g / g_lagrange of one ParamsKZG). */
int h2hip_msm_g1_multi_dev(h2hip_ctx *ctx, const h2hip_bases *const *bases_per_column, const void *const *scalars_dev, size_t n, size_t count,
                           int point_format, void *out_host);
/* the same for scalar columns in HOST memory (an unmodified prover's Vec<Fr>): each column is uploaded on its lane's stream
 * right before its kernels are queued, so the upload of column j+1 overlaps the GPU work of column j */
int h2hip_msm_g1_batch(h2hip_ctx *ctx, const h2hip_bases *bases, const void *const *scalars_host, size_t n, size_t count, int point_format,
                       void *out_host);

/* MSM over G2 (the twist over Fq2).  The reference holds G2 elements only as the verifier half of ParamsKZG (g2, s*g2) and has no
 * prover-side G2 commitment: supported for completeness (8-bit windows, per-bucket lanes), not tuned like the G1 path.
 * points: n x 128 B affine (x.c0, x.c1, y.c0, y.c1 Montgomery limbs; identity all-zero); out: 128 B affine. */
int h2hip_msm_g2(h2hip_ctx *ctx, const void *g2_affine_host, const void *scalars_host, size_t n, void *out_affine_host);
int h2hip_msm_g2_dev(h2hip_ctx *ctx, const void *g2_affine_dev, const void *scalars_dev, size_t n, void *out_affine_host);

/* ---- a2: KZG SRS (ParamsKZG::<Bn256>::setup [UPSTREAM]; reference gen_srs halo2-base/src/utils/mod.rs:439-443,
 *      halo2-base/benches/mul.rs:39).  g[i] = s^i*G1 and g_lagrange[i] = L_i(s)*G1 for i < 2^k are generated on
 *      the GPU (fixed-base window tables) and stay resident as two base sets; `flags` as in h2hip_bases_upload.
 *      The G2 half (g2, s*g2) is verifier-side and stays with the host library. --------------------------- */
int h2hip_params_kzg_setup(h2hip_ctx *ctx, uint32_t k, const void *s_fr, uint32_t flags, h2hip_bases **g_out, h2hip_bases **g_lagrange_out);
/* g_to_lagrange [UPSTREAM poly/kzg/commitment.rs]: Lagrange-basis SRS from the monomial one, g_lagrange[i] = n^-1 * sum_j
 * omega^(-ij) * g[j] with n = 2^k (group-element inverse FFT on the device) — for SRS sources that carry only g
 * (ceremony files, ParamsKZG::from_parts(.., None, ..), downsize).  Uses the first 2^k bases of g. */
int h2hip_g1_to_lagrange(h2hip_ctx *ctx, const h2hip_bases *g, uint32_t k, uint32_t flags, h2hip_bases **g_lagrange_out);
/* out[i] = scalars[i] * base for a fixed G1Affine base (host pointer); scalars and out are device arrays */
int h2hip_g1_fixed_base_mul_batch_dev(h2hip_ctx *ctx, const void *base_affine, const void *scalars_dev, size_t n, void *out_affine_dev);
/* copy the resident affine points back to the host (n x 64 B) — ParamsKZG::write / tests */
int h2hip_bases_download(h2hip_ctx *ctx, const h2hip_bases *bases, void *out_host);

/* SRS files (ParamsKZG::read / gen_srs, reference halo2-base/src/utils/mod.rs:401-443).  RawBytes files carry 64-byte Montgomery points:
 * validate counts the entries that are not canonical on-curve points (the identity (0,0) is allowed).  Processed files carry 32-byte
 * compressed points: decompress fails with H2HIP_ERR_INVALID on any malformed encoding.  sign_bit / inf_bit: positions of the y-sign and
 * identity flags in the top byte (6 / 7 in halo2curves' encoding). */
int h2hip_g1_validate_dev(h2hip_ctx *ctx, const void *points_dev, size_t n, size_t *invalid);
int h2hip_g1_decompress_batch_dev(h2hip_ctx *ctx, const void *compressed_dev, size_t n, void *out_affine_dev, uint32_t sign_bit, uint32_t inf_bit);

/* Sum of n Jacobian points resident on the device (multi-GPU: the all-gathered per-GPU partial MSM results;
 * RCCL has no group-law reduction). */
int h2hip_g1_sum_jacobian_dev(h2hip_ctx *ctx, const void *points_dev, size_t n, int point_format, void *out_host);
/* the host half of a point-range sharded round: out[j] = sum_r gathered[r * count + j] over `world` ranks' Jacobian partials (the layout
 * h2hip_comm_allgather_host leaves), all `count` columns in one call, host arithmetic (no device round trip).  Output Jacobian (z in {0, 1}) or affine. */
int h2hip_g1_sum_partials_host(const void *gathered_jacobian, size_t world, size_t count, int point_format, void *out);

/* ---- K2/K3: NTT family  (replaces arithmetic::best_fft and EvaluationDomain::{ifft, coeff_to_extended,
 *      extended_to_coeff} [UPSTREAM]) ------------------------------------------------ */
/* in-place, natural order in and out, no scaling:  a[k] <- sum_j a[j] * omega^(jk),  len(a) = 2^log_n */
int h2hip_best_fft(h2hip_ctx *ctx, void *a_host, const void *omega, uint32_t log_n);
int h2hip_best_fft_dev(h2hip_ctx *ctx, void *a_dev, const void *omega, uint32_t log_n);
/* EvaluationDomain::ifft: best_fft with omega_inv, then every element times `divisor` (= 2^-log_n) */
int h2hip_ifft(h2hip_ctx *ctx, void *a_host, const void *omega_inv, uint32_t log_n, const void *divisor);
int h2hip_ifft_dev(h2hip_ctx *ctx, void *a_dev, const void *omega_inv, uint32_t log_n, const void *divisor);
/* EvaluationDomain::coeff_to_extended: out[0..2^ext_k) = NTT_{ext_omega}( coeffs[i] * zeta^(i mod 3), zero padded ) */
int h2hip_coeff_to_extended(h2hip_ctx *ctx, const void *coeffs_host, uint32_t k, void *out_host, uint32_t ext_k, const void *ext_omega,
                            const void *zeta);
int h2hip_coeff_to_extended_dev(h2hip_ctx *ctx, const void *coeffs_dev, uint32_t k, void *out_dev, uint32_t ext_k,
                                const void *ext_omega, const void *zeta);
/* ifft / coeff_to_extended over `count` equal-size columns at once (`cols_dev`, `coeffs_dev`, `outs_dev`: HOST arrays of device pointers):
 * 32 columns per kernel launch.  What create_proof does with every advice / permuted / product column (upstream transforms them one by
 * one on the CPU's thread pool, [UPSTREAM-RECALL plonk/prover.rs]); a wide halo2-base shape (bench_pairing.config: k = 14, 211 + 27 advice
 * columns) has ~400 columns of 2^14 rows, each far too small to fill the chip alone. */
int h2hip_ifft_batch_dev(h2hip_ctx *ctx, void *const *cols_dev, size_t count, const void *omega_inv, uint32_t log_n, const void *divisor);
int h2hip_coeff_to_extended_batch_dev(h2hip_ctx *ctx, const void *const *coeffs_dev, uint32_t k, void *const *outs_dev, uint32_t ext_k, size_t count,
                                      const void *ext_omega, const void *zeta);
/* EvaluationDomain::extended_to_coeff (without the final truncation, which is a length change on the
 * caller's Vec): in-place iNTT with ext_omega_inv, times ext_divisor, times [1, zeta_inv, zeta_inv^2][i mod 3]
 * (zeta_inv = zeta^2 because zeta^3 = 1) */
int h2hip_extended_to_coeff(h2hip_ctx *ctx, void *a_host, uint32_t ext_k, const void *ext_omega_inv, const void *ext_divisor,
                            const void *zeta_inv);
int h2hip_extended_to_coeff_dev(h2hip_ctx *ctx, void *a_dev, uint32_t ext_k, const void *ext_omega_inv, const void *ext_divisor,
                                const void *zeta_inv);

/* ---- K8: witness-column F_r batches (halo2-base GateInstructions add/sub/mul/mul_add on column values,
 *      reference halo2-base/src/gates/flex_gate/mod.rs:158-277); out may alias an input ----------------- */
int h2hip_fr_add_batch_dev(h2hip_ctx *ctx, void *out_dev, const void *a_dev, const void *b_dev, size_t n);
int h2hip_fr_sub_batch_dev(h2hip_ctx *ctx, void *out_dev, const void *a_dev, const void *b_dev, size_t n);
int h2hip_fr_mul_batch_dev(h2hip_ctx *ctx, void *out_dev, const void *a_dev, const void *b_dev, size_t n);
int h2hip_fr_mul_add_batch_dev(h2hip_ctx *ctx, void *out_dev, const void *a_dev, const void *b_dev, const void *c_dev, size_t n);
/* polynomial linear combinations (Polynomial * F and += of scaled polynomials in the multiopen argument):
 * y[i] += a * x[i]   and   y[i] *= s ; a, s: host pointers to one Montgomery Fr */
int h2hip_fr_axpy_dev(h2hip_ctx *ctx, void *y_dev, const void *a, const void *x_dev, size_t n);
int h2hip_fr_scale_dev(h2hip_ctx *ctx, void *y_dev, const void *s, size_t n);
/* y[i] = s * y[i] + a * x[i]  (Horner steps over polynomial pieces, e.g. h(X) = sum_i x^(n i) h_i(X)) */
int h2hip_fr_axpby_dev(h2hip_ctx *ctx, void *y_dev, const void *s, const void *a, const void *x_dev, size_t n);
/* out[i] = sum_j coeffs[j] * polys[j][i]: `polys` a HOST array of `count` device columns, `coeffs` `count` host Fr; every operand is read
 * once per 15 terms (the multiopen argument's sum_j y^j P_j(X) per rotation set and its linearisation
 * [UPSTREAM-RECALL poly/kzg/multiopen/shplonk/prover.rs]; `out` may be one of the operands). count = 0 gives zeros. */
int h2hip_fr_linear_combination_dev(h2hip_ctx *ctx, void *out_dev, const void *const *polys_dev, const void *coeffs_host, size_t count, size_t n);
/* y[i] -= low[i] for i < m <= 8 (`low_host`: m elements): P(X) - r(X) for the low-degree interpolant r of an opening set */
int h2hip_fr_sub_low_dev(h2hip_ctx *ctx, void *y_dev, const void *low_host, uint32_t m);

/* ---- K4: BatchInvert, in place, 0 -> 0 (ff::BatchInvert / batch_invert_assigned [UPSTREAM]; the deferred
 *      denominators come from reference halo2-base/src/gates/flex_gate/mod.rs:677-681,791-795) ---------- */
int h2hip_fr_batch_invert_dev(h2hip_ctx *ctx, void *a_dev, size_t n);

/* a5/a6: advice cells arrive as Assigned<F> — Trivial values and Rational(num, den) with the inversion deferred (reference
 * halo2-base/src/gates/flex_gate/mod.rs:677-681,791-795; columns laid out by threads/single_phase.rs:273-312).  out = num * den^-1
 * with 0^-1 := 0 (batch_invert_assigned [UPSTREAM]); Trivial cells carry den = 1.  out may alias den. */
int h2hip_assigned_resolve_dev(h2hip_ctx *ctx, void *out_dev, const void *num_dev, const void *den_dev, size_t n);

/* ---- K5: permutation / lookup grand products: out[i] = prod_{j<=i} in[j], and
 *      z[0] = 1, z[i+1] = z[i]*num[i]/den[i] (z has n+1 elements) ---------------------------------------- */
int h2hip_fr_prefix_product_dev(h2hip_ctx *ctx, void *out_dev, const void *in_dev, size_t n);
int h2hip_fr_grand_product_dev(h2hip_ctx *ctx, void *z_dev, const void *num_dev, const void *den_dev, size_t n);
/* `segments` grand products of seg_len factors each in a handful of launches (one batched inversion, one segmented prefix product):
 * num_dev / den_dev hold the factors of all segments back to back, z_dev (HOST array of device pointers) receives seg_len + 1 values per
 * segment.  chained != 0: z[s][0] = z[s-1][seg_len], z[0][0] = 1 — the sets of the permutation argument, whose chain then needs neither a
 * host round trip nor a rescaling pass; chained == 0: every z[s][0] = 1 — the lookup arguments of one proof. */
int h2hip_fr_grand_products_dev(h2hip_ctx *ctx, void *const *z_dev, const void *num_dev, const void *den_dev, size_t segments, size_t seg_len,
                                int chained);

/* factors of ONE permutation set's grand product over rows [0, rows) (columns = the equality-enabled columns of
 * halo2-base's configs, flex_gate/mod.rs:69,124-128, range/mod.rs:104):  num[i] = prod_j (v_j[i] + beta*delta^(first_col_index+j)*omega^i
 * + gamma),  den[i] = prod_j (v_j[i] + beta*sigma_j[i] + gamma);  cols / sigmas: host arrays of ncols (<= 8) device pointers.
 * z = h2hip_fr_grand_product_dev(num, den), scaled by the previous set's last value. */
int h2hip_permutation_product_terms_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *const *cols_dev, const void *const *sigmas_dev,
                                        uint32_t ncols, uint32_t first_col_index, size_t rows, const void *beta, const void *gamma, const void *delta,
                                        const void *omega);
/* every set of the permutation argument at once: set s owns columns [s*chunk_len, (s+1)*chunk_len) of cols_dev / sigmas_dev (HOST arrays of
 * num_columns device pointers) and writes rows [s*rows, (s+1)*rows) of num_dev / den_dev — the layout h2hip_fr_grand_products_dev reads */
int h2hip_permutation_product_terms_sets_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *const *cols_dev, const void *const *sigmas_dev,
                                             uint32_t num_columns, uint32_t chunk_len, size_t rows, const void *beta, const void *gamma,
                                             const void *delta, const void *omega);
/* ... restricted to the rows [row0, row0 + rows) of the (full-length) columns: num / den hold `rows` factors per set (the sharded prover's rank
 * forms the factors of its row range only) */
int h2hip_permutation_product_terms_rows_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *const *cols_dev, const void *const *sigmas_dev,
                                             uint32_t num_columns, uint32_t chunk_len, size_t row0, size_t rows, const void *beta, const void *gamma,
                                             const void *delta, const void *omega);
/* factors of a lookup's grand product: num[i] = (a[i]+beta)(s[i]+gamma), den[i] = (a'[i]+beta)(s'[i]+gamma) */
int h2hip_lookup_product_terms_dev(h2hip_ctx *ctx, void *num_dev, void *den_dev, const void *a_dev, const void *s_dev, const void *a_perm_dev,
                                   const void *s_perm_dev, size_t rows, const void *beta, const void *gamma);

/* ---- K7: arithmetic::eval_polynomial and arithmetic::kate_division [UPSTREAM] ------------------------- */
int h2hip_fr_eval_polynomial_dev(h2hip_ctx *ctx, const void *coeffs_dev, size_t n, const void *x, void *out_host);
/* `count` evaluations in one pass: out[j] = poly_j(points[j]); coeffs_dev / lens: host arrays of device pointers and lengths, points: count
 * Montgomery elements (host), out_host: count elements — the prover's whole evaluation round with one synchronisation */
int h2hip_fr_eval_polynomial_batch_dev(h2hip_ctx *ctx, const void *const *coeffs_dev, const size_t *lens, const void *points, size_t count,
                                       void *out_host);
/* q[0..n-1) = (f(X) - f(b)) / (X - b); q_dev must not alias coeffs_dev */
int h2hip_fr_kate_division_dev(h2hip_ctx *ctx, void *q_dev, const void *coeffs_dev, size_t n, const void *b);

/* q[0..n-1) = sum_j weights[j] * (f(X) - f(points[j])) / (X - points[j]) for m <= 8 points (host arrays of m elements).  With weights[j] =
 * 1 / prod_{i != j} (points[j] - points[i]) this is (f(X) - r(X)) / prod_j (X - points[j]), r the interpolant of f on the points: the
 * quotient of one SHPLONK rotation set in a single pass over f. */
int h2hip_fr_kate_division_multi_dev(h2hip_ctx *ctx, void *q_dev, const void *coeffs_dev, size_t n, const void *points, const void *weights, uint32_t m);
/* q[0..n-1) += the same sum (SHPLONK's v-weighted sum over the rotation sets: v^i folded into the weights, no pass of its own) */
int h2hip_fr_kate_division_multi_acc_dev(h2hip_ctx *ctx, void *q_dev, const void *coeffs_dev, size_t n, const void *points, const void *weights, uint32_t m);
/* ... and for `nsets` polynomials at once (coeffs_dev: host array of device pointers, all of n coefficients; set i has set_sizes[i] points / weights, taken
 * from the flat host arrays in order): q[0..n-1) (accumulate ? += : =) the sum over the sets — SHPLONK's v-weighted sum over its rotation sets in one call */
int h2hip_fr_kate_division_sets_dev(h2hip_ctx *ctx, void *q_dev, const void *const *coeffs_dev, size_t n, const void *points, const void *weights,
                                    const uint32_t *set_sizes, size_t nsets, int accumulate);
/* The same quotient for ONE COEFFICIENT RANGE [lo, lo + n) of f (the multi-GPU prover, where a rank holds the coefficient range of its SRS
 * slice): coeffs_dev = that range; carries[j] = sum_{i >= lo + n} f_i points[j]^(i - lo - n), i.e. what the ranges above contribute (the caller
 * assembles it from the other ranks' partial evaluations; zeros for the top range); q_dev[0..n) = the quotient's coefficients lo .. lo + n - 1. */
int h2hip_fr_kate_division_range_dev(h2hip_ctx *ctx, void *q_dev, const void *coeffs_dev, size_t n, const void *points, const void *weights,
                                     const void *carries, uint32_t m);

/* ---- K6: the halo2-base custom gate's term of the quotient numerator on the extended domain:
 *      acc[i] = acc[i]*y + q[i]*(a[i] + a[i+s]*a[i+2s] - a[i+3s]), s = 2^(ext_k-k)
 *      (gate q*(a+b*c-d) at rotations 0..3, reference halo2-base/src/gates/flex_gate/mod.rs:80-91) ----- */
int h2hip_quotient_flex_gate_dev(h2hip_ctx *ctx, void *acc_dev, const void *q_dev, const void *a_dev, uint32_t ext_k, uint32_t k,
                                 const void *y);

/* EvaluationDomain::divide_by_vanishing_poly: a[i] *= 1 / ((zeta * ext_omega^i)^n - 1) over the 2^ext_k extended-domain
 * evaluations (n = 2^k); the 2^(ext_k-k) distinct inverses are computed on the device. */
int h2hip_divide_by_vanishing_poly_dev(h2hip_ctx *ctx, void *a_dev, uint32_t ext_k, uint32_t k, const void *ext_omega, const void *zeta);

/* Lookup argument's five identities, folded in upstream's order: l0*(1-z); l_last*(z^2-z);
 * active*(z(wX)(a'+beta)(s'+gamma) - z(a+beta)(s+gamma)); l0*(a'-s'); active*(a'-s')(a'-a'(w^-1 X)); active = 1-(l_last+l_blind).
 * All arrays: 2^ext_k extended-domain evaluations (halo2-base's lookups: halo2-base/src/gates/range/mod.rs:131-150). */
int h2hip_quotient_lookup_dev(h2hip_ctx *ctx, void *acc_dev, const void *z_dev, const void *a_dev, const void *s_dev, const void *a_perm_dev,
                              const void *s_perm_dev, const void *l0_dev, const void *l_last_dev, const void *l_blind_dev, uint32_t ext_k,
                              uint32_t k, const void *beta, const void *gamma, const void *y);
/* Terms of ONE permutation set, selected by `terms` and folded into acc by y in this order:
 *   H2HIP_PERM_FIRST    l0*(1 - z)                                   (upstream: first set only)
 *   H2HIP_PERM_LAST     l_last*(z^2 - z)                             (last set only)
 *   H2HIP_PERM_CHAIN    l0*(z - z_prev(w^last_rotation X))           (every set but the first; needs z_prev_dev)
 *   H2HIP_PERM_PRODUCT  active*(z(wX)*prod_j(p_j + beta*sigma_j + gamma) - z*prod_j(p_j + delta^(first_col_index+j)*beta*X + gamma))
 * with X = zeta*ext_omega^i, active = 1 - (l_last + l_blind).  Upstream's evaluate_h folds FIRST (set 0), LAST (last set),
 * CHAIN (sets 1..) and then PRODUCT (all sets) as four separate loops over the sets: issue one call per (loop, set) to
 * reproduce that order with several sets; a single-set argument is one call with FIRST|LAST|PRODUCT.
 * cols / sigmas: host arrays of ncols (<= 8) device pointers (only read for PRODUCT). */
#define H2HIP_PERM_FIRST 1u
#define H2HIP_PERM_LAST 2u
#define H2HIP_PERM_CHAIN 4u
#define H2HIP_PERM_PRODUCT 8u
int h2hip_quotient_permutation_set_dev(h2hip_ctx *ctx, void *acc_dev, const void *z_dev, const void *z_prev_dev, const void *const *cols_dev,
                                       const void *const *sigmas_dev, uint32_t ncols, uint32_t first_col_index, const void *l0_dev,
                                       const void *l_last_dev, const void *l_blind_dev, uint32_t ext_k, uint32_t k, uint32_t terms,
                                       int32_t last_rotation, const void *beta, const void *gamma, const void *delta, const void *zeta,
                                       const void *ext_omega, const void *y);
/* The same identities for all gate columns / all lookups / the whole permutation argument of a proof: every launch folds up to 64 gate
 * columns, 32 lookups or 12 (set, term) pairs into the accumulator in evaluate_h's order (acc = acc*y + term per identity, exactly the
 * values of one call per column / lookup / (loop, set)), reading and writing the accumulator once.  A wide halo2-base shape
 * (bench_pairing.config: k = 14, 211 gate columns, 80 permutation sets, 27 lookups) needs ~20 launches instead of ~400.
 * q/a, z/a/s/a_perm/s_perm, z/cols/sigmas: HOST arrays of device pointers; permutation_sets: set i owns columns [i*chunk_len, (i+1)*chunk_len)
 * of cols_dev / sigmas_dev (num_sets = ceil(num_columns / chunk_len), chunk_len <= 8), z_dev[i] is its grand product. */
int h2hip_quotient_flex_gate_batch_dev(h2hip_ctx *ctx, void *acc_dev, const void *const *q_dev, const void *const *a_dev, size_t count, uint32_t ext_k,
                                       uint32_t k, const void *y);
int h2hip_quotient_lookups_dev(h2hip_ctx *ctx, void *acc_dev, const void *const *z_dev, const void *const *a_dev, const void *const *s_dev,
                               const void *const *a_perm_dev, const void *const *s_perm_dev, size_t count, const void *l0_dev, const void *l_last_dev,
                               const void *l_blind_dev, uint32_t ext_k, uint32_t k, const void *beta, const void *gamma, const void *y);
int h2hip_quotient_permutation_sets_dev(h2hip_ctx *ctx, void *acc_dev, const void *const *z_dev, uint32_t num_sets, const void *const *cols_dev,
                                        const void *const *sigmas_dev, uint32_t num_columns, uint32_t chunk_len, const void *l0_dev,
                                        const void *l_last_dev, const void *l_blind_dev, uint32_t ext_k, uint32_t k, int32_t last_rotation,
                                        const void *beta, const void *gamma, const void *delta, const void *zeta, const void *ext_omega, const void *y);

/* permute_expression_pair of the lookup argument: a_perm = sort(a[..usable]); s_perm[i] = a_perm[i] on
 * run starts, the other rows take the unconsumed table elements (ascending) from the last repeated row backwards.
 * Rows >= usable_rows are left untouched (blinding).  H2HIP_ERR_INVALID if an input value is missing from the table. */
int h2hip_lookup_permute_dev(h2hip_ctx *ctx, const void *a_dev, const void *s_dev, size_t usable_rows, void *a_perm_dev, void *s_perm_dev);

/* The table of a lookup is a FIXED column: its sorted keys can be prepared once (keygen) and reused by every proof.
 * sorted_out_dev: h2hip_lookup_sorted_table_bytes(usable_rows) bytes of device memory. */
size_t h2hip_lookup_sorted_table_bytes(size_t usable_rows);
int h2hip_lookup_table_sort_dev(h2hip_ctx *ctx, const void *s_dev, size_t usable_rows, void *sorted_out_dev);
int h2hip_lookup_permute_presorted_dev(h2hip_ctx *ctx, const void *a_dev, const void *sorted_table_dev, size_t usable_rows, void *a_perm_dev,
                                       void *s_perm_dev);
/* `count` input columns against ONE presorted table (all range lookups of a halo2-base circuit read the same table column,
 * halo2-base/src/gates/range/mod.rs:131-150): the multiset checks of the whole batch come back in one host synchronisation.
 * a_dev / a_perm_dev / s_perm_dev: HOST arrays of `count` device pointers. */
int h2hip_lookup_permute_presorted_batch_dev(h2hip_ctx *ctx, const void *const *a_dev, const void *sorted_table_dev, size_t usable_rows,
                                             void *const *a_perm_dev, void *const *s_perm_dev, size_t count);

/* ---- K8: Poseidon permutation batches (halo2-base PoseidonState::permutation, reference
 *      halo2-base/src/poseidon/hasher/state.rs:35-83,124-160).  The caller supplies the spec its
 *      OptimizedPoseidonSpec was derived from (hasher/spec.rs:88-175): (r_f+r_p)*t round constants and
 *      the t x t MDS matrix (row major), Montgomery limbs; t in {3, 5}.  states: n x t elements, updated
 *      in place; inputs: n x num_inputs elements (num_inputs <= t-1) added to s[1..] with the padding 1
 *      after the last input when num_inputs < t-1. -------------------------------------------------------- */
int h2hip_poseidon_set_spec(h2hip_ctx *ctx, uint32_t t, uint32_t r_f, uint32_t r_p, const void *round_constants, const void *mds);
int h2hip_poseidon_permute_batch_dev(h2hip_ctx *ctx, void *states_dev, const void *inputs_dev, uint32_t num_inputs, size_t n);

/* ---- a1: plonk::create_proof for halo2-base circuits, resident on the GPU -------------------------------------------------
 * Replaces create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<_>, Challenge255<_>, _, Blake2bWrite<_, _, _>, _> as the
 * reference calls it (halo2-base/src/utils/testing.rs:32-50) for circuits configured by BaseConfig::configure
 * (halo2-base/src/gates/circuit/mod.rs:70-96: FlexGateConfig's gate q*(a + b*c - d), RangeConfig's lookups, the equality-enabled
 * columns) — the only constraint system halo2-lib builds.  First phase only (halo2-ecc's ECDSA / pairing circuits use no
 * challenge phases).  The Rust side keeps what it owns: circuit synthesis (it hands over the advice columns assign_witnesses
 * produced, halo2-base/src/gates/flex_gate/threads/single_phase.rs:273-312), the RNG (called back for every Fr::random the
 * prover draws, in upstream's order), the verifying key's transcript representation, and the proof bytes.
 * Everything between — 12 MSMs, ~11 NTTs, the lookup sort, grand products, h(X), evaluations, SHPLONK — runs on the device with
 * every polynomial resident in HBM; the host part is the Blake2b transcript and the O(#openings) bookkeeping of the multiopen. */
/* LIMITS, stated once: (1) ONE challenge phase.  The reference's params are per-phase vectors with MAX_PHASE = 3
 * (num_advice_per_phase / num_lookup_advice_per_phase, halo2-base/src/gates/flex_gate/mod.rs:28,100,132-137, gates/range/mod.rs:87-108); this
 * struct carries phase 0 only, so a circuit that uses SecondPhase / ThirdPhase columns (none of halo2-ecc's benchmark circuits does) cannot
 * be expressed and must stay on the CPU prover — the Rust shim checks `params.num_advice_per_phase.len() == 1` (ffi/rust/h2hip-sys/src/safe.rs)
 * and returns an error otherwise.  (2) Every gate column's q_enable keeps a fixed column of its own: h2hip_plonk_keygen returns
 * H2HIP_ERR_INVALID for selector activations that upstream's compress_selectors would merge (two gate columns never enabled on a common
 * row, e.g. an empty gate column).  (3) lookup_bits: the table 0..2^lookup_bits must fit 2^k - (blinding_factors + 3) rows, as in
 * RangeConfig::configure (gates/range/mod.rs:117-121). */
typedef struct {
    uint32_t k;                 /* BaseCircuitParams (gates/circuit/mod.rs:25-45), first phase */
    uint32_t num_advice;        /* num_advice_per_phase[0] */
    uint32_t num_lookup_advice; /* num_lookup_advice_per_phase[0] */
    uint32_t num_fixed;
    uint32_t num_instance;      /* num_instance_columns */
    int32_t lookup_bits;        /* < 0: None (no RangeConfig table / lookups) */
} h2hip_base_circuit_params;

/* the ConstraintSystem BaseConfig::configure derives from the params (column order = creation order in the reference's configure) */
typedef struct {
    uint32_t num_advice_total;   /* gate advice columns, then dedicated lookup-advice columns (none when num_advice == 1: range/mod.rs:93-95) */
    uint32_t num_fixed_total;    /* [table] [constants...] [q_lookup] [q_enable per gate column] */
    int32_t table_col, first_constant_col, q_lookup_col, first_q_enable_col;   /* fixed-column indices, -1 = absent */
    uint32_t num_lookups, num_perm_columns, num_perm_sets;   /* permutation columns: constants, gate advice, lookup advice, instance */
    uint32_t degree, extended_k, blinding_factors, usable_rows, quotient_pieces;
    uint32_t num_commitments, num_evals;   /* what one proof carries: proof bytes = 32 * (num_commitments + num_evals) */
} h2hip_plonk_shape;
int h2hip_plonk_shape_of(const h2hip_base_circuit_params *params, h2hip_plonk_shape *out);

typedef struct h2hip_plonk_pk h2hip_plonk_pk;
/* keygen_vk + keygen_pk [UPSTREAM], reference halo2-base/src/utils/testing.rs:224-227.  fixed_host: num_fixed_total columns of 2^k
 * Montgomery Fr (Lagrange values, as the circuit's synthesize assigned them).  copies: ncopies x 4 u32 = (column, row, column, row)
 * with `column` indexing the permutation columns — the copy constraints in the order the circuit emitted them (the cycle
 * representation depends on it).  g / g_lagrange: ParamsKZG's resident base sets (borrowed: they must outlive the key).
 * Builds sigma polynomials, all coefficient / extended-domain forms and the l_0 / l_last / l_blind cosets on the device. */
int h2hip_plonk_keygen(h2hip_ctx *ctx, const h2hip_base_circuit_params *params, const h2hip_bases *g, const h2hip_bases *g_lagrange,
                       const void *const *fixed_host, const uint32_t *copies, size_t ncopies, h2hip_plonk_pk **out);
void h2hip_plonk_pk_free(h2hip_ctx *ctx, h2hip_plonk_pk *pk);
/* VerifyingKey contents: fixed_commitments (num_fixed_total x 64 B affine) and permutation commitments (num_perm_columns x 64 B) */
int h2hip_plonk_pk_commitments(const h2hip_plonk_pk *pk, void *fixed_out, void *permutation_out);
/* vk.transcript_repr — upstream hashes the Debug rendering of the pinned key; it is computed by the Rust side and handed over */
int h2hip_plonk_pk_set_transcript_repr(h2hip_plonk_pk *pk, const void *fr);

/* ---- multi-GPU: one process per GPU (north star: "MSM ranges and independent transforms shard across the 8 GPUs of one node
 * with RCCL over xGMI").  h2hip_comm is the exchange step, inside the library so that the Rust host needs no collective library of its own:
 *   - RCCL transport: librccl.so is dlopen'ed on first use (libh2hip does not link it); rank 0 obtains a 128-byte id with
 *     h2hip_comm_rccl_unique_id and hands it to the other ranks over any channel the host has; h2hip_comm_init_rccl is collective.
 *     ncclAllGather then runs on the context's stream over device buffers (xGMI peer-to-peer, no host staging);
 *   - callback transport: allgather(user, local, bytes, all) gathers `bytes` of host memory from every rank into all[rank * bytes ...] and
 *     returns 0 — torch.distributed / gloo in the CPU tests, MPI, ...; device payloads are staged through pinned memory.
 * The reference has no counterpart (single process, rayon threads). */
typedef struct h2hip_comm h2hip_comm;
typedef int (*h2hip_allgather_fn)(void *user, const void *local, size_t bytes, void *all);
int h2hip_comm_rccl_unique_id(void *out128);
int h2hip_comm_init_rccl(h2hip_ctx *ctx, const void *unique_id128, int world, int rank, h2hip_comm **out);
int h2hip_comm_init_callback(int world, int rank, h2hip_allgather_fn allgather, void *user, h2hip_comm **out);
int h2hip_comm_info(const h2hip_comm *comm, int *world, int *rank, int *is_rccl);
void h2hip_comm_destroy(h2hip_comm *comm);
/* recv[r * bytes ...] = rank r's send[0 .. bytes): device buffers, ordered on the context's stream (RCCL: queued there, returns at once) /
 * host buffers (the 96-byte commitment partials of a round; RCCL: staged through device memory) */
int h2hip_comm_allgather_dev(h2hip_comm *comm, h2hip_ctx *ctx, const void *send_dev, size_t bytes, void *recv_dev);
int h2hip_comm_allgather_host(h2hip_comm *comm, h2hip_ctx *ctx, const void *send_host, size_t bytes, void *recv_host);
/* all-to-all over device buffers: recv[p * bytes ...] = the block rank p put at its send[me * bytes ...] (RCCL: one group of ncclSend / ncclRecv
 * per peer on the context's stream — point-to-point xGMI links, 1 / world of the matching all-gather's inbound traffic; callback transport:
 * through the all-gather callback).  The sharded prover routes the grand products' rows to the ranks that own the columns' lagrange_to_coeff. */
int h2hip_comm_alltoall_dev(h2hip_comm *comm, h2hip_ctx *ctx, const void *send_dev, size_t bytes, void *recv_dev);

/* Sharded create_proof: all ranks run the same h2hip_plonk_create_proof call on the same inputs (same circuit, same RNG stream) and emit
 * identical proof bytes.
 *   - every commitment is a partial MSM over this rank's point range [offset, offset + len) of the SRS — g_shard / g_lagrange_shard hold
 *     just that slice (1/N of the table memory) — followed by ONE all-gather of the 96-byte Jacobian partials per commitment round and
 *     an N-term sum (RCCL has no group-law reduction);
 *   - with H2HIP_SHARD_QUOTIENT, h(X)'s numerator is evaluated by cosets: the extended domain of 2^(ek-k) cosets of the original domain is
 *     dealt round-robin to the ranks (coset c to rank c mod N), each rank runs coeff_to_extended and the quotient identities for its
 *     cosets only (identities are pointwise up to rotations, which stay inside a coset), ONE device-to-device all-gather (2^ek x 32 B in
 *     total) precedes extended_to_coeff.  Ranks >= 2^(ek-k) take no part in this stage.
 *   - evaluations and SHPLONK's polynomial work run on the rank's COEFFICIENT range (= its point range): partial evaluations x^lo * sum_j c[lo + j] x^j
 *     are exchanged and summed (32 bytes per query); the divisions by (X - root) take the ranges above as a carry assembled from one exchange of
 *     partial evaluations (h2hip_fr_kate_division_range_dev);
 *   - with H2HIP_SHARD_PRODUCTS the grand products of the permutation and lookup arguments are formed by ROW range: local factors, inversion
 *     and prefix products, one 32-byte exchange per product, a scaling by everything before the range, ONE device-to-device all-gather of
 *     the rows (every rank needs the complete columns for their coefficient forms);
 *   - with H2HIP_SHARD_QUOTIENT extended_to_coeff runs by cosets too (the size-n inverse transform of a coset where the coset is, a pointwise
 *     combine on every rank after the all-gather); with H2HIP_SHARD_NTT_COLUMNS lagrange_to_coeff is dealt by column;
 *   - replicated on every rank: uploads, the lookup permutation (and lagrange_to_coeff without H2HIP_SHARD_NTT_COLUMNS).
 * Safety: at its first exchange a proof checks that all ranks agree on the shape and the RNG stream and that the point ranges tile
 * [0, 2^k) (H2HIP_ERR_INVALID otherwise); every host exchange carries a status word, and a rank that fails takes part in the next
 * exchange with an error status, so that all ranks return (H2HIP_ERR_PEER on the others) instead of waiting in a collective.
 * comm == NULL or world 1 switches sharding off.  The bases and the communicator must outlive the key (or the next set_sharding call). */
#define H2HIP_SHARD_QUOTIENT 1u
#define H2HIP_SHARD_PRODUCTS 4u /* the grand products (permutation and lookup arguments) by row range: local prefix products, one 32-byte exchange per
                                 product, one device-to-device all-gather of the columns */
#define H2HIP_SHARD_NTT_COLUMNS 8u /* lagrange_to_coeff by column: column j on rank j mod N, one device-to-device all-gather of the coefficient forms per
                                    batch (first-round columns, product columns); pays when a transform outlasts moving a column between GPUs */
#define H2HIP_SHARD_FORCE 2u /* run the sharded code path even with a one-rank communicator (tests: a 1-GPU box exercises RCCL and the coset kernels) */
int h2hip_plonk_pk_set_sharding(h2hip_plonk_pk *pk, h2hip_comm *comm, const h2hip_bases *g_shard, const h2hip_bases *g_lagrange_shard, size_t offset,
                                size_t len, uint32_t flags);
/* the host exchanges of the key's last sharded proof in order — payload bytes per rank, each sent with an 8-byte status word (count: how many;
 * sizes: the first min(count, cap)): hello, the commitment rounds, the products' totals, the go-ahead, evaluations, SHPLONK's carries ... */
int h2hip_plonk_pk_last_exchanges(const h2hip_plonk_pk *pk, size_t *sizes, size_t cap, size_t *count);
/* coefficient / extended-domain helpers of the sharded prover: f(X) -> f(s X) for `count` columns of n coefficients; the listed cosets
 * (count <= 16) of an (n << log_cosets)-point array one after the other; and the inverse with slots[c] = position of coset c in `in` */
int h2hip_fr_coset_scale_batch_dev(h2hip_ctx *ctx, void *const *outs_dev, const void *const *ins_dev, size_t count, size_t n, const void *s);
int h2hip_fr_coset_gather_dev(h2hip_ctx *ctx, void *out_dev, const void *in_dev, const uint32_t *cosets, uint32_t count, uint32_t log_cosets, size_t n);
int h2hip_fr_coset_interleave_dev(h2hip_ctx *ctx, void *out_dev, const void *in_dev, const uint32_t *slots, uint32_t log_cosets, size_t n);
/* ... and the coefficients of a polynomial of degree < n << log_cosets from its PER-COSET inverse transforms P_c (iNTT of size n of coset c's
 * evaluations, then the scaling by s_c^-t; at in[slots[c] * n ...]): out[q n + t] = zeta_n_inv^q / C * sum_c P_c[t] * rho_inv^(c q), C = 2^log_cosets,
 * rho = ext_omega^n, zeta_n_inv = zeta^-n — extended_to_coeff with the size-n transforms done where the cosets are */
int h2hip_fr_coset_combine_dev(h2hip_ctx *ctx, void *out_dev, const void *in_dev, const uint32_t *slots, uint32_t log_cosets, size_t n, const void *rho_inv,
                               const void *zeta_n_inv);

/* `Fr::random(rng)` x n into out (Montgomery limbs); called in upstream's draw order */
typedef void (*h2hip_rng_fill_fn)(void *user, void *out_fr, size_t n);
/* a ready-made h2hip_rng_fill_fn for callers that pre-draw their randomness (tests, deterministic replays, a host RNG running ahead of the
 * prover): serves `count` Montgomery Fr from `values` in order; past the end it writes zeros and sets `exhausted`, which the caller checks
 * after create_proof.  `user` = pointer to the h2hip_array_rng. */
typedef struct {
    const void *values;
    size_t count, pos;
    int exhausted;
} h2hip_array_rng;
void h2hip_array_rng_fill(void *user, void *out_fr, size_t n);
/* The `Fr::random(&mut rng)` stream of a seeded rand_chacha generator (the reference's prover RNG, halo2-base/src/utils/testing.rs:38
 * `StdRng::seed_from_u64(0)` = ChaCha12; gen_srs's `ChaCha20Rng::from_seed([0; 32])`, halo2-base/src/utils/mod.rs:441): element j of the
 * stream = keystream block j (counter mode) reduced mod r as Fr::from_u512 does.  `pos` = the number of elements drawn so far.
 * h2hip_chacha_rng_fill is a ready-made h2hip_rng_fill_fn, `user` = the h2hip_chacha_rng, that generates on ONE host thread; when it is the
 * function handed to h2hip_plonk_create_proof the prover generates its large draws (the n scalars of the random polynomial) ON THE DEVICE
 * with h2hip_rng_chacha_fill_dev instead — same values, same final `pos`, no host generation and no upload.  The block function is pinned
 * to RFC 8439's vectors; the stream layout is [UPSTREAM-RECALL]. */
typedef struct {
    uint8_t seed[32];
    int32_t rounds;   /* 12: rand 0.8's StdRng; 20: ChaCha20Rng; 8: ChaCha8Rng */
    uint64_t pos;
} h2hip_chacha_rng;
void h2hip_rng_seed_from_u64(uint64_t state, uint8_t *seed_out32);   /* rand_core's SeedableRng::seed_from_u64 (PCG32 expansion) */
void h2hip_chacha_rng_init(h2hip_chacha_rng *rng, const uint8_t *seed32, int rounds);
void h2hip_chacha_block(const uint8_t *seed32, uint64_t counter, uint64_t stream, int rounds, uint8_t *out64);
void h2hip_chacha_rng_fill(void *user, void *out_fr, size_t n);
/* out_dev[i] = stream element first_block + i, i < n (Montgomery Fr), on the context's stream */
int h2hip_rng_chacha_fill_dev(h2hip_ctx *ctx, void *out_dev, size_t n, const uint8_t *seed32, int rounds, uint64_t first_block);
#define H2HIP_PLONK_STAGES 12
const char *h2hip_plonk_stage_name(int stage);
/* advice: num_advice_total columns of 2^k Montgomery Fr (host pointers, or device pointers when advice_on_device != 0); rows >=
 * usable_rows are ignored (blinding rows).  instances: num_instance host arrays of instance_lens[i] Fr.  proof_out: capacity
 * proof_cap bytes, *proof_len receives 32 * (num_commitments + num_evals).  stage_ms (optional, H2HIP_PLONK_STAGES doubles):
 * host wall-clock per stage, synchronising the stream at stage boundaries. */
int h2hip_plonk_create_proof(h2hip_ctx *ctx, h2hip_plonk_pk *pk, const void *const *advice, int advice_on_device, const void *const *instances_host,
                             const size_t *instance_lens, h2hip_rng_fill_fn rng, void *rng_user, uint8_t *proof_out, size_t proof_cap,
                             size_t *proof_len, double *stage_ms);

/* verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<_>, Challenge255<_>, Blake2bRead<_, _, _>, SingleStrategy<_>> as the reference runs
 * it after every proof (check_proof, halo2-base/src/utils/testing.rs:64-88).  Host code (the reference verifies on the CPU too): transcript
 * replay, the quotient identity rebuilt from the openings, SHPLONK's folded opening and one pairing check.  fixed / permutation commitments:
 * the verifying key (h2hip_plonk_pk_commitments); g1: params.g[0] (64 B); g2, s_g2: 128 B each, SerdeFormat::RawBytes (x.c0, x.c1, y.c0, y.c1).
 * *accepted = 1 iff the proof verifies; a malformed proof is a rejection, not an error. */
int h2hip_plonk_verify_proof(const h2hip_base_circuit_params *params, const void *fixed_commitments, const void *permutation_commitments,
                             const void *transcript_repr, const void *g1, const void *g2, const void *s_g2, const void *const *instances_host,
                             const size_t *instance_lens, const uint8_t *proof, size_t proof_len, int *accepted);

/* The final CPU-side pairing check of the north star as an entry of its own: *is_one = 1 iff prod_i e(P_i, Q_i) == 1 in Fq12 (what
 * DualMSM::check / halo2curves' multi_miller_loop + final_exponentiation decide for KZG's two pairs).  g1_points: n x 64 B G1Affine
 * (Montgomery, identity all-zero); g2_points: n x 128 B G2Affine in SerdeFormat::RawBytes order (x.c0, x.c1, y.c0, y.c1; identity all-zero).
 * Host code.  Points off the curve / twist are H2HIP_ERR_INVALID.  Pinned by an EIP-197 vector (tests/test_external_vectors.py). */
int h2hip_pairing_check(const void *g1_points, const void *g2_points, size_t n, int *is_one);
/* BLAKE2b (RFC 7693), unkeyed, digest_len 1..64, optional 16-byte personalisation (NULL = none): the hash of the Blake2bWrite / Blake2bRead
 * transcripts (personalisation "Halo2-Transcript"), exported so that it is pinned by the RFC's vectors, not only by proofs. */
int h2hip_blake2b(const void *personal16, unsigned digest_len, const void *msg, size_t len, void *out);

/* ---- diagnostics: 254-bit Montgomery multiplier throughput (the integer roofline bench.py quotes) ------ */
int h2hip_bench_modmul(h2hip_ctx *ctx, uint32_t blocks, uint32_t iters, uint32_t chains, double *elapsed_ms, double *modmuls);
/* HBM-counter calibration probes: kind 0 = coalesced stream of table_bytes, kind 64 / 128 = lanes * per_lane random gathers of aligned
 * 64- / 128-byte entries from a table of table_bytes (the base-table access pattern of the MSM's accumulation) */
int h2hip_bench_gather(h2hip_ctx *ctx, uint32_t kind, size_t table_bytes, uint32_t lanes, uint32_t per_lane, double *elapsed_ms, double *useful_bytes);
/* the same probe on the unsaturated 9 x 29-bit representation the MSM / NTT kernels multiply in (chains: 1 or 2) */
int h2hip_bench_modmul29(h2hip_ctx *ctx, uint32_t blocks, uint32_t iters, uint32_t chains, double *elapsed_ms, double *modmuls);

#ifdef __cplusplus
}
#endif
#endif /* H2HIP_H */
