//! Raw bindings to `include/h2hip.h` (one declaration per exported symbol) + the safe layer in `safe.rs`.
//! NOT COMPILED in this repository's environment (no Rust toolchain) — see ffi/rust/README.md.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_uint, c_void};

pub mod safe;

#[repr(C)]
pub struct h2hip_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct h2hip_bases {
    _private: [u8; 0],
}
#[repr(C)]
pub struct h2hip_plonk_pk {
    _private: [u8; 0],
}
/// BaseCircuitParams, first phase (halo2-base/src/gates/circuit/mod.rs:25-45); `lookup_bits < 0` = None
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct h2hip_base_circuit_params {
    pub k: u32,
    pub num_advice: u32,
    pub num_lookup_advice: u32,
    pub num_fixed: u32,
    pub num_instance: u32,
    pub lookup_bits: i32,
}
/// state of libh2hip's ready-made array RNG (`h2hip_array_rng_fill` as the `h2hip_rng_fill_fn`)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct h2hip_array_rng {
    pub values: *const c_void,
    pub count: usize,
    pub pos: usize,
    pub exhausted: c_int,
}
/// state of libh2hip's seeded ChaCha generator (`h2hip_chacha_rng_fill` as the `h2hip_rng_fill_fn`): the `Fr::random` stream of
/// `StdRng` (rounds = 12) / `ChaCha20Rng` (rounds = 20); `pos` = elements drawn so far
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct h2hip_chacha_rng {
    pub seed: [u8; 32],
    pub rounds: i32,
    pub pos: u64,
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct h2hip_plonk_shape {
    pub num_advice_total: u32,
    pub num_fixed_total: u32,
    pub table_col: i32,
    pub first_constant_col: i32,
    pub q_lookup_col: i32,
    pub first_q_enable_col: i32,
    pub num_lookups: u32,
    pub num_perm_columns: u32,
    pub num_perm_sets: u32,
    pub degree: u32,
    pub extended_k: u32,
    pub blinding_factors: u32,
    pub usable_rows: u32,
    pub quotient_pieces: u32,
    pub num_commitments: u32,
    pub num_evals: u32,
}
pub type h2hip_rng_fill_fn = Option<unsafe extern "C" fn(user: *mut c_void, out_fr: *mut c_void, n: usize)>;
pub type h2hip_allgather_fn = Option<unsafe extern "C" fn(user: *mut c_void, local: *const c_void, bytes: usize, all: *mut c_void) -> c_int>;
pub const H2HIP_PLONK_STAGES: usize = 12;

#[repr(C)]
pub struct h2hip_comm {
    _private: [u8; 0],
}
pub const H2HIP_SHARD_QUOTIENT: u32 = 1;
pub const H2HIP_SHARD_FORCE: u32 = 2;
pub const H2HIP_SHARD_PRODUCTS: u32 = 4;
pub const H2HIP_SHARD_NTT_COLUMNS: u32 = 8;
pub const H2HIP_ERR_PEER: c_int = -5;
pub const H2HIP_OK: c_int = 0;
pub const H2HIP_ERR_INVALID: c_int = -1;
pub const H2HIP_ERR_HIP: c_int = -2;
pub const H2HIP_ERR_NOMEM: c_int = -3;
pub const H2HIP_ERR_NO_DEVICE: c_int = -4;
pub const H2HIP_POINT_JACOBIAN: c_int = 0;
pub const H2HIP_POINT_AFFINE: c_int = 1;
pub const H2HIP_BASES_PLAIN: u32 = 0;
pub const H2HIP_BASES_PRECOMPUTE: u32 = 1;
pub const H2HIP_PERM_FIRST: u32 = 1;
pub const H2HIP_PERM_LAST: u32 = 2;
pub const H2HIP_PERM_CHAIN: u32 = 4;
pub const H2HIP_PERM_PRODUCT: u32 = 8;

extern "C" {
    pub fn h2hip_last_error() -> *const c_char;
    pub fn h2hip_version() -> c_int;
    pub fn h2hip_device_count(count: *mut c_int) -> c_int;
    pub fn h2hip_init(device: c_int, hip_stream: *mut c_void, out: *mut *mut h2hip_ctx) -> c_int;
    pub fn h2hip_destroy(ctx: *mut h2hip_ctx);
    pub fn h2hip_sync(ctx: *mut h2hip_ctx) -> c_int;
    pub fn h2hip_set_param(ctx: *mut h2hip_ctx, name: *const c_char, value: c_int) -> c_int;
    pub fn h2hip_get_param(ctx: *mut h2hip_ctx, name: *const c_char, value: *mut c_int) -> c_int;
    pub fn h2hip_malloc(ctx: *mut h2hip_ctx, bytes: usize, dptr: *mut *mut c_void) -> c_int;
    pub fn h2hip_free(ctx: *mut h2hip_ctx, dptr: *mut c_void) -> c_int;
    pub fn h2hip_upload(ctx: *mut h2hip_ctx, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn h2hip_download(ctx: *mut h2hip_ctx, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn h2hip_host_register(ctx: *mut h2hip_ctx, host_ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn h2hip_host_unregister(ctx: *mut h2hip_ctx, host_ptr: *mut c_void) -> c_int;
    // K1 — arithmetic::best_multiexp / ParamsKZG::{commit, commit_lagrange}
    pub fn h2hip_bases_upload(ctx: *mut h2hip_ctx, g1_affine_host: *const c_void, n: usize, flags: u32, out: *mut *mut h2hip_bases) -> c_int;
    pub fn h2hip_bases_from_device(ctx: *mut h2hip_ctx, g1_affine_dev: *const c_void, n: usize, flags: u32, out: *mut *mut h2hip_bases) -> c_int;
    pub fn h2hip_bases_free(ctx: *mut h2hip_ctx, bases: *mut h2hip_bases);
    pub fn h2hip_bases_len(bases: *const h2hip_bases) -> usize;
    pub fn h2hip_bases_download(ctx: *mut h2hip_ctx, bases: *const h2hip_bases, out_host: *mut c_void) -> c_int;
    pub fn h2hip_msm_g1(ctx: *mut h2hip_ctx, bases: *const h2hip_bases, scalars_host: *const c_void, n: usize, point_format: c_int, out_host: *mut c_void) -> c_int;
    pub fn h2hip_msm_g1_dev(ctx: *mut h2hip_ctx, bases: *const h2hip_bases, scalars_dev: *const c_void, n: usize, point_format: c_int, out_host: *mut c_void) -> c_int;
    pub fn h2hip_msm_g1_batch(ctx: *mut h2hip_ctx, bases: *const h2hip_bases, scalars_host: *const *const c_void, n: usize, count: usize, point_format: c_int, out_host: *mut c_void) -> c_int;
    pub fn h2hip_msm_g1_batch_dev(ctx: *mut h2hip_ctx, bases: *const h2hip_bases, scalars_dev: *const *const c_void, n: usize, count: usize, point_format: c_int, out_host: *mut c_void) -> c_int;
    pub fn h2hip_msm_g1_multi_dev(ctx: *mut h2hip_ctx, bases_per_column: *const *const h2hip_bases, scalars_dev: *const *const c_void, n: usize, count: usize, point_format: c_int, out_host: *mut c_void) -> c_int;
    pub fn h2hip_g1_sum_jacobian_dev(ctx: *mut h2hip_ctx, points_dev: *const c_void, n: usize, point_format: c_int, out_host: *mut c_void) -> c_int;
    pub fn h2hip_g1_sum_partials_host(gathered_jacobian: *const c_void, world: usize, count: usize, point_format: c_int, out: *mut c_void) -> c_int;
    pub fn h2hip_msm_g2(ctx: *mut h2hip_ctx, g2_affine_host: *const c_void, scalars_host: *const c_void, n: usize, out_affine_host: *mut c_void) -> c_int;
    pub fn h2hip_msm_g2_dev(ctx: *mut h2hip_ctx, g2_affine_dev: *const c_void, scalars_dev: *const c_void, n: usize, out_affine_host: *mut c_void) -> c_int;
    // a2 — ParamsKZG::setup
    pub fn h2hip_g1_to_lagrange(ctx: *mut h2hip_ctx, g: *const h2hip_bases, k: u32, flags: u32, g_lagrange_out: *mut *mut h2hip_bases) -> c_int;
    pub fn h2hip_params_kzg_setup(ctx: *mut h2hip_ctx, k: u32, s_fr: *const c_void, flags: u32, g_out: *mut *mut h2hip_bases, g_lagrange_out: *mut *mut h2hip_bases) -> c_int;
    pub fn h2hip_g1_fixed_base_mul_batch_dev(ctx: *mut h2hip_ctx, base_affine: *const c_void, scalars_dev: *const c_void, n: usize, out_affine_dev: *mut c_void) -> c_int;
    pub fn h2hip_g1_validate_dev(ctx: *mut h2hip_ctx, points_dev: *const c_void, n: usize, invalid: *mut usize) -> c_int;
    pub fn h2hip_g1_decompress_batch_dev(ctx: *mut h2hip_ctx, compressed_dev: *const c_void, n: usize, out_affine_dev: *mut c_void, sign_bit: u32, inf_bit: u32) -> c_int;
    // K2/K3 — arithmetic::best_fft, EvaluationDomain::*
    pub fn h2hip_best_fft(ctx: *mut h2hip_ctx, a_host: *mut c_void, omega: *const c_void, log_n: u32) -> c_int;
    pub fn h2hip_best_fft_dev(ctx: *mut h2hip_ctx, a_dev: *mut c_void, omega: *const c_void, log_n: u32) -> c_int;
    pub fn h2hip_ifft(ctx: *mut h2hip_ctx, a_host: *mut c_void, omega_inv: *const c_void, log_n: u32, divisor: *const c_void) -> c_int;
    pub fn h2hip_ifft_dev(ctx: *mut h2hip_ctx, a_dev: *mut c_void, omega_inv: *const c_void, log_n: u32, divisor: *const c_void) -> c_int;
    pub fn h2hip_coeff_to_extended(ctx: *mut h2hip_ctx, coeffs_host: *const c_void, k: u32, out_host: *mut c_void, ext_k: u32, ext_omega: *const c_void, zeta: *const c_void) -> c_int;
    pub fn h2hip_coeff_to_extended_dev(ctx: *mut h2hip_ctx, coeffs_dev: *const c_void, k: u32, out_dev: *mut c_void, ext_k: u32, ext_omega: *const c_void, zeta: *const c_void) -> c_int;
    pub fn h2hip_extended_to_coeff(ctx: *mut h2hip_ctx, a_host: *mut c_void, ext_k: u32, ext_omega_inv: *const c_void, ext_divisor: *const c_void, zeta_inv: *const c_void) -> c_int;
    pub fn h2hip_extended_to_coeff_dev(ctx: *mut h2hip_ctx, a_dev: *mut c_void, ext_k: u32, ext_omega_inv: *const c_void, ext_divisor: *const c_void, zeta_inv: *const c_void) -> c_int;
    // K4-K8
    pub fn h2hip_fr_batch_invert_dev(ctx: *mut h2hip_ctx, a_dev: *mut c_void, n: usize) -> c_int;
    pub fn h2hip_fr_prefix_product_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, in_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_fr_grand_product_dev(ctx: *mut h2hip_ctx, z_dev: *mut c_void, num_dev: *const c_void, den_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_fr_grand_products_dev(ctx: *mut h2hip_ctx, z_dev: *const *mut c_void, num_dev: *const c_void, den_dev: *const c_void, segments: usize,
                                       seg_len: usize, chained: c_int) -> c_int;
    pub fn h2hip_fr_eval_polynomial_dev(ctx: *mut h2hip_ctx, coeffs_dev: *const c_void, n: usize, x: *const c_void, out_host: *mut c_void) -> c_int;
    pub fn h2hip_fr_kate_division_dev(ctx: *mut h2hip_ctx, q_dev: *mut c_void, coeffs_dev: *const c_void, n: usize, b: *const c_void) -> c_int;
    pub fn h2hip_fr_kate_division_multi_dev(ctx: *mut h2hip_ctx, q_dev: *mut c_void, coeffs_dev: *const c_void, n: usize, points: *const c_void, weights: *const c_void, m: u32) -> c_int;
    pub fn h2hip_fr_kate_division_multi_acc_dev(ctx: *mut h2hip_ctx, q_dev: *mut c_void, coeffs_dev: *const c_void, n: usize, points: *const c_void, weights: *const c_void, m: u32) -> c_int;
    pub fn h2hip_fr_kate_division_sets_dev(ctx: *mut h2hip_ctx, q_dev: *mut c_void, coeffs_dev: *const *const c_void, n: usize, points: *const c_void, weights: *const c_void, set_sizes: *const u32, nsets: usize, accumulate: c_int) -> c_int;
    pub fn h2hip_fr_kate_division_range_dev(ctx: *mut h2hip_ctx, q_dev: *mut c_void, coeffs_dev: *const c_void, n: usize, points: *const c_void, weights: *const c_void, carries: *const c_void, m: u32) -> c_int;
    pub fn h2hip_quotient_flex_gate_dev(ctx: *mut h2hip_ctx, acc_dev: *mut c_void, q_dev: *const c_void, a_dev: *const c_void, ext_k: u32, k: u32, y: *const c_void) -> c_int;
    pub fn h2hip_fr_add_batch_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_fr_sub_batch_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_fr_mul_batch_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_fr_mul_add_batch_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, a_dev: *const c_void, b_dev: *const c_void, c_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_poseidon_set_spec(ctx: *mut h2hip_ctx, t: u32, r_f: u32, r_p: u32, round_constants: *const c_void, mds: *const c_void) -> c_int;
    pub fn h2hip_poseidon_permute_batch_dev(ctx: *mut h2hip_ctx, states_dev: *mut c_void, inputs_dev: *const c_void, num_inputs: u32, n: usize) -> c_int;
    // polynomial linear combinations (multiopen), h(X) = numerator / (X^n - 1)
    pub fn h2hip_fr_axpy_dev(ctx: *mut h2hip_ctx, y_dev: *mut c_void, a: *const c_void, x_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_fr_scale_dev(ctx: *mut h2hip_ctx, y_dev: *mut c_void, s: *const c_void, n: usize) -> c_int;
    pub fn h2hip_divide_by_vanishing_poly_dev(ctx: *mut h2hip_ctx, a_dev: *mut c_void, ext_k: u32, k: u32, ext_omega: *const c_void, zeta: *const c_void) -> c_int;
    // lookup / permutation arguments
    pub fn h2hip_quotient_lookup_dev(ctx: *mut h2hip_ctx, acc_dev: *mut c_void, z_dev: *const c_void, a_dev: *const c_void, s_dev: *const c_void,
                                     a_perm_dev: *const c_void, s_perm_dev: *const c_void, l0_dev: *const c_void, l_last_dev: *const c_void,
                                     l_blind_dev: *const c_void, ext_k: u32, k: u32, beta: *const c_void, gamma: *const c_void, y: *const c_void) -> c_int;
    pub fn h2hip_quotient_permutation_set_dev(ctx: *mut h2hip_ctx, acc_dev: *mut c_void, z_dev: *const c_void, z_prev_dev: *const c_void,
                                              cols_dev: *const *const c_void, sigmas_dev: *const *const c_void, ncols: u32, first_col_index: u32,
                                              l0_dev: *const c_void, l_last_dev: *const c_void, l_blind_dev: *const c_void, ext_k: u32, k: u32,
                                              terms: u32, last_rotation: i32, beta: *const c_void, gamma: *const c_void, delta: *const c_void,
                                              zeta: *const c_void, ext_omega: *const c_void, y: *const c_void) -> c_int;
    pub fn h2hip_quotient_flex_gate_batch_dev(ctx: *mut h2hip_ctx, acc_dev: *mut c_void, q_dev: *const *const c_void, a_dev: *const *const c_void,
                                              count: usize, ext_k: u32, k: u32, y: *const c_void) -> c_int;
    pub fn h2hip_quotient_lookups_dev(ctx: *mut h2hip_ctx, acc_dev: *mut c_void, z_dev: *const *const c_void, a_dev: *const *const c_void,
                                      s_dev: *const *const c_void, a_perm_dev: *const *const c_void, s_perm_dev: *const *const c_void, count: usize,
                                      l0_dev: *const c_void, l_last_dev: *const c_void, l_blind_dev: *const c_void, ext_k: u32, k: u32,
                                      beta: *const c_void, gamma: *const c_void, y: *const c_void) -> c_int;
    pub fn h2hip_quotient_permutation_sets_dev(ctx: *mut h2hip_ctx, acc_dev: *mut c_void, z_dev: *const *const c_void, num_sets: u32,
                                               cols_dev: *const *const c_void, sigmas_dev: *const *const c_void, num_columns: u32, chunk_len: u32,
                                               l0_dev: *const c_void, l_last_dev: *const c_void, l_blind_dev: *const c_void, ext_k: u32, k: u32,
                                               last_rotation: i32, beta: *const c_void, gamma: *const c_void, delta: *const c_void,
                                               zeta: *const c_void, ext_omega: *const c_void, y: *const c_void) -> c_int;
    pub fn h2hip_lookup_permute_presorted_batch_dev(ctx: *mut h2hip_ctx, a_dev: *const *const c_void, sorted_table_dev: *const c_void,
                                                    usable_rows: usize, a_perm_dev: *const *mut c_void, s_perm_dev: *const *mut c_void,
                                                    count: usize) -> c_int;
    pub fn h2hip_lookup_permute_dev(ctx: *mut h2hip_ctx, a_dev: *const c_void, s_dev: *const c_void, usable_rows: usize, a_perm_dev: *mut c_void,
                                    s_perm_dev: *mut c_void) -> c_int;
    // prover steps between the big kernels
    pub fn h2hip_fr_axpby_dev(ctx: *mut h2hip_ctx, y_dev: *mut c_void, s: *const c_void, a: *const c_void, x_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_ifft_batch_dev(ctx: *mut h2hip_ctx, cols_dev: *const *mut c_void, count: usize, omega_inv: *const c_void, log_n: u32,
                                divisor: *const c_void) -> c_int;
    pub fn h2hip_coeff_to_extended_batch_dev(ctx: *mut h2hip_ctx, coeffs_dev: *const *const c_void, k: u32, outs_dev: *const *mut c_void, ext_k: u32,
                                             count: usize, ext_omega: *const c_void, zeta: *const c_void) -> c_int;
    pub fn h2hip_fr_linear_combination_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, polys_dev: *const *const c_void, coeffs_host: *const c_void,
                                           count: usize, n: usize) -> c_int;
    pub fn h2hip_fr_sub_low_dev(ctx: *mut h2hip_ctx, y_dev: *mut c_void, low_host: *const c_void, m: u32) -> c_int;
    pub fn h2hip_assigned_resolve_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, num_dev: *const c_void, den_dev: *const c_void, n: usize) -> c_int;
    pub fn h2hip_permutation_product_terms_sets_dev(ctx: *mut h2hip_ctx, num_dev: *mut c_void, den_dev: *mut c_void, cols_dev: *const *const c_void,
                                                    sigmas_dev: *const *const c_void, num_columns: u32, chunk_len: u32, rows: usize,
                                                    beta: *const c_void, gamma: *const c_void, delta: *const c_void, omega: *const c_void) -> c_int;
    pub fn h2hip_permutation_product_terms_rows_dev(ctx: *mut h2hip_ctx, num_dev: *mut c_void, den_dev: *mut c_void, cols_dev: *const *const c_void,
                                                    sigmas_dev: *const *const c_void, num_columns: u32, chunk_len: u32, row0: usize, rows: usize,
                                                    beta: *const c_void, gamma: *const c_void, delta: *const c_void, omega: *const c_void) -> c_int;
    pub fn h2hip_permutation_product_terms_dev(ctx: *mut h2hip_ctx, num_dev: *mut c_void, den_dev: *mut c_void, cols_dev: *const *const c_void,
                                               sigmas_dev: *const *const c_void, ncols: u32, first_col_index: u32, rows: usize, beta: *const c_void,
                                               gamma: *const c_void, delta: *const c_void, omega: *const c_void) -> c_int;
    pub fn h2hip_lookup_product_terms_dev(ctx: *mut h2hip_ctx, num_dev: *mut c_void, den_dev: *mut c_void, a_dev: *const c_void, s_dev: *const c_void,
                                          a_perm_dev: *const c_void, s_perm_dev: *const c_void, rows: usize, beta: *const c_void,
                                          gamma: *const c_void) -> c_int;
    pub fn h2hip_fr_eval_polynomial_batch_dev(ctx: *mut h2hip_ctx, coeffs_dev: *const *const c_void, lens: *const usize, points: *const c_void,
                                              count: usize, out_host: *mut c_void) -> c_int;
    // a1 — plonk::{keygen_vk, keygen_pk, create_proof} for BaseConfig circuits, resident on the GPU
    pub fn h2hip_plonk_shape_of(params: *const h2hip_base_circuit_params, out: *mut h2hip_plonk_shape) -> c_int;
    pub fn h2hip_plonk_keygen(ctx: *mut h2hip_ctx, params: *const h2hip_base_circuit_params, g: *const h2hip_bases, g_lagrange: *const h2hip_bases,
                              fixed_host: *const *const c_void, copies: *const u32, ncopies: usize, out: *mut *mut h2hip_plonk_pk) -> c_int;
    pub fn h2hip_plonk_pk_free(ctx: *mut h2hip_ctx, pk: *mut h2hip_plonk_pk);
    pub fn h2hip_plonk_pk_commitments(pk: *const h2hip_plonk_pk, fixed_out: *mut c_void, permutation_out: *mut c_void) -> c_int;
    pub fn h2hip_plonk_pk_set_transcript_repr(pk: *mut h2hip_plonk_pk, fr: *const c_void) -> c_int;
    pub fn h2hip_plonk_pk_set_sharding(pk: *mut h2hip_plonk_pk, comm: *mut h2hip_comm, g_shard: *const h2hip_bases, g_lagrange_shard: *const h2hip_bases,
                                       offset: usize, len: usize, flags: u32) -> c_int;
    pub fn h2hip_plonk_pk_last_exchanges(pk: *const h2hip_plonk_pk, sizes: *mut usize, cap: usize, count: *mut usize) -> c_int;
    pub fn h2hip_comm_rccl_unique_id(out128: *mut c_void) -> c_int;
    pub fn h2hip_comm_init_rccl(ctx: *mut h2hip_ctx, unique_id128: *const c_void, world: c_int, rank: c_int, out: *mut *mut h2hip_comm) -> c_int;
    pub fn h2hip_comm_init_callback(world: c_int, rank: c_int, allgather: h2hip_allgather_fn, user: *mut c_void, out: *mut *mut h2hip_comm) -> c_int;
    pub fn h2hip_comm_info(comm: *const h2hip_comm, world: *mut c_int, rank: *mut c_int, is_rccl: *mut c_int) -> c_int;
    pub fn h2hip_comm_destroy(comm: *mut h2hip_comm);
    pub fn h2hip_comm_allgather_dev(comm: *mut h2hip_comm, ctx: *mut h2hip_ctx, send_dev: *const c_void, bytes: usize, recv_dev: *mut c_void) -> c_int;
    pub fn h2hip_comm_alltoall_dev(comm: *mut h2hip_comm, ctx: *mut h2hip_ctx, send_dev: *const c_void, bytes: usize, recv_dev: *mut c_void) -> c_int;
    pub fn h2hip_comm_allgather_host(comm: *mut h2hip_comm, ctx: *mut h2hip_ctx, send_host: *const c_void, bytes: usize, recv_host: *mut c_void) -> c_int;
    pub fn h2hip_fr_coset_scale_batch_dev(ctx: *mut h2hip_ctx, outs_dev: *const *mut c_void, ins_dev: *const *const c_void, count: usize, n: usize,
                                          s: *const c_void) -> c_int;
    pub fn h2hip_fr_coset_gather_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, in_dev: *const c_void, cosets: *const u32, count: u32, log_cosets: u32,
                                     n: usize) -> c_int;
    pub fn h2hip_fr_coset_interleave_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, in_dev: *const c_void, slots: *const u32, log_cosets: u32, n: usize) -> c_int;
    pub fn h2hip_fr_coset_combine_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, in_dev: *const c_void, slots: *const u32, log_cosets: u32, n: usize, rho_inv: *const c_void, zeta_n_inv: *const c_void) -> c_int;
    pub fn h2hip_array_rng_fill(user: *mut c_void, out_fr: *mut c_void, n: usize);
    pub fn h2hip_rng_seed_from_u64(state: u64, seed_out32: *mut u8);
    pub fn h2hip_chacha_rng_init(rng: *mut h2hip_chacha_rng, seed32: *const u8, rounds: c_int);
    pub fn h2hip_chacha_block(seed32: *const u8, counter: u64, stream: u64, rounds: c_int, out64: *mut u8);
    pub fn h2hip_chacha_rng_fill(user: *mut c_void, out_fr: *mut c_void, n: usize);
    pub fn h2hip_rng_chacha_fill_dev(ctx: *mut h2hip_ctx, out_dev: *mut c_void, n: usize, seed32: *const u8, rounds: c_int, first_block: u64) -> c_int;
    pub fn h2hip_plonk_stage_name(stage: c_int) -> *const c_char;
    pub fn h2hip_plonk_create_proof(ctx: *mut h2hip_ctx, pk: *mut h2hip_plonk_pk, advice: *const *const c_void, advice_on_device: c_int,
                                    instances_host: *const *const c_void, instance_lens: *const usize, rng: h2hip_rng_fill_fn, rng_user: *mut c_void,
                                    proof_out: *mut u8, proof_cap: usize, proof_len: *mut usize, stage_ms: *mut f64) -> c_int;
    pub fn h2hip_lookup_sorted_table_bytes(usable_rows: usize) -> usize;
    pub fn h2hip_lookup_table_sort_dev(ctx: *mut h2hip_ctx, s_dev: *const c_void, usable_rows: usize, sorted_out_dev: *mut c_void) -> c_int;
    pub fn h2hip_lookup_permute_presorted_dev(ctx: *mut h2hip_ctx, a_dev: *const c_void, sorted_table_dev: *const c_void, usable_rows: usize,
                                              a_perm_dev: *mut c_void, s_perm_dev: *mut c_void) -> c_int;
    pub fn h2hip_plonk_verify_proof(params: *const h2hip_base_circuit_params, fixed_commitments: *const c_void, permutation_commitments: *const c_void,
                                    transcript_repr: *const c_void, g1: *const c_void, g2: *const c_void, s_g2: *const c_void,
                                    instances_host: *const *const c_void, instance_lens: *const usize, proof: *const u8, proof_len: usize,
                                    accepted: *mut c_int) -> c_int;
    pub fn h2hip_pairing_check(g1_points: *const c_void, g2_points: *const c_void, n: usize, is_one: *mut c_int) -> c_int;
    pub fn h2hip_blake2b(personal16: *const c_void, digest_len: c_uint, msg: *const c_void, len: usize, out: *mut c_void) -> c_int;
    // timing / diagnostics
    pub fn h2hip_profile_enable(ctx: *mut h2hip_ctx, on: c_int) -> c_int;
    pub fn h2hip_profile_reset(ctx: *mut h2hip_ctx) -> c_int;
    pub fn h2hip_profile_filter(ctx: *mut h2hip_ctx, prefix: *const c_char) -> c_int;
    pub fn h2hip_profile_get_busy(ctx: *mut h2hip_ctx, prefix: *const c_char, busy_ms: *mut f64) -> c_int;
    pub fn h2hip_profile_get(ctx: *mut h2hip_ctx, prefix: *const c_char, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn h2hip_profile_dump(ctx: *mut h2hip_ctx, out: *mut c_char, cap: usize, needed: *mut usize) -> c_int;
    pub fn h2hip_timer_start(ctx: *mut h2hip_ctx) -> c_int;
    pub fn h2hip_timer_stop(ctx: *mut h2hip_ctx, elapsed_ms: *mut f64) -> c_int;
    pub fn h2hip_bench_modmul(ctx: *mut h2hip_ctx, blocks: u32, iters: u32, chains: u32, elapsed_ms: *mut f64, modmuls: *mut f64) -> c_int;
    pub fn h2hip_bench_gather(ctx: *mut h2hip_ctx, kind: u32, table_bytes: usize, lanes: u32, per_lane: u32, elapsed_ms: *mut f64, useful_bytes: *mut f64) -> c_int;
    pub fn h2hip_bench_modmul29(ctx: *mut h2hip_ctx, blocks: u32, iters: u32, chains: u32, elapsed_ms: *mut f64, modmuls: *mut f64) -> c_int;
}
