//! Safe layer: what the `halo2-axiom-hip` fork calls instead of its CPU kernels.
//! NOT COMPILED in this repository's environment (no Rust toolchain) — see ffi/rust/README.md.
use super::*;
use halo2curves::bn256::{Fr, G1Affine, G1};
use std::ffi::CStr;
use std::ptr;

#[derive(Debug)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}
fn check(rc: c_int) -> Result<(), HipError> {
    if rc == H2HIP_OK {
        Ok(())
    } else {
        let message = unsafe { CStr::from_ptr(h2hip_last_error()) }.to_string_lossy().into_owned();
        Err(HipError { code: rc, message })
    }
}

/// One per GPU and process; work is serialised on its HIP stream (create_proof is driven from one thread,
/// reference halo2-base/src/utils/testing.rs:32-50).
pub struct Backend {
    ctx: *mut h2hip_ctx,
}
unsafe impl Send for Backend {}
impl Backend {
    pub fn new(device: i32) -> Result<Self, HipError> {
        let mut ctx = ptr::null_mut();
        check(unsafe { h2hip_init(device, ptr::null_mut(), &mut ctx) })?;
        Ok(Self { ctx })
    }
}
impl Drop for Backend {
    fn drop(&mut self) {
        unsafe { h2hip_destroy(self.ctx) }
    }
}

/// Resident SRS column (`ParamsKZG::g` or `::g_lagrange`), uploaded once with precomputed window tables.
pub struct ResidentBases<'b> {
    be: &'b Backend,
    h: *mut h2hip_bases,
}
impl<'b> ResidentBases<'b> {
    pub fn upload(be: &'b Backend, points: &[G1Affine]) -> Result<Self, HipError> {
        let mut h = ptr::null_mut();
        check(unsafe { h2hip_bases_upload(be.ctx, points.as_ptr().cast(), points.len(), H2HIP_BASES_PRECOMPUTE, &mut h) })?;
        Ok(Self { be, h })
    }
    pub fn len(&self) -> usize {
        unsafe { h2hip_bases_len(self.h) }
    }
    /// `best_multiexp(coeffs, &bases[..coeffs.len()])` — the body of `ParamsKZG::commit` / `commit_lagrange`
    /// (KZG ignores the blind).  Returns the projective point like upstream.
    pub fn multiexp(&self, coeffs: &[Fr]) -> Result<G1, HipError> {
        let mut out = G1::default();
        check(unsafe {
            h2hip_msm_g1(self.be.ctx, self.h, coeffs.as_ptr().cast(), coeffs.len(), H2HIP_POINT_JACOBIAN, (&mut out as *mut G1).cast())
        })?;
        Ok(out)
    }
}
impl Drop for ResidentBases<'_> {
    fn drop(&mut self) {
        unsafe { h2hip_bases_free(self.be.ctx, self.h) }
    }
}

/// Replacement body of `arithmetic::best_fft(a, omega, log_n)`.
pub fn best_fft(be: &Backend, a: &mut [Fr], omega: Fr, log_n: u32) -> Result<(), HipError> {
    assert_eq!(a.len(), 1 << log_n);
    check(unsafe { h2hip_best_fft(be.ctx, a.as_mut_ptr().cast(), (&omega as *const Fr).cast(), log_n) })
}
/// Replacement body of `EvaluationDomain::ifft` (omega_inv and the 2^-k divisor come from the domain).
pub fn ifft(be: &Backend, a: &mut [Fr], omega_inv: Fr, log_n: u32, divisor: Fr) -> Result<(), HipError> {
    assert_eq!(a.len(), 1 << log_n);
    check(unsafe { h2hip_ifft(be.ctx, a.as_mut_ptr().cast(), (&omega_inv as *const Fr).cast(), log_n, (&divisor as *const Fr).cast()) })
}
/// Replacement body of `EvaluationDomain::coeff_to_extended`.
pub fn coeff_to_extended(be: &Backend, coeffs: &[Fr], k: u32, extended_k: u32, extended_omega: Fr, zeta: Fr) -> Result<Vec<Fr>, HipError> {
    assert_eq!(coeffs.len(), 1 << k);
    let mut out = vec![Fr::zero(); 1 << extended_k];
    check(unsafe {
        h2hip_coeff_to_extended(be.ctx, coeffs.as_ptr().cast(), k, out.as_mut_ptr().cast(), extended_k,
                                (&extended_omega as *const Fr).cast(), (&zeta as *const Fr).cast())
    })?;
    Ok(out)
}
/// Replacement body of `EvaluationDomain::extended_to_coeff` (including upstream's final truncate).
pub fn extended_to_coeff(be: &Backend, mut a: Vec<Fr>, extended_k: u32, extended_omega_inv: Fr, extended_ifft_divisor: Fr, zeta_inv: Fr,
                         n: usize, quotient_poly_degree: usize) -> Result<Vec<Fr>, HipError> {
    assert_eq!(a.len(), 1 << extended_k);
    check(unsafe {
        h2hip_extended_to_coeff(be.ctx, a.as_mut_ptr().cast(), extended_k, (&extended_omega_inv as *const Fr).cast(),
                                (&extended_ifft_divisor as *const Fr).cast(), (&zeta_inv as *const Fr).cast())
    })?;
    a.truncate(n * quotient_poly_degree);
    Ok(a)
}

// ------------------------------------------------------------------------------------------------------------------
// The rest of INTEGRATION.md §2's call-site table.  Polynomials that stay on the device between calls are `DeviceVec`s.

/// A column / polynomial resident in HBM.
pub struct DeviceVec<'b> {
    be: &'b Backend,
    ptr: *mut c_void,
    len: usize,
}
impl<'b> DeviceVec<'b> {
    pub fn zeroed(be: &'b Backend, len: usize) -> Result<Self, HipError> {
        let mut ptr = ptr::null_mut();
        check(unsafe { h2hip_malloc(be.ctx, 32 * len.max(1), &mut ptr) })?;
        let v = Self { be, ptr, len };
        v.upload(&vec![Fr::zero(); len])?;
        Ok(v)
    }
    pub fn from_slice(be: &'b Backend, values: &[Fr]) -> Result<Self, HipError> {
        let mut ptr = ptr::null_mut();
        check(unsafe { h2hip_malloc(be.ctx, 32 * values.len().max(1), &mut ptr) })?;
        let v = Self { be, ptr, len: values.len() };
        v.upload(values)?;
        Ok(v)
    }
    pub fn upload(&self, values: &[Fr]) -> Result<(), HipError> {
        assert_eq!(values.len(), self.len);
        check(unsafe { h2hip_upload(self.be.ctx, self.ptr, values.as_ptr().cast(), 32 * self.len) })
    }
    pub fn to_vec(&self) -> Result<Vec<Fr>, HipError> {
        let mut out = vec![Fr::zero(); self.len];
        check(unsafe { h2hip_download(self.be.ctx, out.as_mut_ptr().cast(), self.ptr, 32 * self.len) })?;
        Ok(out)
    }
    pub fn len(&self) -> usize {
        self.len
    }
}
impl Drop for DeviceVec<'_> {
    fn drop(&mut self) {
        unsafe { h2hip_free(self.be.ctx, self.ptr) };
    }
}
fn fr_ptr(v: &Fr) -> *const c_void {
    (v as *const Fr).cast()
}

impl<'b> ResidentBases<'b> {
    /// all commitments of one prover round (`advice.iter().map(|p| params.commit_lagrange(p))` upstream) as ONE pipelined batch
    pub fn multiexp_many(&self, columns: &[&DeviceVec<'b>]) -> Result<Vec<G1>, HipError> {
        let n = columns.first().map_or(0, |c| c.len);
        assert!(columns.iter().all(|c| c.len == n));
        let ptrs: Vec<*const c_void> = columns.iter().map(|c| c.ptr as *const c_void).collect();
        let mut out = vec![G1::default(); columns.len()];
        check(unsafe { h2hip_msm_g1_batch_dev(self.be.ctx, self.h, ptrs.as_ptr(), n, columns.len(), H2HIP_POINT_JACOBIAN, out.as_mut_ptr().cast()) })?;
        Ok(out)
    }
}

/// `batch_invert_assigned`: Assigned::Rational(num, den) columns -> values, 0^-1 := 0 (halo2-base/src/gates/flex_gate/mod.rs:677-681)
pub fn assigned_resolve<'b>(be: &'b Backend, num: &DeviceVec<'b>, den: &DeviceVec<'b>) -> Result<DeviceVec<'b>, HipError> {
    assert_eq!(num.len, den.len);
    let out = DeviceVec::zeroed(be, num.len)?;
    check(unsafe { h2hip_assigned_resolve_dev(be.ctx, out.ptr, num.ptr, den.ptr, num.len) })?;
    Ok(out)
}
/// `ff::BatchInvert` on a resident column, in place
pub fn batch_invert(be: &Backend, a: &mut DeviceVec) -> Result<(), HipError> {
    check(unsafe { h2hip_fr_batch_invert_dev(be.ctx, a.ptr, a.len) })
}
/// permutation argument: z over the usable rows of one column set, starting at `first` (the previous set's last value)
pub fn permutation_product<'b>(be: &'b Backend, columns: &[&DeviceVec<'b>], sigmas: &[&DeviceVec<'b>], first_col_index: u32, usable_rows: usize,
                               beta: Fr, gamma: Fr, delta: Fr, omega: Fr, first: Fr) -> Result<DeviceVec<'b>, HipError> {
    assert_eq!(columns.len(), sigmas.len());
    let (num, den) = (DeviceVec::zeroed(be, usable_rows)?, DeviceVec::zeroed(be, usable_rows)?);
    let c: Vec<*const c_void> = columns.iter().map(|v| v.ptr as *const c_void).collect();
    let s: Vec<*const c_void> = sigmas.iter().map(|v| v.ptr as *const c_void).collect();
    check(unsafe {
        h2hip_permutation_product_terms_dev(be.ctx, num.ptr, den.ptr, c.as_ptr(), s.as_ptr(), c.len() as u32, first_col_index, usable_rows,
                                            fr_ptr(&beta), fr_ptr(&gamma), fr_ptr(&delta), fr_ptr(&omega))
    })?;
    let z = DeviceVec::zeroed(be, usable_rows + 1)?;
    check(unsafe { h2hip_fr_grand_product_dev(be.ctx, z.ptr, num.ptr, den.ptr, usable_rows) })?;
    check(unsafe { h2hip_fr_scale_dev(be.ctx, z.ptr, fr_ptr(&first), usable_rows + 1) })?;
    Ok(z)
}
/// lookup argument: `permute_expression_pair` over the usable rows (Err = upstream's ConstraintSystemFailure)
pub fn permute_expression_pair<'b>(be: &'b Backend, input: &DeviceVec<'b>, table: &DeviceVec<'b>, usable_rows: usize)
                                   -> Result<(DeviceVec<'b>, DeviceVec<'b>), HipError> {
    let (a, s) = (DeviceVec::zeroed(be, input.len)?, DeviceVec::zeroed(be, input.len)?);
    check(unsafe { h2hip_lookup_permute_dev(be.ctx, input.ptr, table.ptr, usable_rows, a.ptr, s.ptr) })?;
    Ok((a, s))
}
/// lookup argument: the grand product z over the usable rows
pub fn lookup_product<'b>(be: &'b Backend, input: &DeviceVec<'b>, table: &DeviceVec<'b>, permuted_input: &DeviceVec<'b>,
                          permuted_table: &DeviceVec<'b>, usable_rows: usize, beta: Fr, gamma: Fr) -> Result<DeviceVec<'b>, HipError> {
    let (num, den) = (DeviceVec::zeroed(be, usable_rows)?, DeviceVec::zeroed(be, usable_rows)?);
    check(unsafe {
        h2hip_lookup_product_terms_dev(be.ctx, num.ptr, den.ptr, input.ptr, table.ptr, permuted_input.ptr, permuted_table.ptr, usable_rows,
                                       fr_ptr(&beta), fr_ptr(&gamma))
    })?;
    let z = DeviceVec::zeroed(be, usable_rows + 1)?;
    check(unsafe { h2hip_fr_grand_product_dev(be.ctx, z.ptr, num.ptr, den.ptr, usable_rows) })?;
    Ok(z)
}
/// evaluate_h, gate term: acc = acc*y + q*(a + a(wX)*a(w^2 X) - a(w^3 X)) on the extended domain
pub fn quotient_flex_gate(be: &Backend, acc: &mut DeviceVec, q: &DeviceVec, a: &DeviceVec, extended_k: u32, k: u32, y: Fr) -> Result<(), HipError> {
    check(unsafe { h2hip_quotient_flex_gate_dev(be.ctx, acc.ptr, q.ptr, a.ptr, extended_k, k, fr_ptr(&y)) })
}
/// evaluate_h, the lookup argument's five identities
#[allow(clippy::too_many_arguments)]
pub fn quotient_lookup(be: &Backend, acc: &mut DeviceVec, z: &DeviceVec, input: &DeviceVec, table: &DeviceVec, permuted_input: &DeviceVec,
                       permuted_table: &DeviceVec, l0: &DeviceVec, l_last: &DeviceVec, l_blind: &DeviceVec, extended_k: u32, k: u32, beta: Fr,
                       gamma: Fr, y: Fr) -> Result<(), HipError> {
    check(unsafe {
        h2hip_quotient_lookup_dev(be.ctx, acc.ptr, z.ptr, input.ptr, table.ptr, permuted_input.ptr, permuted_table.ptr, l0.ptr, l_last.ptr,
                                  l_blind.ptr, extended_k, k, fr_ptr(&beta), fr_ptr(&gamma), fr_ptr(&y))
    })
}
/// evaluate_h, one permutation set's terms (`terms`: mask of H2HIP_PERM_*)
#[allow(clippy::too_many_arguments)]
pub fn quotient_permutation_set(be: &Backend, acc: &mut DeviceVec, z: &DeviceVec, z_prev: Option<&DeviceVec>, columns: &[&DeviceVec],
                                sigmas: &[&DeviceVec], first_col_index: u32, l0: &DeviceVec, l_last: &DeviceVec, l_blind: &DeviceVec,
                                extended_k: u32, k: u32, terms: u32, last_rotation: i32, beta: Fr, gamma: Fr, delta: Fr, zeta: Fr,
                                extended_omega: Fr, y: Fr) -> Result<(), HipError> {
    let c: Vec<*const c_void> = columns.iter().map(|v| v.ptr as *const c_void).collect();
    let s: Vec<*const c_void> = sigmas.iter().map(|v| v.ptr as *const c_void).collect();
    check(unsafe {
        h2hip_quotient_permutation_set_dev(be.ctx, acc.ptr, z.ptr, z_prev.map_or(ptr::null(), |v| v.ptr as *const c_void), c.as_ptr(), s.as_ptr(),
                                           c.len() as u32, first_col_index, l0.ptr, l_last.ptr, l_blind.ptr, extended_k, k, terms, last_rotation,
                                           fr_ptr(&beta), fr_ptr(&gamma), fr_ptr(&delta), fr_ptr(&zeta), fr_ptr(&extended_omega), fr_ptr(&y))
    })
}
/// `EvaluationDomain::divide_by_vanishing_poly`
pub fn divide_by_vanishing_poly(be: &Backend, a: &mut DeviceVec, extended_k: u32, k: u32, extended_omega: Fr, zeta: Fr) -> Result<(), HipError> {
    check(unsafe { h2hip_divide_by_vanishing_poly_dev(be.ctx, a.ptr, extended_k, k, fr_ptr(&extended_omega), fr_ptr(&zeta)) })
}
/// `arithmetic::eval_polynomial` for a whole evaluation round: out[j] = polys[j](points[j])
pub fn eval_polynomials(be: &Backend, polys: &[&DeviceVec], points: &[Fr]) -> Result<Vec<Fr>, HipError> {
    assert_eq!(polys.len(), points.len());
    let p: Vec<*const c_void> = polys.iter().map(|v| v.ptr as *const c_void).collect();
    let l: Vec<usize> = polys.iter().map(|v| v.len).collect();
    let mut out = vec![Fr::zero(); polys.len()];
    check(unsafe { h2hip_fr_eval_polynomial_batch_dev(be.ctx, p.as_ptr(), l.as_ptr(), points.as_ptr().cast(), polys.len(), out.as_mut_ptr().cast()) })?;
    Ok(out)
}
/// `arithmetic::kate_division`: (f(X) - f(b)) / (X - b)
pub fn kate_division<'b>(be: &'b Backend, f: &DeviceVec<'b>, b: Fr) -> Result<DeviceVec<'b>, HipError> {
    let q = DeviceVec::zeroed(be, f.len.saturating_sub(1))?;
    check(unsafe { h2hip_fr_kate_division_dev(be.ctx, q.ptr, f.ptr, f.len, fr_ptr(&b)) })?;
    Ok(q)
}
/// `poly * scalar` / `poly += other * scalar` of the multiopen argument
pub fn axpy(be: &Backend, y: &mut DeviceVec, a: Fr, x: &DeviceVec) -> Result<(), HipError> {
    check(unsafe { h2hip_fr_axpy_dev(be.ctx, y.ptr, fr_ptr(&a), x.ptr, x.len.min(y.len)) })
}
/// `EvaluationDomain::lagrange_to_coeff` over many columns at once (32 per kernel launch)
pub fn lagrange_to_coeff_many(be: &Backend, columns: &mut [&mut DeviceVec], omega_inv: Fr, k: u32, ifft_divisor: Fr) -> Result<(), HipError> {
    let p: Vec<*mut c_void> = columns.iter().map(|v| v.ptr).collect();
    check(unsafe { h2hip_ifft_batch_dev(be.ctx, p.as_ptr(), p.len(), fr_ptr(&omega_inv), k, fr_ptr(&ifft_divisor)) })
}
/// `EvaluationDomain::coeff_to_extended` over many columns at once
pub fn coeff_to_extended_many<'b>(be: &'b Backend, coeffs: &[&DeviceVec<'b>], k: u32, extended_k: u32, extended_omega: Fr, zeta: Fr)
                                  -> Result<Vec<DeviceVec<'b>>, HipError> {
    let outs = coeffs.iter().map(|_| DeviceVec::zeroed(be, 1usize << extended_k)).collect::<Result<Vec<_>, _>>()?;
    let pi: Vec<*const c_void> = coeffs.iter().map(|v| v.ptr as *const c_void).collect();
    let po: Vec<*mut c_void> = outs.iter().map(|v| v.ptr).collect();
    check(unsafe { h2hip_coeff_to_extended_batch_dev(be.ctx, pi.as_ptr(), k, po.as_ptr(), extended_k, pi.len(), fr_ptr(&extended_omega), fr_ptr(&zeta)) })?;
    Ok(outs)
}
/// all grand products of one argument: `num` / `den` hold the factors of `segments` products back to back; chained = the permutation
/// argument's sets (z_i(0) = z_{i-1}(last usable row)), unchained = the lookup arguments
pub fn grand_products<'b>(be: &'b Backend, num: &DeviceVec<'b>, den: &DeviceVec<'b>, segments: usize, chained: bool) -> Result<Vec<DeviceVec<'b>>, HipError> {
    assert!(segments > 0 && num.len == den.len && num.len % segments == 0);
    let seg = num.len / segments;
    let zs = (0..segments).map(|_| DeviceVec::zeroed(be, seg + 1)).collect::<Result<Vec<_>, _>>()?;
    let pz: Vec<*mut c_void> = zs.iter().map(|v| v.ptr).collect();
    check(unsafe { h2hip_fr_grand_products_dev(be.ctx, pz.as_ptr(), num.ptr, den.ptr, segments, seg, chained as c_int) })?;
    Ok(zs)
}
/// sum_j coeffs[j] * polys[j]: a rotation set's sum_j y^j P_j(X) of the multiopen argument in one pass
pub fn linear_combination<'b>(be: &'b Backend, polys: &[&DeviceVec<'b>], coeffs: &[Fr]) -> Result<DeviceVec<'b>, HipError> {
    assert_eq!(polys.len(), coeffs.len());
    let n = polys.iter().map(|v| v.len).min().unwrap_or(0);
    let p: Vec<*const c_void> = polys.iter().map(|v| v.ptr as *const c_void).collect();
    let out = DeviceVec::zeroed(be, n)?;
    check(unsafe { h2hip_fr_linear_combination_dev(be.ctx, out.ptr, p.as_ptr(), coeffs.as_ptr().cast(), polys.len(), n) })?;
    Ok(out)
}

// ------------------------------------------------------------------------------------------------------------------
// plonk::{keygen_pk, create_proof}: the whole prover on the device.  This is what `halo2_proofs::plonk::create_proof` becomes in the fork
// for `ConcreteCircuit = BaseCircuitBuilder<Fr>` (the reference's only circuit type, halo2-base/src/utils/testing.rs:32-50): synthesis, the
// RNG and the transcript's verifying-key hash stay in Rust, everything else is one FFI call.

/// The C key stores the raw `h2hip_bases*` of `g` and `g_lagrange` and reads them on every commitment (include/h2hip.h: "the bases must
/// outlive the key"): the key therefore BORROWS both base sets for its whole life (`'g`), so that safe code cannot drop a `ResidentBases`
/// (whose `Drop` frees the tables) while a key that points into it is alive.
pub struct ProvingKeyHip<'b, 'g> {
    be: &'b Backend,
    pk: *mut h2hip_plonk_pk,
    pub params: h2hip_base_circuit_params,
    pub shape: h2hip_plonk_shape,
    _g: &'g ResidentBases<'b>,
    _g_lagrange: &'g ResidentBases<'b>,
}
fn invalid(message: String) -> HipError {
    HipError { code: H2HIP_ERR_INVALID, message }
}
impl<'b, 'g> ProvingKeyHip<'b, 'g> {
    /// `keygen_vk` + `keygen_pk`: `fixed` = the fixed columns after synthesis (table, constants, selector columns), `copies` = the copy
    /// constraints as (permutation column, row, permutation column, row) in emission order.
    /// `phases` = `BaseCircuitParams::num_advice_per_phase.len()`: libh2hip proves first-phase circuits only (include/h2hip.h, LIMITS).
    pub fn keygen(be: &'b Backend, params: h2hip_base_circuit_params, phases: usize, g: &'g ResidentBases<'b>, g_lagrange: &'g ResidentBases<'b>,
                  fixed: &[Vec<Fr>], copies: &[[u32; 4]], transcript_repr: impl FnOnce(&[G1Affine], &[G1Affine]) -> Fr) -> Result<Self, HipError> {
        if phases > 1 {
            return Err(invalid(format!("libh2hip proves first-phase circuits only; the circuit uses {phases} challenge phases")));
        }
        let mut shape = h2hip_plonk_shape::default();
        check(unsafe { h2hip_plonk_shape_of(&params, &mut shape) })?;
        // the C side reads num_fixed_total pointers and 2^k elements behind each: check the shapes here, in safe code
        let n = 1usize << params.k;
        if fixed.len() != shape.num_fixed_total as usize {
            return Err(invalid(format!("keygen: {} fixed columns, the shape has {}", fixed.len(), shape.num_fixed_total)));
        }
        if let Some(c) = fixed.iter().position(|c| c.len() != n) {
            return Err(invalid(format!("keygen: fixed column {c} has {} rows, expected 2^k = {n}", fixed[c].len())));
        }
        if g.len() < n || g_lagrange.len() < n {
            return Err(invalid(format!("keygen: the SRS holds fewer than 2^k = {n} bases")));
        }
        let cols: Vec<*const c_void> = fixed.iter().map(|c| c.as_ptr().cast()).collect();
        let mut pk = ptr::null_mut();
        check(unsafe { h2hip_plonk_keygen(be.ctx, &params, g.h, g_lagrange.h, cols.as_ptr(), copies.as_ptr().cast(), copies.len(), &mut pk) })?;
        // from here on `key`'s Drop frees the handle on every early return and on a panic inside the caller's `transcript_repr`
        let key = Self { be, pk, params, shape, _g: g, _g_lagrange: g_lagrange };
        let mut fc = vec![G1Affine::default(); shape.num_fixed_total as usize];
        let mut pc = vec![G1Affine::default(); (shape.num_perm_columns as usize).max(1)];
        check(unsafe { h2hip_plonk_pk_commitments(key.pk, fc.as_mut_ptr().cast(), pc.as_mut_ptr().cast()) })?;
        pc.truncate(shape.num_perm_columns as usize);
        let repr = transcript_repr(&fc, &pc);   // VerifyingKey::transcript_repr, computed by the Rust side from the pinned key
        check(unsafe { h2hip_plonk_pk_set_transcript_repr(key.pk, fr_ptr(&repr)) })?;
        Ok(key)
    }
    /// `create_proof(params, pk, &[circuit], &[instances], rng, &mut transcript)` after synthesis: returns what
    /// `transcript.finalize()` would.  `rng_fill` is called for every batch of `Fr::random(rng)` draws, in upstream's order.
    pub fn create_proof<R: FnMut(&mut [Fr])>(&self, advice: &[Vec<Fr>], instances: &[&[Fr]], mut rng_fill: R) -> Result<Vec<u8>, HipError> {
        unsafe extern "C" fn trampoline<R: FnMut(&mut [Fr])>(user: *mut c_void, out: *mut c_void, n: usize) {
            let f = &mut *(user as *mut R);
            f(std::slice::from_raw_parts_mut(out as *mut Fr, n));
        }
        // the C side reads num_advice_total column pointers with usable_rows elements each and num_instance instance arrays
        if advice.len() != self.shape.num_advice_total as usize {
            return Err(invalid(format!("create_proof: {} advice columns, the shape has {}", advice.len(), self.shape.num_advice_total)));
        }
        if let Some(c) = advice.iter().position(|c| c.len() < self.shape.usable_rows as usize) {
            return Err(invalid(format!("create_proof: advice column {c} has {} rows, fewer than the {} usable rows", advice[c].len(), self.shape.usable_rows)));
        }
        if instances.len() != self.params.num_instance as usize {
            return Err(invalid(format!("create_proof: {} instance columns, the circuit has {}", instances.len(), self.params.num_instance)));
        }
        if let Some(c) = instances.iter().position(|c| c.len() > self.shape.usable_rows as usize) {
            return Err(invalid(format!("create_proof: instance column {c} is longer than the usable rows")));
        }
        let adv: Vec<*const c_void> = advice.iter().map(|c| c.as_ptr().cast()).collect();
        let ins: Vec<*const c_void> = instances.iter().map(|c| c.as_ptr().cast()).collect();
        let lens: Vec<usize> = instances.iter().map(|c| c.len()).collect();
        let mut proof = vec![0u8; 32 * (self.shape.num_commitments + self.shape.num_evals) as usize];
        let mut len = 0usize;
        check(unsafe {
            h2hip_plonk_create_proof(self.be.ctx, self.pk, adv.as_ptr(), 0, ins.as_ptr(), lens.as_ptr(), Some(trampoline::<R>),
                                     (&mut rng_fill as *mut R).cast(), proof.as_mut_ptr(), proof.len(), &mut len, ptr::null_mut())
        })?;
        proof.truncate(len);
        Ok(proof)
    }
}
/// `verify_proof(params, vk, SingleStrategy::new(params), &[instances], &mut Blake2bRead::init(proof))` (check_proof,
/// halo2-base/src/utils/testing.rs:64-88): `g1` = params.get_g()[0], `g2` / `s_g2` = the verifier half of the SRS in RawBytes form.
pub fn verify_proof(params: h2hip_base_circuit_params, fixed_commitments: &[G1Affine], permutation_commitments: &[G1Affine], transcript_repr: Fr,
                    g1: G1Affine, g2: &[u8; 128], s_g2: &[u8; 128], instances: &[&[Fr]], proof: &[u8]) -> Result<bool, HipError> {
    // the C side reads num_fixed_total / num_perm_columns commitments and num_instance instance arrays: check the slices first
    let mut shape = h2hip_plonk_shape::default();
    check(unsafe { h2hip_plonk_shape_of(&params, &mut shape) })?;
    if fixed_commitments.len() != shape.num_fixed_total as usize || permutation_commitments.len() != shape.num_perm_columns as usize {
        return Err(invalid(format!("verify_proof: {} fixed / {} permutation commitments, the shape has {} / {}", fixed_commitments.len(),
                                   permutation_commitments.len(), shape.num_fixed_total, shape.num_perm_columns)));
    }
    if instances.len() != params.num_instance as usize {
        return Err(invalid(format!("verify_proof: {} instance columns, the circuit has {}", instances.len(), params.num_instance)));
    }
    let ins: Vec<*const c_void> = instances.iter().map(|c| c.as_ptr().cast()).collect();
    let lens: Vec<usize> = instances.iter().map(|c| c.len()).collect();
    let mut ok: c_int = 0;
    check(unsafe {
        h2hip_plonk_verify_proof(&params, fixed_commitments.as_ptr().cast(), permutation_commitments.as_ptr().cast(), fr_ptr(&transcript_repr),
                                 (&g1 as *const G1Affine).cast(), g2.as_ptr().cast(), s_g2.as_ptr().cast(), ins.as_ptr(), lens.as_ptr(), proof.as_ptr(),
                                 proof.len(), &mut ok)
    })?;
    Ok(ok != 0)
}

impl Drop for ProvingKeyHip<'_, '_> {
    fn drop(&mut self) {
        unsafe { h2hip_plonk_pk_free(self.be.ctx, self.pk) }
    }
}
