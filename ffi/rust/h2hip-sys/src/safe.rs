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
