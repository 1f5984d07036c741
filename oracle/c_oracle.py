"""
ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes view of oracle/libh2oracle.so (the plain-C restatement in
h2_oracle.c).  All arrays are numpy uint64: field elements (n,4), affine points (n,8), Montgomery form.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False, native: bool = False) -> str:
    """make -C oracle; native=True builds a -march=native copy (bench cpu_baseline on the box's own cores)."""
    name = "libh2oracle_native.so" if native else "libh2oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "h2_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        if native:
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-pthread", "-shared", "-o", path, src, "-lm", "-lpthread"])
        else:
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return path


def lib(native: bool = False):
    global _LIB
    if native:
        return C.CDLL(build(native=True))
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _fe(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def _pt(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 8)


def fr_mul(a, b):
    a, b = _fe(a), _fe(b)
    out = np.empty_like(a)
    lib().orc_fr_mul_batch(_p(out), _p(a), _p(b), C.c_size_t(len(a)))
    return out


def fr_add(a, b):
    a, b = _fe(a), _fe(b)
    out = np.empty_like(a)
    lib().orc_fr_add_batch(_p(out), _p(a), _p(b), C.c_size_t(len(a)))
    return out


def fr_sub(a, b):
    a, b = _fe(a), _fe(b)
    out = np.empty_like(a)
    lib().orc_fr_sub_batch(_p(out), _p(a), _p(b), C.c_size_t(len(a)))
    return out


def fq_mul(a, b):
    a, b = _fe(a), _fe(b)
    out = np.empty_like(a)
    lib().orc_fq_mul_batch(_p(out), _p(a), _p(b), C.c_size_t(len(a)))
    return out


def fr_batch_invert(a):
    a = _fe(a).copy()
    lib().orc_fr_batch_invert(_p(a), C.c_size_t(len(a)))
    return a


def fr_grand_product(num, den):
    num, den = _fe(num), _fe(den)
    z = np.empty((len(num) + 1, 4), dtype=np.uint64)
    lib().orc_fr_grand_product(_p(z), _p(num), _p(den), C.c_size_t(len(num)))
    return z


def fr_eval_polynomial(coeffs, x):
    coeffs, x = _fe(coeffs), _fe(x)
    out = np.empty((1, 4), dtype=np.uint64)
    lib().orc_fr_eval_polynomial(_p(out), _p(coeffs), C.c_size_t(len(coeffs)), _p(x))
    return out


def fr_kate_division(coeffs, b):
    coeffs, b = _fe(coeffs), _fe(b)
    q = np.empty((len(coeffs) - 1, 4), dtype=np.uint64)
    lib().orc_fr_kate_division(_p(q), _p(coeffs), C.c_size_t(len(coeffs)), _p(b))
    return q


def best_fft(a, log_n, omega, threads=1, l=None):
    a = _fe(a).copy()
    assert len(a) == 1 << log_n
    (l or lib()).orc_best_fft(_p(a), C.c_uint32(log_n), _p(_fe(omega)), C.c_int(threads))
    return a


def ifft(a, log_n, omega, threads=1):
    a = _fe(a).copy()
    lib().orc_ifft(_p(a), C.c_uint32(log_n), _p(_fe(omega)), C.c_int(threads))
    return a


def coeff_to_extended(a, k, ext_k, ext_omega, zeta, threads=1):
    a = _fe(a)
    out = np.empty((1 << ext_k, 4), dtype=np.uint64)
    lib().orc_coeff_to_extended(_p(out), _p(a), C.c_uint32(k), C.c_uint32(ext_k), _p(_fe(ext_omega)), _p(_fe(zeta)), C.c_int(threads))
    return out


def extended_to_coeff(a, ext_k, ext_omega, zeta, threads=1):
    a = _fe(a).copy()
    lib().orc_extended_to_coeff(_p(a), C.c_uint32(ext_k), _p(_fe(ext_omega)), _p(_fe(zeta)), C.c_int(threads))
    return a


def g1_add(a, b):
    out = np.empty((1, 8), dtype=np.uint64)
    lib().orc_g1_add(_p(out), _p(_pt(a)), _p(_pt(b)))
    return out


def g1_mul(p, scalar_mont):
    out = np.empty((1, 8), dtype=np.uint64)
    lib().orc_g1_mul(_p(out), _p(_pt(p)), _p(_fe(scalar_mont)))
    return out


def g1_fixed_base_batch(base, scalars_mont, threads=1):
    """scalars[i] * base for every i (8-bit fixed-base windows + batch normalisation; h2_oracle.c:orc_g1_fixed_base_batch)"""
    s = _fe(scalars_mont)
    out = np.empty((len(s), 8), dtype=np.uint64)
    lib().orc_g1_fixed_base_batch(_p(out), _p(_pt(base)), _p(s), C.c_size_t(len(s)), C.c_int(threads))
    return out


def g1_is_on_curve(p) -> bool:
    return bool(lib().orc_g1_is_on_curve(_p(_pt(p))))


def known_dlog_bases(n, k0_mont, d_mont):
    out = np.empty((n, 8), dtype=np.uint64)
    lib().orc_known_dlog_bases(_p(out), C.c_size_t(n), _p(_fe(k0_mont)), _p(_fe(d_mont)))
    return out


def best_multiexp(scalars, bases, threads=1, l=None):
    scalars, bases = _fe(scalars), _pt(bases)
    assert len(scalars) == len(bases)
    out = np.empty((1, 8), dtype=np.uint64)
    (l or lib()).orc_best_multiexp(_p(out), _p(scalars), _p(bases), C.c_size_t(len(scalars)), C.c_int(threads))
    return out


# ------------------------------------------------------------------ create_proof vector steps (see h2_oracle.c, last section)
def _opt(x):
    return None if x is None else _p(_fe(x))


def fr_lincomb(a, sa=None, b=None, sb=None, c=None, threads=1):
    """a*sa (+ b*sb) (+ c); sa, sb, c: single elements or None"""
    a = _fe(a)
    out = np.empty_like(a)
    keep = [_fe(v) if v is not None else None for v in (sa, b, sb, c)]
    lib().orc_fr_lincomb(_p(out), _p(a), *[None if v is None else _p(v) for v in keep], C.c_size_t(len(a)), C.c_int(threads))
    return out


def fr_mul_mt(a, b, threads=1):
    a, b = _fe(a), _fe(b)
    out = np.empty_like(a)
    lib().orc_fr_mul_batch_mt(_p(out), _p(a), _p(b), C.c_size_t(len(a)), C.c_int(threads))
    return out


def fr_geom(start, ratio, n, threads=1):
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_fr_geom(_p(out), _p(_fe(start)), _p(_fe(ratio)), C.c_size_t(n), C.c_int(threads))
    return out


def fr_batch_invert_mt(a, threads=1):
    a = _fe(a).copy()
    lib().orc_fr_batch_invert_mt(_p(a), C.c_size_t(len(a)), C.c_int(threads))
    return a


def fr_running_product(first, vals):
    vals = _fe(vals)
    z = np.empty((len(vals) + 1, 4), dtype=np.uint64)
    lib().orc_fr_running_product(_p(z), _p(_fe(first)), _p(vals), C.c_size_t(len(vals)))
    return z


def fr_axpy(y, a, x, threads=1):
    """y + a*x (a: one element); returns a new array"""
    y, x = _fe(y).copy(), _fe(x)
    assert len(x) <= len(y)
    lib().orc_fr_axpy(_p(y), _p(_fe(a)), _p(x), C.c_size_t(len(x)), C.c_int(threads))
    return y


def fr_scale(y, s, threads=1):
    y = _fe(y).copy()
    lib().orc_fr_scale(_p(y), _p(_fe(s)), C.c_size_t(len(y)), C.c_int(threads))
    return y


def permute_expression_pair(a, s, usable):
    a, s = _fe(a), _fe(s)
    ap, sp = np.empty((usable, 4), dtype=np.uint64), np.empty((usable, 4), dtype=np.uint64)
    if lib().orc_permute_expression_pair(_p(ap), _p(sp), _p(a), _p(s), C.c_size_t(usable)) != 0:
        raise ValueError("input value missing from the table")
    return ap, sp


def _pp(arrs):
    keep = [_fe(a) for a in arrs]
    return keep, (C.c_void_p * len(keep))(*[a.ctypes.data for a in keep])


def quotient_gate(acc, q, a, y, step, threads=1):
    acc = _fe(acc)   # in place
    lib().orc_quotient_gate(_p(acc), _p(_fe(q)), _p(_fe(a)), _p(_fe(y)), C.c_size_t(len(acc)), C.c_size_t(step), C.c_int(threads))
    return acc


def quotient_permutation(acc, z_sets, cols, sigmas, chunk_len, l0, l_last, l_active, step, last_rotation, beta, gamma, y, delta, zeta, ext_omega,
                         threads=1):
    acc = _fe(acc)
    kz, pz = _pp(z_sets)
    kc, pc = _pp(cols)
    ks, ps = _pp(sigmas)
    lib().orc_quotient_permutation(_p(acc), pz, C.c_uint32(len(kz)), pc, ps, C.c_uint32(len(kc)), C.c_uint32(chunk_len), _p(_fe(l0)), _p(_fe(l_last)),
                                   _p(_fe(l_active)), C.c_size_t(len(acc)), C.c_size_t(step), C.c_int32(last_rotation), _p(_fe(beta)), _p(_fe(gamma)),
                                   _p(_fe(y)), _p(_fe(delta)), _p(_fe(zeta)), _p(_fe(ext_omega)), C.c_int(threads))
    return acc


def quotient_lookup(acc, z, inp, tab, ap, sp, l0, l_last, l_active, step, beta, gamma, y, threads=1):
    acc = _fe(acc)
    lib().orc_quotient_lookup(_p(acc), _p(_fe(z)), _p(_fe(inp)), _p(_fe(tab)), _p(_fe(ap)), _p(_fe(sp)), _p(_fe(l0)), _p(_fe(l_last)),
                              _p(_fe(l_active)), C.c_size_t(len(acc)), C.c_size_t(step), _p(_fe(beta)), _p(_fe(gamma)), _p(_fe(y)), C.c_int(threads))
    return acc


def divide_by_vanishing(a, ext_k, k, ext_omega, zeta, threads=1):
    a = _fe(a)
    lib().orc_divide_by_vanishing(_p(a), C.c_uint32(ext_k), C.c_uint32(k), _p(_fe(ext_omega)), _p(_fe(zeta)), C.c_int(threads))
    return a
