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
