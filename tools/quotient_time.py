"""time the quotient kernels on a 2^21 extended domain (k = 19)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from bench import synthetic_scalars
ctx = H.Context(0)
k, ek = 19, 21
ne = 1 << ek
dom = HP.EvaluationDomain(ctx, 5, k)
arrs = [ctx.to_device(synthetic_scalars(ne, 10 + i)) for i in range(6)]
acc = ctx.to_device(synthetic_scalars(ne, 99))
y, x = synthetic_scalars(1, 5), synthetic_scalars(1, 6)
cols3, sig3 = (C.c_void_p * 3)(arrs[0], arrs[1], arrs[2]), (C.c_void_p * 3)(arrs[2], arrs[3], arrs[4])
def perm(terms):
    ctx._chk(ctx.lib.h2hip_quotient_permutation_set_dev(ctx.handle, acc, arrs[3], arrs[5], cols3, sig3, 3, 0, arrs[4], arrs[4], arrs[4], ek, k, terms, -6,
                                                        y.ctypes.data, x.ctypes.data, y.ctypes.data, dom.g_coset.ctypes.data, dom.extended_omega.ctypes.data, y.ctypes.data))
def lookup():
    ctx._chk(ctx.lib.h2hip_quotient_lookup_dev(ctx.handle, acc, arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], arrs[4], arrs[4], arrs[4], ek, k,
                                               y.ctypes.data, x.ctypes.data, y.ctypes.data))
def gate():
    ctx._chk(ctx.lib.h2hip_quotient_flex_gate_dev(ctx.handle, acc, arrs[1], arrs[0], ek, k, y.ctypes.data))
def vanish():
    ctx._chk(ctx.lib.h2hip_divide_by_vanishing_poly_dev(ctx.handle, acc, ek, k, dom.extended_omega.ctypes.data, dom.g_coset.ctypes.data))
for name, fn in (("perm FIRST|LAST|PRODUCT", lambda: perm(11)), ("perm PRODUCT", lambda: perm(8)), ("perm CHAIN", lambda: perm(4)), ("lookup", lookup), ("gate", gate), ("vanish", vanish)):
    fn(); ctx.sync(); ctx.timer_start()
    for _ in range(10): fn()
    print(f"{name}: {ctx.timer_stop() / 10:.4f} ms", flush=True)
