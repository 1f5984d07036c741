"""A/B of ntt_wave_local: forward NTT at 2^19..2^23 and the prover's coset extension 2^19 -> 2^21"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from bench import synthetic_scalars
from halo2_lib_amd import halo2_proofs as HP

ctx = H.Context(0)
for log_n in (19, 21, 22, 23):
    n = 1 << log_n
    dom = HP.EvaluationDomain(ctx, 2, log_n)
    d = ctx.to_device(synthetic_scalars(n, 3))
    for wl in (0, 1, 0, 1):
        ctx.set_param("ntt_wave_local", wl)
        ctx.best_fft_dev(d, dom.omega, log_n)
        ctx.timer_start()
        for _ in range(20): ctx.best_fft_dev(d, dom.omega, log_n)
        print("NTT 2^%d wave_local=%d: %.4f ms" % (log_n, wl, ctx.timer_stop() / 20), flush=True)
    ctx.free(d)
k, ek = 19, 21
dom = HP.EvaluationDomain(ctx, 5, k)
src, dst = ctx.to_device(synthetic_scalars(1 << k, 4)), ctx.malloc(32 << ek)
for wl in (0, 1, 0, 1):
    ctx.set_param("ntt_wave_local", wl)
    ctx.coeff_to_extended_dev(src, k, dst, ek, dom.extended_omega, dom.g_coset)
    ctx.timer_start()
    for _ in range(20): ctx.coeff_to_extended_dev(src, k, dst, ek, dom.extended_omega, dom.g_coset)
    print("coeff_to_extended 2^19 -> 2^21 wave_local=%d: %.4f ms" % (wl, ctx.timer_stop() / 20), flush=True)
