"""The host-side mirror of the halo2_proofs names (halo2-lib_amd/halo2_proofs.py): emulated kernels on CPU,
the real library on the GPU (same test bodies)."""
import os

import numpy as np
import pytest

import halo2_lib_amd as H
from halo2_lib_amd import halo2_proofs as HP
from oracle import bn254 as O
from oracle import c_oracle as CO
from tests.util import R, fr, rand_fr


def _run_all(ctx, k, tmp_path):
    n = 1 << k
    # EvaluationDomain for the k=19-ECDSA-shaped circuit: degree 5 -> quotient degree 4 -> extended_k = k + 2
    dom = HP.EvaluationDomain(ctx, 5, k)
    assert dom.extended_k == k + 2 and dom.quotient_poly_degree == 4
    assert HP.EvaluationDomain(ctx, 4, k).extended_k == k + 2   # degree 4 -> 3n -> also k + 2
    vals = rand_fr(n, 1)
    coeffs = dom.lagrange_to_coeff(vals)
    assert np.array_equal(coeffs, CO.ifft(vals, k, dom.omega, threads=4))
    ext = dom.coeff_to_extended(coeffs)
    assert np.array_equal(ext, CO.coeff_to_extended(coeffs, k, k + 2, dom.extended_omega, dom.g_coset, threads=4))
    back = dom.extended_to_coeff(ext)
    assert len(back) == 4 * n and np.array_equal(back[:n], coeffs) and not back[n:].any()
    # arithmetic::*
    w = dom.omega
    assert np.array_equal(HP.best_fft(ctx, coeffs, w, k), vals)
    x = rand_fr(1, 5)
    assert np.array_equal(HP.eval_polynomial(ctx, coeffs, x), CO.fr_eval_polynomial(coeffs, x))
    assert np.array_equal(HP.kate_division(ctx, coeffs, x), CO.fr_kate_division(coeffs, x))
    with pytest.raises(AssertionError):
        HP.best_fft(ctx, coeffs[:-1], w, k)
    pm = HP.ParamsKZG.setup(ctx, k, 0xC0FFEE, precompute=True)
    many = pm.commit_many([vals, coeffs, vals], lagrange=True, point_format=H.POINT_AFFINE)
    assert np.array_equal(many[0:1], pm.commit_lagrange(vals, H.POINT_AFFINE)) and np.array_equal(many[1:2], pm.commit_lagrange(coeffs, H.POINT_AFFINE))
    assert np.array_equal(many[2], many[0])
    pm.free()
    # ParamsKZG: commit_lagrange(values) == commit(coeffs) == p(s)*G  (closed form for SRS-shaped bases)
    s = 0x1234567890ABCDEF1234567890ABCDEF
    for pre in (False, True):
        params = HP.ParamsKZG.setup(ctx, k, s, precompute=pre)
        c1 = params.commit_lagrange(vals, H.POINT_AFFINE)
        c2 = params.commit(coeffs, H.POINT_AFFINE)
        assert np.array_equal(c1, c2)
        ps = O.eval_polynomial(O.limbs_to_ints(coeffs, R), s)
        assert O.limbs_to_points(c1) == [O.g1_mul(O.G1_GEN, ps)]
        # best_multiexp mirror + its length assertion
        assert np.array_equal(HP.best_multiexp(ctx, coeffs, params.g, H.POINT_AFFINE), c2)
        with pytest.raises(AssertionError):
            HP.best_multiexp(ctx, coeffs[:-1], params.g)
        # shorter polynomial commits against a prefix of the bases
        short = params.commit(coeffs[: n // 2], H.POINT_AFFINE)
        assert np.array_equal(short, CO.best_multiexp(coeffs[: n // 2], ctx.bases_download(params.g)[: n // 2], threads=4))
        if not pre:
            path = os.path.join(tmp_path, "kzg_bn254_%d.srs" % k)
            params.write(path)
            again = HP.ParamsKZG.read(ctx, path, precompute=False)
            assert again.k == k
            assert np.array_equal(ctx.bases_download(again.g_lagrange), ctx.bases_download(params.g_lagrange))
            assert np.array_equal(again.commit_lagrange(vals, H.POINT_AFFINE), c1)
            again.free()
        params.free()


def test_host_mirror_emulated(tmp_path):
    from tests.emu_util import emu_context

    ctx = emu_context()
    _run_all(ctx, 7, str(tmp_path))
    ctx.close()


@pytest.mark.gpu
def test_host_mirror_gpu(tmp_path):
    ctx = H.Context(0)
    _run_all(ctx, 14, str(tmp_path))
    ctx.close()


@pytest.mark.gpu
def test_params_kzg_setup_k16_matches_oracle():
    ctx = H.Context(0)
    k, s = 16, 0xDEADBEEFCAFEBABE1234
    params = HP.ParamsKZG.setup(ctx, k, s, precompute=False)
    g = ctx.bases_download(params.g)
    n = 1 << k
    # g[i] = s^i G: spot-check against the oracle's scalar multiplication, and every point is on the curve
    for i in (0, 1, 2, 777, n - 1):
        assert O.limbs_to_points(g[i:i + 1]) == [O.g1_mul(O.G1_GEN, pow(s, i, R))]
    gl = ctx.bases_download(params.g_lagrange)
    w = O.omega_for(k)
    for i in (0, 1, 4097, n - 1):
        li = (pow(s, n, R) - 1) * pow(n, -1, R) * pow(w, i, R) * pow(s - pow(w, i, R), -1, R) % R
        assert O.limbs_to_points(gl[i:i + 1]) == [O.g1_mul(O.G1_GEN, li)]
    params.free()
    ctx.close()


@pytest.mark.gpu
def test_commit_identity_at_k21():
    """BASELINE config #5 size (k = 21): SRS generation, precomputed tables, iNTT and two 2^21-point MSMs tied together by
    the size-independent identity commit_lagrange(values) == commit(lagrange_to_coeff(values)); the Lagrange commitment
    is additionally checked against the closed form sum_i v_i L_i(s) * G."""
    ctx = H.Context(0)
    k, s = 21, 0xABCDEF0123456789ABCDEF
    n = 1 << k
    params = HP.ParamsKZG.setup(ctx, k, s, precompute=True)
    dom = HP.EvaluationDomain(ctx, 4, k)
    vals = rand_fr(n, 3)
    vals[5:100000] = 0   # a sparse stretch, like a padded advice column
    c1 = params.commit_lagrange(vals, H.POINT_AFFINE)
    coeffs = dom.lagrange_to_coeff(vals)
    c2 = params.commit(coeffs, H.POINT_AFFINE)
    assert np.array_equal(c1, c2) and c1.any()
    # closed form p(s)*G with p(s) = sum_i v_i * L_i(s), evaluated on the host from the (non-zero) values
    w, sn = O.omega_for(k), pow(s, n, R)
    mult = (sn - 1) * pow(n, -1, R) % R
    vi = O.limbs_to_ints(vals, R)
    wi, acc = 1, 0
    dens, nums = [], []
    for i in range(n):
        if vi[i]:
            nums.append(vi[i] * wi % R)
            dens.append((s - wi) % R)
        wi = wi * w % R
    # batch inversion on the host (python ints)
    pref, run = [], 1
    for d in dens:
        pref.append(run)
        run = run * d % R
    inv = pow(run, -1, R)
    for j in range(len(dens) - 1, -1, -1):
        acc = (acc + nums[j] * (inv * pref[j] % R)) % R
        inv = inv * dens[j] % R
    assert O.limbs_to_points(c1) == [O.g1_mul(O.G1_GEN, mult * acc % R)]
    params.free()
    ctx.close()


def _g_to_lagrange_check(ctx, k):
    """g_to_lagrange(g) must reproduce, bit for bit, the Lagrange basis the setup derives from the known toxic waste
    (L_i(s)*G by fixed-base multiplications): two independent routes to the same affine points."""
    params = HP.ParamsKZG.setup(ctx, k, 0x5EED5EED5EED0001 + k, precompute=False)
    derived = HP.ParamsKZG.from_parts(ctx, k, params.g, None, precompute=False)
    assert np.array_equal(ctx.bases_download(derived.g_lagrange), ctx.bases_download(params.g_lagrange))
    vals = rand_fr(1 << k, 3)
    assert np.array_equal(derived.commit_lagrange(vals, H.POINT_AFFINE), params.commit_lagrange(vals, H.POINT_AFFINE))
    derived.g_lagrange.free()
    params.free()


def test_g_to_lagrange_emulated():
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        _g_to_lagrange_check(ctx, 3)
        _g_to_lagrange_check(ctx, 5)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 6, 12])
def test_g_to_lagrange_gpu(k):
    ctx = H.Context()
    try:
        _g_to_lagrange_check(ctx, k)
    finally:
        ctx.close()


def _srs_files(ctx, k, tmp_path):
    """ParamsKZG files (reference halo2-base/src/utils/mod.rs:401-443): both element encodings, written by the product AND by an independent
    writer on the oracle side (oracle/transcript.py's compression, big-int Montgomery conversion), round trips, corruption, gen_srs"""
    import struct

    from oracle import pairing as PR
    from oracle import transcript as OT

    n = 1 << k
    s = HP.default_srs_secret()
    params = HP.ParamsKZG.setup(ctx, k, s, precompute=False)
    g, gl = ctx.bases_download(params.g), ctx.bases_download(params.g_lagrange)
    # g[i] = s^i * G by definition (spot checks), s_g2 = s * G2 in the file tail
    for i in (0, 1, n - 1):
        assert O.limbs_to_points(g[i:i + 1]) == [O.g1_mul(O.G1_GEN, pow(s, i, R))]
    raw_g2 = lambda P: b"".join(((c << 256) % O.Q_MOD).to_bytes(32, "little") for c in (P[0][0], P[0][1], P[1][0], P[1][1]))
    assert params.g2_raw == raw_g2(PR.G2_GEN) + raw_g2(PR.g2_mul(PR.G2_GEN, s))
    # files from an independent writer
    pts = O.limbs_to_points(np.concatenate([g, gl]))
    mont = lambda v: ((v << 256) % O.Q_MOD).to_bytes(32, "little")
    oracle_raw = struct.pack("<I", k) + b"".join(bytes(64) if P is None else mont(P[0]) + mont(P[1]) for P in pts) + params.g2_raw
    oracle_proc = struct.pack("<I", k) + b"".join(OT.g1_compress(P) for P in pts) + params.g2_raw
    for fmt, want in (("raw", oracle_raw), ("processed", oracle_proc)):
        path = os.path.join(tmp_path, "p_%s.srs" % fmt)
        params.write(path, fmt)
        assert open(path, "rb").read() == want, fmt                       # the product writes what the independent writer writes
        for pre in (False, True):
            again = HP.ParamsKZG.read(ctx, path, precompute=pre)
            assert again.k == k and again.g2_raw == params.g2_raw
            assert np.array_equal(ctx.bases_download(again.g), g) and np.array_equal(ctx.bases_download(again.g_lagrange), gl)
            v = rand_fr(n, 3)
            assert np.array_equal(again.commit_lagrange(v, H.POINT_AFFINE), params.commit_lagrange(v, H.POINT_AFFINE))
            again.free()
        # corruption: one flipped bit in a point -> rejected, not silently wrong commitments
        bad = bytearray(want)
        bad[4 + 64 * 3 + 5 if fmt == "raw" else 4 + 32 * 3 + 5] ^= 0x10
        open(path, "wb").write(bytes(bad))
        with pytest.raises(ValueError):
            HP.ParamsKZG.read(ctx, path, precompute=False)
        open(path, "wb").write(want[: len(want) // 3])
        with pytest.raises(ValueError):
            HP.ParamsKZG.read(ctx, path, precompute=False)
    # a non-canonical coordinate (x + q) in a raw file
    bad = bytearray(oracle_raw)
    x = int.from_bytes(bad[4 + 64:4 + 96], "little") + O.Q_MOD
    if x < 1 << 256:
        bad[4 + 64:4 + 96] = x.to_bytes(32, "little")
        path = os.path.join(tmp_path, "noncanonical.srs")
        open(path, "wb").write(bytes(bad))
        with pytest.raises(ValueError):
            HP.ParamsKZG.read(ctx, path, precompute=False)
    # gen_srs: PARAMS_DIR convention, setup with the reference's fixed seed on first use, read afterwards
    d = os.path.join(tmp_path, "params")
    first = HP.gen_srs(ctx, k, params_dir=d, precompute=False)
    assert os.path.exists(os.path.join(d, "kzg_bn254_%d.srs" % k)) and np.array_equal(ctx.bases_download(first.g), g)
    second = HP.gen_srs(ctx, k, params_dir=d, precompute=False)
    assert np.array_equal(ctx.bases_download(second.g_lagrange), gl)
    for p_ in (first, second, params):
        p_.free()


def test_srs_files_emulated(tmp_path):
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        _srs_files(ctx, 5, str(tmp_path))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_srs_files_gpu(tmp_path):
    ctx = H.Context()
    try:
        _srs_files(ctx, 10, str(tmp_path))
    finally:
        ctx.close()
