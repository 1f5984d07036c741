"""The prover's RNG on the device (halo2-lib_amd/csrc/rng.hip, include/h2hip.h: h2hip_chacha_rng*, h2hip_rng_chacha_fill_dev): the
`Fr::random` stream of the seeded rand_chacha generator the reference hands to create_proof (halo2-base/src/utils/testing.rs:38
`StdRng::seed_from_u64(0)`) and draws its SRS secret from (halo2-base/src/utils/mod.rs:441).

  * the ChaCha block function of libh2hip (host code) and of the oracle restatement against RFC 8439's published vectors (section 2.3.2 and
    appendix A.1 #1) — the one part of the construction with an external pin;
  * libh2hip's host generator, its device kernel (emulated build here, the real GPU under -m gpu) and the oracle's numpy restatement agree
    element for element, at offsets, for ChaCha12 and ChaCha20;
  * create_proof with the device-generated stream = create_proof with the same stream served by a host callback = the oracle prover drawing
    from its own restatement: proof bytes equal.
The stream LAYOUT (counter / stream words, word order of next_u64, from_u512) is [UPSTREAM-RECALL] — INTEGRATION.md section 8."""
import ctypes as C

import numpy as np
import pytest

import halo2_lib_amd as H
from oracle import bn254 as O
from oracle import chacha as OC

R = O.R_MOD
RFC_KEY = bytes(range(32))
# RFC 8439 section 2.3.2: key 00..1f, block counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00 -> state words 12..15 = 1, 0x09000000, 0x4a000000, 0
RFC_232 = bytes.fromhex("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
# RFC 8439 appendix A.1, test vector #1: all-zero key and nonce, counter 0
RFC_A1_1 = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                         "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")


def _lib():
    return H.load_library()


def _lib_block(seed, counter, stream, rounds):
    out = (C.c_uint8 * 64)()
    _lib().h2hip_chacha_block((C.c_uint8 * 32)(*seed), C.c_uint64(counter), C.c_uint64(stream), rounds, out)
    return bytes(out)


def test_chacha20_block_matches_rfc8439_vectors():
    ctr, stream = 1 | (0x09000000 << 32), 0x4A000000
    assert _lib_block(RFC_KEY, ctr, stream, 20) == RFC_232
    assert OC.chacha_blocks(RFC_KEY, [ctr], 20, stream).astype("<u4").tobytes() == RFC_232
    assert _lib_block(bytes(32), 0, 0, 20) == RFC_A1_1
    assert OC.chacha_blocks(bytes(32), [0], 20).astype("<u4").tobytes() == RFC_A1_1
    from halo2_lib_amd.halo2_proofs import _chacha20_block_zero_key, default_srs_secret

    assert _chacha20_block_zero_key(0) == RFC_A1_1                         # gen_srs's own block function
    assert default_srs_secret() == OC.fr_random_ints(bytes(32), 0, 1, 20)[0]   # the SRS secret = element 0 of the ChaCha20 zero-seed stream


def test_reduced_round_blocks_agree():
    """ChaCha12 / ChaCha8 have no RFC vector: the library's host code against the independent numpy restatement, many counters"""
    g = np.random.default_rng(5)
    seed = bytes(g.integers(0, 256, size=32, dtype=np.uint8))
    ctrs = [0, 1, 2, (1 << 32) - 1, 1 << 32, (1 << 40) + 12345, (1 << 64) - 1]
    for rounds in (8, 12, 20):
        want = OC.chacha_blocks(seed, ctrs, rounds).astype("<u4")
        for i, c in enumerate(ctrs):
            assert _lib_block(seed, c, 0, rounds) == want[i].tobytes(), (rounds, c)


def test_seed_from_u64():
    buf = (C.c_uint8 * 32)()
    for s in (0, 1, 42, (1 << 64) - 1):
        _lib().h2hip_rng_seed_from_u64(C.c_uint64(s), buf)
        assert bytes(buf) == OC.seed_from_u64(s)
    assert len(set(OC.seed_from_u64(0))) > 8   # not degenerate


def _host_stream(seed, rounds, first, count):
    from halo2_lib_amd.plonk import ChaChaRng

    rng = ChaChaRng(_lib(), seed, rounds)
    rng.state.pos = first
    out = rng.fill(count)
    assert rng.pos == first + count
    return out


def test_host_generator_matches_oracle_stream():
    for seed, rounds in ((0, 12), (bytes(32), 20), (7, 12)):
        sd = OC.seed_from_u64(seed) if isinstance(seed, int) else seed
        for first, count in ((0, 300), (1000, 17), ((1 << 32) - 5, 10)):
            assert np.array_equal(_host_stream(seed, rounds, first, count), O.ints_to_limbs(OC.fr_random_ints(sd, first, count, rounds), R))
    # every element is a canonical Montgomery value (< r)
    vals = O.limbs_to_ints(_host_stream(0, 12, 0, 64), R)
    assert all(0 <= v < R for v in vals) and len(set(vals)) == 64


def _device_stream_checks(ctx, sizes):
    for rounds, seed in ((12, OC.seed_from_u64(0)), (20, bytes(32))):
        for first, n in sizes:
            d = ctx.malloc(32 * n)
            ctx._chk(ctx.lib.h2hip_rng_chacha_fill_dev(ctx.handle, d, n, (C.c_uint8 * 32)(*seed), rounds, C.c_uint64(first)))
            got = ctx.download(d, (n, 4))
            ctx.free(d)
            assert np.array_equal(got, O.ints_to_limbs(OC.fr_random_ints(seed, first, n, rounds), R)), (rounds, first, n)


def _proof_checks(ctx, shape, precompute):
    """device-generated blinding = host-callback blinding = the oracle prover on its own restatement of the stream"""
    from halo2_lib_amd import halo2_proofs as HP
    from halo2_lib_amd import plonk as PL
    from halo2_lib_amd import testing as T
    from oracle import c_oracle as CO
    from oracle import plonk as P

    class _Backend:
        mul = staticmethod(CO.fr_mul)
        add = staticmethod(CO.fr_add)

    sh = P.Shape(*shape)
    kzg = HP.ParamsKZG.setup(ctx, shape[0], 0xFEEDC0DE, precompute=precompute)
    params = P.Params.setup(shape[0], 0xFEEDC0DE, g=ctx.bases_download(kzg.g), g_lagrange=ctx.bases_download(kzg.g_lagrange))
    circ = T.build_circuit(sh, 3, _Backend)
    pk = PL.keygen(kzg, PL.BaseCircuitParams.new(*shape), circ.fixed, circ.copies)
    try:
        dev_rng = PL.ChaChaRng(ctx.lib, 0, 12, device=True)
        host_rng = PL.ChaChaRng(ctx.lib, 0, 12, device=False)
        a = PL.create_proof(pk, circ.advice, circ.instances, dev_rng)
        b = PL.create_proof(pk, circ.advice, circ.instances, host_rng)
        assert a == b and dev_rng.pos == host_rng.pos > (1 << shape[0])
        asm = P.PermutationAssembly(sh)
        for l, r in circ.copies:
            asm.copy(l, r)
        opk = P.keygen(params, sh, circ.fixed, asm, 4)
        orng = OC.ChaChaFrRng(0, 12)
        inst = [O.limbs_to_ints(v, R) for v in circ.instances]
        assert P.create_proof(params, opk, circ.advice, inst, orng, 4) == a, "proof bytes differ from the oracle prover's"
        assert orng.pos == dev_rng.pos
        # a second proof continues the stream where the first left it (like the reference's rng across proofs): different bytes, same on both paths
        a2 = PL.create_proof(pk, circ.advice, circ.instances, dev_rng)
        assert a2 != a and a2 == PL.create_proof(pk, circ.advice, circ.instances, host_rng)
        assert PL.verify_proof(pk, circ.instances, a2)
    finally:
        pk.free()
        kzg.free()


def test_device_kernel_and_prover_path_emulated():
    from tests.emu_util import emu_context

    ctx = emu_context()
    try:
        _device_stream_checks(ctx, [(0, 1), (0, 300), (123456789, 70)])
        _proof_checks(ctx, (6, 2, 1, 1, 1, 4), precompute=False)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_device_stream_gpu():
    ctx = H.Context()
    try:
        _device_stream_checks(ctx, [(0, 1), (5, 1000), ((1 << 32) - 100, 300), (0, (1 << 16) + 3)])
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(12, 2, 1, 1, 1, 11), (15, 1, 1, 1, 0, 14)])
def test_create_proof_device_rng_gpu(shape):
    ctx = H.Context()
    try:
        _proof_checks(ctx, shape, precompute=True)
    finally:
        ctx.close()
