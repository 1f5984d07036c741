"""C++ host mirror (halo2-lib_amd/host/halo2_proofs.hpp) through the C ABI: over the emulated kernels on CPU,
over the real libh2hip.so on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "halo2-lib_amd", "host")


def test_selftest_over_emulated_kernels(tmp_path):
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu

    lib = build_emu.build()
    exe = str(tmp_path / "selftest_emu")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(HOST, "selftest.cpp"), "-L" + os.path.dirname(lib),
                           "-lh2hip_emu", "-Wl,-rpath," + os.path.dirname(lib), "-lpthread"])
    out = subprocess.run([exe, "7"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "selftest OK" in out.stdout, out.stdout + out.stderr
    dump = subprocess.run([exe, "7", "--dump-proof"], capture_output=True, text=True, timeout=300)
    assert dump.returncode == 0, dump.stderr
    assert bytes.fromhex(dump.stdout.strip()) == _oracle_proof_of_selftest_circuit(7, lib)


def _oracle_proof_of_selftest_circuit(k, emu_lib):
    """the circuit and RNG stream of selftest.cpp's prove_small_circuit, rebuilt here and proven by the oracle prover: the C++ host mirror
    (what the Rust shim would call) must produce the same proof bytes"""
    import numpy as np

    import halo2_lib_amd as H
    from halo2_lib_amd import halo2_proofs as HP
    from oracle import bn254 as O
    from oracle import plonk as P

    R = O.R_MOD
    lb = k - 2
    sh = P.Shape(k, 1, 1, 1, 0, lb)
    n, m = sh.n, sh.usable_rows // 4

    def draw(g):
        v = [g.next(), g.next(), g.next(), g.next() >> 4]
        return (v[0] | v[1] << 64 | v[2] << 128 | v[3] << 192) % R

    g = O.SplitMix64(99)
    adv = [0] * n
    fixed = [[0] * n for _ in range(sh.num_fixed_total)]
    for i in range(1 << lb):
        fixed[sh.table_col][i] = i
    for j in range(m):
        a = (g.next() & ((1 << lb) - 1)) if j % 3 == 0 else draw(g)
        b, c = draw(g), draw(g)
        if j == 1:
            b = adv[1]
        adv[4 * j: 4 * j + 4] = [a, b, c, (a + b * c) % R]
        fixed[sh.q_enable_cols[0]][4 * j] = 1
        if j % 3 == 0:
            fixed[sh.q_lookup_col][4 * j] = 1
    asm = P.PermutationAssembly(sh)
    asm.copy((("advice", 0), 1), (("advice", 0), 5))
    for t in range(min(8, m)):
        fixed[sh.constant_cols[0]][t] = adv[4 * t + 2]
        asm.copy((("fixed", sh.constant_cols[0]), t), (("advice", 0), 4 * t + 2))
    ctx = H.Context(lib_path=emu_lib)
    kzg = HP.ParamsKZG.setup(ctx, k, 0x5EED5EED5EED, precompute=False)
    params = P.Params.setup(k, 0x5EED5EED5EED, g=ctx.bases_download(kzg.g), g_lagrange=ctx.bases_download(kzg.g_lagrange))
    kzg.free()
    ctx.close()
    pk = P.keygen(params, sh, [O.ints_to_limbs(c, R) for c in fixed], asm, 2)
    pk.vk.transcript_repr = 0x1234567890ABCDEF

    class Rng:
        def __init__(self):
            self.g = O.SplitMix64(7)

        def next_fr(self):
            return draw(self.g)

        def fill(self, cnt):
            return O.ints_to_limbs([draw(self.g) for _ in range(cnt)], R)

    proof = P.create_proof(params, pk, [O.ints_to_limbs(adv, R)], [], Rng(), 2)
    assert P.verify_proof(params, pk.vk, [], proof)
    return proof


@pytest.mark.gpu
def test_selftest_on_gpu():
    exe = os.path.join(HOST, "selftest")
    if not os.path.exists(exe):   # normally prebuilt by __graft_entry__.build()
        import __graft_entry__ as g

        g.build()
    out = subprocess.run([exe, "14"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "selftest OK" in out.stdout, out.stdout + out.stderr
    dump = subprocess.run([exe, "11", "--dump-proof"], capture_output=True, text=True, timeout=300)
    assert dump.returncode == 0, dump.stderr
    assert bytes.fromhex(dump.stdout.strip()) == _oracle_proof_of_selftest_circuit(11, None)
