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
    out = subprocess.run([exe, "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "selftest OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_selftest_on_gpu():
    exe = os.path.join(HOST, "selftest")
    if not os.path.exists(exe):   # normally prebuilt by __graft_entry__.build()
        import __graft_entry__ as g

        g.build()
    out = subprocess.run([exe, "14"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "selftest OK" in out.stdout, out.stdout + out.stderr
