"""Helpers for the CPU-emulated kernel tests (tests/emu/ is test infrastructure, see its header)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def emu_context():
    import build_emu
    import halo2_lib_amd as H

    return H.Context(lib_path=build_emu.build())
