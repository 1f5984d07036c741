"""Helpers for the CPU-emulated kernel tests (tests/emu/ is test infrastructure, see its header)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def emu_context():
    import build_emu
    import halo2_lib_amd as H

    # H2HIP_EMU_LIB: an alternative build of the same sources (e.g. with -fsanitize=address, see tools/README.md) instead of the default one
    return H.Context(lib_path=os.environ.get("H2HIP_EMU_LIB") or build_emu.build())
