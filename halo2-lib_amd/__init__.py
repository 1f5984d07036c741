"""
halo2-lib_amd — MI355X (gfx950) prover backend for halo2-lib's proving hot path: BN254 G1 Pippenger MSM,
radix-2 NTT family over F_r and the pointwise F_r kernels around them, as hand-written HIP behind the C ABI
declared in include/h2hip.h (built to halo2-lib_amd/csrc/libh2hip.so).

Python here is host-side plumbing only (ctypes over the C ABI, mirroring the halo2_proofs names the reference
uses at halo2-base/src/utils/testing.rs:8-22); there is no CPU implementation and no fallback: importing
works anywhere, creating a Context without the built library or without a GPU raises.
"""
from .h2hip import (  # noqa: F401
    H2HipError,
    Context,
    Bases,
    library_path,
    load_library,
    POINT_AFFINE,
    POINT_JACOBIAN,
)

__all__ = ["H2HipError", "Context", "Bases", "library_path", "load_library", "POINT_AFFINE", "POINT_JACOBIAN"]
