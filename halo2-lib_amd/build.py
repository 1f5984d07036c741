"""Builds halo2-lib_amd/csrc/libh2hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libh2hip.so")
SOURCES = ["capi.hip", "ntt.hip", "msm.hip", "msm_tables.hip", "fr_ops.hip", "srs.hip", "lookup.hip", "prover_ops.hip", "plonk.hip", "verifier.hip", "g2.hip", "comm.hip", "rng.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".cuh", ".h", ".inc"))) + [os.path.join("..", "..", "include", "h2hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS if os.path.exists(os.path.join(CSRC, f)))


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS if os.path.exists(os.path.join(CSRC, f)))
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        # incremental: an object newer than its source and than every header is kept
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_time):
            continue
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
