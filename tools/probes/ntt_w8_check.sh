#!/bin/bash
# compiles r05's wave-owned radix-8 pass (tools/probes/ntt_w8_kernel.inc) for gfx950 against the product's CURRENT ntt.hip: the kernel text is put back
# in front of ntt_direct_twiddle_kernel of a temporary copy.  A compile check only — the probe is not linked into libh2hip.so any more (r06).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); T=$(mktemp -d)
python3 - "$R" "$T" <<'PY'
import sys
r, t = sys.argv[1:3]
s = open(r + "/halo2-lib_amd/csrc/ntt.hip").read()
inc = open(r + "/tools/probes/ntt_w8_kernel.inc").read().split("// ---- the launch code")[0]
# the probe was written against r05's tile types: give it the names it used
shim = "struct TileElem { typedef Fr29L type; };\n"
k = s.index("__global__ void ntt_direct_twiddle_kernel")
open(t + "/ntt_w8_probe.hip", "w").write(s[:k] + shim + inc + s[k:])
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$R/halo2-lib_amd/csrc" -I"$R/include" -c "$T/ntt_w8_probe.hip" -o "$T/ntt_w8_probe.o" && echo "ntt_w8 probe compiles"
rm -rf "$T"
