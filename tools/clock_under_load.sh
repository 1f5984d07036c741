#!/bin/bash
# r05: the shader clock the chip holds under the prover's kernels — rocm-smi sampled every 0.1 s next to (a) back-to-back 2^20-point MSMs, (b) back-to-back k = 19 proofs,
# (c) the multiplier probe, (d) idle.  usage: bash tools/clock_under_load.sh  (on the GPU box, from the repo root)
sample() {   # $1 = label, $2 = seconds
  local n=$(( $2 * 10 )) out=""
  for i in $(seq $n); do out="$out $(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | grep -o '([0-9]*Mhz)' | tr -d '()Mhz' | head -1)"; sleep 0.1; done
  python3 - "$1" $out <<'PY'
import sys
v=[int(x) for x in sys.argv[2:] if x.isdigit()]
print("%-28s sclk MHz: n=%d min %d median %d max %d" % (sys.argv[1], len(v), min(v) if v else 0, sorted(v)[len(v)//2] if v else 0, max(v) if v else 0))
PY
}
sample "idle" 2
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import halo2_lib_amd as H
from bench import synthetic_scalars
from halo2_lib_amd import halo2_proofs as HP
ctx = H.Context(0)
kzg = HP.ParamsKZG.setup(ctx, 20, 0x1D0C0FFEE1234567890ABCDEF, precompute=True)
d = [ctx.to_device(synthetic_scalars(1 << 20, 5 + j)) for j in range(4)]
import ctypes as C
out = (C.c_uint8 * (96 * 4))()
ptrs = (C.c_void_p * 4)(*[C.c_void_p(x) for x in d])
open("/tmp/h2_load_started", "w").write("1")
t0 = time.time()
n = 0
while time.time() - t0 < 9:
    ctx._chk(ctx.lib.h2hip_msm_g1_batch_dev(ctx.handle, kzg.g.handle, ptrs, 1 << 20, 4, 0, out)); n += 4
print("msm loop: %d MSMs of 2^20 in %.1f s = %.3f ms per MSM" % (n, time.time() - t0, (time.time() - t0) * 1e3 / n))
PY
while [ ! -f /tmp/h2_load_started ]; do sleep 0.2; done
sleep 1; sample "2^20-point MSMs (batches of 4)" 5
wait; rm -f /tmp/h2_load_started
( timeout 60 ./tools/probes/valu_rate > /dev/null 2>&1 ) &
sleep 0.3; sample "VALU probe (pure multiply-adds)" 1
wait
sample "idle again" 1
