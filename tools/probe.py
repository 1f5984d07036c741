"""First-contact GPU probe: box facts + multiplier roofline + MSM/NTT timings (writes gpurun_out/probe.json)."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from oracle import bn254 as O, c_oracle as CO
from tests.util import fr, rand_fr, circuit_like_fr, domain_consts

out = {}
def sh(c):
    try: return subprocess.run(c, shell=True, capture_output=True, text=True, timeout=60).stdout
    except Exception as e: return str(e)
out["nproc"] = os.cpu_count()
out["lscpu"] = sh("lscpu | head -20")
out["rocminfo"] = sh("rocminfo | grep -E 'Name:|Compute Unit|Max Clock|Wavefront|LDS|Cacheline' | head -40")
out["rocm_smi"] = sh("rocm-smi --showmeminfo vram --showclocks | head -30")
ctx = H.Context(0)
mm = {}
for chains in (1, 2, 4):
    for blocks in (1024, 4096, 16384):
        ms, n = ctx.bench_modmul(blocks, 256, chains)
        mm[f"chains{chains}_blocks{blocks}"] = {"ms": ms, "modmul_per_s": n / ms * 1e3}
out["modmul"] = mm
print(json.dumps(mm, indent=1))
ctx.profile_enable(True)
res = {}
for log_n in (16, 19, 20, 22):
    n = 1 << log_n
    a = rand_fr(n, log_n); w, winv, div = domain_consts(log_n)
    d = ctx.to_device(a)
    ctx.best_fft_dev(d, w, log_n); ctx.sync()
    ctx.timer_start()
    for _ in range(5): ctx.best_fft_dev(d, w, log_n)
    res[f"ntt_{log_n}_ms"] = ctx.timer_stop() / 5
    ctx.free(d)
for kind in ("uniform", "circuit", "uniform_pre", "circuit_pre"):
    for log_n in (16, 18, 20):
        n = 1 << log_n
        t = time.time(); bases = CO.known_dlog_bases(n, fr([5]), fr([3])); tb = time.time() - t
        s = rand_fr(n, 1) if kind.startswith("uniform") else circuit_like_fr(n, 1)
        t = time.time(); b = ctx.bases_upload(bases, 1 if kind.endswith("_pre") else 0); res[f"upload_{kind}_{log_n}_s"] = time.time() - t
        ds = ctx.to_device(s)
        ctx.msm_dev(b, ds, n)
        ctx.profile_reset()
        ctx.timer_start()
        for _ in range(3): ctx.msm_dev(b, ds, n)
        res[f"msm_{kind}_{log_n}_ms"] = ctx.timer_stop() / 3
        for name in ("msm_digits", "msm_hist_kernel", "msm_hist_scan", "scan_kernels", "msm_scatter", "msm_accum_kernel", "msm_merge", "msm_presum", "msm_seg", "msm_winsum", "msm_fold", "point_finish"):
            ms, cnt = ctx.profile_get(name)
            res[f"msm_{kind}_{log_n}_{name}_ms_per_msm"] = ms / 3
        ctx.free(ds); b.free()
        print(kind, log_n, res[f"msm_{kind}_{log_n}_ms"], "bases gen s", round(tb, 2), flush=True)
out["timings"] = res
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe.json", "w"), indent=1)
