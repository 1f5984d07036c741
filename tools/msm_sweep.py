"""MSM parameter sweep on the GPU (writes gpurun_out/sweep.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import halo2_lib_amd as H
from bench import synthetic_bases, synthetic_scalars
ctx = H.Context(0)
n = 1 << 20
bases_h = synthetic_bases(n, 1); s = synthetic_scalars(n, 2)
ds = ctx.to_device(s)
res = {}
for pre in (1,):
    b = ctx.bases_upload(bases_h, pre)
    for k1, seg, var in ((64, 8, 1), (64, 8, 2), (64, 8, 4), (64, 8, 8), (64, 8, 16)):
        if True:
            ctx.set_param("msm_chunk", k1); ctx.set_param("msm_seg", seg); ctx.set_param("msm_scatter_split", var)
            ctx.msm_dev(b, ds, n)
            ctx.profile_enable(True); ctx.profile_reset()
            ctx.timer_start()
            for _ in range(5): ctx.msm_dev(b, ds, n)
            ms = ctx.timer_stop() / 5
            parts = {nm: round(ctx.profile_get(nm)[0] / 5, 3) for nm in ("msm_scatter", "msm_accum_kernel", "msm_merge", "msm_presum", "msm_seg", "msm_winsum", "msm_fold")}
            ctx.profile_enable(False)
            res[f"pre{pre}_k{k1}_seg{seg}_v{var}"] = {"ms": round(ms, 3), **parts}
            print(f"pre{pre}_k{k1}_seg{seg}_v{var}", round(ms, 3), parts, flush=True)
    b.free()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/sweep.json", "w"), indent=1)
