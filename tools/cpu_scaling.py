"""how the C restatement of best_multiexp scales with threads on this host (cgroup limits, SMT)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO
from bench import synthetic_bases, synthetic_scalars
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: pass
lib = CO.lib(native=True)
n = 1 << 20
b, s = synthetic_bases(n, 1), synthetic_scalars(n, 2)
for t in (1, 8, 32, 64, 128, 256):
    if t == 1:
        m = 1 << 16
        t0 = time.perf_counter(); CO.best_multiexp(s[:m], b[:m], threads=1, l=lib); dt = (time.perf_counter() - t0) * (n / m)
        print(f"threads=1 (2^16 sample, scaled to 2^20 points): {dt:.3f} s", flush=True)
        continue
    CO.best_multiexp(s, b, threads=t, l=lib)
    t0 = time.perf_counter()
    for _ in range(3): CO.best_multiexp(s, b, threads=t, l=lib)
    print(f"threads={t}: {(time.perf_counter() - t0) / 3:.3f} s", flush=True)
