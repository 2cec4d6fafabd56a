"""Host-side probe (GPU box): how the all-core fused CPU leg of bench.py's cpu_baseline scales with threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as ora
print("max threads", ora.max_threads(), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)),
      "OMP_PROC_BIND", os.environ.get("OMP_PROC_BIND"), flush=True)
n, m = 16_000_000, 24
for T in (8, 16, 32, 64, 128):
    if T > ora.max_threads():
        break
    dt, _ = bench._oracle_sample(n, m, T, fused=True, repeat=2)
    by = sum(8.0 * n * (3 * k + 10) for k in range(1, m + 1))
    print(f"threads {T:4d}: {dt:7.3f} s  {by / dt / 1e9:7.1f} GB/s on the fused byte model", flush=True)
