"""double_gram_schmidt_step against WIDE bases (129..512 columns, the lane-split fused sweeps; beyond: column panels of 512):
GB/s of each sweep and of the whole DGS on the ALGORITHMIC 3k+5 columns.   python tools/bench_wide.py [n] [f64|c128] [KEY=INT ...]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
dtype = np.complex128 if len(sys.argv) > 2 and sys.argv[2] == "c128" else np.float64
ctx = lk.Context(device=0)
for kv in sys.argv[3:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
s = np.dtype(dtype).itemsize
ks = (64, 128, 129, 160, 200, 256, 257, 320, 384, 512, 640)
kmax = max(ks)
B = lk.krylov_basis_gpu(n, kmax + 1, dtype, ctx)
for j in range(kmax + 1):
    B[j].rand(True, seed=100 + j)
for k in ks:
    lk.double_gram_schmidt_step(B[kmax], B[:k], False)
    ctx.profile_reset(); ctx.profile_enable(True)
    reps = 4
    for _ in range(reps):
        lk.double_gram_schmidt_step(B[kmax], B[:k], False)
    out, ms_tot = {}, {}
    for tag in ("dgs_sweep1", "dgs_sweep2", "dgs_sweep3", "dgs_sweep*", "dgs"):
        c, ms, by = ctx.profile_get(tag)
        out[tag] = round(by / ms / 1e6) if ms > 0 else None
        ms_tot[tag] = ms / reps
    ctx.profile_enable(False)
    print(json.dumps({"n": n, "dtype": str(np.dtype(dtype)), "k": k, "ms_per_dgs": round(ms_tot["dgs"], 3),
                      "GBps_on_3k+5": round(s * n * (3 * k + 5) / ms_tot["dgs"] / 1e6), "per_sweep_GBps": out}), flush=True)
