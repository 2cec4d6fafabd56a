"""Single-vector double Gram-Schmidt step across sizes: TB/s on the algorithmic 3k+5 columns for a grid of (rows, basis columns), both kinds --
where the fused sweeps sit on the streaming ceiling and where they turn launch-bound or change shape (register tiles, lane split, kc32).
  python tools/scan_dgs.py [f64|c128] [KEY=INT ...]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
dtype = np.complex128 if len(sys.argv) > 1 and sys.argv[1] == "c128" else np.float64
ctx = lk.Context(device=0)
for kv in sys.argv[2:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
s = np.dtype(dtype).itemsize
ks = (4, 8, 16, 24, 32, 33, 48, 64, 96, 128)
for n in (300_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000):
    nn = n if s == 8 else n // 2
    kmax = max(ks)
    B = lk.krylov_basis_gpu(nn, kmax + 1, dtype, ctx)
    for j in range(kmax + 1):
        B[j].rand(True, seed=100 + j)
    row = {}
    for k in ks:
        lk.double_gram_schmidt_step(B[kmax], B[:k], False)
        ctx.profile_reset(); ctx.profile_enable(True)
        reps = 8 if nn <= 3_000_000 else 3
        for _ in range(reps):
            lk.double_gram_schmidt_step(B[kmax], B[:k], False)
        c, ms, by = ctx.profile_get("dgs")
        c2, ms2, by2 = ctx.profile_get("dgs_sweep*")
        ctx.profile_enable(False)
        row[str(k)] = {"dgs_TBps": round(s * nn * (3 * k + 5) / (ms / reps) / 1e9, 2), "sweeps_only_TBps": round(by2 / ms2 / 1e9, 2), "ms": round(ms / reps, 4)}
    print(json.dumps({"dtype": np.dtype(dtype).name, "n": nn, "by_k": row}), flush=True)
    del B
