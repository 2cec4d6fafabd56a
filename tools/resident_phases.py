"""In-kernel timeline of the single-launch Gram-Schmidt step (block 0's 100 MHz clock at the phase boundaries) across sizes.
  python tools/resident_phases.py [f64|c128] [KEY=INT ...] [sizes=..] [ks=..]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
dtype = np.complex128 if len(sys.argv) > 1 and sys.argv[1] == "c128" else np.float64
ctx = lk.Context(device=0)
sizes, ks, knobs = (1000, 30_000, 175_000, 300_000, 1_000_000), (1, 8, 32, 64, 128), {}
for kv in sys.argv[2:]:
    key, val = kv.split("=")
    if key == "sizes": sizes = tuple(int(v) for v in val.split(","))
    elif key == "ks": ks = tuple(int(v) for v in val.split(","))
    else:
        ctx.set_tuning(key, int(val)); knobs[key] = int(val)
s = np.dtype(dtype).itemsize
ctx.set_tuning("resident_max_mb", 4096)
for n in sizes:
    kmax = max(ks)
    B = lk.krylov_basis_gpu(n, kmax + 1, dtype, ctx)
    for j in range(kmax + 1):
        B[j].rand(True, seed=100 + j)
    for k in ks:
        acc = np.zeros(7)
        reps = 6
        _b = ctx.resident_stats()[2]
        for r in range(reps + 1):
            lk.double_gram_schmidt_step(B[kmax], B[:k], False)
            if r: acc += np.array(ctx.resident_phase_us())
        acc /= reps
        _onchip = ctx.resident_stats()[2] > _b
        print(json.dumps({"dtype": np.dtype(dtype).name, "n": n, "k": k, "knobs": knobs, "panel_MB": round(s * n * (k + 1) / 2**20, 1),
                          "onchip": ctx.resident_stats()[2] > 0 and knobs.get("resident_onchip", 1) == 1 and _onchip,
                          "us[ph1,sum1,ph2,sum2,ph3,sum3,scale]": [round(float(v), 1) for v in acc], "total_us": round(float(acc.sum()), 1),
                          "TBps_on_3k+5": round(s * n * (3 * k + 5) / acc.sum() / 1e6, 2)}), flush=True)
    del B
