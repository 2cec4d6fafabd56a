"""Single-vector double Gram-Schmidt step across sizes: TB/s on the algorithmic 3k+5 columns for a grid of (rows, basis columns), both kinds --
where the fused sweeps sit on the streaming ceiling and where they turn launch-bound or change shape (register tiles, lane split, single launch).
  python tools/scan_dgs.py [f64|c128] [KEY=INT ...] [sizes=300000,1000000] [ks=8,32,128]
(round 6: "dgs_sweep*" includes the single launch of csrc/lk_resident.hip.h, tag dgs_sweep_resident; resident=0 gives the three-sweep schedule)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
dtype = np.complex128 if len(sys.argv) > 1 and sys.argv[1] == "c128" else np.float64
ctx = lk.Context(device=0)
ks = (4, 8, 16, 24, 32, 33, 48, 64, 96, 128)
sizes = (300_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000)
knobs = {}
for kv in sys.argv[2:]:
    key, val = kv.split("=")
    if key == "sizes": sizes = tuple(int(v) for v in val.split(","))
    elif key == "ks": ks = tuple(int(v) for v in val.split(","))
    else:
        ctx.set_tuning(key, int(val))
        knobs[key] = int(val)
s = np.dtype(dtype).itemsize
for n in sizes:
    nn = n if s == 8 else n // 2
    kmax = max(ks)
    B = lk.krylov_basis_gpu(nn, kmax + 1, dtype, ctx)
    for j in range(kmax + 1):
        B[j].rand(True, seed=100 + j)
    row = {}
    for k in ks:
        reps = 40 if nn <= 3_000_000 else 3
        for _ in range(10 if nn <= 3_000_000 else 1):      # (the first launches on fresh memory run 20-30 us slower: profiles/r06 notes)
            lk.double_gram_schmidt_step(B[kmax], B[:k], False)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(reps):
            lk.double_gram_schmidt_step(B[kmax], B[:k], False)
        c, ms, by = ctx.profile_get("dgs")
        c2, ms2, by2 = ctx.profile_get("dgs_sweep*")
        ctx.profile_enable(False)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(reps):
            lk.double_gram_schmidt_step(B[kmax], B[:k], False)
        ctx.sync(); wall_us = (time.perf_counter() - t0) / reps * 1e6      # host-visible latency of a synchronous call, profiling off (python included)
        row[str(k)] = {"dgs_TBps": round(s * nn * (3 * k + 5) / (ms / reps) / 1e9, 2), "sweeps_only_TBps": round(by2 / ms2 / 1e9, 2) if ms2 > 0 else None,
                       "ms": round(ms / reps, 4), "wall_us_per_sync_call": round(wall_us, 1), "single_launch": c2 == reps}
    print(json.dumps({"dtype": np.dtype(dtype).name, "n": nn, "knobs": knobs, "by_k": row}), flush=True)
    del B
