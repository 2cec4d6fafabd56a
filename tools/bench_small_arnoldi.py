"""Arnoldi factorisations in the launch-bound regime (the reference's published use case is 1.75 10^5 unknowns,
paper/paper.md:103-113): iterations per second of lk_arnoldi with a diagonal operator, the single-launch Gram-Schmidt step
(csrc/lk_resident.hip.h) against the three-sweep schedule, interleaved in one process.
  python tools/bench_small_arnoldi.py [f64|c128] [KEY=INT ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
dtype = np.complex128 if len(sys.argv) > 1 and sys.argv[1] == "c128" else np.float64
ctx = lk.Context(device=0)
for kv in sys.argv[2:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
CASES = [(175_000, 32), (175_000, 64), (175_000, 128), (300_000, 32), (300_000, 128), (1_000_000, 32), (1_000_000, 64), (1_000_000, 128), (3_000_000, 64)]
for n, m in CASES:
    nn = n if dtype is np.float64 else n // 2
    X = lk.krylov_basis_gpu(nn, m + 1, dtype, ctx)
    A = lk.diag_linop_gpu(n_local=nn, row0=0, d0=1.0, dstep=1.0 / nn, ctx=ctx) if dtype is np.float64 else lk.diag_linop_gpu((1.0 + np.arange(nn) / nn).astype(dtype), ctx)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    out = {"dtype": np.dtype(dtype).name, "n": nn, "m": m}
    best = {0: 1e9, 1: 1e9}
    for rep in range(7):
        for route in (0, 1):
            ctx.set_tuning("resident", route)
            X[0].rand(True, seed=7)
            ctx.sync()
            t0 = time.perf_counter()
            info = lk.arnoldi(A, X, H)
            dt = time.perf_counter() - t0
            assert info == 0
            if rep > 0: best[route] = min(best[route], dt)
    out["three_sweeps_it_per_s"] = round(m / best[0], 1)
    out["single_launch_it_per_s"] = round(m / best[1], 1)
    out["us_per_step"] = [round(best[0] / m * 1e6, 1), round(best[1] / m * 1e6, 1)]
    print(json.dumps(out), flush=True)
    del X
