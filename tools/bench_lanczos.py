"""Lanczos / Golub-Kahan against Arnoldi on the same operator (diagonal, real(dp)): per-object calls of the python mirror, eager
and lazy engine.   python tools/bench_lanczos.py [rows] [m]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
out = {"n": n, "m": m}
for lazy in (0, 1):
    ctx = lk.Context(device=0)
    ctx.set_tuning("lazy", lazy)
    A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=ctx)
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    T = np.zeros((m + 1, m), order="F")
    def run(fn):
        best = 1e9
        for _ in range(3):
            X[0].rand(True, seed=7); ctx.sync()
            t0 = time.perf_counter(); fn(); ctx.sync(); best = min(best, time.perf_counter() - t0)
        return best
    t_l = run(lambda: lk.lanczos(A, X, T))
    t_a = run(lambda: lk.arnoldi(A, X, T))
    out["lazy" if lazy else "eager"] = {"lanczos_ms": round(t_l * 1e3, 2), "arnoldi_ms": round(t_a * 1e3, 2), "lanczos_steps_per_s": round(m / t_l, 1)}
    del X, A
    ctx.close()
print(json.dumps(out))
