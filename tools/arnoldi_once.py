"""A few Arnoldi factorisations of one size (diagonal operator): the target of `rocprofv3 --kernel-trace --stats -- python3 tools/arnoldi_once.py n m [reps] [KEY=INT ...]`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
n, m = int(float(sys.argv[1])), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = lk.Context(device=0)
for kv in sys.argv[4:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=ctx)
H = np.zeros((m + 1, m), order="F")
best = 1e9
for _ in range(reps):
    X[0].rand(True, seed=7)
    ctx.sync()
    t0 = time.perf_counter()
    assert lk.arnoldi(A, X, H) == 0
    best = min(best, time.perf_counter() - t0)
print("arnoldi", n, m, "it/s", round(m / best, 1), ctx.resident_stats())
