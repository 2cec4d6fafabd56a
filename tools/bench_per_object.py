"""Per-object (type-bound-procedure) DGS, as an unchanged LightKrylov drives it, eager vs lazy vs the fused call."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
n, k = (int(float(sys.argv[1])), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000_000, 64)
out = {}
for mode in ("eager", "lazy", "fused"):
    c = lk.Context(device=0)
    c.set_tuning("lazy", 1 if mode == "lazy" else 0)
    B = lk.krylov_basis_gpu(n, k + 1, np.float64, c)
    for j in range(k + 1):
        B[j].rand(True, seed=j)
    X = B[:k] if mode == "fused" else [B[j] for j in range(k)]
    beta = np.zeros(k)
    lk.double_gram_schmidt_step(B[k], X, False, beta)
    c.sync(); t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        lk.double_gram_schmidt_step(B[k], X, False, beta)
    c.sync(); dt = (time.perf_counter() - t0) / reps
    out[mode] = {"ms_per_dgs": dt * 1e3, "GBps_on_algorithmic_3k+5": 8 * n * (3 * k + 5) / dt / 1e9}
    del X, B; c.close()
print(json.dumps({"n": n, "k": k, **out}))
