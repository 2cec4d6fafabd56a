"""krylov_schur-shaped basis update X Z on the GPU: GB/s of lk_lincomb (panel_gemm) at config-4 shape."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
ctx = lk.Context(device=0)
for dtype, n, k, q in ((np.complex128, 1_000_000, 128, 64), (np.float64, 10_000_000, 64, 32), (np.float64, 10_000_000, 128, 1)):
    X = lk.krylov_basis_gpu(n, k, dtype, ctx)
    for j in range(k):
        X[j].rand(True, seed=j)
    Z = np.asfortranarray(np.random.default_rng(0).standard_normal((k, q)).astype(dtype))
    lk.linear_combination(X, Z)
    ctx.profile_reset(); ctx.profile_enable(True)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(5):
        Y = lk.linear_combination(X, Z)
    ctx.sync(); dt = (time.perf_counter() - t0) / 5
    cnt, ms, by = ctx.profile_get("lincomb"); ctx.profile_enable(False)
    print(json.dumps({"kernel_launches": cnt, "kernel_ms_per_call": ms / 5, "kernel_GBps_actual": by / ms / 1e6}))
    s = np.dtype(dtype).itemsize
    print(json.dumps({"dtype": str(np.dtype(dtype)), "n": n, "k": k, "q": q, "ms": dt * 1e3,
                      "GBps_min_traffic(k+q cols)": s * n * (k + q) / dt / 1e9}))
