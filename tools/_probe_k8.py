import sys, os, json, time
import numpy as np
sys.path.insert(0, "/root/repo")
import lightkrylov_amd as lk
ctx = lk.Context(device=0)
for n in (175_000, 300_000):
    for k in (8, 32):
        B = lk.krylov_basis_gpu(n, k + 1, np.float64, ctx)
        for j in range(k + 1):
            B[j].rand(True, seed=100 + j)
        out = {"n": n, "k": k}
        for route in (1, 0):
            ctx.set_tuning("resident", route)
            for _ in range(20): lk.double_gram_schmidt_step(B[k], B[:k], False)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(300): lk.double_gram_schmidt_step(B[k], B[:k], False)
            ctx.sync()
            out["wall_us_per_call_route%d" % route] = round((time.perf_counter() - t0) / 300 * 1e6, 1)
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(50): lk.double_gram_schmidt_step(B[k], B[:k], False)
            c1, ms1, _ = ctx.profile_get("dgs")
            c2, ms2, _ = ctx.profile_get("dgs_sweep*")
            ctx.profile_enable(False)
            out["dgs_tag_us_route%d" % route] = round(ms1 / c1 * 1e3, 1)
            out["kernels_us_route%d" % route] = round(ms2 / c1 * 1e3, 1)
        print(json.dumps(out), flush=True)
