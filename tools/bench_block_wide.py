"""Block double Gram-Schmidt against a basis WIDER than 128 columns (lk_dgs_block, round 5): the panel x panel schedule over column
panels of X (coefficients and updates on the matrix cores, 4k - |last panel| columns of X per group of <= 32 columns of Y) against the
per-column fallback it replaces (xhy_mfma = 0: one three-sweep DGS per column of Y, 3k columns of X each).
  python tools/bench_block_wide.py [rows] [KEY=INT ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = lk.Context(device=0)
panels_only = "panels_only" in sys.argv[2:]          # (PMC passes: the panel schedule alone, default knobs, ONE warm-up + 3 calls per shape)
for kv in sys.argv[2:]:
    if "=" in kv:
        key, val = kv.split("=")
        ctx.set_tuning(key, int(val))
for dtype in (np.float64, np.complex128):
    nn = n if dtype == np.float64 else n // 2
    s = np.dtype(dtype).itemsize
    for k, p in ((256, 32), (256, 4), (192, 8), (512, 32)):
        if (k + p) * nn * s > 200e9:
            continue
        B = lk.krylov_basis_gpu(nn, k, dtype, ctx)
        Y = lk.krylov_basis_gpu(nn, p, dtype, ctx)
        for j in range(k):
            B[j].rand(True, seed=10 + j)
        for j in range(p):
            Y[j].rand(True, seed=500 + j)
        res = {}
        for name, mf, fused in ((("panels_default", 1, 1),) if panels_only else
                                (("per_column", 0, 1), ("panels_4pass", 1, 0), ("panels_fused_last", 1, 2), ("panels_default", 1, 1))):
            ctx.set_tuning("xhy_mfma", mf); ctx.set_tuning("block_fused", fused)
            fn = lambda: lk.double_gram_schmidt_step(Y, B, if_chk_orthonormal=False)   # noqa: E731
            fn(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            ctx.sync()
            res[name + "_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        last = k - (k - 1) // 128 * 128
        cols = (4 * k - last) * ((p + 31) // 32) + 6 * p           # X columns per block DGS + Y traffic (read + write per update, read per product)
        print(json.dumps({"dtype": np.dtype(dtype).name, "n": nn, "k": k, "p": p, "columns_moved_by_the_panel_schedule": cols,
                          "GB_of_X_per_block_dgs_(4k-last)": round((4 * k - last) * ((p + 31) // 32) * nn * s / 1e9, 3),
                          "ms_at_6.5TBps_for_those": round(cols * nn * s / 6.5e12 * 1e3, 3), "per_column_schedule_columns": 3 * k * p + 4 * p, **res}), flush=True)
        del B, Y
