"""Many right-hand sides: Gram / innerprod_matrix / block DGS with the X^H Y pass on the matrix cores (xhy_mfma = 1) against
the VALU schedule (panel_dot_p, <= 4 right-hand sides per pass over X).   python tools/bench_block.py [rows] [KEY=INT ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = lk.Context(device=0)
for kv in sys.argv[2:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
for dtype in (np.float64, np.complex128):
    nn = n if dtype == np.float64 else n // 2
    s = np.dtype(dtype).itemsize
    for k, p in ((128, 16), (128, 32), (64, 8), (32, 32)):
        B = lk.krylov_basis_gpu(nn, k, dtype, ctx)
        Y = lk.krylov_basis_gpu(nn, p, dtype, ctx)
        for j in range(k):
            B[j].rand(True, seed=10 + j)
        for j in range(p):
            Y[j].rand(True, seed=500 + j)
        res = {}
        for mf, fused in ((0, 1), (1, 0), (1, 2)):
            ctx.set_tuning("xhy_mfma", mf); ctx.set_tuning("block_fused", fused)
            out = {}
            for name, fn in (("gram", lambda: lk.Gram(B)), ("innerprod", lambda: lk.innerprod(B, Y)),
                             ("block_dgs", lambda: lk.double_gram_schmidt_step(Y, B, if_chk_orthonormal=False))):
                fn(); ctx.sync()
                t0 = time.perf_counter()
                for _ in range(3):
                    fn()
                ctx.sync()
                out[name + "_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            res[("mfma_3pass" if fused else "mfma_4pass") if mf else "valu"] = out
        one_pass_ms = nn * s * k / 6.5e12 * 1e3
        print(json.dumps({"dtype": np.dtype(dtype).name, "n": nn, "k": k, "p": p, "one_pass_over_X_ms_at_6.5TBps": round(one_pass_ms, 3), **res}), flush=True)
        del B, Y
