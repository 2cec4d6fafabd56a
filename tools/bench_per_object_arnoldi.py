"""Arnoldi as an unchanged LightKrylov drives it -- per-object type-bound procedures: y%norm(), k x X(i)%dot(y),
proj%zero(), k x proj%axpby(h_i, X(i), 1), y%sub(proj), twice per step, then qr's norm + scal -- eager vs lazy,
against the fused lk_arnoldi.  The per-object runs go through the python mirror of the reference's arnoldi /
double_gram_schmidt_step / linear_combination on a python LIST of vectors (the same ABI call sequence the Fortran
plugin issues; temporaries come from the column pool in both)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

n, m = (int(float(sys.argv[1])), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000_000, 64)
out = {"n": n, "m": m}
alg = sum(8.0 * n * (3 * k + 5) for k in range(1, m + 1))
for mode in ("eager", "lazy", "fused"):
    c = lk.Context(device=0)
    c.set_tuning("lazy", 1 if mode == "lazy" else 0)
    A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=c)

    class pyop(lk.abstract_linop):                     # python operator => the python (reference) step loop
        def matvec(self, vi, vo): A.matvec(vi, vo)
    B = lk.krylov_basis_gpu(n, m + 1, np.float64, c)
    X = B if mode == "fused" else [B[j] for j in range(m + 1)]
    op = A if mode == "fused" else pyop()
    H = np.zeros((m + 1, m), order="F")
    best = None
    for rep in range(2):
        B[0].rand(True, seed=7)
        c.sync(); t0 = time.perf_counter()
        info = lk.arnoldi(op, X, H)
        c.sync(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out[mode] = {"seconds": best, "iters_per_s": m / best, "GBps_on_algorithmic_3k+5": alg / best / 1e9, "info": info,
                 "H_fro": float(np.linalg.norm(H))}
    if mode == "lazy":
        out[mode]["lazy_stats"] = c.lazy_stats(); out[mode]["fusion_stats"] = c.lazy_fusion_stats()
    del X, B, A, op
    c.close()
print(json.dumps(out))
