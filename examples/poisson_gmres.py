#!/usr/bin/env python3
"""BASELINE's Poisson case: GMRES(30) on the 5-point Laplacian of an N x N grid, once with the matrix-free stencil
operator and once with the same matrix handed over in CSR (a user's sparse `abstract_linop`).

  python examples/poisson_gmres.py [N=1024]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = N * N
ctx = lk.Context(device=0)
T = sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1])
S = sp.diags([-np.ones(N - 1), -np.ones(N - 1)], [-1, 1])
Acsr = ((sp.kron(sp.identity(N), T) + sp.kron(S, sp.identity(N))) * float((N + 1) ** 2)).tocsr()
b = lk.dense_vector_gpu(n, np.float64, ctx)
b.rand(False, seed=11)
for name, A in (("stencil", lk.laplacian2d_linop_gpu(N, ctx)), ("csr", lk.csr_linop_gpu(Acsr, ctx))):
    x = lk.dense_vector_gpu(n, np.float64, ctx)                      # x0 = 0
    meta = lk.gmres_dp_metadata()
    ctx.sync(); t0 = time.perf_counter()
    info = lk.gmres(A, b, x, rtol=1e-8, options=lk.gmres_dp_opts(kdim=30, maxiter=20), meta=meta)
    ctx.sync(); dt = time.perf_counter() - t0
    r = lk.dense_vector_gpu(n, np.float64, ctx)
    A.apply_matvec(x, r); r.sub(b)
    print(f"{name:8s} info = {info:5d}  {meta.n_iter} iterations in {dt:.3f} s   |b - A x| / |b| = {r.norm() / b.norm():.3e}")
ctx.close()
