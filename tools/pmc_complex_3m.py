"""complex block DGS / innerprod / lincomb once with gemm_3m = V (for PMC passes)"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import lightkrylov_amd as lk
ctx = lk.Context(device=0)
ctx.set_tuning("gemm_3m", int(sys.argv[1]))
n, k, p = 2_000_000, 128, 32
B = lk.krylov_basis_gpu(n, k, np.complex128, ctx); Y = lk.krylov_basis_gpu(n, p, np.complex128, ctx)
for j in range(k): B[j].rand(True, seed=10 + j)
for j in range(p): Y[j].rand(True, seed=500 + j)
for _ in range(3):
    lk.innerprod(B, Y)
Z = np.asfortranarray((np.random.default_rng(0).standard_normal((k, 64)) + 1j * np.random.default_rng(1).standard_normal((k, 64))))
for _ in range(3):
    lk.linear_combination(B, Z)
ctx.sync()
