"""BASELINE.json configs[0..3] on one MI355X, end to end through the host-side mirror of the reference's solvers
(wall clock with the inputs resident in HBM; one warm-up run each).  configs[4] is bench.py --gpus N.
Each line also carries the algorithmic DGS bytes of the run and the fraction of the 8 TB/s line the WHOLE solve
(operator, host LAPACK, synchronisations included) reaches on them."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

ctx = lk.Context(device=0)


def timed(fn, reps=2):
    best, out = None, None
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter()
        out = fn()
        ctx.sync(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def dgs_bytes(s, n, ks):
    return float(sum(s * n * (3 * k + 5) for k in ks))


# configs[0]: eigs on a 1000 x 1000 random real dense operator, kdim = 30 (the reference's CPU-runnable case)
rng = np.random.default_rng(1)
n = 1000
A0 = rng.standard_normal((n, n)) / np.sqrt(n) + np.diag(np.r_[2.0, 1.8, 1.6, 1.4, np.zeros(n - 4)])
x0h = np.random.default_rng(2).standard_normal(n); x0h /= np.linalg.norm(x0h)
A = lk.dense_linop_gpu(A0, ctx)
def cfg1():
    V = lk.krylov_basis_gpu(n, 4, np.float64, ctx)
    x0 = lk.dense_vector_gpu.from_array(x0h, ctx)
    vals, res, info = lk.eigs(A, V, x0=x0, kdim=30, tolerance=1e-10, max_restarts=50)
    v = V[0].to_array()
    # the TRUE residual of the leading pair (the `residuals` eigs returns are the reference's: permuted by the restart that
    # follows convergence, IterativeSolvers.fypp:1100, 1118-1120 -- faithful, but not a convergence measure)
    return vals, float(np.linalg.norm(A0 @ v - vals[0].real * v) / np.linalg.norm(v)), info
dt, (vals, res, info) = timed(cfg1)
print(json.dumps({"config": "configs[0]: eigs, 1000x1000 dense real(dp), kdim 30, nev 4", "seconds": dt, "arnoldi_steps": int(info),
                  "steps_per_s": info / dt, "leading_eigenvalue": [float(vals[0].real), float(vals[0].imag)], "true_residual_of_leading_pair": res}), flush=True)
del A

# configs[1]: arnoldi, diagonal operator, n = 1e7 real(dp), m = 64
n, m = 10_000_000, 64
A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=ctx)
X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
H = np.zeros((m + 1, m), order="F")
def cfg2():
    X[0].rand(True, seed=7)
    return lk.arnoldi(A, X, H)
dt, info = timed(cfg2, 3)
b = dgs_bytes(8, n, range(1, m + 1))
print(json.dumps({"config": "configs[1]: arnoldi, diagonal operator, n = 1e7 real(dp), m = 64", "seconds": dt, "info": int(info),
                  "iters_per_s": m / dt, "dgs_algorithmic_GB": b / 1e9, "whole_solve_frac_of_8TBps_on_dgs_bytes": b / dt / 8e12}), flush=True)
del A, X

# configs[2]: gmres on the 5-point Laplacian, N = 4096 (n = 16.8 M), GMRES(30), maxiter = 2 (3 cycles = 90 inner steps)
N = 4096; n = N * N
A = lk.laplacian2d_linop_gpu(N, ctx)
bvec = lk.dense_vector_gpu(n, np.float64, ctx); bvec.rand(False, seed=11)
def cfg3():
    x = lk.dense_vector_gpu(n, np.float64, ctx)
    meta = lk.gmres_dp_metadata()
    info = lk.gmres(A, bvec, x, rtol=1e-8, options=lk.gmres_dp_opts(kdim=30, maxiter=2), meta=meta)
    return info, meta
dt, (info, meta) = timed(cfg3)
b = dgs_bytes(8, n, list(range(1, 31)) * 3)
print(json.dumps({"config": "configs[2]: gmres(30) x 3 cycles, 5-point Laplacian 4096^2, real(dp)", "seconds": dt, "info": int(info),
                  "inner_steps": int(meta.n_inner), "steps_per_s": meta.n_inner / dt, "residual_first_last": [meta.res[0], meta.res[-1]],
                  "dgs_algorithmic_GB": b / 1e9, "whole_solve_frac_of_8TBps_on_dgs_bytes": b / dt / 8e12}), flush=True)
# the same solve with the Laplacian as an explicit sparse matrix (CSR, 5 entries per row) instead of the matrix-free stencil
import scipy.sparse as sp
T = sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1])
S = sp.diags([-np.ones(N - 1), -np.ones(N - 1)], [-1, 1])
Acsr = ((sp.kron(sp.identity(N), T) + sp.kron(S, sp.identity(N))) * float((N + 1) ** 2)).tocsr()
A_stencil, A = A, lk.csr_linop_gpu(Acsr, ctx)
xx, yy = lk.dense_vector_gpu(n, np.float64, ctx), lk.dense_vector_gpu(n, np.float64, ctx)
xx.rand(False, seed=3)
mv = {}
for name, op in (("stencil", A_stencil), ("csr", A)):
    t, _ = timed(lambda: [op.matvec(xx, yy) for _ in range(20)], 3)
    mv[name] = t / 20
csr_bytes = Acsr.nnz * 12.0 + n * (8 + 8 + 8)            # values + column indices; row pointers, x (gathered, cached), y
dt, (info, meta) = timed(cfg3)
print(json.dumps({"config": "configs[2] with the operator as a CSR sparse matrix (84 M non-zeros)", "seconds": dt, "info": int(info),
                  "inner_steps": int(meta.n_inner), "steps_per_s": meta.n_inner / dt, "residual_first_last": [meta.res[0], meta.res[-1]],
                  "matvec_ms": {k: v * 1e3 for k, v in mv.items()},
                  "csr_matvec_GBps_on_12B_per_nonzero_plus_24B_per_row": csr_bytes / mv["csr"] / 1e9,
                  "stencil_matvec_GBps_on_16B_per_row": 16.0 * n / mv["stencil"] / 1e9}), flush=True)
del A, A_stencil, bvec, Acsr, xx, yy

# configs[3]: eigs on the Ginzburg-Landau stepper (one RK4 step of tau = 0.01), n = 1e6 complex(dp), kdim = 128, nev = 8:
# (a) the 128-step factorisation alone (lk_arnoldi, asynchronous), (b) eigs itself, one Krylov-Schur cycle + restart
n, kdim, nev = 1_000_000, 128, 8
A = lk.ginzburg_landau_linop_gpu(n, ctx, tau=0.01, nsub=1)
X = lk.krylov_basis_gpu(n, kdim + 1, np.complex128, ctx)
H = np.zeros((kdim + 1, kdim), dtype=np.complex128, order="F")
def cfg4a():
    X[0].rand(True, seed=13)
    return lk.arnoldi(A, X, H)
dt, info = timed(cfg4a, 3)
b = dgs_bytes(16, n, range(1, kdim + 1))
print(json.dumps({"config": "configs[3]a: 128-step arnoldi, Ginzburg-Landau RK4 stepper, n = 1e6 complex(dp)", "seconds": dt, "info": int(info),
                  "iters_per_s": kdim / dt, "dgs_algorithmic_GB": b / 1e9, "whole_solve_frac_of_8TBps_on_dgs_bytes": b / dt / 8e12}), flush=True)
del X
def cfg4b():
    V = lk.krylov_basis_gpu(n, nev, np.complex128, ctx)
    x0 = lk.dense_vector_gpu(n, np.complex128, ctx); x0.rand(False, seed=13)
    return lk.eigs(A, V, x0=x0, kdim=kdim, tolerance=1e-10, max_restarts=0)
dt, (vals, res, info) = timed(cfg4b)
print(json.dumps({"config": "configs[3]b: eigs (one Krylov-Schur cycle: 128 steps with a host eig each + restart + eigenvectors), same operator",
                  "seconds": dt, "arnoldi_steps": int(info), "steps_per_s": info / dt, "dgs_algorithmic_GB": b / 1e9,
                  "whole_solve_frac_of_8TBps_on_dgs_bytes": b / dt / 8e12}), flush=True)
ctx.close()
