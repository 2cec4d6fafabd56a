#!/usr/bin/env python3
"""Generates tests/golden/*.npz: OUTPUTS of the CPU oracle (reference schedule) on seeded inputs that
every backend regenerates from the shared counter RNG / closed formulas, so only results are stored.
Provenance: oracle/ (pinned by the reference's KATs and by the reference-run values recorded in
SURVEY.md Appendix A).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as ora  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def seeded(n, dtype, seed):
    x = np.empty(n, dtype=dtype)
    ora.fill_counter(x, seed)
    return x


def diag_values(n, dtype):
    d = (1.0 + np.arange(n) / n).astype(dtype)
    if np.dtype(dtype).kind == "c":
        d = d * np.exp(1j * 0.3 * np.arange(n) / n)
    return d


def arnoldi_diag(n, m, dtype, seed=7):
    d = diag_values(n, dtype)
    X = np.zeros((n, m + 1), dtype=dtype, order="F")
    x0 = seeded(n, dtype, seed)
    X[:, 0] = x0 / np.linalg.norm(x0)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    info = ora.arnoldi(ora.DiagOp(d), X, H)
    return dict(H=H, info=info, ritz=np.sort_complex(np.linalg.eigvals(H[:m, :m])), n=n, m=m, seed=seed,
                orth=np.abs(X.conj().T @ X - np.eye(m + 1)).max())


def cfg1_matrix():
    rng = np.random.default_rng(1)                       # numpy PCG64(seed=1), SURVEY 8d cfg1
    A = rng.standard_normal((1000, 1000)) / np.sqrt(1000)
    A[np.arange(4), np.arange(4)] += np.array([2.0, 1.8, 1.6, 1.4])
    x0 = np.random.default_rng(2).standard_normal(1000)
    return np.asfortranarray(A), x0 / np.linalg.norm(x0)


GL_REF = dict(n=512, dx=200.0 / 513.0, tau=0.01, nsub=1, nu=2.0 + 0.2j, gamma=1.0 - 1.0j, mu_c=0.38 - 0.2 ** 2, mu2=-0.01,
              nev=8, kdim=16, seed=13)


def gl_reference_size():
    """SURVEY 8(c) fixture (5): the reference's Ginzburg-Landau example at ITS OWN size and parameters
    (example/ginzburg_landau/Ginzburg_Landau.f90:23-33: nx = 512, L = 200, nu = 2 + 0.2i, gamma = 1 - i, mu_0 = 0.38, c_mu = 0.2,
    mu_2 = -0.01; main.f90:20,27,66: tau = 0.01, nev = 8, kdim = 2 nev = 16), operator = one classical RK4 step of the reference
    right-hand side incl. its boundary rows (:126-136) -- the reference integrates with rklib's adaptive scheme, an un-vendored
    dependency.  Stored: H(:17, :16) of the first 16-step Arnoldi factorisation, its Ritz values, and what
    eigs(nev = 8, kdim = 16, tolerance = rtol_dp) returns (8 leading eigenvalues, residuals, info)."""
    g = GL_REF
    A = ora.GLOp(g["n"], g["dx"], g["tau"], g["nsub"], g["nu"], g["gamma"], g["mu_c"], g["mu2"])
    x0 = seeded(g["n"], np.complex128, g["seed"])
    m = g["kdim"]
    X = np.zeros((g["n"], m + 1), dtype=np.complex128, order="F")
    X[:, 0] = x0 / np.linalg.norm(x0)
    H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    info = ora.arnoldi(A, X, H)
    vals, res, _V, niter = ora.eigs(A, x0, g["nev"], m)
    return dict(H=H, info=info, ritz=np.linalg.eigvals(H[:m, :m]), eig_vals=vals, eig_res=res, eig_niter=niter)


def main():
    for n, m in ((1000, 8), (100_000, 64), (20_011, 128)):
        for dtype, tag in ((np.float64, "rdp"), (np.complex128, "cdp")):
            np.savez(os.path.join(OUT, f"arnoldi_diag_n{n}_m{m}_{tag}.npz"), **arnoldi_diag(n, m, dtype))
    # cfg1: dense 1000 x 1000, m = 30, then eigs(nev=4, kdim=30, tol=1e-10)
    A, x0 = cfg1_matrix()
    X = np.zeros((1000, 31), order="F"); X[:, 0] = x0
    H = np.zeros((31, 30), order="F")
    info = ora.arnoldi(ora.DenseOp(A), X, H)
    vals, res, _V, niter = ora.eigs(ora.DenseOp(A), x0, 4, 30, 1e-10)
    np.savez(os.path.join(OUT, "cfg1_dense1000_m30_rdp.npz"), H=H, info=info, eig_vals=vals, eig_res=res, eig_niter=niter)
    # Poisson 64^2, GMRES(30), maxiter=2, rtol=1e-8, b from seed 11
    N = 64
    b = seeded(N * N, np.float64, 11)
    x = np.zeros(N * N)
    ginfo, hist = ora.gmres(ora.Lap5Op(N), b, x, rtol=1e-8, kdim=30, maxiter=2)
    np.savez(os.path.join(OUT, "gmres_poisson64_k30.npz"), info=ginfo, res=hist, x_norm=np.linalg.norm(x), x_head=x[:64])
    np.savez(os.path.join(OUT, "gl_nx512_kdim16_cdp.npz"), **gl_reference_size())
    # reference-run values recorded by the survey: reference arnoldi,
    # n=1000, m=8, d_i = 1+(i-1)/n, x0_i = sin(i)/||.||
    np.savez(os.path.join(OUT, "survey_reference_run_n1000_m8.npz"), H11=1.4991929804973552,
             H21=0.28878608972273856, Hlast=0.2505741778943683)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
