#!/usr/bin/env python3
"""The reference's Ginzburg-Landau demo (example/ginzburg_landau/main.f90) on the MI355X engine: leading right and left
eigenpairs of the linearised complex Ginzburg-Landau propagator by Krylov-Schur `eigs`, the spectrum mapped back with
log(lambda)/tau and written in the reference's `.npy` layout (so example/ginzburg_landau/eigenplots.py reads it).

  python examples/ginzburg_landau.py [nx=512] [outdir=.]

Same call sequence as the Fortran program: eigs(A, X, lambda, residuals, info, kdim=2*nev) for the direct problem, the same
with transpose=.true. for the adjoint one; the operator is the engine's RK4 propagator with the reference's parameters
(Ginzburg_Landau.f90:24-33)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
outdir = sys.argv[2] if len(sys.argv) > 2 else "."
tau, nev = 1.0, 8                                   # unit sampling time (40 RK4 sub-steps), 8 eigenvalues as in main.f90:27

ctx = lk.Context(device=0)
A = lk.ginzburg_landau_linop_gpu(nx, ctx, tau=tau, nsub=40)
for transpose, tag in ((False, ""), (True, "adjoint_")):
    X = lk.krylov_basis_gpu(nx, nev, np.complex128, ctx)                       # allocate (X(nev)); call zero_basis(X)
    lam, residuals, info = lk.eigs(A, X, kdim=2 * nev, transpose=transpose)    # main.f90:69 / :91
    lam = np.log(lam) / tau                                                    # unit disk -> complex plane (main.f90:73)
    lk.save_eigenspectrum(lam, residuals, os.path.join(outdir, f"{tag}eigenspectrum.npy"))
    with open(os.path.join(outdir, f"{tag}eigenvectors.npy"), "wb") as f:
        np.save(f, np.asfortranarray(X.download()))
    print(f"{'left' if transpose else 'right'} eigenpairs: {info} Arnoldi steps")
    for l, r in zip(lam, residuals):
        print(f"   lambda = {l.real:+.10f} {l.imag:+.10f}i   residual {r:.2e}")
ctx.close()
