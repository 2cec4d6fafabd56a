"""On-disk outputs of the eigensolvers, byte-compatible with the reference so downstream scripts
(example/ginzburg_landau/eigenplots.py) keep working.  Host-only file I/O, outside the hot path.

  write_results     src/IterativeSolvers/IterativeSolvers.fypp:881-925  (`eigs_output.txt` table)
  save_eigenspectrum                                           :944-963  (`.npy`, n x 3 / n x 2 real array)
"""
from __future__ import annotations

import numpy as np

eigs_output = "eigs_output.txt"          # IterativeSolvers.fypp:44


def _E(x: float, w: int = 16, d: int = 9) -> str:
    """Fortran Ew.d edit descriptor (0.dddddddddE+xx)."""
    if x == 0.0 or not np.isfinite(x):
        body = f"{0.0:.{d}f}E+00" if x == 0.0 else str(x)
    else:
        e = int(np.floor(np.log10(abs(x)))) + 1
        m = abs(x) / 10.0 ** e
        if round(m, d) >= 1.0:
            m /= 10.0
            e += 1
        body = f"{m:.{d}f}E{'+' if e >= 0 else '-'}{abs(e):02d}"
    if x < 0:
        body = "-" + body
    return body.rjust(w)


def write_results(filename: str, vals: np.ndarray, res: np.ndarray, tol: float) -> None:
    """Table of intermediate Ritz values, sorted by residual.  NB like the reference (`sort_index(res, indices)`
    with `res` intent(inout), :906) this SORTS `res` IN PLACE."""
    k = vals.size
    indices = np.argsort(res, kind="stable")
    res[:] = res[indices]
    cplx = np.iscomplexobj(vals)
    with open(filename, "w") as f:
        if cplx:
            f.write(f"{'Iter':>6}" + "".join(f"{h:>18}" for h in ("Re", "Im", "modulus", "residual")) + f"{'conv':>6}\n")
        else:
            f.write(f"{'Iter':>6}" + "".join(f"{h:>18}" for h in ("value", "residual")) + f"{'conv':>6}\n")
        for i in range(k):
            v = vals[indices[i]]
            conv = "T" if res[i] < tol else "F"
            if cplx:
                cols = (v.real, v.imag, float(np.sqrt(v.real ** 2 + v.imag ** 2)), res[i])
            else:
                cols = (float(v), res[i])
            f.write(f"{k:6d}" + "".join("  " + _E(c) for c in cols) + "  " + conv.rjust(4) + "\n")


def save_eigenspectrum(lam: np.ndarray, residuals: np.ndarray, fname: str) -> None:
    """`.npy` with columns (Re, Im, residual) for complex eigenvalues, (value, residual) for real ones.
    stdlib's save_npy writes Fortran-ordered real(dp) arrays; numpy reads either order transparently."""
    lam = np.asarray(lam)
    if np.iscomplexobj(lam):
        arr = np.column_stack([lam.real, lam.imag, np.asarray(residuals, dtype=float)])
    else:
        arr = np.column_stack([lam.astype(float), np.asarray(residuals, dtype=float)])
    with open(fname, "wb") as f:            # np.save would append ".npy"; the reference writes `fname` as given
        np.save(f, np.asfortranarray(arr))
