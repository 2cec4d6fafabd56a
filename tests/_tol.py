"""Tolerances for Hessenberg entries and Ritz values, stated once (north_star: 1e-12 rtol in double precision).

Hessenberg / tridiagonal / bidiagonal columns are compared NORMWISE per column at 1e-12, bare.

Ritz values (eigenvalues of the small projected matrix) are functions of H whose sensitivity is the eigenvalue condition number
kappa_i = 1 / |y_i^H x_i| (unit left / right eigenvectors of H): two factorisations whose H agree to a relative eps in norm have
Ritz values that agree to kappa_i * eps * ||H||_2 to first order (Bauer-Fike / Wilkinson).  The bound used here is therefore
    |lambda_i - lambda_i'| <= 1e-12 * max(1, kappa_i) * ||H||_2,
with kappa_i COMPUTED from the matrix under test and printed; where kappa_i = 1 (normal H) it is the bare 1e-12.

Every comparison appends (label, measured, bound, margin) to $LK_TOL_REPORT when that variable names a file: the GPU run's
evidence of how far inside the bounds the engine is (profiles/r03_parity_margins.txt)."""
import os

import numpy as np
import scipy.linalg as sla

RTOL = 1e-12


def _report(label, measured, bound, extra=""):
    path = os.environ.get("LK_TOL_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"{label}\tmeasured={measured:.3e}\tbound={bound:.3e}\tmargin={bound / max(measured, 1e-300):.1f}x\t{extra}\n")


def column_errors(H, Ho):
    """max |H(:, j) - Ho(:, j)| / max |Ho(:, j)| per column"""
    return np.array([np.abs(H[:, j] - Ho[:, j]).max() / max(np.abs(Ho[:, j]).max(), 1e-300) for j in range(Ho.shape[1])])


def assert_columns_close(H, Ho, label, rtol=RTOL):
    """every column of the projected matrix normwise within rtol (1e-12 unless the caller states why not)"""
    err = column_errors(H, Ho)
    _report(label + " [H columns]", float(err.max()), rtol, f"worst column {int(err.argmax()) + 1} of {len(err)}")
    assert err.max() <= rtol, f"{label}: column {int(err.argmax()) + 1} differs by {err.max():.2e} > {rtol:.1e}"
    return float(err.max())


def assert_close(got, ref, label, rtol=RTOL, scale=None, kappa=1.0):
    """max |got - ref| <= rtol * max(1, kappa) * scale (scale: max |ref| unless given).  For the quantities that are functions of
    the projected matrix with condition number 1 -- eigenvalues of a symmetric T, singular values of B, the Givens residual
    history relative to |r0| -- and for the GMRES solutions of this suite (measured <= 1.5e-14, profiles/r04_parity_margins.txt)
    kappa stays 1 and this is the bare 1e-12; a caller that passes a kappa computes it from the matrix under test and says so."""
    got, ref = np.asarray(got), np.asarray(ref)
    sc = float(np.abs(ref).max()) if scale is None else float(scale)
    err = float(np.abs(got - ref).max()) / max(sc, 1e-300)
    bound = rtol * max(1.0, float(kappa))
    _report(label, err, bound, f"kappa {float(kappa):.2e}" if kappa != 1.0 else "bare")
    assert err <= bound, f"{label}: differs by {err:.2e} > {rtol:.0e} * kappa ({float(kappa):.2e})"
    return err


def ritz_condition(Hm):
    """eigenvalues of Hm and their condition numbers 1 / |y^H x| (unit left / right eigenvectors)"""
    w, vl, vr = sla.eig(Hm, left=True, right=True)
    den = np.abs(np.sum(vl.conj() * vr, axis=0)) / (np.linalg.norm(vl, axis=0) * np.linalg.norm(vr, axis=0))
    return w, 1.0 / np.maximum(den, 1e-300)


def assert_ritz_close(vals, ref, Hm, label, rtol=RTOL, top=None):
    """every value of `ref` (the leading `top` by modulus when given) has a partner in `vals` within
    rtol * max(1, kappa_i) * ||Hm||_2, kappa_i the condition number of the eigenvalue of Hm nearest to it."""
    vals, ref = np.asarray(vals, dtype=complex), np.asarray(ref, dtype=complex)
    w, kap = ritz_condition(np.asarray(Hm))
    hn = np.linalg.norm(Hm, 2)
    order = np.argsort(-np.abs(ref))
    if top is not None:
        order = order[:top]
    worst, worst_ratio, kmax = 0.0, 0.0, 1.0
    for i in order:
        d = np.abs(vals - ref[i]).min()
        k = max(1.0, float(kap[np.abs(w - ref[i]).argmin()]))
        bound = rtol * k * hn
        worst, kmax = max(worst, d / hn), max(kmax, k)
        worst_ratio = max(worst_ratio, d / bound)
        assert d <= bound, f"{label}: Ritz value {ref[i]:.6g} off by {d:.2e} > 1e-12 * kappa ({k:.2e}) * ||H|| ({hn:.3g})"
    _report(label + " [Ritz values]", worst, rtol * kmax, f"max kappa {kmax:.2e}, worst |d|/bound {worst_ratio:.2e}, ||H||_2 {hn:.3g}")
    return worst, kmax
