"""Krylov factorisations -- host-side mirror of src/Krylov/*.fypp (LightKrylov_BaseKrylov).

Same names, argument meaning and `info` convention as the reference (0 ok, >0 informational,
<0 failure; BaseKrylov.fypp:106-109).  Out-arguments that Fortran passes as arrays (`H`,
`R`, `T`, `beta`) are numpy arrays filled in place; `info` is the return value.  Indices in
`kstart` / `kend` are 1-based like the reference; bases are 0-based python sequences
(`X[j]` is the reference's `X(j+1)`).

For `krylov_basis_gpu` bases the orthogonalisation is ONE engine call (three fused panel
sweeps); for any other `abstract_vector` implementation the reference's generic loops run
over the type-bound procedures.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
from scipy.linalg import lapack as _lapack

from . import _capi, _hostlapack
from .constants import atol_dp, rtol_dp
from .linops import _engine_linop, abstract_linop
from .vectors import (Gram, abstract_vector, copy, dense_vector_gpu, innerprod, krylov_basis_gpu,
                      linear_combination, zero_basis)

_DP = C.POINTER(C.c_double)


def _check_beta(beta, shape, what):
    """assert_shape(beta, shape(proj_coefficients), ...)  gram_schmidt.fypp:51-53"""
    if beta is not None and tuple(beta.shape) != tuple(shape):
        raise ValueError(f"{what}: beta has shape {beta.shape}, expected {shape}")


def is_orthonormal(X) -> bool:
    """src/Krylov/utilities.fypp:86-99 (compares against rtol_sp for every kind)."""
    G = Gram(X)
    k = G.shape[0]
    return bool(np.linalg.norm(G - np.eye(k), "fro") < 10.0 ** (-3))  # rtol_sp = sqrt(1e-6)


# ------------------------------------------------------------------------------------------
def orthogonalize_against_basis(y, X, if_chk_orthonormal: bool = True, beta: np.ndarray | None = None) -> int:
    """One classical Gram-Schmidt pass of y (vector or basis) against X.
    src/Krylov/gram_schmidt.fypp:113-200."""
    if if_chk_orthonormal and not is_orthonormal(X):
        raise RuntimeError("Input basis not orthonormal.")
    k = len(X)
    if isinstance(X, krylov_basis_gpu) and isinstance(y, dense_vector_gpu):
        _check_beta(beta, (k,), "orthogonalize_against_basis")
        h = np.zeros(k, dtype=X.dtype)
        info = C.c_int()
        _capi.check(X._lib.lk_orthogonalize(X._h, k, y.basis._h, y.col, h.ctypes.data_as(_DP), C.byref(info)))
        if beta is not None:
            beta[...] = h
        return info.value
    if isinstance(y, abstract_vector):
        _check_beta(beta, (k,), "orthogonalize_against_basis")
        info = 1 if y.norm() < atol_dp else 0                 # :126-127
        h = innerprod(X, y)                                   # :141
        proj = linear_combination(X, h)                       # :144
        y.sub(proj)                                           # :145
        if beta is not None:
            beta[...] = h
        return info
    # basis against basis (:156-200)
    p = len(y)
    _check_beta(beta, (k, p), "orthogonalize_against_basis")
    info = 0
    for j in range(p):
        hj = np.zeros(k, dtype=beta.dtype if beta is not None else complex)
        ij = orthogonalize_against_basis(y[j], X, False, hj)
        if ij:
            info = j + 1
        if beta is not None:
            beta[:, j] = hj
    return info


def double_gram_schmidt_step(y, X, if_chk_orthonormal: bool = True, beta: np.ndarray | None = None,
                             _normalize: bool = False, _norms: list | None = None) -> int:
    """double_gram_schmidt_step(y, X, info, if_chk_orthonormal, beta): two CGS passes, beta = h1 + h2.
    src/Krylov/gram_schmidt.fypp:12-105; interface BaseKrylov.fypp:634-712.  Returns info.
    (`_normalize` / `_norms` are engine extras used by arnoldi: fold the following norm + scal.)"""
    if if_chk_orthonormal and not is_orthonormal(X):          # default .true. like the reference
        raise RuntimeError("Input basis not orthonormal.")
    k = len(X)
    if isinstance(X, krylov_basis_gpu) and isinstance(y, dense_vector_gpu):
        _check_beta(beta, (k,), "double_gram_schmidt_step")
        h = np.zeros(k, dtype=X.dtype)
        norms = (C.c_double * 3)()
        info = C.c_int()
        _capi.check(X._lib.lk_dgs(X._h, k, y.basis._h, y.col, h.ctypes.data_as(_DP), norms,
                                  _capi.LK_DGS_NORMALIZE if _normalize else 0, C.byref(info)))
        if beta is not None:
            beta[...] = h
        if _norms is not None:
            _norms[:] = [norms[0], norms[1], norms[2]]
        return info.value
    if isinstance(X, krylov_basis_gpu) and isinstance(y, krylov_basis_gpu):
        p = len(y)
        _check_beta(beta, (k, p), "double_gram_schmidt_step")
        h = np.zeros((k, p), dtype=X.dtype, order="F")
        info = C.c_int()
        _capi.check(X._lib.lk_dgs_block(X._h, k, y._h, 0, p, h.ctypes.data_as(_DP), C.byref(info)))
        if beta is not None:
            beta[...] = h
        return info.value
    # generic abstract_vector path: the reference's schedule
    isvec = isinstance(y, abstract_vector)
    shape = (k,) if isvec else (k, len(y))
    _check_beta(beta, shape, "double_gram_schmidt_step")
    dt = beta.dtype if beta is not None else complex
    h1 = np.zeros(shape, dtype=dt)
    h2 = np.zeros(shape, dtype=dt)
    orthogonalize_against_basis(y, X, False, h1)               # :40-43
    info = orthogonalize_against_basis(y, X, False, h2)        # :45-47
    if beta is not None:
        beta[...] = h1 + h2                                    # :49
    return info


def _panel_columns(Q):
    """Q -> (panel, first column, count) when Q is (a view of) ONE device panel, else None"""
    if isinstance(Q, krylov_basis_gpu):
        return Q, 0, len(Q)
    return None


# ------------------------------------------------------------------------------------------
def qr(Q, R: np.ndarray, tol: float = atol_dp) -> int:
    """qr_no_pivoting: in-place DGS-based QR of the basis Q, R upper triangular.
    src/Krylov/qr.fypp:116-167.  Returns info (index of the first colinear column, 1-based)."""
    # columns of ONE device panel: the whole factorisation inside the engine (lk_qr)
    cols = _panel_columns(Q)
    if cols is not None and R.flags.f_contiguous and R.dtype == cols[0].dtype and R.shape[0] >= len(Q):
        B, j0, p = cols
        cinfo = C.c_int()
        _capi.check(B._lib.lk_qr(B._h, j0, p, R.ctypes.data_as(_DP), R.shape[0], float(tol), C.byref(cinfo)))
        return cinfo.value
    info, flag = 0, False
    R[...] = 0
    for j in range(len(Q)):
        qj = Q[j]
        if j > 0:
            bj = np.zeros(j, dtype=R.dtype)
            double_gram_schmidt_step(qj, Q[:j], if_chk_orthonormal=False, beta=bj)   # :131-134
            R[:j, j] = bj
        beta = qj.norm()
        if np.isnan(beta):
            raise FloatingPointError("|beta| = NaN detected! Abort")               # :137-143
        if abs(beta) < tol:
            if not flag:
                flag, info = True, j + 1
            R[j, j] = 0
            qj.rand()
            if j > 0:
                double_gram_schmidt_step(qj, Q[:j], if_chk_orthonormal=False)
            beta = qj.norm()
        else:
            R[j, j] = beta
        qj.scal(1.0 / beta)                                                         # :164
    return info


# ------------------------------------------------------------------------------------------
def arnoldi(A: abstract_linop, X, H: np.ndarray, kstart: int = 1, kend: int | None = None,
            tol: float = atol_dp, transpose: bool = False, blksize: int = 1, _segments=None, _progress=None) -> int:
    """(block) Arnoldi factorisation A X(:, :k) = X(:, :k+1) H(:k+1, :k).
    src/Krylov/arnoldi.fypp:8-76.  X holds (kdim+1)*blksize vectors, H is ((kdim+1)p, kdim p).
    Returns info: 0, or kp when the residual block is below tol (invariant subspace).
    (`_segments` / `_progress`: engine extra on the fused path -- lk_arnoldi_segments: `_progress(kfirst, klast)` is called as soon as the
    columns kfirst..klast of H are final, segment by segment (`_segments` = last step of each), while the device runs the later steps.)"""
    p = int(blksize)
    kdim = (len(X) - p) // p                                                       # :26
    kend = kdim if kend is None else kend
    info = 0

    # whole step loop inside the engine: operator kernel -> 3 DGS sweeps -> normalise
    if (p == 1 and isinstance(X, krylov_basis_gpu) and isinstance(A, _engine_linop)
            and H.flags.f_contiguous and H.dtype == X.dtype and H.shape[0] >= kdim + 1):
        cinfo = C.c_int()
        if _progress is not None:
            segs = [int(b) for b in (_segments or []) if kstart <= int(b) <= kend]
            arr = (C.c_int * max(len(segs), 1))(*segs)
            failure = []

            reported = [int(kstart) - 1]

            def _cb(_user, kfirst, klast):
                reported[0] = int(klast)
                try:
                    return 1 if _progress(int(kfirst), int(klast)) else 0          # a true return value asks the engine to stop
                except BaseException as exc:  # noqa: BLE001 - must not propagate through C; re-raised below
                    failure.append(exc)
                    return 1
            cb = _capi.PROGRESS_FN(_cb)
            _capi.check(X._lib.lk_arnoldi_segments(A._h, X._h, H.ctypes.data_as(_DP), H.shape[0], int(kstart), int(kend), float(tol),
                                                   1 if transpose else 0, arr, len(segs), cb, None, C.byref(cinfo)))
            if failure:
                raise failure[0]
            n_rep = reported[0] - int(kstart) + 1                                  # (a stopped factorisation: the steps reported; up to 24 more ran)
            if transpose:
                A.rmatvec_counter += max(n_rep, 0)
            else:
                A.matvec_counter += max(n_rep, 0)
            return cinfo.value
        else:
            _capi.check(X._lib.lk_arnoldi(A._h, X._h, H.ctypes.data_as(_DP), H.shape[0], int(kstart), int(kend),
                                          float(tol), 1 if transpose else 0, C.byref(cinfo)))
        n_steps = (cinfo.value if cinfo.value else kend) - kstart + 1
        if transpose:
            A.rmatvec_counter += max(n_steps, 0)
        else:
            A.matvec_counter += max(n_steps, 0)
        return cinfo.value

    # block factorisation inside the engine: p operator applications, the panel x panel Gram-Schmidt and the p-column qr of every step
    # enqueued asynchronously, one host synchronisation per call (lk_arnoldi_block)
    if (p > 1 and isinstance(X, krylov_basis_gpu) and isinstance(A, _engine_linop) and _progress is None
            and H.flags.f_contiguous and H.dtype == X.dtype and H.shape[0] >= (kdim + 1) * p):
        cinfo = C.c_int()
        _capi.check(X._lib.lk_arnoldi_block(A._h, X._h, H.ctypes.data_as(_DP), H.shape[0], p, int(kstart), int(kend), float(tol),
                                            1 if transpose else 0, C.byref(cinfo)))
        n_steps = (cinfo.value // p if cinfo.value else kend) - kstart + 1
        if transpose:
            A.rmatvec_counter += max(n_steps, 0) * p
        else:
            A.matvec_counter += max(n_steps, 0) * p
        return cinfo.value

    gpu = isinstance(X, krylov_basis_gpu)
    for k in range(kstart, kend + 1):
        kpm, kp, kpp = (k - 1) * p, k * p, (k + 1) * p                              # :36
        for i in range(p):                                                          # :39-47
            if transpose:
                A.apply_rmatvec(X[kpm + i], X[kp + i])
            else:
                A.apply_matvec(X[kpm + i], X[kp + i])
        if p == 1:
            hcol = np.zeros(kp, dtype=H.dtype)
            norms: list = []
            if gpu:
                double_gram_schmidt_step(X[kp], X[:kp], False, hcol, _normalize=True, _norms=norms)  # :50-55 fused
                H[:kp, kpm] = hcol
                beta = norms[2]
                if beta < atol_dp:                                                  # qr.fypp:146-162
                    H[kp, kpm] = 0
                    X[kp].rand(True, seed=0x5EED + k)
                else:
                    H[kp, kpm] = beta
            else:
                double_gram_schmidt_step(X[kp], X[:kp], False, hcol)                # :50-52
                H[:kp, kpm] = hcol
                Rb = np.zeros((1, 1), dtype=H.dtype)
                qr([X[kp]], Rb)                                                     # :55
                H[kp, kpm] = Rb[0, 0]
        else:
            hblk = np.zeros((kp, p), dtype=H.dtype, order="F")
            double_gram_schmidt_step(X[kp:kpp], X[:kp], False, hblk)
            H[:kp, kpm:kp] = hblk
            Rb = np.zeros((p, p), dtype=H.dtype, order="F")
            qr(X[kp:kpp], Rb)
            H[kp:kpp, kpm:kp] = Rb
        res = np.array([np.real(H[kp + i, kpm + i]) for i in range(p)])            # :58-62
        if np.min(np.abs(res)) < tol:                                               # :65-71
            info = kp
            break
    return info


def lanczos(A: abstract_linop, X, T: np.ndarray, kstart: int = 1, kend: int | None = None,
            tol: float = atol_dp) -> int:
    """lanczos_tridiagonalization for symmetric / Hermitian operators.  src/Krylov/lanczos.fypp:7-64."""
    kdim = len(X) - 1
    kend = kdim if kend is None else kend
    info = 0
    # whole step loop inside the engine (asynchronous, one synchronisation per call, for the first 512 basis columns; one round trip per
    # step beyond)
    if (isinstance(X, krylov_basis_gpu) and isinstance(A, _engine_linop) and T.flags.f_contiguous and T.dtype == X.dtype
            and T.shape[0] >= kdim + 1 and kstart <= kend):
        cinfo = C.c_int()
        _capi.check(X._lib.lk_lanczos(A._h, X._h, T.ctypes.data_as(_DP), T.shape[0], int(kstart), int(kend), float(tol),
                                      C.byref(cinfo)))
        A.matvec_counter += (cinfo.value if cinfo.value else kend) - kstart + 1
        return cinfo.value
    for k in range(kstart, kend + 1):
        A.apply_matvec(X[k - 1], X[k])                                              # :26
        for i in range(max(1, k - 1), k + 1):                                       # :57-60
            T[i - 1, k - 1] = X[i - 1].dot(X[k])
            X[k].axpby(-T[i - 1, k - 1], X[i - 1], 1.0)
        double_gram_schmidt_step(X[k], X[:k], if_chk_orthonormal=False)             # :62
        beta = X[k].norm()
        T[k, k - 1] = beta                                                          # :29
        if beta < tol:
            info = k
            break
        X[k].scal(1.0 / beta)                                                       # :39
    return info


def bidiagonalization(A: abstract_linop, U, V, B: np.ndarray, kstart: int = 1, kend: int | None = None,
                      tol: float = atol_dp) -> int:
    """lanczos_bidiagonalization (Golub-Kahan): A V = U B, full re-orthogonalisation of both bases.
    src/Krylov/golub_kahan.fypp:7-64.  U has kdim+1 vectors, V kdim(+1), B is (kdim+1, kdim)."""
    kdim = len(U) - 1
    kend = kdim if kend is None else kend
    info = 0
    gpu = isinstance(U, krylov_basis_gpu) and isinstance(V, krylov_basis_gpu)
    # whole step loop inside the engine (asynchronous, one synchronisation per call, for the first 512 basis columns; one round trip per
    # step beyond)
    if (gpu and isinstance(A, _engine_linop) and B.flags.f_contiguous and B.dtype == U.dtype and B.shape[0] >= kdim + 1
            and tol >= atol_dp and kstart <= kend and len(V) >= kdim):
        cinfo = C.c_int()
        _capi.check(U._lib.lk_bidiag(A._h, U._h, V._h, B.ctypes.data_as(_DP), B.shape[0], int(kstart), int(kend), float(tol),
                                     C.byref(cinfo)))
        ksteps = (cinfo.value if cinfo.value else kend) - kstart + 1
        A.rmatvec_counter += ksteps
        A.matvec_counter += ksteps if not cinfo.value or B[cinfo.value - 1, cinfo.value - 1].real > tol else ksteps - 1
        return cinfo.value
    for k in range(kstart, kend + 1):
        A.apply_rmatvec(U[k - 1], V[k - 1])                                         # :27
        norms: list = []
        if k > 1:
            double_gram_schmidt_step(V[k - 1], V[:k - 1], if_chk_orthonormal=False,  # :30-33
                                     _norms=norms if gpu else None)
        alpha = norms[2] if norms else V[k - 1].norm()                              # :36
        B[k - 1, k - 1] = alpha
        if abs(alpha) > tol:
            V[k - 1].scal(1.0 / alpha)
        else:
            info = k
            break
        A.apply_matvec(V[k - 1], U[k])                                              # :45
        norms = []
        double_gram_schmidt_step(U[k], U[:k], if_chk_orthonormal=False,             # :48-49
                                 _norms=norms if gpu else None)
        beta = norms[2] if norms else U[k].norm()                                   # :52
        B[k, k - 1] = beta
        if abs(beta) > tol:
            U[k].scal(1.0 / beta)
        else:
            info = k
            break
    return info


# ------------------------------------------------------------------------------------------
def _schur(Hm: np.ndarray):
    """stdlib `schur` = LAPACK gees without sorting (BaseKrylov.fypp:807).  Through ctypes when the library is reachable
    (bit-identical to scipy's wrapper, and the interpreter lock is free meanwhile: _hostlapack.gees)."""
    if _hostlapack.threaded():
        return _hostlapack.gees(Hm)
    if Hm.dtype == np.float64:
        T, _sdim, wr, wi, Z, _work, info = _lapack.dgees(lambda *a: False, np.asfortranarray(Hm), sort_t=0)
        w = wr + 1j * wi
    else:
        T, _sdim, w, Z, _work, info = _lapack.zgees(lambda *a: False, np.asfortranarray(Hm), sort_t=0)
    if info != 0:
        raise RuntimeError(f"GEES failed, info={info}")
    return T, Z, w


def _ordschur(T: np.ndarray, Q: np.ndarray, selected: np.ndarray):
    """ordschur = LAPACK trsen(job='N', compq='V').  submodule_utility_functions.fypp:90-117"""
    fn = _lapack.dtrsen if T.dtype == np.float64 else _lapack.ztrsen
    out = fn(np.asarray(selected, dtype=np.int32), np.asfortranarray(T), np.asfortranarray(Q), job="N", wantq=1)
    if out[-1] != 0:
        raise RuntimeError(f"TRSEN failed, info={out[-1]}")
    return out[0], out[1]


def krylov_schur_host_part(H: np.ndarray, kdim: int, select_eigs):
    """The small-matrix half of krylov_schur (gees, the selector, trsen: BaseKrylov.fypp:807-813) as a pure function of H:
    returns (T, Tk, Z, n).  eigs runs it on a spare host thread as soon as the device has delivered the last column of H, beside
    the per-step Ritz tests still in flight, and hands the result to krylov_schur -- same LAPACK calls on the same data."""
    m = H.shape[1]
    T, Z, eigvals = _schur(H[:m, :])                                                # :807
    Hc = np.array(H, order="F", copy=True)                                          # what H holds after `H(:m, :) = T`
    Hc[:m, :] = T
    selected = np.asarray(select_eigs(eigvals), dtype=bool)                         # :810
    n = int(np.count_nonzero(selected))
    Tk, Z = _ordschur(Hc[:kdim, :], Z, selected)                                    # :813
    return T, Tk, Z, n


@_hostlapack.small_problems
def krylov_schur(X, H: np.ndarray, select_eigs, _host_part=None) -> int:
    """Krylov-Schur restart: re-order the Schur form of H and compress the basis.
    src/Krylov/BaseKrylov.fypp:782-834.  Returns n, the number of selected eigenvalues.
    (`_host_part`: the result of krylov_schur_host_part on THIS H, computed ahead by the caller.)"""
    kdim = len(X) - 1
    m = H.shape[1]
    T, Tk, Z, n = _host_part if _host_part is not None else krylov_schur_host_part(H, kdim, select_eigs)
    H[:m, :] = T
    H[:kdim, :] = Tk
    # basis update (:816-824): Xwrk = X(:m) Z(:, :n); X(:n) = Xwrk; X(n+1) = X(kdim+1); X(n+2:) = 0
    if n > 0:
        Xw = linear_combination(X[:m], np.asfortranarray(Z[:, :n]))
        copy(X[:n], Xw if not isinstance(Xw, krylov_basis_gpu) else Xw[:n])
    copy(X[n], X[kdim])
    if n + 1 < len(X):
        zero_basis(X[n + 1:])
    b = H[kdim, :] @ Z                                                              # :827
    H[n, :] = b
    H[n + 1:, :] = 0
    H[:, n:] = 0
    return n
