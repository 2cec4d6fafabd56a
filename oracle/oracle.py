"""CPU ORACLE for the LightKrylov hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Python face of ``oracle/lk_oracle.c`` (plain C, reference schedule, one thread) plus a
restatement of the reference's *callers* of that path (arnoldi / lanczos / gmres / eigs /
krylov_schur) with the small host LAPACK work done by scipy, exactly where the reference
calls stdlib's LAPACK.  Every function cites the reference file:line it follows (paths are
relative to /root/reference; nothing is read from there at run time).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  ``lightkrylov_amd`` never does.

Parity pinning: see the header of lk_oracle.c -- pinned by the reference's known-answer
tests (tests/test_oracle_kat.py); no oracle/_ref build exists because the Fortran
reference needs fortran-lang/stdlib, which this image lacks.

Bases are numpy arrays of shape (n, ncols), Fortran order, dtype float64 / complex128
(column j == reference ``X(j+1)``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
from scipy.linalg import lapack as _lp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ATOL_DP = 1.0e-15                     # src/Constants.f90:35
RTOL_DP = float(np.sqrt(ATOL_DP))     # src/Constants.f90:37

MATVEC_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liblk_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("lk_oracle.c", "lk_oracle_body.inc", "lk_oracle_fast.inc")]
    if force or not os.path.exists(so) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in src
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liblk_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for sfx in ("_d", "_z"):
            getattr(_LIB, "ora_norm" + sfx).restype = C.c_double
            getattr(_LIB, "ora_orthogonalize" + sfx).restype = C.c_int
            getattr(_LIB, "ora_dgs" + sfx).restype = C.c_int
            getattr(_LIB, "ora_arnoldi" + sfx).restype = C.c_int
            getattr(_LIB, "ora_dgs_fast" + sfx).restype = C.c_int
            getattr(_LIB, "ora_arnoldi_fast" + sfx).restype = C.c_int
            getattr(_LIB, "ora_norm_mode" + sfx).restype = C.c_double
            getattr(_LIB, "ora_arnoldi_fused" + sfx).restype = C.c_int
    return _LIB


SEQUENTIAL, COMPENSATED = 0, 1          # dot modes of the *_fast entry points (lk_oracle_fast.inc)


def set_threads(nt: int) -> int:
    """Threads for the ``fast=True`` paths and the element-wise operators (default 1 = the
    reference).  Never changes a result in SEQUENTIAL mode.  Returns the value in effect."""
    lib().ora_set_threads(C.c_int(int(nt)))
    return int(lib().ora_get_threads())


def max_threads() -> int:
    return int(lib().ora_max_threads())


def _sfx(a: np.ndarray) -> str:
    if a.dtype == np.float64:
        return "_d"
    if a.dtype == np.complex128:
        return "_z"
    raise TypeError(f"oracle supports float64/complex128 only, got {a.dtype}")


def _p(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def _scalar(val, dtype) -> np.ndarray:
    return np.array([val], dtype=dtype)


def _ld(X: np.ndarray) -> int:
    assert X.flags.f_contiguous or X.ndim == 1
    return X.strides[1] // X.itemsize if X.ndim == 2 and X.shape[1] > 1 else X.shape[0]


# ----------------------------------------------------------------------------------------
# abstract_vector primitives as dense_vector implements them
# ----------------------------------------------------------------------------------------
def scal(x: np.ndarray, alpha) -> None:
    """AbstractVectors.fypp:505-512"""
    a = _scalar(alpha, x.dtype)
    getattr(lib(), "ora_scal" + _sfx(x))(C.c_int64(x.size), _p(a), _p(x))


def axpby(alpha, x: np.ndarray, beta, y: np.ndarray) -> None:
    """y <- alpha*x + beta*y the way dense_axpby does it.  AbstractVectors.fypp:514-536"""
    a, b = _scalar(alpha, y.dtype), _scalar(beta, y.dtype)
    getattr(lib(), "ora_dense_axpby" + _sfx(y))(C.c_int64(y.size), _p(a), _p(x), _p(b), _p(y))


def dot(x: np.ndarray, y: np.ndarray):
    """x%dot(y) = sum conj(x) y.  AbstractVectors.fypp:538-555"""
    out = np.zeros(1, dtype=x.dtype)
    getattr(lib(), "ora_dot" + _sfx(x))(C.c_int64(x.size), _p(x), _p(y), _p(out))
    return out[0]


def norm(x: np.ndarray) -> float:
    """AbstractVectors.fypp:424-432"""
    return float(getattr(lib(), "ora_norm" + _sfx(x))(C.c_int64(x.size), _p(x)))


def innerprod(X: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """M = X^H Y, one dot per entry.  AbstractVectors.fypp:659-695"""
    Y2 = Y.reshape(Y.shape[0], -1, order="F")
    k, p = X.shape[1], Y2.shape[1]
    M = np.zeros((k, p), dtype=X.dtype, order="F")
    getattr(lib(), "ora_innerprod" + _sfx(X))(
        C.c_int64(X.shape[0]), C.c_int(k), _p(X), C.c_int64(_ld(X)), C.c_int(p), _p(Y2),
        C.c_int64(_ld(Y2)), _p(M))
    return M[:, 0] if Y.ndim == 1 else M


def linear_combination(X: np.ndarray, v: np.ndarray) -> np.ndarray:
    """y = X v by k axpby calls.  AbstractVectors.fypp:571-603"""
    v = np.ascontiguousarray(v, dtype=X.dtype)
    out = np.empty(X.shape[0], dtype=X.dtype)
    getattr(lib(), "ora_lincomb" + _sfx(X))(
        C.c_int64(X.shape[0]), C.c_int(X.shape[1]), _p(X), C.c_int64(_ld(X)), _p(v), _p(out))
    return out


def gram(X: np.ndarray) -> np.ndarray:
    """AbstractVectors.fypp:645-657 (mirrors without conjugation)."""
    k = X.shape[1]
    G = np.zeros((k, k), dtype=X.dtype, order="F")
    getattr(lib(), "ora_gram" + _sfx(X))(C.c_int64(X.shape[0]), C.c_int(k), _p(X),
                                        C.c_int64(_ld(X)), _p(G))
    return G


def orthogonalize_against_basis(y: np.ndarray, X: np.ndarray):
    """One CGS pass; returns (h, info).  gram_schmidt.fypp:113-154"""
    h = np.zeros(X.shape[1], dtype=X.dtype)
    info = getattr(lib(), "ora_orthogonalize" + _sfx(X))(
        C.c_int64(X.shape[0]), C.c_int(X.shape[1]), _p(X), C.c_int64(_ld(X)), _p(y), _p(h))
    return h, int(info)


def double_gram_schmidt_step(y: np.ndarray, X: np.ndarray, fast: bool = False, mode: int = SEQUENTIAL):
    """Returns (beta = h1 + h2, info).  y is updated in place.  gram_schmidt.fypp:12-57.
    fast=True: the multi-threaded evaluation of the same schedule (bit-identical in SEQUENTIAL mode)."""
    k = X.shape[1]
    h = np.zeros(max(k, 1), dtype=X.dtype)
    wrk = np.zeros(max(k, 1), dtype=X.dtype)
    if fast or mode != SEQUENTIAL:
        info = getattr(lib(), "ora_dgs_fast" + _sfx(X))(
            C.c_int64(X.shape[0]), C.c_int(k), _p(X), C.c_int64(_ld(X)), _p(y), _p(h), _p(wrk), C.c_int(mode))
    else:
        info = getattr(lib(), "ora_dgs" + _sfx(X))(
            C.c_int64(X.shape[0]), C.c_int(k), _p(X), C.c_int64(_ld(X)), _p(y), _p(h), _p(wrk))
    return h[:k], int(info)


def dot_mode(x: np.ndarray, y: np.ndarray, mode: int):
    out = np.zeros(1, dtype=x.dtype)
    getattr(lib(), "ora_dot_mode" + _sfx(x))(C.c_int64(x.size), _p(x), _p(y), _p(out), C.c_int(mode))
    return out[0]


def first_touch(X: np.ndarray) -> None:
    """Parallel first touch (zeros) of a freshly allocated real F-order array by the current thread team: NUMA placement for the
    all-core leg of bench.py's cpu_baseline."""
    X2 = X.reshape(X.shape[0], -1, order="F")
    assert X2.dtype == np.float64
    lib().ora_first_touch(C.c_int64(X2.shape[0]), C.c_int(X2.shape[1]), _p(X2), C.c_int64(_ld(X2)))


def thread_cpus() -> list:
    """CPU number each thread of the current team runs on (diagnostic of the binding)."""
    nt = int(lib().ora_get_threads())
    out = (C.c_int * nt)()
    lib().ora_thread_cpus(out)
    return list(out)


def fill_counter(x: np.ndarray, seed: int, i0: int = 0) -> None:
    """x_i = 2u-1 with u = (splitmix64(seed*2^32 + i0+i) >> 11) 2^-53."""
    getattr(lib(), "ora_fill_counter" + _sfx(x))(C.c_int64(x.size), C.c_int64(i0),
                                                C.c_uint64(seed), _p(x))


# ----------------------------------------------------------------------------------------
# operators standing in for a user's abstract_linop (AbstractLinops.fypp:58-87)
# ----------------------------------------------------------------------------------------
class _OpBase:
    def c_matvec(self):          # -> (function pointer as c_void_p-castable, op struct pointer)
        raise NotImplementedError

    def matvec(self, x: np.ndarray, y: np.ndarray) -> None:
        fn, op = self.c_matvec()
        fn(op, C.c_int64(x.size), _p(x), _p(y))


class _DiagStruct(C.Structure):
    _fields_ = [("d", C.c_void_p)]


class _DenseStruct(C.Structure):
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int64)]


class _Lap5Struct(C.Structure):
    _fields_ = [("N", C.c_int64)]


class DiagOp(_OpBase):
    def __init__(self, d: np.ndarray):
        self.d = np.ascontiguousarray(d)
        self._s = _DiagStruct(self.d.ctypes.data)

    def c_matvec(self):
        return getattr(lib(), "ora_matvec_diag" + _sfx(self.d)), C.byref(self._s)


class _DiagLinStruct(C.Structure):
    _fields_ = [("d0", C.c_double), ("dstep", C.c_double), ("row0", C.c_int64)]


class DiagLinOp(_OpBase):
    """d_i = fma(dstep, row0 + i, d0), generated on the fly (no n-sized array): the operator of
    BASELINE configs 2 and 5 exactly as the engine's diag_linop_gpu(d0=, dstep=) evaluates it."""

    def __init__(self, d0: float, dstep: float, row0: int = 0):
        self._s = _DiagLinStruct(d0, dstep, row0)

    def c_matvec(self):
        return lib().ora_matvec_diaglin_d, C.byref(self._s)


class DenseOp(_OpBase):
    """dense_linop: gemv('N').  AbstractLinops.fypp:608-631"""

    def __init__(self, A: np.ndarray):
        self.A = np.asfortranarray(A)
        self._s = _DenseStruct(self.A.ctypes.data, self.A.shape[0])

    def c_matvec(self):
        return getattr(lib(), "ora_matvec_dense" + _sfx(self.A)), C.byref(self._s)


class Lap5Op(_OpBase):
    def __init__(self, N: int):
        self.N = N
        self._s = _Lap5Struct(N)

    def c_matvec(self):
        return lib().ora_matvec_lap5_d, C.byref(self._s)


class GLOp(_OpBase):
    """Ginzburg-Landau propagator: nsub classical RK4 steps of the reference right-hand side
    (example/ginzburg_landau/Ginzburg_Landau.f90:126-136, adjoint :170-179), numpy restatement."""

    def __init__(self, n, dx, tau, nsub, nu, gamma, mu_c, mu2, adjoint=False):
        self.n, self.dx, self.tau, self.nsub = n, dx, tau, nsub
        self.nu, self.gamma, self.adjoint = nu, gamma, adjoint
        L = dx * (n + 1)
        x = -L / 2 + dx * np.arange(1, n + 1)                      # linspace(-L/2, L/2, n+2)(2:n+1)
        self.mu = mu_c + 0.5 * mu2 * x * x

        def _cb(_op, nn, xp, yp):
            xin = np.ctypeslib.as_array(C.cast(xp, C.POINTER(C.c_double)), shape=(2 * nn,)).view(np.complex128)
            yout = np.ctypeslib.as_array(C.cast(yp, C.POINTER(C.c_double)), shape=(2 * nn,)).view(np.complex128)
            yout[:] = self.apply(xin)
        self._cb = MATVEC_FN(_cb)

    def rhs(self, u):
        dx, n = self.dx, self.n
        cu = np.empty_like(u)
        d2u = np.empty_like(u)
        cu[1:-1] = (u[2:] - u[:-2]) / (2 * dx)
        d2u[1:-1] = (u[2:] - 2 * u[1:-1] + u[:-2]) / dx ** 2
        cu[0] = u[1] / (2 * dx)
        d2u[0] = (u[1] - 2 * u[0]) / dx ** 2
        cu[-1] = -u[n - 2] / (2 * dx)
        d2u[-1] = (-2 * u[n - 1] + u[n - 2]) / (2 * dx)            # sic, :131
        if self.adjoint:
            return np.conj(self.nu) * cu + np.conj(self.gamma) * d2u + self.mu * u
        return -self.nu * cu + self.gamma * d2u + self.mu * u

    def apply(self, u):
        dt = self.tau / self.nsub
        u = u.copy()
        for _ in range(self.nsub):
            k1 = self.rhs(u)
            k2 = self.rhs(u + 0.5 * dt * k1)
            k3 = self.rhs(u + 0.5 * dt * k2)
            k4 = self.rhs(u + dt * k3)
            u = u + dt / 6 * k1 + dt / 3 * k2 + dt / 3 * k3 + dt / 6 * k4
        return u

    def c_matvec(self):
        return self._cb, None

    def matvec(self, x, y):
        y[:] = self.apply(x)


class PyOp(_OpBase):
    """Any python callable f(x)->y as an operator (small cases only)."""

    def __init__(self, f, dtype):
        self.f, self.dtype = f, np.dtype(dtype)

        def _cb(_op, n, xp, yp):
            x = np.ctypeslib.as_array(C.cast(xp, C.POINTER(C.c_double)),
                                      shape=(n * (2 if self.dtype.kind == "c" else 1),)).view(self.dtype)
            y = np.ctypeslib.as_array(C.cast(yp, C.POINTER(C.c_double)),
                                      shape=(n * (2 if self.dtype.kind == "c" else 1),)).view(self.dtype)
            y[:] = self.f(x)
        self._cb = MATVEC_FN(_cb)

    def c_matvec(self):
        return self._cb, None


# ----------------------------------------------------------------------------------------
# Krylov factorisations
# ----------------------------------------------------------------------------------------
def arnoldi(A: _OpBase, X: np.ndarray, H: np.ndarray, kstart: int = 1, kend: int | None = None,
            tol: float = ATOL_DP, rand_seed: int = 12345, fast: bool = False, mode: int = SEQUENTIAL) -> int:
    """arnoldi (blksize 1).  src/Krylov/arnoldi.fypp:8-76.  X: (n, m+1) F-order, H: (m+1, m) F-order.
    fast=True runs the multi-threaded evaluation (set_threads), bit-identical in SEQUENTIAL mode;
    mode=COMPENSATED uses twice-working-precision dots (not the reference's arithmetic)."""
    m = X.shape[1] - 1
    kend = m if kend is None else kend
    fn, op = A.c_matvec()
    fnp = C.cast(fn, C.c_void_p)
    if fast or mode != SEQUENTIAL:
        return int(getattr(lib(), "ora_arnoldi_fast" + _sfx(X))(
            C.c_int64(X.shape[0]), C.c_int(m), _p(X), C.c_int64(_ld(X)), _p(H), C.c_int64(H.shape[0]),
            C.c_int(kstart), C.c_int(kend), C.c_double(tol), fnp, op, C.c_uint64(rand_seed), C.c_int(mode)))
    info = getattr(lib(), "ora_arnoldi" + _sfx(X))(
        C.c_int64(X.shape[0]), C.c_int(m), _p(X), C.c_int64(_ld(X)), _p(H), C.c_int64(H.shape[0]),
        C.c_int(kstart), C.c_int(kend), C.c_double(tol), fnp, op, C.c_uint64(rand_seed))
    return int(info)


def orthogonalize_basis_against_basis(Y: np.ndarray, X: np.ndarray):
    """One CGS pass of a BLOCK Y (n x p) against X (n x k); returns (coefficients k x p, info).  gram_schmidt.fypp:156-200:
    zero-vector flag per column (:170-173), proj_coefficients = innerprod(X, Y) (:188), proj = linear_combination(X, coefficients)
    column by column (AbstractVectors.fypp:605-643) and Y(j) <- -proj(j) + Y(j) (axpby_basis, :190-191)."""
    info = 0
    for i in range(Y.shape[1]):
        if norm(Y[:, i]) < ATOL_DP:
            info = i + 1
    M = innerprod(X, Y).reshape(X.shape[1], Y.shape[1], order="F")
    for j in range(Y.shape[1]):
        proj = linear_combination(X, M[:, j])
        axpby(-1.0, proj, 1.0, Y[:, j])
    return M, info


def double_gram_schmidt_step_block(Y: np.ndarray, X: np.ndarray):
    """DGS_basis_against_basis: two passes of the above, beta = pass 1 + pass 2, info of the second.  gram_schmidt.fypp:59-105"""
    M1, _ = orthogonalize_basis_against_basis(Y, X)
    M2, info = orthogonalize_basis_against_basis(Y, X)
    return M1 + M2, info


def qr_no_pivoting(Q: np.ndarray, R: np.ndarray, tol: float = ATOL_DP, rand_seed: int = 12345, column_seed=None) -> int:
    """In-place double-Gram-Schmidt QR of the columns of Q, R upper triangular.  qr.fypp:116-167 (a colinear column is replaced by
    counter-RNG numbers -- the reference draws from the unseeded intrinsic generator there, :146-162).
    `column_seed(j)`: the counter stream of the re-draw of column j, filling the column AS STORED (complex: re, im interleaved) -- the
    stream the engine uses for that column (lk_qr / lk_arnoldi_block: 0x5EED + panel column + 1), so that a colinear case can be compared
    entry by entry; default: the streams rand_seed + j (real), rand_seed + 2j / + 2j + 1 (complex: re and im filled separately)."""
    info, flag = 0, False
    R[...] = 0
    for j in range(Q.shape[1]):
        if j > 0:
            h, _ = double_gram_schmidt_step(Q[:, j], Q[:, :j])
            R[:j, j] = h
        beta = norm(Q[:, j])
        if beta != beta:
            raise FloatingPointError("|beta| = NaN detected! Abort")
        if abs(beta) < tol:
            if not flag:
                flag, info = True, j + 1
            R[j, j] = 0
            if column_seed is not None:
                col = np.empty(Q.shape[0], dtype=Q.dtype)
                fill_counter(col, int(column_seed(j)))
                Q[:, j] = col
            elif np.iscomplexobj(Q):
                re, im = np.empty(Q.shape[0]), np.empty(Q.shape[0])
                fill_counter(re, rand_seed + 2 * j); fill_counter(im, rand_seed + 2 * j + 1)
                Q[:, j] = re + 1j * im
            else:
                fill_counter(Q[:, j], rand_seed + j)
            if j > 0:
                double_gram_schmidt_step(Q[:, j], Q[:, :j])
            beta = norm(Q[:, j])
        else:
            R[j, j] = beta
        scal(Q[:, j], 1.0 / beta)
    return info


def arnoldi_block(A: _OpBase, X: np.ndarray, H: np.ndarray, blksize: int, kstart: int = 1, kend: int | None = None,
                  tol: float = ATOL_DP, engine_streams: bool = False) -> int:
    """arnoldi with blksize = p > 1.  src/Krylov/arnoldi.fypp:20-73: p matvecs (:39-47), the batch double Gram-Schmidt step with
    beta = H(:kp, kpm+1:kp) (:50-51), qr of the new block into H(kp+1:kpp, kpm+1:kp) (:55), breakdown on the smallest |diagonal| of
    that block (:58-71).  X: (n, (kdim+1) p) F-order, H: ((kdim+1) p, kdim p) F-order."""
    p = blksize
    kdim = (X.shape[1] - p) // p
    kend = kdim if kend is None else kend
    info = 0
    for k in range(kstart, kend + 1):
        kpm, kp, kpp = (k - 1) * p, k * p, (k + 1) * p
        for i in range(p):
            A.matvec(X[:, kpm + i], X[:, kp + i])
        beta, _ = double_gram_schmidt_step_block(X[:, kp:kpp], X[:, :kp])
        H[:kp, kpm:kp] = beta
        R = np.zeros((p, p), dtype=X.dtype, order="F")
        # engine_streams: colinear columns are re-drawn from the counter streams lk_arnoldi_block uses (0x5EED + basis column + 1)
        qr_no_pivoting(X[:, kp:kpp], R, column_seed=(lambda j, kp=kp: 0x5EED + kp + j + 1) if engine_streams else None)
        H[kp:kpp, kpm:kp] = R
        if np.min(np.abs(np.diag(R))) < tol:
            info = kp
            break
    return info


def arnoldi_fused_allcores(A: _OpBase, X: np.ndarray, H: np.ndarray, tol: float = ATOL_DP) -> int:
    """NOT the reference's arithmetic and never a checker: the engine's three-sweep fused CGS2 schedule on
    the host cores (set_threads), timed by bench.py as the all-core leg of cpu_baseline."""
    m = X.shape[1] - 1
    fn, op = A.c_matvec()
    return int(getattr(lib(), "ora_arnoldi_fused" + _sfx(X))(
        C.c_int64(X.shape[0]), C.c_int(m), _p(X), C.c_int64(_ld(X)), _p(H), C.c_int64(H.shape[0]),
        C.c_double(tol), C.cast(fn, C.c_void_p), op))


def lanczos(A: _OpBase, X: np.ndarray, T: np.ndarray, kstart: int = 1, kend: int | None = None,
            tol: float = ATOL_DP) -> int:
    """lanczos_tridiagonalization.  src/Krylov/lanczos.fypp:7-64"""
    kdim = X.shape[1] - 1
    kend = kdim if kend is None else kend
    info = 0
    for k in range(kstart, kend + 1):
        A.matvec(X[:, k - 1], X[:, k])                                   # :26
        for i in range(max(1, k - 1), k + 1):                            # :57-60
            T[i - 1, k - 1] = dot(X[:, i - 1], X[:, k])
            axpby(-T[i - 1, k - 1], X[:, i - 1], 1.0, X[:, k])
        double_gram_schmidt_step(X[:, k], X[:, :k])                      # :62
        beta = norm(X[:, k])
        T[k, k - 1] = beta                                               # :29
        if beta < tol:
            info = k
            break
        scal(X[:, k], 1.0 / beta)                                        # :39
    return info


def bidiagonalization(A: _OpBase, Ah: _OpBase, U: np.ndarray, V: np.ndarray, B: np.ndarray,
                      tol: float = ATOL_DP, kstart: int = 1, kend: int | None = None) -> int:
    """lanczos_bidiagonalization.  src/Krylov/golub_kahan.fypp:7-64.  `Ah` applies A^H (rmatvec)."""
    kdim = U.shape[1] - 1
    kend = kdim if kend is None else kend
    info = 0
    for k in range(kstart, kend + 1):
        Ah.matvec(U[:, k - 1], V[:, k - 1])                                # :27
        if k > 1:
            double_gram_schmidt_step(V[:, k - 1], V[:, :k - 1])            # :30-33
        alpha = norm(V[:, k - 1])
        B[k - 1, k - 1] = alpha
        if abs(alpha) > tol:
            scal(V[:, k - 1], 1.0 / alpha)
        else:
            info = k
            break
        A.matvec(V[:, k - 1], U[:, k])                                     # :45
        double_gram_schmidt_step(U[:, k], U[:, :k])                        # :48-49
        beta = norm(U[:, k])
        B[k, k - 1] = beta
        if abs(beta) > tol:
            scal(U[:, k], 1.0 / beta)
        else:
            info = k
            break
    return info


# ----------------------------------------------------------------------------------------
# small host LAPACK pieces (src/Utilities/submodule_utility_functions.fypp)
# ----------------------------------------------------------------------------------------
def eig(Hk: np.ndarray):
    """geev, right vectors in LAPACK layout (real pairs NOT combined).  :55-85"""
    if Hk.dtype == np.float64:
        wr, wi, _vl, vr, info = _lp.dgeev(Hk, compute_vl=0, compute_vr=1)
        assert info == 0
        return wr + 1j * wi, vr
    w, _vl, vr, info = _lp.zgeev(Hk, compute_vl=0, compute_vr=1)
    assert info == 0
    return w, vr


def apply_givens_rotation(h: np.ndarray, c: np.ndarray, s: np.ndarray) -> None:
    """:173-204.  Real: lasr('L','V','F') + lartg; complex: the in-house unconjugated rotation."""
    k = h.size - 1
    if h.dtype == np.float64:
        for j in range(k - 1):                    # dlasr, SIDE=L, PIVOT=V, DIRECT=F
            t = h[j + 1]
            h[j + 1] = c[j] * t - s[j] * h[j]
            h[j] = s[j] * t + c[j] * h[j]
        cc, ss, r = _lp.dlartg(h[k - 1], h[k])
        c[k - 1], s[k - 1] = cc, ss
        h[k - 1], h[k] = r, 0.0
    else:
        for i in range(k - 1):
            t = c[i] * h[i] + s[i] * h[i + 1]
            h[i + 1] = -s[i] * h[i] + c[i] * h[i + 1]
            h[i] = t
        g = np.array([h[k - 1], h[k]])
        g = g / np.sqrt(np.sum(np.abs(g) ** 2))   # givens_rotation: x / norm(x,2)
        c[k - 1], s[k - 1] = g[0], g[1]
        h[k - 1] = c[k - 1] * h[k - 1] + s[k - 1] * h[k]
        h[k] = 0.0


def gmres(A: _OpBase, b: np.ndarray, x: np.ndarray, rtol: float = RTOL_DP, atol: float = ATOL_DP,
          kdim: int = 30, maxiter: int = 10, fast: bool = False):
    """Restarted GMRES, no preconditioner.  src/IterativeSolvers/GMRES/gmres.fypp:105-239.
    Returns (info, residual history); x updated in place.  fast=True: the Gram-Schmidt steps run through the
    multi-threaded, bit-identical evaluation (set_threads)."""
    n, dt = b.size, b.dtype
    tol = atol + rtol * norm(b)                                          # :106
    V = np.zeros((n, kdim + 1), dtype=dt, order="F")
    H = np.zeros((kdim + 1, kdim), dtype=dt, order="F")
    res, n_iter, n_outer, converged = [], 0, 0, False
    while (not converged) and n_outer <= maxiter:                        # :131
        H[:] = 0
        V[:] = 0
        if norm(x) != 0.0:
            A.matvec(x, V[:, 0])                                         # :134-140
        axpby(-1.0, b, 1.0, V[:, 0])
        scal(V[:, 0], -1.0)                                              # :141
        e = np.zeros(kdim + 1, dtype=dt)
        beta = norm(V[:, 0])
        e[0] = beta
        scal(V[:, 0], 1.0 / beta)
        c = np.zeros(kdim, dtype=dt)
        s = np.zeros(kdim, dtype=dt)
        if n_outer == 0:
            res.append(abs(beta))
        k = 0
        for k in range(1, kdim + 1):
            wrk = V[:, k - 1].copy()                                     # :155
            A.matvec(wrk, V[:, k])
            h, _ = double_gram_schmidt_step(V[:, k], V[:, :k], fast=fast)  # :167-168
            H[:k, k - 1] = h
            H[k, k - 1] = norm(V[:, k])                                  # :171
            if abs(H[k, k - 1]) > tol:
                scal(V[:, k], 1.0 / H[k, k - 1])                         # :172
            apply_givens_rotation(H[:k + 1, k - 1], c[:k], s[:k])        # :178
            e[k] = -s[k - 1] * e[k - 1]
            e[k - 1] = c[k - 1] * e[k - 1]                               # :180
            beta = abs(e[k])
            n_iter += 1
            res.append(abs(beta))
            if abs(beta) < tol:
                converged = True
                break
        k = min(k, kdim)
        trtrs = _lp.dtrtrs if dt == np.float64 else _lp.ztrtrs
        yk, info = trtrs(H[:k, :k], e[:k], lower=0, trans=0, unitdiag=0)  # :199-200
        assert info == 0
        dx = linear_combination(V[:, :k], yk)                            # :201
        axpby(1.0, dx, 1.0, x)
        A.matvec(x, V[:, 0])                                             # :205-210
        axpby(-1.0, b, 1.0, V[:, 0])
        scal(V[:, 0], -1.0)
        beta = norm(V[:, 0])
        if abs(beta) > 0.0:
            scal(V[:, 0], 1.0 / beta)
        n_iter += 1
        n_outer += 1
        res.append(abs(beta))
        if abs(beta) < tol:
            converged = True
            break
    info = n_iter if converged else -n_iter                              # :234-238
    return info, np.array(res)


def _lincomb_columns(X: np.ndarray, Zc: np.ndarray, threads: int = 1) -> np.ndarray:
    """Xw(:, j) = linear_combination(X, Zc(:, j)) for every column j -- each column is its own chain of axpbys in the reference's
    order (AbstractVectors.fypp:605-643 loops over the columns), so running the COLUMNS on several host threads is bit-identical
    to the sequential loop (ctypes releases the interpreter lock for the duration of a call)."""
    q = Zc.shape[1]
    Xw = np.zeros((X.shape[0], q), dtype=X.dtype, order="F")

    def one(j):
        Xw[:, j] = linear_combination(X, np.ascontiguousarray(Zc[:, j]))
    if threads > 1 and q > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(threads, q)) as ex:
            list(ex.map(one, range(q)))
    else:
        for j in range(q):
            one(j)
    return Xw


def krylov_schur(X: np.ndarray, H: np.ndarray, select, threads: int = 1):
    """src/Krylov/BaseKrylov.fypp:782-834.  Returns n (number of selected eigenvalues).  `threads` > 1: the columns of the basis
    update on several host threads (bit-identical, see _lincomb_columns)."""
    kdim = X.shape[1] - 1
    m = H.shape[1]
    if H.dtype == np.float64:
        T, sdim, wr, wi, Z, work, info = _lp.dgees(lambda *a: False, H[:m, :].copy(order="F"), sort_t=0)
        eigvals = wr + 1j * wi
    else:
        T, sdim, eigvals, Z, work, info = _lp.zgees(lambda *a: False, H[:m, :].copy(order="F"), sort_t=0)
    assert info == 0
    H[:m, :] = T
    selected = np.asarray(select(eigvals), dtype=bool)
    n = int(np.count_nonzero(selected))
    Hk = np.asfortranarray(H[:kdim, :])
    if H.dtype == np.float64:
        out = _lp.dtrsen(selected.astype(np.int32), Hk, np.asfortranarray(Z), job="N", wantq=1)
    else:
        out = _lp.ztrsen(selected.astype(np.int32), Hk, np.asfortranarray(Z), job="N", wantq=1)
    assert out[-1] == 0
    H[:kdim, :] = out[0]
    Z = out[1]
    # basis update: Xwrk = X(:m) Z(:, :n) by n*m axpby;  X(:n) = Xwrk ; X(n+1) = X(kdim+1) ; rest zero
    Xw = _lincomb_columns(X[:, :m], np.asarray(Z)[:, :n], threads)
    X[:, :n] = Xw
    X[:, n] = X[:, kdim]
    X[:, n + 1:] = 0
    b = H[kdim, :] @ Z
    H[n, :] = b
    H[n + 1:, :] = 0
    H[:, n:] = 0
    return n


def eigs(A: _OpBase, x0: np.ndarray, nev: int, kdim: int | None = None, tolerance: float = RTOL_DP,
         max_restarts: int = 1000, fast: bool = False, stop_after_cycles: int | None = None):
    """Krylov-Schur eigensolver.  src/IterativeSolvers/IterativeSolvers.fypp:972-1143.
    Returns (eigvals[nev], residuals[nev], eigvecs (n, nev), info=niter).
    fast=True: the Arnoldi steps and the columns of the restart's basis update run through the multi-threaded, bit-identical
    evaluation (set_threads).  stop_after_cycles = c: leave the loop after c Arnoldi cycles (each followed by its restart) whether
    or not `nev` pairs have converged -- the reference loops for ever; the engine mirror's `max_restarts` = c - 1 is the same cut, so
    that a fixed amount of RESTARTED work can be compared at sizes where convergence is out of reach."""
    n, dt = x0.size, x0.dtype
    kdim = 4 * nev if kdim is None else kdim
    X = np.zeros((n, kdim + 1), dtype=dt, order="F")
    X[:, 0] = x0
    scal(X[:, 0], 1.0 / norm(x0))                                        # :1036-1038
    H = np.zeros((kdim + 1, kdim), dtype=dt, order="F")
    res = np.zeros(kdim)
    kstart, conv, niter, k = 1, 0, 0, 0
    restarts = 0
    nthr = int(lib().ora_get_threads()) if fast else 1
    while conv < nev:
        if stop_after_cycles is not None and restarts >= stop_after_cycles:
            break
        for k in range(kstart, kdim + 1):
            info = arnoldi(A, X, H, kstart=k, kend=k, fast=fast)         # :1059
            w, vr = eig(np.asfortranarray(H[:k, :k]))                    # :1065
            beta = H[k, k - 1]
            if dt == np.complex128:
                res[:k] = np.abs(beta * vr[k - 1, :k])                   # :1071
            else:
                for i in range(k):                                       # :1073-1082
                    if w[i].imag > 0:
                        alpha = abs(complex(vr[k - 1, i], vr[k - 1, i + 1]))
                    elif w[i].imag < 0:
                        alpha = abs(complex(vr[k - 1, i - 1], vr[k - 1, i]))
                    else:
                        alpha = abs(vr[k - 1, i])
                    res[i] = abs(beta * alpha)
            niter += 1
            conv = int(np.count_nonzero(res[:k] < tolerance))
            if conv >= nev:
                break
        restarts += 1
        if restarts > max_restarts:
            raise RuntimeError("oracle eigs: too many restarts")
        # NB the reference restarts unconditionally, also after convergence (:1100)
        nsel = krylov_schur(X, H, lambda lam: np.abs(lam) > np.median(np.abs(lam)), threads=nthr)
        kstart = nsel + 1
    k = min(k, kdim)
    w, vr = eig(np.asfortranarray(H[:k, :k]))                            # :1115
    wfull = np.zeros(kdim, dtype=np.complex128)
    wfull[:k] = w
    idx = np.argsort(-np.abs(wfull), kind="stable")                      # :1118-1120
    vfull = np.zeros((kdim, kdim), dtype=vr.dtype)
    vfull[:k, :k] = vr
    wsorted, vsorted, rsorted = wfull[idx], vfull[:, idx], res[idx]
    vecs = _lincomb_columns(X[:, :k], np.asfortranarray(vsorted[:k, :nev].astype(dt)), nthr)   # :1127-1132
    return wsorted[:nev], rsorted[:nev], vecs, niter


# ----------------------------------------------------------------------------------------
# the remaining solver families (callers of the same primitives; restated for the host-side mirror's tests)
# ----------------------------------------------------------------------------------------
def cg(A: _OpBase, b: np.ndarray, x: np.ndarray, rtol: float = RTOL_DP, atol: float = ATOL_DP, maxiter: int = 100):
    """Conjugate gradient without preconditioner.  src/IterativeSolvers/CG/CG.fypp:98-200.
    Returns (info, residual history); x updated in place."""
    tol = atol + rtol * norm(b)                                          # :110
    r = np.zeros_like(b)
    Ap = np.zeros_like(b)
    if norm(x) > 0:
        A.matvec(x, r)                                                   # :130
    axpby(-1.0, b, 1.0, r)
    scal(r, -1.0)                                                        # r = b - A x   :131
    p = r.copy()                                                         # p = r        :137
    rr_old = dot(r, r)
    res = [float(np.sqrt(abs(rr_old)))]
    n_iter, converged = 0, False
    for _ in range(maxiter):
        A.matvec(p, Ap)                                                  # :146
        alpha = rr_old / dot(p, Ap)                                      # :148
        axpby(alpha, p, 1.0, x)                                          # :150
        axpby(-alpha, Ap, 1.0, r)                                        # :152
        rr_new = dot(r, r)
        residual = float(np.sqrt(abs(rr_new)))
        n_iter += 1
        res.append(residual)
        if residual < tol:
            converged = True
            break
        beta = rr_new / rr_old                                           # :174
        axpby(1.0, r, beta, p)                                           # p = r + beta p   :180
        rr_old = rr_new
    return (n_iter if converged else -n_iter), np.array(res)


def eighs(A: _OpBase, x0: np.ndarray, nev: int, kdim: int | None = None, tolerance: float = RTOL_DP):
    """Lanczos eigensolver for symmetric / Hermitian operators.  src/IterativeSolvers/EIGHS/eighs.fypp:46-140.
    Returns (eigvals[nev], residuals[nev], X[n, nev], info = Lanczos steps)."""
    from scipy.linalg import eigh
    n, dt = x0.size, x0.dtype
    kdim = 4 * nev if kdim is None else kdim
    Xw = np.zeros((n, kdim + 1), dtype=dt, order="F")
    Xw[:, 0] = x0
    scal(Xw[:, 0], 1.0 / norm(x0))
    T = np.zeros((kdim + 1, kdim), dtype=dt, order="F")
    vals = np.zeros(kdim)
    vecs = np.zeros((kdim, kdim), dtype=dt)
    res = np.zeros(kdim)
    k = 0
    for k in range(1, kdim + 1):
        lanczos(A, Xw, T, kstart=k, kend=k)
        vals[:] = 0
        vecs[:] = 0
        w, v = eigh(T[:k, :k], lower=False, driver="ev")                 # stdlib eigh = syev / heev on the upper triangle (upper_a defaults to .true.), ascending
        vals[:k], vecs[:k, :k] = w, v
        res[:k] = np.abs(T[k, k - 1] * vecs[k - 1, :k])
        if np.count_nonzero(res[:k] < tolerance) >= nev:
            break
    idx = np.argsort(-vals, kind="stable")                               # sort_index(eigvals_wrk, reverse=.true.) over ALL kdim entries
    vals, vecs, res = vals[idx], vecs[:, idx], res[idx]
    k = min(k, kdim)
    X = np.zeros((n, nev), dtype=dt, order="F")
    for i in range(nev):
        for j in range(k):
            axpby(vecs[j, i], Xw[:, j], 1.0, X[:, i])
    return vals[:nev].copy(), res[:nev].copy(), X, k


def svds(A: _OpBase, Ah: _OpBase, u0: np.ndarray, nsv: int, kdim: int | None = None, tolerance: float = RTOL_DP):
    """Golub-Kahan singular value solver.  src/IterativeSolvers/SVDS/svd_solvers.fypp:44-150.
    Returns (S[nsv], residuals[nsv], U[n, nsv], V[n, nsv], info)."""
    from scipy.linalg import svd
    n, dt = u0.size, u0.dtype
    kdim = 4 * nsv if kdim is None else kdim
    Uw = np.zeros((n, kdim + 1), dtype=dt, order="F")
    Vw = np.zeros((n, kdim + 1), dtype=dt, order="F")
    Uw[:, 0] = u0
    scal(Uw[:, 0], 1.0 / norm(u0))
    B = np.zeros((kdim + 1, kdim), dtype=dt, order="F")
    sv = np.zeros(kdim)
    um = np.zeros((kdim, kdim), dtype=dt)
    vm = np.zeros((kdim, kdim), dtype=dt)
    res = np.zeros(kdim)
    k = 0
    for k in range(1, kdim + 1):
        bidiagonalization(A, Ah, Uw, Vw, B, tol=tolerance, kstart=k, kend=k)
        sv[:] = 0
        um[:] = 0
        vm[:] = 0
        u, s, vh = svd(B[:k, :k])
        sv[:k], um[:k, :k] = s, u
        vm[:k, :k] = vh.conj().T                                         # vmat = hermitian(vmat)
        res[:k] = np.abs(B[k, k - 1] * vm[k - 1, :k])
        if np.count_nonzero(res[:k] < tolerance) >= nsv:
            break
    k = min(k, kdim)
    U = np.zeros((n, nsv), dtype=dt, order="F")
    V = np.zeros((n, nsv), dtype=dt, order="F")
    for i in range(nsv):
        for j in range(k):
            axpby(um[j, i], Uw[:, j], 1.0, U[:, i])
            axpby(vm[j, i], Vw[:, j], 1.0, V[:, i])
    return sv[:nsv].copy(), res[:nsv].copy(), U, V, k
