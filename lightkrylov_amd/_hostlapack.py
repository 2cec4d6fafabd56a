"""Host LAPACK `geev` for the small Hessenberg problems of `eigs`, callable from SEVERAL host threads at once.

The reference calls `geev` on H(1:k, 1:k) after every Arnoldi step (IterativeSolvers.fypp:1065,
submodule_utility_functions.fypp:55-85): O(k^3) on the host, 128 times per Krylov-Schur cycle at kdim = 128 --
about a second of LAPACK next to 0.08 s of GPU work for BASELINE's configs[3].  scipy's `lapack.zgeev` wrapper
holds the interpreter lock, so those independent problems cannot run side by side through it.  This module
calls the SAME routine of the SAME library scipy uses (`scipy_dgeev_` / `scipy_zgeev_` in the OpenBLAS that
ships inside the scipy wheel) through ctypes, which releases the lock for the duration of the call; results are
bit-identical to `scipy.linalg.lapack.{d,z}geev` (tests/test_host_logic.py).  If that library cannot be located
the scipy wrapper is used (one problem at a time).  Host-side control flow only: nothing here touches vectors.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import threading

import numpy as np
from scipy.linalg import lapack as _lapack

_lib = None
_lock = threading.Lock()
_tried = False


def _load():
    global _lib, _tried
    with _lock:
        if _tried:
            return _lib
        _tried = True
        try:
            import scipy
            cands = sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so*")))
            for path in cands:
                lib = C.CDLL(path)
                if all(hasattr(lib, f) for f in ("scipy_dgeev_", "scipy_zgeev_", "scipy_dsyev_", "scipy_zheev_", "scipy_dgesdd_", "scipy_zgesdd_")):
                    _lib = lib
                    break
        except Exception:  # noqa: BLE001
            _lib = None
        return _lib


def threaded() -> bool:
    """True when geev() can run concurrently from several threads."""
    return _load() is not None


_controller = None


def _all_blas():
    """threadpoolctl's controller over EVERY BLAS loaded in the process (numpy and scipy each ship their own OpenBLAS, each
    with its own worker threads), built once; None when threadpoolctl is not installed."""
    global _controller
    if _controller is None:
        try:
            import threadpoolctl
            _controller = threadpoolctl.ThreadpoolController()
        except Exception:  # noqa: BLE001
            _controller = False
    return _controller or None


class blas_threads:
    """Context manager: the BLAS libraries' own worker threads off while WE parallelise over problems (OpenBLAS' threading
    only slows 128 x 128 problems down: 2.3 s vs 1.3 s for the 128 problems of one cycle on 8 cores)."""

    def __init__(self, n: int = 1):
        self.n, self.prev, self.cm = n, None, None

    def __enter__(self):
        ctl = _all_blas()
        if ctl is not None:
            self.cm = ctl.limit(limits=self.n)
            self.cm.__enter__()
            return self
        lib = _load()
        if lib is not None and hasattr(lib, "scipy_openblas_set_num_threads"):
            self.prev = lib.scipy_openblas_get_num_threads()
            lib.scipy_openblas_set_num_threads(C.c_int(self.n))
        return self

    def __exit__(self, *exc):
        if self.cm is not None:
            self.cm.__exit__(*exc)
            self.cm = None
        elif self.prev is not None:
            _load().scipy_openblas_set_num_threads(C.c_int(self.prev))
        return False


def small_problems(fn):
    """Decorator for host routines whose LAPACK work is all on matrices of at most kdim x kdim (<= a few hundred): run them
    with OpenBLAS' own threading off.  Its threading gains nothing at that size, and its workers busy-wait for a while after
    every parallel call (numpy's copy as much as scipy's): on the GPU box (128 hardware threads) the Arnoldi batch of an `eigs` call that started right after
    the previous call's Schur step took 98-116 ms instead of 74 -- the spinning workers delay the HIP runtime's threads."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with blas_threads(1):
            return fn(*args, **kwargs)
    return wrapper


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def geev(Hk: np.ndarray):
    """(vr, vals) = right eigenvectors in LAPACK layout (real pairs NOT combined) and eigenvalues of Hk --
    the same contract as solvers.eig."""
    lib = _load()
    k = Hk.shape[0]
    if lib is None or k == 0:
        if Hk.dtype == np.float64:
            wr, wi, _vl, vr, info = _lapack.dgeev(np.asfortranarray(Hk), compute_vl=0, compute_vr=1)
            vals = wr + 1j * wi
        else:
            vals, _vl, vr, info = _lapack.zgeev(np.asfortranarray(Hk), compute_vl=0, compute_vr=1)
        if info != 0:
            raise RuntimeError(f"GEEV failed, info={info}")
        return vr, vals
    cplx = Hk.dtype == np.complex128
    a = np.array(Hk, dtype=Hk.dtype, order="F", copy=True)          # geev overwrites its input
    vr = np.empty((k, k), dtype=Hk.dtype, order="F")
    vl = np.empty((1, 1), dtype=Hk.dtype, order="F")
    n, lda, ldvl, ldvr, info = C.c_int(k), C.c_int(k), C.c_int(1), C.c_int(k), C.c_int(0)
    jobvl, jobvr = C.c_char(b"N"), C.c_char(b"V")
    one = C.c_size_t(1)
    if cplx:
        w = np.empty(k, dtype=np.complex128)
        rwork = np.empty(2 * k, dtype=np.float64)

        def call(work, lw):
            lib.scipy_zgeev_(C.byref(jobvl), C.byref(jobvr), C.byref(n), _p(a), C.byref(lda), _p(w), _p(vl), C.byref(ldvl),
                             _p(vr), C.byref(ldvr), _p(work), C.byref(lw), _p(rwork), C.byref(info), one, one)
        lw = C.c_int(max(2 * k, 1))                                 # scipy's default lwork: same code path, same rounding
        call(np.empty(lw.value, dtype=np.complex128), lw)
        vals = w
    else:
        wr, wi = np.empty(k), np.empty(k)

        def call(work, lw):
            lib.scipy_dgeev_(C.byref(jobvl), C.byref(jobvr), C.byref(n), _p(a), C.byref(lda), _p(wr), _p(wi), _p(vl),
                             C.byref(ldvl), _p(vr), C.byref(ldvr), _p(work), C.byref(lw), C.byref(info), one, one)
        lw = C.c_int(max(4 * k, 1))
        call(np.empty(lw.value, dtype=np.float64), lw)
        vals = wr + 1j * wi
    if info.value != 0:
        raise RuntimeError(f"GEEV failed, info={info.value}")
    return vr, vals


def gees(Hm: np.ndarray):
    """(T, Z, w) = Schur form, Schur vectors and eigenvalues of Hm by LAPACK gees without sorting (jobvs = 'V', sort = 'N') --
    what stdlib's `schur` (BaseKrylov.fypp:807) calls -- with scipy's default workspace size (lwork = 3 n: same code path, same
    rounding as scipy.linalg.lapack.{d,z}gees, tests/test_host_logic.py), through ctypes so that the interpreter lock is released
    for the duration of the call: eigs runs the restart's Schur factorisation beside the last Ritz tests of a cycle."""
    lib = _load()
    k = Hm.shape[0]
    if lib is None or k == 0 or not (hasattr(lib, "scipy_dgees_") and hasattr(lib, "scipy_zgees_")):
        if Hm.dtype == np.float64:
            T, _sdim, wr, wi, Z, _work, info = _lapack.dgees(lambda *a: False, np.asfortranarray(Hm), sort_t=0)
            w = wr + 1j * wi
        else:
            T, _sdim, w, Z, _work, info = _lapack.zgees(lambda *a: False, np.asfortranarray(Hm), sort_t=0)
        if info != 0:
            raise RuntimeError(f"GEES failed, info={info}")
        return T, Z, w
    cplx = Hm.dtype == np.complex128
    a = np.array(Hm, dtype=Hm.dtype, order="F", copy=True)          # overwritten by T
    vs = np.empty((k, k), dtype=Hm.dtype, order="F")
    n, lda, ldvs, sdim, info = C.c_int(k), C.c_int(k), C.c_int(k), C.c_int(0), C.c_int(0)
    jobvs, sort = C.c_char(b"V"), C.c_char(b"N")
    one = C.c_size_t(1)
    lw = C.c_int(max(3 * k, 1))
    work = np.empty(lw.value, dtype=Hm.dtype)
    bwork = np.empty(max(k, 1), dtype=np.int32)
    nosel = C.c_void_p(None)                                        # SELECT is not referenced when SORT = 'N'
    if cplx:
        w = np.empty(k, dtype=np.complex128)
        rwork = np.empty(max(k, 1), dtype=np.float64)
        lib.scipy_zgees_(C.byref(jobvs), C.byref(sort), nosel, C.byref(n), _p(a), C.byref(lda), C.byref(sdim), _p(w), _p(vs), C.byref(ldvs),
                         _p(work), C.byref(lw), _p(rwork), _p(bwork), C.byref(info), one, one)
    else:
        wr, wi = np.empty(k), np.empty(k)
        lib.scipy_dgees_(C.byref(jobvs), C.byref(sort), nosel, C.byref(n), _p(a), C.byref(lda), C.byref(sdim), _p(wr), _p(wi), _p(vs),
                         C.byref(ldvs), _p(work), C.byref(lw), _p(bwork), C.byref(info), one, one)
        w = wr + 1j * wi
    if info.value != 0:
        raise RuntimeError(f"GEES failed, info={info.value}")
    return a, vs, w


def syev(Tk: np.ndarray):
    """(w, v) = eigenvalues (ascending) and orthonormal eigenvectors of the symmetric / Hermitian matrix whose UPPER triangle is
    Tk's -- LAPACK syev / heev with a workspace query, i.e. what stdlib's `eigh(a, lambda, vectors)` (the routine `eighs` calls,
    EIGHS/eighs.fypp:87; stdlib is not vendored with the reference: `upper_a` defaults to .true. there) does.  Callable from
    several threads at once like geev(); falls back to scipy's wrapper of the same routine."""
    lib = _load()
    k = Tk.shape[0]
    cplx = Tk.dtype == np.complex128
    if lib is None or k == 0:
        from scipy.linalg import eigh
        return eigh(Tk, lower=False, driver="ev")
    a = np.array(Tk, dtype=Tk.dtype, order="F", copy=True)          # overwritten by the eigenvectors
    w = np.empty(k, dtype=np.float64)
    n, lda, info = C.c_int(k), C.c_int(k), C.c_int(0)
    jobz, uplo = C.c_char(b"V"), C.c_char(b"U")
    one = C.c_size_t(1)
    rwork = np.empty(max(1, 3 * k - 2), dtype=np.float64)

    def call(work, lw):
        if cplx:
            lib.scipy_zheev_(C.byref(jobz), C.byref(uplo), C.byref(n), _p(a), C.byref(lda), _p(w), _p(work), C.byref(lw), _p(rwork),
                             C.byref(info), one, one)
        else:
            lib.scipy_dsyev_(C.byref(jobz), C.byref(uplo), C.byref(n), _p(a), C.byref(lda), _p(w), _p(work), C.byref(lw),
                             C.byref(info), one, one)
    q = np.empty(1, dtype=Tk.dtype)
    call(q, C.c_int(-1))                                             # workspace query
    lw = C.c_int(max(1, int(q[0].real)))
    call(np.empty(lw.value, dtype=Tk.dtype), lw)
    if info.value != 0:
        raise RuntimeError(f"SYEV/HEEV failed, info={info.value}")
    return w, a


def gesdd(Bk: np.ndarray):
    """(u, s, vh) = full singular value decomposition of the square matrix Bk by LAPACK gesdd (jobz = 'A') with a workspace
    query: what stdlib's `svd(a, s, u, vt)` (SVDS/svd_solvers.fypp:102) and scipy.linalg.svd's default driver call.  Callable
    from several threads at once; bit-identical to scipy.linalg.svd(Bk) (tests/test_host_logic.py)."""
    lib = _load()
    k = Bk.shape[0]
    if lib is None or k == 0:
        from scipy.linalg import svd
        return svd(Bk)
    cplx = Bk.dtype == np.complex128
    a = np.array(Bk, dtype=Bk.dtype, order="F", copy=True)
    sv = np.empty(k, dtype=np.float64)
    u = np.empty((k, k), dtype=Bk.dtype, order="F")
    vt = np.empty((k, k), dtype=Bk.dtype, order="F")
    n, ld, info = C.c_int(k), C.c_int(k), C.c_int(0)
    jobz = C.c_char(b"A")
    one = C.c_size_t(1)
    iwork = np.empty(8 * k, dtype=np.int32)
    rwork = np.empty(max(1, 5 * k * k + 5 * k), dtype=np.float64)

    def call(work, lw):
        if cplx:
            lib.scipy_zgesdd_(C.byref(jobz), C.byref(n), C.byref(n), _p(a), C.byref(ld), _p(sv), _p(u), C.byref(ld), _p(vt), C.byref(ld),
                              _p(work), C.byref(lw), _p(rwork), _p(iwork), C.byref(info), one)
        else:
            lib.scipy_dgesdd_(C.byref(jobz), C.byref(n), C.byref(n), _p(a), C.byref(ld), _p(sv), _p(u), C.byref(ld), _p(vt), C.byref(ld),
                              _p(work), C.byref(lw), _p(iwork), C.byref(info), one)
    q = np.empty(1, dtype=Bk.dtype)
    call(q, C.c_int(-1))
    lw = C.c_int(max(1, int(q[0].real)))
    call(np.empty(lw.value, dtype=Bk.dtype), lw)
    if info.value != 0:
        raise RuntimeError(f"GESDD failed, info={info.value}")
    return u, sv, vt
