"""abstract_linop and GPU operators -- mirror of src/AbstractTypes/AbstractLinops.fypp.

`abstract_linop` keeps the reference contract: deferred `matvec(vec_in, vec_out)` /
`rmatvec`, and the counting wrappers `apply_matvec` / `apply_rmatvec` every Krylov routine
calls (AbstractLinops.fypp:58-87, 391-424).  Operators whose matvec is a kernel of the HIP
engine carry an engine handle (`_h`), which lets `arnoldi` run its whole step loop inside
the library (lk_arnoldi) with no Python between the matvec and the DGS sweeps.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .context import Context, default_context
from .vectors import _DT, abstract_vector, dense_vector_gpu


class abstract_linop:
    """AbstractLinops.fypp:30-87"""

    def __init__(self):
        self.matvec_counter = 0
        self.rmatvec_counter = 0

    def matvec(self, vec_in: abstract_vector, vec_out: abstract_vector) -> None:
        raise NotImplementedError

    def rmatvec(self, vec_in: abstract_vector, vec_out: abstract_vector) -> None:
        raise NotImplementedError

    def apply_matvec(self, vec_in, vec_out) -> None:
        """AbstractLinops.fypp:391-407"""
        self.matvec_counter += 1
        with _user_kernel_stream(self, vec_in):
            self.matvec(vec_in, vec_out)

    def apply_rmatvec(self, vec_in, vec_out) -> None:
        """AbstractLinops.fypp:409-424"""
        self.rmatvec_counter += 1
        with _user_kernel_stream(self, vec_in):
            self.rmatvec(vec_in, vec_out)

    def get_counter(self, trans: bool = False) -> int:
        return self.rmatvec_counter if trans else self.matvec_counter

    def reset_counter(self, trans: bool = False, _procedure: str = "") -> None:
        if trans:
            self.rmatvec_counter = 0
        else:
            self.matvec_counter = 0


def _user_kernel_stream(op, vec_in):
    """A user's operator may run torch ops / its own kernels on the vectors' device memory (dense_vector_gpu.as_torch):
    those must be ordered with the engine, so a python-level operator on GPU vectors runs with torch's current stream set
    to the engine's.  Engine operators (and processes that never imported torch) need nothing."""
    import contextlib
    import sys
    basis = getattr(vec_in, "basis", None)
    if isinstance(op, _engine_linop) or basis is None or "torch" not in sys.modules:
        return contextlib.nullcontext()
    return basis.ctx.torch_stream()


# ---- composite operators built on the vector contract only (AbstractLinops.fypp; generated .f90 lines cited) ----
class Id(abstract_linop):
    """Identity: `copy(vec_out, vec_in)`.  AbstractLinops.f90:350-357, 974-980"""

    def matvec(self, vec_in, vec_out) -> None:
        from .vectors import copy
        copy(vec_out, vec_in)

    rmatvec = matvec


class scaled_linop(abstract_linop):
    """B = sigma A (the adjoint is scaled by sigma too, NOT conj(sigma), like the reference).
    AbstractLinops.f90:395-403, 1011-1025"""

    def __init__(self, A: abstract_linop, sigma):
        super().__init__()
        self.A, self.sigma = A, sigma

    def matvec(self, vec_in, vec_out) -> None:
        self.A.apply_matvec(vec_in, vec_out)
        vec_out.scal(self.sigma)

    def rmatvec(self, vec_in, vec_out) -> None:
        self.A.apply_rmatvec(vec_in, vec_out)
        vec_out.scal(self.sigma)


class axpby_linop(abstract_linop):
    """C = alpha op(A) + beta op(B).  AbstractLinops.f90:451-459, 1125-1191"""

    def __init__(self, A: abstract_linop, B: abstract_linop, alpha=1.0, beta=1.0, transA: bool = False, transB: bool = False):
        super().__init__()
        self.A, self.B, self.alpha, self.beta, self.transA, self.transB = A, B, alpha, beta, transA, transB

    def _apply(self, vec_in, vec_out, adjoint: bool) -> None:
        wrk = vec_in.zeros_like()
        wrk.zero()
        (self.A.apply_rmatvec if self.transA != adjoint else self.A.apply_matvec)(vec_in, wrk)
        (self.B.apply_rmatvec if self.transB != adjoint else self.B.apply_matvec)(vec_in, vec_out)
        vec_out.axpby(self.alpha, wrk, self.beta)                  # y = alpha*w + beta*y

    def matvec(self, vec_in, vec_out) -> None:
        self._apply(vec_in, vec_out, False)

    def rmatvec(self, vec_in, vec_out) -> None:
        self._apply(vec_in, vec_out, True)


class adjoint_linop(abstract_linop):
    """matvec and rmatvec of A switched (does not compute an adjoint).  AbstractLinops.f90:155-163, 1372-1386"""

    def __init__(self, A: abstract_linop):
        super().__init__()
        self.A = A

    def matvec(self, vec_in, vec_out) -> None:
        self.A.apply_rmatvec(vec_in, vec_out)

    def rmatvec(self, vec_in, vec_out) -> None:
        self.A.apply_matvec(vec_in, vec_out)


class _engine_linop(abstract_linop):
    """An operator implemented by a kernel of the HIP engine."""

    def __init__(self, ctx: Context | None):
        super().__init__()
        self.ctx = ctx or default_context()
        self._lib = _capi.load()
        self._h = C.c_void_p()

    def _apply(self, trans: int, vec_in, vec_out) -> None:
        if not (isinstance(vec_in, dense_vector_gpu) and isinstance(vec_out, dense_vector_gpu)):
            raise TypeError("engine operators act on dense_vector_gpu")  # type_error(...)
        _capi.check(self._lib.lk_linop_apply(self._h, trans, vec_in.basis._h, vec_in.col,
                                             vec_out.basis._h, vec_out.col))

    def matvec(self, vec_in, vec_out) -> None:
        self._apply(_capi.LK_OP_N, vec_in, vec_out)

    def rmatvec(self, vec_in, vec_out) -> None:
        self._apply(_capi.LK_OP_H, vec_in, vec_out)

    def close(self) -> None:
        if self._h:
            self._lib.lk_linop_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _row_starts(ctx: Context, n_global: int, row_starts):
    """(nranks + 1) int64 row offsets of the contiguous row blocks; default = `row_partition` of the context's ranks."""
    from .context import row_partition
    if row_starts is None:
        row_starts = [row_partition(n_global, ctx.nranks, r)[0] for r in range(ctx.nranks)] + [n_global]
    rs = np.ascontiguousarray(row_starts, dtype=np.int64)
    if rs.shape != (ctx.nranks + 1,):
        raise ValueError(f"row_starts needs nranks + 1 = {ctx.nranks + 1} entries")
    return rs


class dense_linop_gpu(_engine_linop):
    """dense_linop_{rdp,cdp}: y = A x via gemv('N') / ('T'|'C').  AbstractLinops.fypp:265-271, 608-660.
    Row-sharded (one process per GPU): pass this rank's ROW BLOCK `A[row0:row0 + n_local, :]` and `n_global`; x is
    all-gathered for matvec, the ranks' A_rows^H x_rows are summed for rmatvec (lk_linop_dense_create_sharded)."""

    def __init__(self, A: np.ndarray, ctx: Context | None = None, n_global: int | None = None, row_starts=None):
        super().__init__(ctx)
        A = np.asfortranarray(A)
        if A.dtype not in _DT or A.ndim != 2:
            raise TypeError("dense_linop_gpu needs a float64/complex128 matrix")
        self.dtype = A.dtype
        if n_global is None and self.ctx.nranks == 1:
            if A.shape[0] != A.shape[1]:
                raise TypeError("dense_linop_gpu needs a square matrix")
            self.n = A.shape[0]
            _capi.check(self._lib.lk_linop_dense_create(self.ctx._h, _DT[A.dtype], self.n,
                                                        A.ctypes.data_as(C.c_void_p), A.shape[0], C.byref(self._h)))
            return
        n_global = A.shape[1] if n_global is None else int(n_global)
        rs = _row_starts(self.ctx, n_global, row_starts)
        self.n = int(rs[self.ctx.rank + 1] - rs[self.ctx.rank])
        if A.shape != (self.n, n_global):
            raise TypeError(f"this rank's block must be {self.n} x {n_global}, got {A.shape}")
        _capi.check(self._lib.lk_linop_dense_create_sharded(self.ctx._h, _DT[A.dtype], n_global, rs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                            A.ctypes.data_as(C.c_void_p), max(A.shape[0], 1), C.byref(self._h)))

    @classmethod
    def from_device_panel(cls, panel, n_global: int | None = None, row_starts=None):
        """The operator on a matrix that already lies in HBM: `panel` is a krylov_basis_gpu whose columns are the COLUMNS of
        this rank's row block (n_local rows x n_global columns).  Not copied; the panel must outlive the operator."""
        self = cls.__new__(cls)
        _engine_linop.__init__(self, panel.ctx)
        self.dtype = np.dtype(panel.dtype)
        n_global = panel.ncols if n_global is None else int(n_global)
        rs = _row_starts(self.ctx, n_global, row_starts)
        self.n = int(rs[self.ctx.rank + 1] - rs[self.ctx.rank])
        if panel.n_local != self.n or panel.ncols != n_global:
            raise TypeError("panel shape does not match this rank's row block")
        dt, nl, nc, ld, ptr = C.c_int(), C.c_int64(), C.c_int(), C.c_int64(), C.c_void_p()
        _capi.check(self._lib.lk_basis_info(panel._h, C.byref(dt), C.byref(nl), C.byref(nc), C.byref(ld), C.byref(ptr)))
        _capi.check(self._lib.lk_linop_dense_wrap_sharded(self.ctx._h, dt.value, n_global, rs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                          ptr, ld.value, C.byref(self._h)))
        self._panel = panel
        return self


class csr_linop_gpu(_engine_linop):
    """A user's sparse `abstract_linop` (AbstractLinops.fypp:58-87) in CSR: y = A x / A^H x on the device.
    `A`: anything with `.indptr`, `.indices`, `.data`, `.shape` in CSR layout (a scipy.sparse.csr_matrix / csr_array), or
    a tuple (rowptr, colind, vals) with 0-based indices.  Row-sharded: pass this rank's rows (GLOBAL column indices) and
    `n_global` (lk_linop_csr_create_sharded)."""

    def __init__(self, A, ctx: Context | None = None, n_global: int | None = None, row_starts=None):
        super().__init__(ctx)
        sharded = not (n_global is None and self.ctx.nranks == 1)
        try:
            if isinstance(A, tuple):
                rowptr, colind, vals = A
                n = len(rowptr) - 1
                ncols = n if n_global is None else int(n_global)
            else:
                if getattr(A, "format", "csr") != "csr":
                    A = A.tocsr()
                rowptr, colind, vals, n, ncols = A.indptr, A.indices, A.data, A.shape[0], A.shape[1]
                if n_global is not None and ncols != n_global:
                    raise TypeError("csr_linop_gpu: the block's column count must be n_global")
            vals = np.ascontiguousarray(vals)
            if vals.dtype not in _DT:
                raise TypeError("csr_linop_gpu needs float64 / complex128 values")
            rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
            colind = np.ascontiguousarray(colind, dtype=np.int32)
            self.dtype, self.n, self.nnz = vals.dtype, int(n), int(rowptr[-1])
            if not sharded:
                if n != ncols:
                    raise TypeError("csr_linop_gpu needs a square matrix")
                _capi.check(self._lib.lk_linop_csr_create(self.ctx._h, _DT[vals.dtype], self.n, rowptr.ctypes.data_as(C.c_void_p),
                                                          colind.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p),
                                                          C.byref(self._h)))
                return
            rs = _row_starts(self.ctx, int(ncols), row_starts)
            if int(rs[self.ctx.rank + 1] - rs[self.ctx.rank]) != self.n:
                raise TypeError("csr_linop_gpu: the number of rows passed is not this rank's block")
        except Exception:  # noqa: BLE001 - ANY local failure, then re-raised
            # the sharded creation is COLLECTIVE: a rank whose input is unusable -- for whatever reason: a wrong type or shape, an
            # object without tocsr / indptr (AttributeError), an index that does not fit int32 (OverflowError), a failed allocation in
            # ascontiguousarray (MemoryError) -- must not leave the others waiting in the metadata exchange: it joins the library's
            # status agreement with a null row block (every rank then fails) and raises its own error afterwards
            if sharded and self.ctx.nranks > 1:
                try:
                    rs = _row_starts(self.ctx, int(n_global) if n_global is not None else 0, row_starts)
                    self._lib.lk_linop_csr_create_sharded(self.ctx._h, _capi.LK_F64, int(rs[-1]), rs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                          None, None, None, C.byref(self._h))
                except (TypeError, ValueError):
                    pass                    # the partition itself is unusable: the same arguments fail on every rank alike
            raise
        _capi.check(self._lib.lk_linop_csr_create_sharded(self.ctx._h, _DT[vals.dtype], int(ncols), rs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                          rowptr.ctypes.data_as(C.c_void_p), colind.ctypes.data_as(C.c_void_p),
                                                          vals.ctypes.data_as(C.c_void_p), C.byref(self._h)))


class diag_linop_gpu(_engine_linop):
    """y = d .* x (config "arnoldi with synthetic diagonal linop")."""

    def __init__(self, d: np.ndarray | None = None, ctx: Context | None = None, *, n_local: int | None = None,
                 row0: int = 0, d0: float = 1.0, dstep: float = 0.0):
        super().__init__(ctx)
        if d is not None:
            d = np.ascontiguousarray(d)
            self.dtype, self.n = d.dtype, d.shape[0]
            _capi.check(self._lib.lk_linop_diag_create(self.ctx._h, _DT[d.dtype], self.n,
                                                       d.ctypes.data_as(C.c_void_p), C.byref(self._h)))
        else:
            # d_i = d0 + dstep * (row0 + i), never materialised in HBM
            self.dtype, self.n = np.dtype(np.float64), int(n_local)
            _capi.check(self._lib.lk_linop_diag_linspace_create(self.ctx._h, self.n, int(row0), float(d0),
                                                                float(dstep), C.byref(self._h)))


class laplacian2d_linop_gpu(_engine_linop):
    """5-point Laplacian on an N x N grid, Dirichlet, scaled by (N+1)^2 (config 3, Poisson).
    Row-sharded runs pass the grid lines this rank owns (`j0`, `nj`, see `grid_partition`): vector rows
    [j0*N, (j0+nj)*N); one line is exchanged with each neighbouring rank per application (halo exchange)."""

    def __init__(self, N: int, ctx: Context | None = None, j0: int | None = None, nj: int | None = None):
        super().__init__(ctx)
        self.N = int(N)
        if j0 is None:
            self.dtype, self.n = np.dtype(np.float64), N * N
            _capi.check(self._lib.lk_linop_lap5_create(self.ctx._h, int(N), C.byref(self._h)))
        else:
            self.dtype, self.n = np.dtype(np.float64), int(nj) * N
            _capi.check(self._lib.lk_linop_lap5_create_sharded(self.ctx._h, int(N), int(j0), int(nj), C.byref(self._h)))

    def rmatvec(self, vec_in, vec_out) -> None:  # symmetric
        self._apply(_capi.LK_OP_N, vec_in, vec_out)


def grid_partition(N: int, nranks: int, rank: int):
    """Grid lines [j0, j0 + nj) of an N x N grid owned by `rank` (whole lines, so a rank's vector block is nj*N rows)."""
    base = N // nranks
    j0 = base * rank
    nj = base if rank < nranks - 1 else N - base * (nranks - 1)
    return j0, nj


class ginzburg_landau_linop_gpu(_engine_linop):
    """Exponential propagator of the linearised complex Ginzburg-Landau equation over `tau`
    (example/ginzburg_landau: `exponential_prop`, Ginzburg_Landau.f90:259-290), as `nsub` classical RK4
    steps of the reference right-hand side (stencil + boundary rows, :126-136).  Defaults are the example's
    constants (:23-33) with dx fixed at 200/513 and mu_2 rescaled by (200/L)^2 so that mu(x) keeps the
    reference's range when n grows."""

    def __init__(self, n: int, ctx: Context | None = None, tau: float = 0.01, nsub: int = 1,
                 nu: complex = 2.0 + 0.2j, gamma: complex = 1.0 - 1.0j, mu_0: float = 0.38, c_mu: float = 0.2,
                 mu_2: float | None = None, dx: float = 200.0 / 513.0, row0: int | None = None, n_local: int | None = None):
        """`n` is the GLOBAL size; row-sharded runs also pass this rank's block (`row0`, `n_local`) -- every RK4 stage then
        exchanges one point with each neighbouring rank (halo exchange)."""
        super().__init__(ctx)
        n_global = int(n)
        self.dtype = np.dtype(np.complex128)
        self.n = n_global if n_local is None else int(n_local)
        L = dx * (n_global + 1)
        self.mu_2 = -0.01 * (200.0 / L) ** 2 if mu_2 is None else mu_2
        self.params = dict(n=n_global, dx=dx, tau=tau, nsub=nsub, nu=nu, gamma=gamma, mu_c=mu_0 - c_mu ** 2, mu2=self.mu_2)
        nu_a = (C.c_double * 2)(nu.real, nu.imag)
        ga_a = (C.c_double * 2)(gamma.real, gamma.imag)
        if row0 is None:
            _capi.check(self._lib.lk_linop_gl_create(self.ctx._h, n_global, float(dx), float(tau), int(nsub), nu_a, ga_a,
                                                     float(mu_0 - c_mu ** 2), float(self.mu_2), C.byref(self._h)))
        else:
            _capi.check(self._lib.lk_linop_gl_create_sharded(self.ctx._h, n_global, int(row0), int(self.n), float(dx), float(tau),
                                                             int(nsub), nu_a, ga_a, float(mu_0 - c_mu ** 2), float(self.mu_2),
                                                             C.byref(self._h)))
