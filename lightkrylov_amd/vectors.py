"""abstract_vector / dense_vector_gpu -- host-side mirror of src/AbstractTypes/AbstractVectors.fypp.

`abstract_vector` restates the reference's extension contract (six deferred procedures +
derived norm/add/sub/chsgn, AbstractVectors.fypp:295-320, 424-460).  `dense_vector_gpu` is
the concrete MI355X type: a (panel, column) pair in HBM whose procedures are single calls
into the HIP engine.  A Krylov basis `X(:)` is a `krylov_basis_gpu` = one column-contiguous
panel; `X[j]` is a view of column j (0-based; the reference's X(j+1)).

The free functions (`innerprod`, `linear_combination`, `Gram`, `copy`, `zero_basis`, ...)
keep the reference's names and meaning.  Given GPU bases they are ONE fused panel kernel
each; given any other `abstract_vector` implementation they run the reference's generic
loops over the type-bound procedures (that is the extension API, not a fallback: this
package ships no CPU vector type).
"""
from __future__ import annotations

import ctypes as C
import itertools
from typing import Sequence

import numpy as np

from . import _capi
from .constants import atol_dp  # noqa: F401  (re-export convenience)
from .context import Context, default_context

_DT = {np.dtype(np.float64): _capi.LK_F64, np.dtype(np.complex128): _capi.LK_C128}


def _sc(val, dtype) -> "C.Array":
    """Scalar -> 1 or 2 doubles as the C ABI wants them."""
    if np.dtype(dtype).kind == "c":
        z = complex(val)
        return (C.c_double * 2)(z.real, z.imag)
    v = complex(val)
    if v.imag != 0.0:
        raise TypeError("complex scalar passed to a real(dp) vector")
    return (C.c_double * 2)(v.real, 0.0)


# ------------------------------------------------------------------------------------------
class abstract_vector:
    """AbstractVectors.fypp:295-381.  Subclasses implement the six deferred procedures."""

    def zero(self) -> None: raise NotImplementedError
    def rand(self, ifnorm: bool = False) -> None: raise NotImplementedError
    def scal(self, alpha) -> None: raise NotImplementedError
    def axpby(self, alpha, vec: "abstract_vector", beta) -> None: raise NotImplementedError
    def dot(self, vec: "abstract_vector"): raise NotImplementedError
    def get_size(self) -> int: raise NotImplementedError

    # new vector of the same dynamic type and size (`allocate(y, source=X(1))` + zero)
    def zeros_like(self) -> "abstract_vector": raise NotImplementedError

    # -- derived (AbstractVectors.fypp:424-460)
    def norm(self) -> float:
        return float(np.sqrt(abs(self.dot(self))))

    def sub(self, vec) -> None:
        self.axpby(-1.0, vec, 1.0)

    def add(self, vec) -> None:
        self.axpby(1.0, vec, 1.0)

    def chsgn(self) -> None:
        self.scal(-1.0)


# ------------------------------------------------------------------------------------------
class krylov_basis_gpu:
    """`class(abstract_vector), allocatable :: X(:)` as one HBM panel (n_local x ncols)."""

    def __init__(self, n_local: int, ncols: int, dtype=np.float64, ctx: Context | None = None,
                 _handle=None, _owner=None):
        self.ctx = ctx or default_context()
        self.dtype = np.dtype(dtype)
        if self.dtype not in _DT:
            raise TypeError(f"dense_vector_gpu supports float64 / complex128, got {self.dtype}")
        self.n_local, self.ncols = int(n_local), int(ncols)
        self._lib = _capi.load()
        self._owner = _owner  # keeps the parent panel alive for column-offset views
        if _handle is not None:
            self._h = _handle
        else:
            self._h = C.c_void_p()
            _capi.check(self._lib.lk_basis_create(self.ctx._h, _DT[self.dtype], self.n_local, self.ncols,
                                                  C.byref(self._h)))

    # --- python container protocol: X[j] -> vector view, X[a:b] -> basis view
    def __len__(self) -> int:
        return self.ncols

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.ncols)
            if step != 1:
                raise IndexError("basis slices must be contiguous")
            return self.view(start, stop - start)
        j = int(idx)
        if j < 0:
            j += self.ncols
        if not 0 <= j < self.ncols:
            raise IndexError(j)
        return dense_vector_gpu(_basis=self, _col=j)

    def __iter__(self):
        return (self[j] for j in range(self.ncols))

    def info(self):
        dt, n, nc, ld, ptr = C.c_int(), C.c_int64(), C.c_int(), C.c_int64(), C.c_void_p()
        _capi.check(self._lib.lk_basis_info(self._h, C.byref(dt), C.byref(n), C.byref(nc), C.byref(ld), C.byref(ptr)))
        return dt.value, n.value, nc.value, ld.value, ptr.value

    def view(self, col0: int, ncols: int) -> "krylov_basis_gpu":
        """Columns [col0, col0+ncols) as a basis of their own (no copy)."""
        if col0 == 0 and ncols == self.ncols:
            return self
        if ncols < 1 or col0 < 0 or col0 + ncols > self.ncols:
            raise IndexError(f"basis view [{col0}:{col0 + ncols}) out of range")
        if col0 == 0:
            return _prefix_view(self, ncols)
        _dt, _n, _nc, ld, ptr = self.info()
        h = C.c_void_p()
        off = col0 * ld * self.dtype.itemsize
        _capi.check(self._lib.lk_basis_wrap(self.ctx._h, _DT[self.dtype], self.n_local, ncols, ld,
                                            C.c_void_p(ptr + off), C.byref(h)))
        return krylov_basis_gpu(self.n_local, ncols, self.dtype, self.ctx, _handle=h, _owner=self)

    # --- host <-> device
    def upload(self, host: np.ndarray, col0: int = 0) -> None:
        a = np.asfortranarray(host, dtype=self.dtype).reshape(self.n_local, -1, order="F")
        _capi.check(self._lib.lk_basis_upload(self._h, col0, a.shape[1], a.ctypes.data_as(C.c_void_p), a.shape[0]))

    def download(self, col0: int = 0, ncols: int | None = None) -> np.ndarray:
        ncols = self.ncols - col0 if ncols is None else ncols
        out = np.empty((self.n_local, ncols), dtype=self.dtype, order="F")
        _capi.check(self._lib.lk_basis_download(self._h, col0, ncols, out.ctypes.data_as(C.c_void_p), max(self.n_local, 1)))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) and isinstance(self._h, C.c_void_p) and self._h.value:
            self._lib.lk_basis_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _prefix_view(krylov_basis_gpu):
    """X(:k): shares the parent's handle; only the logical column count differs."""

    def __init__(self, parent: krylov_basis_gpu, k: int):  # noqa: D401 - no super().__init__ on purpose
        self.ctx, self.dtype = parent.ctx, parent.dtype
        self.n_local, self.ncols = parent.n_local, int(k)
        self._lib, self._h, self._owner = parent._lib, parent._h, parent

    def close(self) -> None:  # the parent owns the handle
        pass


_pool_tags = itertools.count(1)


class _pool_column(krylov_basis_gpu):
    """One column of a pool slab (lk_pool_*, the header's section on object ownership).  The temporaries of the
    per-object path -- `allocate(y, source=X(1))` in linear_combination (AbstractVectors.fypp:595-598), `wrk` in
    gmres -- live here exactly as the Fortran plugin's do: no allocation per call, and they sit in panels."""

    def __init__(self, n_local: int, dtype, ctx: Context | None):  # noqa: D401 - no super().__init__ on purpose
        self.ctx = ctx or default_context()
        self.dtype = np.dtype(dtype)
        self.n_local = int(n_local)
        self._lib, self._owner = _capi.load(), None
        slab, col = C.c_void_p(), C.c_int()
        _capi.check(self._lib.lk_pool_acquire(self.ctx._h, _DT[self.dtype], self.n_local, C.c_uint64(next(_pool_tags)),
                                              C.byref(slab), C.byref(col)))
        self._h, self.col = slab, col.value
        self.ncols = self.info()[2]

    def close(self) -> None:  # the column goes back to the pool; the slab belongs to the context
        if getattr(self, "_h", None) and self._h.value and self.ctx._h:
            self._lib.lk_pool_release(self.ctx._h, self._h, self.col)
        self._h = C.c_void_p()


class dense_vector_gpu(abstract_vector):
    """extends(abstract_vector_{rdp,cdp}): the GPU counterpart of dense_vector
    (AbstractVectors.fypp:390-407, 476-562)."""

    def __init__(self, n_local: int | None = None, dtype=np.float64, ctx: Context | None = None,
                 _basis: krylov_basis_gpu | None = None, _col: int = 0, seed: int = 0):
        if _basis is None:
            _basis = krylov_basis_gpu(int(n_local), 1, dtype, ctx)
        self.basis, self.col = _basis, int(_col)
        self.dtype = _basis.dtype
        self._lib = _basis._lib
        self._seed = seed

    @classmethod
    def from_array(cls, x: np.ndarray, ctx: Context | None = None) -> "dense_vector_gpu":
        """dense_vector(x) constructor (AbstractVectors.fypp:469-474)."""
        x = np.asarray(x)
        v = cls(x.shape[0], x.dtype, ctx)
        v.basis.upload(x.reshape(-1, 1))
        return v

    def to_array(self) -> np.ndarray:
        return self.basis.download(self.col, 1)[:, 0]

    def as_torch(self, access: str = "rw"):
        """The vector as a torch tensor that ALIASES its device memory (float64 of n rows, or complex128), for operators
        written with torch ops or custom kernels: `class my_op(lk.abstract_linop): def matvec(self, vi, vo):
        torch.mul(d, vi.as_torch("r"), out=vo.as_torch("w"))`.  `access`: "r" read, "w" overwrite (previous contents
        not read), "rw".  Pending engine work on the vector is applied first (lk_vec_device_ptr).  The tensor must be used
        on the ENGINE's stream: inside an operator's matvec / rmatvec called through apply_matvec / apply_rmatvec that is
        already the case; elsewhere wrap the torch code in `with ctx.torch_stream():` (checked here -- a torch op on
        another stream would race with the engine silently).  Do not keep it across later engine calls that write the vector."""
        import torch
        from .context import _DevMem
        ctx = self.basis.ctx
        if torch.cuda.current_stream(ctx.device).cuda_stream != ctx.engine_stream():
            raise RuntimeError("as_torch: torch's current stream is not the engine's stream; use `with ctx.torch_stream():` "
                               "(operators applied through apply_matvec / apply_rmatvec already run inside one)")
        acc = {"r": _capi.LK_ACCESS_READ, "w": _capi.LK_ACCESS_OVERWRITE, "rw": _capi.LK_ACCESS_READWRITE}[access]
        ptr = C.c_void_p()
        _capi.check(self._lib.lk_vec_device_ptr(self.basis._h, self.col, acc, C.byref(ptr)))
        n = self.basis.n_local
        cplx = self.dtype == np.complex128
        t = torch.as_tensor(_DevMem(int(ptr.value or 0), n * (2 if cplx else 1)), device=f"cuda:{self.basis.ctx.device}")
        return torch.view_as_complex(t.view(n, 2)) if cplx else t

    def zeros_like(self) -> "dense_vector_gpu":
        pc = _pool_column(self.basis.n_local, self.dtype, self.basis.ctx)
        v = dense_vector_gpu(_basis=pc, _col=pc.col)
        v.zero()
        return v

    # -- six deferred procedures
    def zero(self) -> None:
        _capi.check(self._lib.lk_vec_zero(self.basis._h, self.col))

    def rand(self, ifnorm: bool = False, seed: int | None = None) -> None:
        if seed is None:
            # every un-seeded call draws a fresh stream (the reference's random_number does): a per-context
            # counter keeps runs reproducible and identical on every rank of a sharded vector
            ctx = self.basis.ctx
            ctx._rand_calls = getattr(ctx, "_rand_calls", 0) + 1
            s = self._seed + 7919 * ctx._rand_calls
        else:
            s = seed
        _capi.check(self._lib.lk_vec_rand(self.basis._h, self.col, C.c_uint64(s),
                                          C.c_int64(getattr(self.basis.ctx, "row0", 0)), 1 if ifnorm else 0))

    def scal(self, alpha) -> None:
        _capi.check(self._lib.lk_vec_scal(self.basis._h, self.col, _sc(alpha, self.dtype)))

    def axpby(self, alpha, vec: "abstract_vector", beta) -> None:
        if not isinstance(vec, dense_vector_gpu):
            # type_error('vec','dense_vector','IN',...)   AbstractVectors.fypp:533-535
            raise TypeError("axpby: vec must be a dense_vector_gpu")
        _capi.check(self._lib.lk_vec_axpby(_sc(alpha, self.dtype), vec.basis._h, vec.col, _sc(beta, self.dtype),
                                           self.basis._h, self.col))

    def dot(self, vec: "abstract_vector"):
        if not isinstance(vec, dense_vector_gpu):
            raise TypeError("dot: vec must be a dense_vector_gpu")
        out = (C.c_double * 2)()
        _capi.check(self._lib.lk_vec_dot(self.basis._h, self.col, vec.basis._h, vec.col, out))
        return complex(out[0], out[1]) if self.dtype.kind == "c" else float(out[0])

    def get_size(self) -> int:
        return self.basis.n_local

    def norm(self) -> float:
        out = C.c_double()
        _capi.check(self._lib.lk_vec_norm(self.basis._h, self.col, C.byref(out)))
        return float(out.value)


# ------------------------------------------------------------------------------------------
# free functions of LightKrylov_AbstractVectors (AbstractVectors.fypp:53-59)
# ------------------------------------------------------------------------------------------
def _is_gpu_basis(X) -> bool:
    return isinstance(X, krylov_basis_gpu)


def _as_gpu_cols(Y):
    """Y: dense_vector_gpu | krylov_basis_gpu -> (basis, col0, p, is_vector)"""
    if isinstance(Y, dense_vector_gpu):
        return Y.basis, Y.col, 1, True
    if isinstance(Y, krylov_basis_gpu):
        return Y, 0, Y.ncols, False
    return None


def innerprod(X, Y):
    """y = X^H v  /  M = X^H Y.   AbstractVectors.fypp:659-695"""
    g = _as_gpu_cols(Y) if _is_gpu_basis(X) else None
    if g is not None:
        By, j0, p, isvec = g
        k = X.ncols
        M = np.zeros((k, p), dtype=X.dtype, order="F")
        _capi.check(X._lib.lk_innerprod(X._h, k, By._h, j0, p, M.ctypes.data_as(C.POINTER(C.c_double))))
        return M[:, 0].copy() if isvec else M
    if isinstance(Y, abstract_vector):
        return np.array([x.dot(Y) for x in X])
    return np.array([[xi.dot(yj) for yj in Y] for xi in X])


def linear_combination(X, v):
    """y = X v (new vector) / Y = X B (new basis).   AbstractVectors.fypp:571-643"""
    v = np.asarray(v)
    if _is_gpu_basis(X):
        k = X.ncols
        if v.shape[0] != k:
            raise ValueError("Krylov basis X and low-dimensional vector v have different sizes.")
        Bm = np.asfortranarray(v.reshape(k, -1, order="F"), dtype=X.dtype)
        q = Bm.shape[1]
        Y = krylov_basis_gpu(X.n_local, q, X.dtype, X.ctx)
        _capi.check(X._lib.lk_lincomb(X._h, k, Bm.ctypes.data_as(C.POINTER(C.c_double)), q, Y._h, 0))
        return Y[0] if v.ndim == 1 else Y
    if v.ndim == 1:
        if len(X) != v.shape[0]:
            raise ValueError("Krylov basis X and low-dimensional vector v have different sizes.")
        y = X[0].zeros_like()
        y.zero()
        for i in range(len(X)):
            y.axpby(v[i], X[i], 1.0)
        return y
    return [linear_combination(X, v[:, j]) for j in range(v.shape[1])]


def Gram(X):
    """G(i,j) = X(i)%dot(X(j)), upper triangle mirrored without conjugation.  :645-657"""
    if _is_gpu_basis(X):
        k = X.ncols
        G = np.zeros((k, k), dtype=X.dtype, order="F")
        _capi.check(X._lib.lk_gram(X._h, k, G.ctypes.data_as(C.POINTER(C.c_double))))
        return G
    k = len(X)
    G = np.zeros((k, k), dtype=type(X[0].dot(X[0])))
    for i in range(k):
        for j in range(i, k):
            G[i, j] = X[i].dot(X[j])
            G[j, i] = G[i, j]
    return G


def copy(out, frm) -> None:
    """copy(out, from) = out%axpby(1, from, 0) on an intent(out) target.  :717-723
    Accepts vectors or equal-length sequences (the reference's procedure is elemental)."""
    if isinstance(out, dense_vector_gpu) and isinstance(frm, dense_vector_gpu):
        _capi.check(out._lib.lk_vec_copy(out.basis._h, out.col, frm.basis._h, frm.col))
        return
    if isinstance(out, abstract_vector):
        out.zero()
        out.axpby(1.0, frm, 0.0)
        return
    if len(out) != len(frm):
        raise ValueError("copy: bases have different sizes")
    for o, f in zip(out, frm):
        copy(o, f)


def zero_basis(X) -> None:
    """:711-715"""
    for x in ([X] if isinstance(X, abstract_vector) else X):
        x.zero()


def axpby_basis(alpha, X: Sequence[abstract_vector], beta, Y: Sequence[abstract_vector]) -> None:
    """Y <- alpha X + beta Y, elemental.  :697-709"""
    if isinstance(Y, abstract_vector):
        Y.axpby(alpha, X, beta)
        return
    for x, y in zip(X, Y):
        y.axpby(alpha, x, beta)


def rand_basis(X, ifnorm: bool = False) -> None:
    """:725-730"""
    for x in ([X] if isinstance(X, abstract_vector) else X):
        x.rand(ifnorm)


def verify_vector_axioms(x: abstract_vector, ntrials: int = 100, tolerance: float = 10.0 ** (-14), seed: int = 0) -> bool:
    """Conformance harness for a user vector type: the eight vector-space axiom blocks of
    AbstractVectors.fypp:733-927, `ntrials` random trials each, absolute tolerance 10^-(precision-1) on the norm
    of the defect (the reference's default).  Returns True when every trial of every block passes."""
    rng = np.random.default_rng(seed)
    cplx = np.dtype(getattr(x, "dtype", np.float64)).kind == "c"

    def new():
        v = x.zeros_like()
        v.rand()
        return v

    def clone(u):
        v = u.zeros_like()
        copy(v, u)
        return v

    def scalar():
        return complex(rng.random(), rng.random()) if cplx else float(rng.random())

    for _ in range(ntrials):
        # addition: associativity ("addition_distributivity", :756-779)
        u, v, w = new(), new(), new()
        wrk1, wrk2 = clone(v), clone(v)
        wrk1.add(w); wrk2.add(u)
        u.add(wrk1); w.add(wrk2)
        u.sub(w)
        if not u.norm() <= tolerance:
            return False
        # addition: commutativity (:781-797)
        u, v = new(), new()
        w = clone(v)
        v.add(u); u.add(w)
        u.sub(v)
        if not u.norm() <= tolerance:
            return False
        # additive identity (:799-813)
        u = new()
        v = clone(u)
        z = u.zeros_like(); z.zero()
        u.add(z); u.sub(v)
        if not u.norm() <= tolerance:
            return False
        # additive inverse (:815-826)
        u = new()
        v = clone(u)
        u.sub(v)
        if not u.norm() <= tolerance:
            return False
        # scaling identity (:828-842)
        u = new()
        v = clone(u)
        v.scal(1.0); u.sub(v)
        if not u.norm() <= tolerance:
            return False
        # scaling compatibility a(bu) = (ab)u (:844-868)
        a, b = scalar(), scalar()
        u = x.zeros_like(); u.rand(True)
        v = clone(u)
        v.scal(b); v.scal(a); u.scal(a * b)
        u.sub(v)
        if not u.norm() <= tolerance:
            return False
        # distributivity a(u+v) = au + av (:870-893)
        a = scalar()
        u, v = new(), new()
        w = clone(u)
        w.add(v); w.scal(a)
        u.scal(a); v.scal(a); v.add(u)
        v.sub(w)
        if not v.norm() <= tolerance:
            return False
        # distributivity (a+b)u = au + bu through axpby (:895-921)
        a, b = scalar(), scalar()
        u = new()
        v = clone(u)
        v.axpby(a, u, b)
        u.scal(a + b)
        v.sub(u)
        if not v.norm() <= tolerance:
            return False
    return True
