"""Shared helpers of the GPU test files (test infrastructure: inputs come from the oracle's counter RNG, so CPU and GPU see identical data)."""
import ctypes as C

import numpy as np

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora

KINDS = [np.float64, np.complex128]


def seeded(n, dtype, seed):
    x = np.empty(n, dtype=dtype)
    ora.fill_counter(x, seed)
    return x


def basis(n, k, dtype, seed):
    X = np.empty((n, k), dtype=dtype, order="F")
    for j in range(k):
        ora.fill_counter(X[:, j], seed + j)
    return X


def orthonormal_basis(n, k, dtype, seed):
    Q, _ = np.linalg.qr(basis(n, k, dtype, seed))
    return np.asfortranarray(Q)


def _spd(n, seed, lead=(8.0, 6.0, 4.0)):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n)) / np.sqrt(n)
    A = M.T @ M + np.eye(n)
    A[:len(lead), :len(lead)] += np.diag(lead)
    return np.asfortranarray(A)


def _lap5_csr(N):
    import scipy.sparse as sp
    T = sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1])
    S = sp.diags([-np.ones(N - 1), -np.ones(N - 1)], [-1, 1])
    return ((sp.kron(sp.identity(N), T) + sp.kron(S, sp.identity(N))) * float((N + 1) ** 2)).tocsr()


def _pool_fns(ctx):
    lib = _capi.load()

    def acquire(dtype, n, tag):
        slab, col = C.c_void_p(), C.c_int()
        _capi.check(lib.lk_pool_acquire(ctx._h, dtype, n, C.c_uint64(tag), C.byref(slab), C.byref(col)))
        return slab.value, col.value

    def info(slab, col):
        t, g = C.c_uint64(), C.c_uint64()
        _capi.check(lib.lk_pool_column_info(ctx._h, C.c_void_p(slab), col, C.byref(t), C.byref(g)))
        return t.value, g.value

    return lib, acquire, info


def _arnoldi_h(ctx, n=400_003, m=12):
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=ctx)
    H = np.zeros((m + 1, m), order="F")
    X[0].rand(True, seed=7)
    assert lk.arnoldi(A, X, H) == 0
    return H, X.download()
