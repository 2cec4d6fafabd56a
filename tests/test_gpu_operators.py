"""Operators beside the graded path (AbstractLinops.fypp:58-87): a user's sparse matrix as a CSR abstract_linop (matvec / rmatvec
against scipy, the stencil operator reproduced, malformed input refused), a user's own operator written with torch on device pointers,
the row-sharded stencil operators' partition check, and the operator's own time inside the asynchronous batch (`matvec` profile tag)."""
import ctypes as C
import os

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, seeded, _lap5_csr
from tests._tol import assert_close, assert_columns_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_csr_linop_matvec_and_rmatvec_against_scipy(ctx, dtype):
    """y = A x and y = A^H x for random sparse matrices with empty rows, short rows and a few very long ones
    (every lanes-per-row setting from 2 to 64), against scipy's CSR product."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    n = 6_007
    for density, longrows in ((0.0004, 0), (0.002, 3), (0.01, 0), (0.03, 5)):
        A = sp.random(n, n, density=density, random_state=rng, format="lil", dtype=np.float64)
        for r in rng.integers(0, n, longrows):
            A[r, rng.integers(0, n, 900)] = 1.0
        if longrows:
            A[11, rng.integers(0, n, 4000)] = 1.0                      # longer than one CSR-stream block holds: a block of its own
        A[7, :] = 0.0                                                   # an empty row
        A = A.tocsr()
        vals = rng.standard_normal(A.nnz)
        if np.dtype(dtype).kind == "c":
            vals = vals + 1j * rng.standard_normal(A.nnz)
        A = sp.csr_matrix((vals.astype(dtype), A.indices, A.indptr), shape=(n, n))
        A.sort_indices()
        op = lk.csr_linop_gpu(A, ctx)
        ctx.set_tuning("csr_stream", int(density < 0.02))               # both kernels: through LDS (short rows) / lanes per row
        xh = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if np.dtype(dtype).kind == "c" else 0)).astype(dtype)
        x = lk.dense_vector_gpu.from_array(xh, ctx)
        y = lk.dense_vector_gpu(n, dtype, ctx)
        scale = abs(A).dot(np.abs(xh)).max() + 1e-300
        op.apply_matvec(x, y)
        assert np.abs(y.to_array() - A @ xh).max() <= 1e-13 * scale
        op.apply_rmatvec(x, y)
        scale_h = abs(A).T.dot(np.abs(xh)).max() + 1e-300
        assert np.abs(y.to_array() - A.conj().T @ xh).max() <= 1e-13 * scale_h
        assert (op.matvec_counter, op.rmatvec_counter) == (1, 1)
    ctx.set_tuning("csr_stream", 1)


def test_csr_laplacian_reproduces_the_stencil_operator_in_gmres_and_arnoldi(ctx):
    """BASELINE's "5-point Laplacian SpMV linop" literally as a sparse matrix: same products as the matrix-free
    lk_linop_lap5 and the same GMRES(30) residual history and Arnoldi factorisation (whole step loop in the engine)."""
    N = 96
    n = N * N
    A = _lap5_csr(N)
    Ac, As = lk.csr_linop_gpu(A, ctx), lk.laplacian2d_linop_gpu(N, ctx)
    bh = np.empty(n); ora.fill_counter(bh, 11)
    x = lk.dense_vector_gpu.from_array(bh, ctx)
    y1, y2 = lk.dense_vector_gpu(n, np.float64, ctx), lk.dense_vector_gpu(n, np.float64, ctx)
    Ac.apply_matvec(x, y1); As.apply_matvec(x, y2)
    assert np.abs(y1.to_array() - y2.to_array()).max() <= 1e-13 * np.abs(y2.to_array()).max()
    out = []
    for op in (Ac, As):
        xs = lk.dense_vector_gpu(n, np.float64, ctx)
        meta = lk.gmres_dp_metadata()
        info = lk.gmres(op, lk.dense_vector_gpu.from_array(bh, ctx), xs, rtol=1e-8, options=lk.gmres_dp_opts(kdim=30, maxiter=2),
                        meta=meta)
        out.append((info, np.array(meta.res), xs.to_array()))
    assert out[0][0] == out[1][0] and len(out[0][1]) == len(out[1][1])
    assert_close(out[0][1], out[1][1], "gmres on the Laplacian as CSR vs stencil: residual history", scale=out[1][1][0])
    assert_close(out[0][2], out[1][2], "gmres on the Laplacian as CSR vs stencil: solution")
    m = 20
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    X[0].rand(True, seed=5)
    H = np.zeros((m + 1, m), order="F")
    assert lk.arnoldi(Ac, X, H) == 0
    Xo = np.zeros((n, m + 1), order="F"); Xo[:, 0] = X.download(0, 1)[:, 0]
    Ho = np.zeros((m + 1, m), order="F")
    assert ora.arnoldi(ora.PyOp(lambda v: A @ v, np.float64), Xo, Ho) == 0
    for j in range(m):
        assert np.abs(H[:, j] - Ho[:, j]).max() <= 1e-12 * np.abs(Ho[:, j]).max()


def test_csr_linop_rejects_malformed_input(ctx):
    rowptr = np.array([0, 2, 3], dtype=np.int64)
    vals = np.array([1.0, 2.0, 3.0])
    with pytest.raises(_capi.LightKrylovHipError, match="out of range"):
        lk.csr_linop_gpu((rowptr, np.array([0, 5, 1], dtype=np.int32), vals), ctx)
    with pytest.raises(_capi.LightKrylovHipError, match="0-based"):
        lk.csr_linop_gpu((rowptr + 1, np.array([0, 1, 1], dtype=np.int32), vals), ctx)
    with pytest.raises(_capi.LightKrylovHipError, match="decreases"):
        lk.csr_linop_gpu((np.array([0, 3, 2], dtype=np.int64), np.array([0, 1, 1], dtype=np.int32), vals), ctx)
    with pytest.raises(TypeError):
        lk.csr_linop_gpu((rowptr, np.array([0, 1, 1], dtype=np.int32), vals.astype(np.float32)), ctx)


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_user_operator_written_with_torch_on_device_pointers(dtype):
    """A user's own abstract_linop whose matvec runs on the vectors' device memory (lk_vec_device_ptr through
    dense_vector_gpu.as_torch): per-object Arnoldi in lazy mode -- where the engine defers updates, so the accessor must
    first apply what it still owes the vector -- equals the engine's diagonal operator and the oracle."""
    import torch
    n, m = 60_013, 14
    c = lk.Context(device=0)
    c.set_tuning("lazy", 1)
    g = np.arange(n) / n
    d = (1.0 + g) * (np.exp(0.4j * g) if np.dtype(dtype).kind == "c" else 1.0)
    d = d.astype(dtype)
    dt = torch.as_tensor(d, device="cuda:0")

    class torch_diag(lk.abstract_linop):
        def matvec(self, vi, vo):
            torch.mul(dt, vi.as_torch("r"), out=vo.as_torch("w"))

    x0 = seeded(n, dtype, 7); x0 /= np.linalg.norm(x0)
    B = lk.krylov_basis_gpu(n, m + 1, dtype, c); B.upload(x0.reshape(-1, 1), 0)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.arnoldi(torch_diag(), [B[j] for j in range(m + 1)], H) == 0
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F"); Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.arnoldi(ora.DiagOp(d), Xo, Ho) == 0
    for j in range(m):
        assert np.abs(H[:, j] - Ho[:, j]).max() <= 1e-12 * np.abs(Ho[:, j]).max()
    assert c.lazy_fusion_stats()[0] == 2 * m                      # the fast path survived the foreign kernels
    # a pending update is applied before the pointer is handed out: y%sub(proj) then a torch read of y
    y, T = B[m], lk.dense_vector_gpu(n, dtype, c)
    before = y.to_array()
    T.zero(); T.axpby(0.5, B[0], 1.0); T.axpby(-2.0, B[1], 1.0)
    y.sub(T)
    with pytest.raises(RuntimeError, match="torch_stream"):          # outside an operator the stream must be named
        y.as_torch("r")
    with c.torch_stream():
        got = y.as_torch("r").cpu().numpy()
    X = B.download()
    assert np.abs(got - (before - 0.5 * X[:, 0] + 2.0 * X[:, 1])).max() <= 1e-14
    # and a write through the pointer invalidates what the engine remembered about the vector
    nrm = y.norm()
    with c.torch_stream():
        y.as_torch("rw").mul_(3.0)
    assert abs(y.norm() - 3.0 * nrm) <= 1e-13 * nrm
    del B, T, y
    c.close()


def test_sharded_stencil_operators_reject_partitions_out_of_rank_order():
    """The halo exchange addresses neighbours by rank: rank r must own the r-th block."""
    lib = _capi.load()
    cb = _capi.ALLREDUCE_FN(lambda _u, _p, _n, _s: 0)
    c = lk.Context(device=0)
    _capi.check(lib.lk_set_allreduce(c._h, cb, None, 2, 0))             # this context is rank 0 of 2
    op = C.c_void_p()
    assert lib.lk_linop_lap5_create_sharded(c._h, 64, 32, 32, C.byref(op)) != 0        # the UPPER half on rank 0
    assert b"rank order" in lib.lk_last_error()
    _capi.check(lib.lk_linop_lap5_create_sharded(c._h, 64, 0, 32, C.byref(op)))
    _capi.check(lib.lk_linop_destroy(op))
    nu = (C.c_double * 2)(2.0, 0.2); ga = (C.c_double * 2)(1.0, -1.0)
    assert lib.lk_linop_gl_create_sharded(c._h, 1000, 500, 500, 0.4, 0.01, 1, nu, ga, 0.34, -0.01, C.byref(op)) != 0
    assert b"rank order" in lib.lk_last_error()
    _capi.check(lib.lk_set_allreduce(c._h, _capi.ALLREDUCE_FN(), None, 1, 0))
    c.close()


def test_operator_time_is_measured_inside_the_asynchronous_batch(ctx):
    """bench.py's `matvec` figure: the operator launches of an asynchronous Arnoldi batch carry profiling events too."""
    n, m = 1_000_003, 12
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    X[0].rand(True, seed=7)
    H = np.zeros((m + 1, m), order="F")
    A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=ctx)
    ctx.profile_reset(); ctx.profile_enable(True)
    assert lk.arnoldi(A, X, H) == 0
    ctx.sync()
    cnt, ms, by = ctx.profile_get("matvec")
    ctx.profile_enable(False)
    assert cnt == m and ms > 0.0 and by == pytest.approx(m * 2 * 8.0 * n)
