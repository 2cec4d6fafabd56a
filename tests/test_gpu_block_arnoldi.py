"""lk_arnoldi_block / lk_qr: the block Arnoldi factorisation (arnoldi.fypp:20-73 with blksize = p) and qr_no_pivoting (qr.fypp:116-167)
as engine calls -- every step enqueued asynchronously, one host synchronisation per call -- against the oracle (1e-12, normwise per
column), against the host-synchronous schedule (bit for bit), and on the reference's own colinear case (test/TestKrylov.fypp:244-296:
p = 2, kdim = 64 in dimension 128 -- the last block has nothing left to span)."""
import ctypes as C

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, basis, seeded

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _dense(n, dtype, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    if np.dtype(dtype).kind == "c":
        A = A + 1j * rng.standard_normal((n, n)) / np.sqrt(n)
    return np.asfortranarray(A.astype(dtype))


@pytest.fixture(params=[1, 0], ids=["single_launch", "three_sweeps"])
def bctx(request, ctx):
    ctx.set_tuning("resident", request.param)
    yield ctx
    ctx.set_tuning("resident", 1)
    ctx.set_tuning("async_arnoldi", 1)


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("p", [2, 4, 5, 8])
def test_block_arnoldi_engine_call_against_oracle_and_the_synchronous_schedule(bctx, dtype, p):
    ctx = bctx
    n, kdim = 3001, 7
    A = _dense(n, dtype, 11)
    Q0 = np.asfortranarray(np.linalg.qr(basis(n, p, dtype, 70))[0])
    ncol = (kdim + 1) * p
    Xo = np.zeros((n, ncol), dtype=dtype, order="F")
    Xo[:, :p] = Q0
    Ho = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
    assert ora.arnoldi_block(ora.DenseOp(A), Xo, Ho, p) == 0
    Hs = {}
    for asynchronous in (1, 0):
        ctx.set_tuning("async_arnoldi", asynchronous)
        X = lk.krylov_basis_gpu(n, ncol, dtype, ctx)
        X.upload(Q0, 0)
        H = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
        Aop = lk.dense_linop_gpu(A, ctx)
        assert lk.arnoldi(Aop, X, H, blksize=p) == 0
        assert Aop.matvec_counter == kdim * p
        for j in range(kdim * p):
            assert np.abs(H[:, j] - Ho[:, j]).max() <= RTOL * np.abs(Ho[:, j]).max(), (asynchronous, j)
        Xg = X.download()
        assert np.abs(A @ Xg[:, :kdim * p] - Xg @ H).max() <= 1e-12
        assert np.abs(Xg.conj().T @ Xg - np.eye(ncol)).max() <= 1e-12
        Hs[asynchronous] = H
    assert np.array_equal(Hs[0], Hs[1])                   # same kernels, same order: the batch changes WHEN the host waits, nothing else
    # continued ranges equal the one-shot factorisation
    ctx.set_tuning("async_arnoldi", 1)
    X = lk.krylov_basis_gpu(n, ncol, dtype, ctx)
    X.upload(Q0, 0)
    H = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
    Aop = lk.dense_linop_gpu(A, ctx)
    assert lk.arnoldi(Aop, X, H, blksize=p, kstart=1, kend=3) == 0
    assert lk.arnoldi(Aop, X, H, blksize=p, kstart=4, kend=4) == 0
    assert lk.arnoldi(Aop, X, H, blksize=p, kstart=5, kend=kdim) == 0
    assert np.array_equal(H, Hs[1])


@pytest.mark.parametrize("dtype", KINDS)
def test_block_arnoldi_wide_enough_for_the_matrix_core_route_and_several_panels(bctx, dtype):
    """p = 6 (>= 5 right-hand sides: X^H Y and the updates on the FP64 MFMAs) up to 150 basis columns (two column panels of X)"""
    ctx = bctx
    n, p, kdim = 2003, 6, 24
    A = _dense(n, dtype, 5)
    Q0 = np.asfortranarray(np.linalg.qr(basis(n, p, dtype, 31))[0])
    ncol = (kdim + 1) * p
    X = lk.krylov_basis_gpu(n, ncol, dtype, ctx)
    X.upload(Q0, 0)
    H = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
    assert lk.arnoldi(lk.dense_linop_gpu(A, ctx), X, H, blksize=p) == 0
    Xo = np.zeros((n, ncol), dtype=dtype, order="F")
    Xo[:, :p] = Q0
    Ho = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
    assert ora.arnoldi_block(ora.DenseOp(A), Xo, Ho, p) == 0
    for j in range(kdim * p):
        assert np.abs(H[:, j] - Ho[:, j]).max() <= RTOL * np.abs(Ho[:, j]).max(), j
    Xg = X.download()
    assert np.abs(Xg.conj().T @ Xg - np.eye(ncol)).max() <= 1e-12


@pytest.mark.parametrize("dtype", KINDS)
def test_the_reference_test_case_p2_kdim64_in_dimension_128(bctx, dtype):
    """test/TestKrylov.fypp:244-296: 130 vectors in a 128-dimensional space.  The factorisation holds on the first p kdim columns and
    they are orthonormal (the reference's two checks, at its rtol); the last block is colinear with what is there: info = p kdim, its
    diagonal carries a 0, the columns were re-drawn (finite, unit norm)."""
    ctx = bctx
    n, p, kdim = 128, 2, 64
    A = _dense(n, dtype, 3)
    Q0 = np.asfortranarray(np.linalg.qr(basis(n, p, dtype, 9))[0])
    ncol = (kdim + 1) * p
    for asynchronous in (1, 0):
        ctx.set_tuning("async_arnoldi", asynchronous)
        X = lk.krylov_basis_gpu(n, ncol, dtype, ctx)
        X.upload(Q0, 0)
        H = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
        info = lk.arnoldi(lk.dense_linop_gpu(A, ctx), X, H, blksize=p, tol=lk.atol_dp)
        Xg = X.download()
        rtol = np.sqrt(lk.atol_dp)
        assert np.abs(A @ Xg[:, :kdim * p] - Xg @ H).max() < rtol
        assert np.abs(Xg[:, :kdim * p].conj().T @ Xg[:, :kdim * p] - np.eye(kdim * p)).max() < rtol
        assert np.isfinite(Xg).all()
        assert info in (0, kdim * p)                      # (whether the last diagonal falls below 1e-15 or just near it is rounding)
        if info:
            assert min(abs(H[kdim * p + i, (kdim - 1) * p + i]) for i in range(p)) < lk.atol_dp
            assert np.abs(np.linalg.norm(Xg[:, kdim * p:], axis=0) - 1.0).max() <= 1e-12


@pytest.mark.parametrize("dtype", KINDS)
def test_breakdown_inside_a_block_stops_the_batch_and_the_host_finishes_the_block(bctx, dtype):
    """An operator of rank 3 applied to a block of p = 2: step 2 produces only ONE new direction -- its second column is colinear
    (qr.fypp:146-162: R(2,2) = 0, re-draw), arnoldi exits with info = kp = 4 (arnoldi.fypp:65-71); nothing beyond that block is touched.
    The asynchronous batch must agree with the host-synchronous schedule bit for bit (same re-draw stream)."""
    ctx = bctx
    n, p, kdim = 4001, 2, 6
    U = np.asfortranarray(np.linalg.qr(basis(n, 3, dtype, 50))[0])
    V = np.asfortranarray(np.linalg.qr(basis(n, 3, dtype, 60))[0])
    A = np.asfortranarray((U * np.array([3.0, 2.0, 1.0])) @ V.conj().T)
    Q0 = np.asfortranarray(np.linalg.qr(basis(n, p, dtype, 70))[0])
    ncol = (kdim + 1) * p
    out = {}
    for asynchronous in (1, 0):
        ctx.set_tuning("async_arnoldi", asynchronous)
        X = lk.krylov_basis_gpu(n, ncol, dtype, ctx)
        X.upload(Q0, 0)
        H = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
        info = lk.arnoldi(lk.dense_linop_gpu(A, ctx), X, H, blksize=p, tol=1e-10)
        out[asynchronous] = (info, H, X.download())
    info, H, Xg = out[1]
    # the oracle with the engine's re-draw streams: the WHOLE result -- the re-drawn column, the R entries behind it -- entry by entry
    Xo = np.zeros((n, ncol), dtype=dtype, order="F")
    Xo[:, :p] = Q0
    Ho = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
    info_o = ora.arnoldi_block(ora.DenseOp(A), Xo, Ho, p, tol=1e-10, engine_streams=True)
    assert info_o == info
    for j in range(2 * p):
        # (the colinear column's own norm before the re-draw is rounding noise on both sides: its R(j, j) is set to 0 by both)
        assert np.abs(H[:, j] - Ho[:, j]).max() <= RTOL * np.abs(Ho[:, j]).max(), j
    assert np.abs(Xg[:, :3 * p] - Xo[:, :3 * p]).max() <= 1e-10
    # X(:, :2) = Q0; step 1 adds two directions of range(A); step 2 can add only the third: info = 2 p
    assert info == out[0][0] == 2 * p
    assert np.array_equal(H, out[0][1]) and np.array_equal(Xg, out[0][2])
    assert abs(H[2 * p + 1, p + 1]) < 1e-10
    assert not H[:, 2 * p:].any() and not Xg[:, 3 * p:].any()
    m1 = 3 * p - 1
    assert np.abs(Xg[:, :m1].conj().T @ Xg[:, :m1] - np.eye(m1)).max() <= 1e-10
    # the re-drawn column is orthogonalised against ITS BLOCK only -- qr_no_pivoting knows nothing of the basis before it (qr.fypp:157-160):
    # unit norm, orthogonal to the column before it, and that is all the reference guarantees
    assert abs(np.linalg.norm(Xg[:, m1]) - 1.0) <= 1e-12 and abs(np.vdot(Xg[:, m1 - 1], Xg[:, m1])) <= 1e-12
    assert np.abs(A @ Xg[:, :2 * p] - Xg[:, :3 * p] @ H[:3 * p, :2 * p]).max() <= 1e-10


@pytest.mark.parametrize("dtype", KINDS)
def test_qr_engine_call_against_oracle_including_colinear_columns(bctx, dtype):
    ctx = bctx
    n, p = 5003, 7
    Y = basis(n, p, dtype, 200)
    B = lk.krylov_basis_gpu(n, p + 3, dtype, ctx)
    B.upload(Y, 2)                                         # columns [2, 2 + p) of a wider panel
    R = np.zeros((p, p), dtype=dtype, order="F")
    info = C.c_int()
    lib = _capi.load()
    _capi.check(lib.lk_qr(B._h, 2, p, R.ctypes.data_as(C.POINTER(C.c_double)), p, lk.atol_dp, C.byref(info)))
    Yo = Y.copy(order="F")
    Ro = np.zeros((p, p), dtype=dtype, order="F")
    assert ora.qr_no_pivoting(Yo, Ro) == info.value == 0
    for j in range(p):
        assert np.abs(R[:, j] - Ro[:, j]).max() <= RTOL * np.abs(Ro[:, j]).max()
    Qg = B.download(2, p)
    assert np.abs(Qg - Yo).max() <= 1e-12
    assert np.abs(Y - Qg @ R).max() <= 1e-12 * np.abs(Y).max() * p
    # two colinear columns: info = the first of them, R(j, j) = 0 there, the panel comes out orthonormal all the same
    Y2 = Y.copy(order="F")
    Y2[:, 3] = 2.0 * Y2[:, 1]
    Y2[:, 5] = Y2[:, 0] - Y2[:, 2]
    B.upload(Y2, 2)
    R[...] = 0
    _capi.check(lib.lk_qr(B._h, 2, p, R.ctypes.data_as(C.POINTER(C.c_double)), p, 1e-10, C.byref(info)))
    assert info.value == 4 and R[3, 3] == 0 and R[5, 5] == 0
    # ... and entry by entry against the oracle drawing from the engine's streams (0x5EED + panel column + 1; the block starts at column 2)
    Yo2 = Y2.copy(order="F")
    Ro2 = np.zeros((p, p), dtype=dtype, order="F")
    assert ora.qr_no_pivoting(Yo2, Ro2, tol=1e-10, column_seed=lambda j: 0x5EED + 2 + j + 1) == 4
    for j in range(p):
        assert np.abs(R[:, j] - Ro2[:, j]).max() <= RTOL * max(np.abs(Ro2[:, j]).max(), np.linalg.norm(Y2[:, j])), j
    assert np.abs(B.download(2, p) - Yo2).max() <= 1e-10
    Qg = B.download(2, p)
    assert np.abs(Qg.conj().T @ Qg - np.eye(p)).max() <= 1e-12
    keep = [j for j in range(p) if j not in (3, 5)]
    assert np.abs(Y2[:, keep] - Qg @ R[:, keep]).max() <= 1e-12 * np.abs(Y2).max() * p
    # the mirror's qr on a view of the panel takes the same entry
    B.upload(Y, 2)
    R2 = np.zeros((p, p), dtype=dtype, order="F")
    assert lk.qr(B[2:2 + p], R2) == 0
    for j in range(p):
        assert np.abs(R2[:, j] - Ro[:, j]).max() <= RTOL * np.abs(Ro[:, j]).max()
