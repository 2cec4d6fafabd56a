"""The row-sharded ENGINE path on ONE GPU: two contexts (= two ranks, each owning half the rows) driven by
two threads of one process; the all-reduce hook the engine calls after every sweep is emulated with a
thread barrier + a device-side sum of the two ranks' reduction buffers.  (RCCL itself refuses two ranks on
one device; the real multi-GPU launch is the driver's.)  What this pins: every place the engine must
all-reduce (h1, h2, the three norms, rand's normalisation), the partition-independent counter RNG
(row0 offsets), and that a sharded Arnoldi / GMRES reproduces the single-context result."""
import ctypes as C
import threading

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from lightkrylov_amd.context import _DevMem
from tests._tol import assert_close

pytestmark = pytest.mark.gpu


class _EmulatedGroup:
    """Sum all-reduce between `nranks` threads; buffers live on the same device."""

    def __init__(self, nranks):
        import torch
        self.torch = torch
        self.n = nranks
        self.barrier = threading.Barrier(nranks)
        self.slots = [None] * nranks
        self.calls = 0

    def hook(self, rank, ctx):
        torch = self.torch

        def _cb(_user, dev_ptr, count, stream_ptr):
            try:
                ctx.sync_stream_only()
                self.slots[rank] = torch.as_tensor(_DevMem(int(dev_ptr), int(count)), device="cuda:0")
                self.barrier.wait(timeout=120)
                if rank == 0:
                    total = self.slots[0].clone()
                    for r in range(1, self.n):
                        total += self.slots[r]
                    for r in range(self.n):
                        self.slots[r].copy_(total)
                    torch.cuda.synchronize()
                    self.calls += 1
                self.barrier.wait(timeout=120)
                return 0
            except Exception as exc:  # noqa: BLE001
                print("emulated all-reduce failed:", repr(exc))
                self.barrier.abort()
                return 1
        return _capi.ALLREDUCE_FN(_cb)


def _sharded(n, nranks, body):
    """Run body(rank, ctx, row0, n_local) on `nranks` threads with an emulated all-reduce; returns results."""
    lib = _capi.load()
    grp = _EmulatedGroup(nranks)
    out, errs = [None] * nranks, []

    def worker(rank):
        try:
            ctx = lk.Context(device=0, use_torch_stream=False)           # own stream per rank
            ctx.sync_stream_only = lambda: _capi.check(lib.lk_sync(ctx._h))
            cb = grp.hook(rank, ctx)
            _capi.check(lib.lk_set_allreduce(ctx._h, cb, None, nranks, rank))
            ctx._cb, ctx.nranks, ctx.rank = cb, nranks, rank
            row0, nl = lk.row_partition(n, nranks, rank)
            ctx.set_partition(row0, n)
            out[rank] = body(rank, ctx, row0, nl)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            grp.barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    return out, grp


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_arnoldi_matches_single_context(ctx, dtype, nranks):
    n, m = 300_003, 40

    def dvals(row0, nl):
        g = (row0 + np.arange(nl)) / n
        d = 1.0 + g
        return (d * np.exp(0.3j * g)).astype(dtype) if np.dtype(dtype).kind == "c" else d

    def body(rank, c, row0, nl):
        X = lk.krylov_basis_gpu(nl, m + 1, dtype, c)
        X[0].rand(True, seed=7)                                   # global vector, partition independent
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        info = lk.arnoldi(lk.diag_linop_gpu(dvals(row0, nl), c), X, H)
        return info, H, X.download()

    res, grp = _sharded(n, nranks, body)
    X1 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
    X1[0].rand(True, seed=7)
    H1 = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.arnoldi(lk.diag_linop_gpu(dvals(0, n), ctx), X1, H1) == 0
    assert grp.calls >= 3 * m                                      # >= 3 all-reduces per Arnoldi step
    for info, H, _ in res:
        assert info == 0
        assert np.array_equal(H, res[0][1])                        # every rank holds the same H
        for j in range(m):
            assert np.abs(H[:, j] - H1[:, j]).max() <= 1e-13 * np.abs(H1[:, j]).max()
    Xs = np.concatenate([r[2] for r in res], axis=0)               # stitch the row blocks back together
    assert np.abs(Xs - X1.download()).max() <= 1e-12
    assert np.abs(Xs.conj().T @ Xs - np.eye(m + 1)).max() <= 1e-12


@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_lanczos_and_bidiagonalization_match_single_context(ctx, nranks):
    """lk_lanczos and lk_bidiag row-sharded: every reduction of their asynchronous batches (the local one-column passes, the
    first step's norm, the three sweeps, both halves of a Golub-Kahan step) must be all-reduced; T, B and the stitched bases
    against a single context."""
    n, m = 200_003, 24
    dtype = np.float64

    def dvals(row0, nl):
        return 1.0 + (row0 + np.arange(nl)) / n

    def body(rank, c, row0, nl):
        A = lk.diag_linop_gpu(dvals(row0, nl), c)
        X = lk.krylov_basis_gpu(nl, m + 1, dtype, c)
        X[0].rand(True, seed=7)
        T = np.zeros((m + 1, m), dtype=dtype, order="F")
        i1 = lk.lanczos(A, X, T)
        U = lk.krylov_basis_gpu(nl, m + 1, dtype, c); V = lk.krylov_basis_gpu(nl, m + 1, dtype, c)
        U[0].rand(True, seed=9)
        B = np.zeros((m + 1, m), dtype=dtype, order="F")
        i2 = lk.bidiagonalization(A, U, V, B)
        return i1, T, X.download(), i2, B, U.download(), V.download()

    res, _grp = _sharded(n, nranks, body)
    A1 = lk.diag_linop_gpu(dvals(0, n), ctx)
    X1 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X1[0].rand(True, seed=7)
    T1 = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.lanczos(A1, X1, T1) == 0
    U1 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); V1 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); U1[0].rand(True, seed=9)
    B1 = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.bidiagonalization(A1, U1, V1, B1) == 0
    for i1, T, _x, i2, B, _u, _v in res:
        assert i1 == 0 and i2 == 0
        assert np.array_equal(T, res[0][1]) and np.array_equal(B, res[0][4])       # every rank holds the same T and B
        assert np.abs(T - T1).max() <= 1e-13 * np.abs(T1).max() and np.abs(B - B1).max() <= 1e-13 * np.abs(B1).max()
    for idx, ref in ((2, X1), (5, U1), (6, V1)):
        stitched = np.concatenate([r[idx] for r in res], axis=0)
        assert np.abs(stitched - ref.download()).max() <= 1e-11


def test_sharded_blas1_and_gmres(ctx):
    n = 200_001

    def body(rank, c, row0, nl):
        d = 2.0 + (row0 + np.arange(nl)) / n
        b = lk.dense_vector_gpu(nl, np.float64, c); b.rand(False, seed=11)
        w = lk.dense_vector_gpu(nl, np.float64, c); w.rand(False, seed=12)
        dots = (b.dot(w), b.norm())
        x = b.zeros_like()
        meta = lk.gmres_dp_metadata()
        info = lk.gmres(lk.diag_linop_gpu(d, c), b, x, rtol=1e-10, options=lk.gmres_dp_opts(kdim=15, maxiter=3), meta=meta)
        return dots, info, np.array(meta.res), x.to_array()

    res, _ = _sharded(n, 2, body)
    d = 2.0 + np.arange(n) / n
    b = lk.dense_vector_gpu(n, np.float64, ctx); b.rand(False, seed=11)
    w = lk.dense_vector_gpu(n, np.float64, ctx); w.rand(False, seed=12)
    x = b.zeros_like()
    meta = lk.gmres_dp_metadata()
    info = lk.gmres(lk.diag_linop_gpu(d, ctx), b, x, rtol=1e-10, options=lk.gmres_dp_opts(kdim=15, maxiter=3), meta=meta)
    for dots, inf, hist, _ in res:
        assert abs(dots[0] - b.dot(w)) <= 1e-12 * b.norm() * w.norm() and abs(dots[1] - b.norm()) <= 1e-12 * b.norm()
        assert inf == info and len(hist) == len(meta.res)
        assert np.abs(hist - np.array(meta.res)).max() <= 1e-12 * meta.res[0]
    xs = np.concatenate([r[3] for r in res])
    assert np.abs(xs - x.to_array()).max() <= 1e-12 * np.abs(x.to_array()).max()


# ----------------------------------------------------------------------------- stencil operators with a halo exchange
class _EmulatedHalo:
    """Nearest-neighbour exchange between `nranks` threads (lk_halo_fn): what ncclSend / ncclRecv do on a real node."""

    def __init__(self, nranks):
        import torch
        self.torch = torch
        self.n = nranks
        self.barrier = threading.Barrier(nranks)
        self.sends = [None] * nranks
        self.calls = 0

    def hook(self, rank, ctx):
        torch = self.torch

        def view(ptr, count):
            return torch.as_tensor(_DevMem(int(ptr), int(count)), device="cuda:0") if ptr else None

        def _cb(_user, send_lo, send_hi, recv_lo, recv_hi, count, _stream):
            try:
                ctx.sync_stream_only()
                self.sends[rank] = (view(send_lo, count), view(send_hi, count))
                self.barrier.wait(timeout=120)
                if recv_lo:
                    view(recv_lo, count).copy_(self.sends[rank - 1][1])      # rank-1's send_hi
                if recv_hi:
                    view(recv_hi, count).copy_(self.sends[rank + 1][0])      # rank+1's send_lo
                torch.cuda.synchronize()
                if rank == 0:
                    self.calls += 1
                self.barrier.wait(timeout=120)
                return 0
            except Exception as exc:  # noqa: BLE001
                print("emulated halo exchange failed:", repr(exc))
                self.barrier.abort()
                return 1
        return _capi.HALO_FN(_cb)


def _sharded_with_halo(nranks, body):
    lib = _capi.load()
    grp, halo = _EmulatedGroup(nranks), _EmulatedHalo(nranks)
    out, errs = [None] * nranks, []

    def worker(rank):
        try:
            ctx = lk.Context(device=0, use_torch_stream=False)
            ctx.sync_stream_only = lambda: _capi.check(lib.lk_sync(ctx._h))
            cb = grp.hook(rank, ctx)
            _capi.check(lib.lk_set_allreduce(ctx._h, cb, None, nranks, rank))
            ctx._cb, ctx.nranks, ctx.rank = cb, nranks, rank
            ctx.set_halo_exchange(halo.hook(rank, ctx))
            out[rank] = body(rank, ctx)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            grp.barrier.abort(); halo.barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    return out, halo


@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_laplacian_matvec_and_gmres(ctx, nranks):
    """BASELINE config 3's operator row-sharded by grid lines: matvec and a GMRES cycle reproduce the single-context run."""
    N = 151
    n = N * N
    b_full = np.empty(n); from oracle import oracle as ora; ora.fill_counter(b_full, 11)

    def body(rank, c):
        j0, nj = lk.grid_partition(N, nranks, rank)
        c.set_partition(j0 * N, n)
        A = lk.laplacian2d_linop_gpu(N, c, j0=j0, nj=nj)
        b = lk.dense_vector_gpu.from_array(b_full[j0 * N:(j0 + nj) * N], c)
        y = b.zeros_like()
        A.apply_matvec(b, y)
        x = b.zeros_like()
        meta = lk.gmres_dp_metadata()
        info = lk.gmres(A, b, x, rtol=1e-8, options=lk.gmres_dp_opts(kdim=20, maxiter=1), meta=meta)
        return y.to_array(), x.to_array(), info, np.array(meta.res)

    res, halo = _sharded_with_halo(nranks, body)
    A1 = lk.laplacian2d_linop_gpu(N, ctx)
    b1 = lk.dense_vector_gpu.from_array(b_full, ctx)
    y1 = b1.zeros_like(); A1.apply_matvec(b1, y1)
    ys = np.concatenate([r[0] for r in res])
    assert np.array_equal(ys, y1.to_array())                        # same arithmetic per point: bit-identical
    assert np.abs(ys - ora_lap5(N, b_full)).max() <= 1e-12 * np.abs(ys).max()
    x1 = b1.zeros_like(); m1 = lk.gmres_dp_metadata()
    info1 = lk.gmres(A1, b1, x1, rtol=1e-8, options=lk.gmres_dp_opts(kdim=20, maxiter=1), meta=m1)
    xs = np.concatenate([r[1] for r in res])
    assert all(r[2] == info1 for r in res)
    assert np.abs(res[0][3] - np.array(m1.res)).max() <= 1e-12 * m1.res[0]
    assert_close(xs, x1.to_array(), "sharded stencil gmres: solution vs single rank")
    assert halo.calls >= 20


def ora_lap5(N, u):
    from oracle import oracle as ora
    v = np.empty_like(u)
    ora.Lap5Op(N).matvec(u, v)
    return v


@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_ginzburg_landau_stepper_and_arnoldi(ctx, nranks):
    """BASELINE config 4's operator row-sharded: one point per RK4 stage crosses each rank boundary; the propagator, its
    adjoint and an Arnoldi factorisation reproduce the single-context run."""
    n, m = 30_001, 12
    x_full = np.empty(n, dtype=np.complex128); from oracle import oracle as ora; ora.fill_counter(x_full, 13)
    x_full /= np.linalg.norm(x_full)

    def body(rank, c):
        row0, nl = lk.row_partition(n, nranks, rank)
        c.set_partition(row0, n)
        A = lk.ginzburg_landau_linop_gpu(n, c, tau=0.05, nsub=2, row0=row0, n_local=nl)
        v = lk.dense_vector_gpu.from_array(x_full[row0:row0 + nl], c)
        w = v.zeros_like(); wt = v.zeros_like()
        A.apply_matvec(v, w); A.apply_rmatvec(v, wt)
        X = lk.krylov_basis_gpu(nl, m + 1, np.complex128, c)
        X.upload(x_full[row0:row0 + nl].reshape(-1, 1), 0)
        H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
        info = lk.arnoldi(A, X, H)
        return w.to_array(), wt.to_array(), info, H

    res, halo = _sharded_with_halo(nranks, body)
    A1 = lk.ginzburg_landau_linop_gpu(n, ctx, tau=0.05, nsub=2)
    v1 = lk.dense_vector_gpu.from_array(x_full, ctx)
    w1 = v1.zeros_like(); wt1 = v1.zeros_like()
    A1.apply_matvec(v1, w1); A1.apply_rmatvec(v1, wt1)
    assert np.array_equal(np.concatenate([r[0] for r in res]), w1.to_array())
    assert np.array_equal(np.concatenate([r[1] for r in res]), wt1.to_array())
    X1 = lk.krylov_basis_gpu(n, m + 1, np.complex128, ctx); X1.upload(x_full.reshape(-1, 1), 0)
    H1 = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    assert lk.arnoldi(A1, X1, H1) == 0
    for info, H in ((r[2], r[3]) for r in res):
        assert info == 0
        for j in range(m):
            assert np.abs(H[:, j] - H1[:, j]).max() <= 1e-12 * np.abs(H1[:, j]).max()
    assert halo.calls == 8 * (m + 2)                              # 4 stages x 2 sub-steps per application, m + 2 applications


def test_sharded_per_object_arnoldi_in_lazy_mode_matches_single_context(ctx):
    """The per-object (type-bound-procedure) schedule with the lazy path on two emulated ranks: the fused
    update + dot + norm sweep all-reduces inside y%norm() -- the same point of the call sequence on every rank --,
    the catch-all flush never does.  Same H on every rank, equal to the single-context fused factorisation."""
    n, m, nranks = 120_007, 10, 2

    def body(rank, c, row0, nl):
        c.set_tuning("lazy", 1)
        A = lk.diag_linop_gpu(1.0 + (row0 + np.arange(nl)) / n, c)

        class pyop(lk.abstract_linop):                            # python operator => the python (reference) step loop
            def matvec(self, vi, vo): A.matvec(vi, vo)
        B = lk.krylov_basis_gpu(nl, m + 1, np.float64, c)
        B[0].rand(True, seed=7)
        X = [B[j] for j in range(m + 1)]
        H = np.zeros((m + 1, m), order="F")
        info = lk.arnoldi(pyop(), X, H)
        return info, H, c.lazy_fusion_stats()

    res, grp = _sharded(n, nranks, body)
    X1 = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    X1[0].rand(True, seed=7)
    H1 = np.zeros((m + 1, m), order="F")
    assert lk.arnoldi(lk.diag_linop_gpu(1.0 + np.arange(n) / n, ctx), X1, H1) == 0
    for info, H, fs in res:
        assert info == 0 and fs[0] == 2 * m and fs[1] == 0 and fs[3] == 0
        assert np.array_equal(H, res[0][1])
        for j in range(m):
            assert np.abs(H[:, j] - H1[:, j]).max() <= 1e-12 * np.abs(H1[:, j]).max()


# ----------------------------------------------------------------------------- dense and CSR operators with an all-gather of x
class _EmulatedAllGather:
    """All-gather of row blocks between `nranks` threads (lk_allgather_fn): what ncclAllGather / ncclSend+ncclRecv do on a node."""

    def __init__(self, nranks):
        import torch
        self.torch = torch
        self.n = nranks
        self.barrier = threading.Barrier(nranks)
        self.sends = [None] * nranks
        self.calls = 0

    def hook(self, rank, ctx):
        torch = self.torch

        def _cb(_user, send, recv, counts, displs, nranks, _stream):
            try:
                ctx.sync_stream_only()
                cnt = [int(counts[r]) for r in range(nranks)]
                dsp = [int(displs[r]) for r in range(nranks)]
                self.sends[rank] = torch.as_tensor(_DevMem(int(send), cnt[rank]), device="cuda:0") if cnt[rank] else None
                self.barrier.wait(timeout=120)
                for r in range(nranks):
                    if cnt[r]:
                        torch.as_tensor(_DevMem(int(recv) + 8 * dsp[r], cnt[r]), device="cuda:0").copy_(self.sends[r])
                torch.cuda.synchronize()
                if rank == 0:
                    self.calls += 1
                self.barrier.wait(timeout=120)
                return 0
            except Exception as exc:  # noqa: BLE001
                print("emulated all-gather failed:", repr(exc))
                self.barrier.abort()
                return 1
        return _capi.ALLGATHER_FN(_cb)


def _sharded_with_allgather(n, nranks, body):
    lib = _capi.load()
    grp, ag = _EmulatedGroup(nranks), _EmulatedAllGather(nranks)
    out, errs = [None] * nranks, []

    def worker(rank):
        try:
            ctx = lk.Context(device=0, use_torch_stream=False)
            ctx.sync_stream_only = lambda: _capi.check(lib.lk_sync(ctx._h))
            cb = grp.hook(rank, ctx)
            _capi.check(lib.lk_set_allreduce(ctx._h, cb, None, nranks, rank))
            ctx._cb, ctx.nranks, ctx.rank = cb, nranks, rank
            ctx.set_allgather(ag.hook(rank, ctx))
            row0, nl = lk.row_partition(n, nranks, rank)
            ctx.set_partition(row0, n)
            out[rank] = body(rank, ctx, row0, nl)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            grp.barrier.abort(); ag.barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    return out, ag


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_dense_linop_matvec_rmatvec_and_arnoldi(ctx, dtype, nranks):
    """dense_linop (AbstractLinops.fypp:608-660) row-sharded: a row block of A per rank, x all-gathered for matvec, the ranks'
    A_rows^H x_rows summed for rmatvec.  matvec is bit-identical to the single-context operator (same products, same order per
    row), rmatvec and a whole Arnoldi factorisation agree to rounding; also against numpy."""
    n, m = 2051, 12
    rng = np.random.default_rng(5)
    cplx = np.dtype(dtype).kind == "c"
    Afull = rng.standard_normal((n, n)) / np.sqrt(n) + (1j * rng.standard_normal((n, n)) / np.sqrt(n) if cplx else 0)
    Afull = np.asfortranarray(Afull.astype(dtype))
    from oracle import oracle as ora
    x_full = np.empty(n, dtype=dtype); ora.fill_counter(x_full, 13); x_full /= np.linalg.norm(x_full)

    def body(rank, c, row0, nl):
        A = lk.dense_linop_gpu(Afull[row0:row0 + nl, :], c, n_global=n)
        v = lk.dense_vector_gpu.from_array(x_full[row0:row0 + nl], c)
        w = v.zeros_like(); wt = v.zeros_like()
        A.apply_matvec(v, w); A.apply_rmatvec(v, wt)
        X = lk.krylov_basis_gpu(nl, m + 1, dtype, c)
        X.upload(x_full[row0:row0 + nl].reshape(-1, 1), 0)
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        info = lk.arnoldi(A, X, H)
        Ht = np.zeros((m + 1, m), dtype=dtype, order="F")
        Xt = lk.krylov_basis_gpu(nl, m + 1, dtype, c)
        Xt.upload(x_full[row0:row0 + nl].reshape(-1, 1), 0)
        info_t = lk.arnoldi(A, Xt, Ht, transpose=True)
        return w.to_array(), wt.to_array(), info, H, info_t, Ht

    res, ag = _sharded_with_allgather(n, nranks, body)
    A1 = lk.dense_linop_gpu(Afull, ctx)
    v1 = lk.dense_vector_gpu.from_array(x_full, ctx)
    w1 = v1.zeros_like(); wt1 = v1.zeros_like()
    A1.apply_matvec(v1, w1); A1.apply_rmatvec(v1, wt1)
    ws, wts = np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])
    assert np.array_equal(ws, w1.to_array())
    assert np.abs(wts - wt1.to_array()).max() <= 1e-14 * np.abs(wt1.to_array()).max() * np.sqrt(n)
    assert np.abs(ws - Afull @ x_full).max() <= 1e-13 and np.abs(wts - Afull.conj().T @ x_full).max() <= 1e-13
    X1 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X1.upload(x_full.reshape(-1, 1), 0)
    H1 = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.arnoldi(A1, X1, H1) == 0
    X1.upload(np.zeros((n, m + 1), dtype=dtype)); X1.upload(x_full.reshape(-1, 1), 0)
    H1t = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.arnoldi(A1, X1, H1t, transpose=True) == 0
    for r in res:
        assert r[2] == 0 and r[4] == 0
        for j in range(m):
            assert np.abs(r[3][:, j] - H1[:, j]).max() <= 1e-12 * np.abs(H1[:, j]).max()
            assert np.abs(r[5][:, j] - H1t[:, j]).max() <= 1e-12 * np.abs(H1t[:, j]).max()
    assert ag.calls == 1 + m                                       # one all-gather per matvec; rmatvec uses the all-reduce


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_csr_linop_matvec_rmatvec_and_gmres(ctx, dtype, nranks):
    """A user's sparse operator row-sharded (rows of A per rank with global column indices): matvec / rmatvec against scipy and
    the single-context operator; the 5-point Laplacian as a sharded CSR matrix reproduces the single-context GMRES run."""
    import scipy.sparse as sp
    n = 3001
    rng = np.random.default_rng(2)
    cplx = np.dtype(dtype).kind == "c"
    M = sp.random(n, n, density=0.004, format="csr", random_state=3, dtype=np.float64)
    if cplx:
        M = (M + 1j * sp.random(n, n, density=0.004, format="csr", random_state=4, dtype=np.float64)).tocsr()
    M = (M + sp.diags(np.linspace(1.0, 2.0, n))).tocsr().astype(dtype)
    M.sort_indices()
    from oracle import oracle as ora
    x_full = np.empty(n, dtype=dtype); ora.fill_counter(x_full, 17)

    def body(rank, c, row0, nl):
        A = lk.csr_linop_gpu(M[row0:row0 + nl, :], c, n_global=n)
        v = lk.dense_vector_gpu.from_array(x_full[row0:row0 + nl], c)
        w = v.zeros_like(); wt = v.zeros_like()
        A.apply_matvec(v, w); A.apply_rmatvec(v, wt)
        return w.to_array(), wt.to_array()

    res, _ag = _sharded_with_allgather(n, nranks, body)
    A1 = lk.csr_linop_gpu(M, ctx)
    v1 = lk.dense_vector_gpu.from_array(x_full, ctx)
    w1 = v1.zeros_like(); wt1 = v1.zeros_like()
    A1.apply_matvec(v1, w1); A1.apply_rmatvec(v1, wt1)
    ws, wts = np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])
    assert np.array_equal(ws, w1.to_array())                        # a row's entries are summed in the same order
    scale = np.abs(x_full).max() * 4
    assert np.abs(wts - wt1.to_array()).max() <= 1e-14 * scale
    assert np.abs(ws - M @ x_full).max() <= 1e-13 * scale and np.abs(wts - M.conj().T @ x_full).max() <= 1e-13 * scale
    if cplx:
        return
    # the Laplacian of config 3 as a sharded CSR matrix in GMRES
    N = 61
    nn = N * N
    T = sp.diags([-1.0, 4.0, -1.0], [-1, 0, 1], shape=(N, N))
    L = ((sp.kron(sp.identity(N), T) + sp.kron(sp.diags([-1.0, -1.0], [-1, 1], shape=(N, N)), sp.identity(N))) * float((N + 1) ** 2)).tocsr()
    L.sort_indices()
    b_full = np.empty(nn); ora.fill_counter(b_full, 11)

    def body2(rank, c, row0, nl):
        A = lk.csr_linop_gpu(L[row0:row0 + nl, :], c, n_global=nn)
        b = lk.dense_vector_gpu.from_array(b_full[row0:row0 + nl], c)
        x = b.zeros_like()
        meta = lk.gmres_dp_metadata()
        info = lk.gmres(A, b, x, rtol=1e-8, options=lk.gmres_dp_opts(kdim=20, maxiter=1), meta=meta)
        return x.to_array(), info, np.array(meta.res)

    res2, _ = _sharded_with_allgather(nn, nranks, body2)
    A1 = lk.laplacian2d_linop_gpu(N, ctx)
    b1 = lk.dense_vector_gpu.from_array(b_full, ctx)
    x1 = b1.zeros_like(); m1 = lk.gmres_dp_metadata()
    info1 = lk.gmres(A1, b1, x1, rtol=1e-8, options=lk.gmres_dp_opts(kdim=20, maxiter=1), meta=m1)
    xs = np.concatenate([r[0] for r in res2])
    assert all(r[1] == info1 for r in res2)
    assert_close(res2[0][2], np.array(m1.res), "sharded CSR gmres: residual history vs single rank", scale=m1.res[0])
    assert_close(xs, x1.to_array(), "sharded CSR gmres: solution vs single rank")


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_csr_compressed_exchange_on_a_banded_matrix(ctx, dtype, nranks):
    """A banded sparse matrix row-sharded: only the entries of x that another rank's rows reference travel (a few rows of band per
    rank boundary, packed and all-gathered), not all of x.  matvec bit-identical to the single-context operator, rmatvec to
    rounding; the exchange really is compressed (all-gather volume << n); a matrix with one dense row falls back to the whole x."""
    import scipy.sparse as sp
    n, bw = 20_011, 7
    rng = np.random.default_rng(4)
    cplx = np.dtype(dtype).kind == "c"
    diags = [rng.standard_normal(n - abs(o)) + (1j * rng.standard_normal(n - abs(o)) if cplx else 0) for o in range(-bw, bw + 1)]
    M = sp.diags(diags, list(range(-bw, bw + 1)), shape=(n, n), format="csr").astype(dtype)
    M.sort_indices()
    Mdense_row = M.tolil(); Mdense_row[n // 2, :] = 1.0; Mdense_row = Mdense_row.tocsr().astype(dtype); Mdense_row.sort_indices()
    from oracle import oracle as ora
    x_full = np.empty(n, dtype=dtype); ora.fill_counter(x_full, 21)
    volumes = {}

    def make_body(mat, tag):
        def body(rank, c, row0, nl):
            A = lk.csr_linop_gpu(mat[row0:row0 + nl, :], c, n_global=n)
            v = lk.dense_vector_gpu.from_array(x_full[row0:row0 + nl], c)
            w = v.zeros_like(); wt = v.zeros_like()
            A.apply_matvec(v, w); A.apply_matvec(v, w); A.apply_rmatvec(v, wt)
            return w.to_array(), wt.to_array()
        return body

    class _Counting(_EmulatedAllGather):
        def hook(self, rank, ctx_):
            inner = super().hook(rank, ctx_)

            def _cb(user, send, recv, counts, displs, nr, stream):
                if rank == 0:
                    self.volume = getattr(self, "volume", [])
                    self.volume.append(sum(int(counts[r]) for r in range(nr)))
                return inner(user, send, recv, counts, displs, nr, stream)
            self._keep = getattr(self, "_keep", []) + [inner]
            return _capi.ALLGATHER_FN(_cb)

    for mat, tag in ((M, "banded"), (Mdense_row, "one dense row")):
        lib = _capi.load()
        grp, ag = _EmulatedGroup(nranks), _Counting(nranks)
        out, errs = [None] * nranks, []

        def worker(rank, mat=mat, tag=tag):
            try:
                c = lk.Context(device=0, use_torch_stream=False)
                c.sync_stream_only = lambda: _capi.check(lib.lk_sync(c._h))
                cb = grp.hook(rank, c)
                _capi.check(lib.lk_set_allreduce(c._h, cb, None, nranks, rank))
                c._cb, c.nranks, c.rank = cb, nranks, rank
                c.set_allgather(ag.hook(rank, c))
                row0, nl = lk.row_partition(n, nranks, rank)
                c.set_partition(row0, n)
                out[rank] = make_body(mat, tag)(rank, c, row0, nl)
            except Exception as exc:  # noqa: BLE001
                errs.append(exc)
                grp.barrier.abort(); ag.barrier.abort()
        ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
        [t.start() for t in ts]
        [t.join(600) for t in ts]
        assert not errs, errs
        A1 = lk.csr_linop_gpu(mat, ctx)
        v1 = lk.dense_vector_gpu.from_array(x_full, ctx)
        w1 = v1.zeros_like(); wt1 = v1.zeros_like()
        A1.apply_matvec(v1, w1); A1.apply_rmatvec(v1, wt1)
        ws, wts = np.concatenate([r[0] for r in out]), np.concatenate([r[1] for r in out])
        assert np.array_equal(ws, w1.to_array()), tag
        scale = np.abs(x_full).max() * np.abs(mat).sum(axis=1).max()
        assert np.abs(wts - wt1.to_array()).max() <= 1e-14 * scale, tag
        assert np.abs(ws - mat @ x_full).max() <= 1e-13 * scale, tag
        volumes[tag] = ag.volume
    ED = 2 if cplx else 1
    # banded: the two matvecs each gathered ~2 * bandwidth entries per rank boundary (after the creation-time metadata exchange)
    assert volumes["banded"][-1] == volumes["banded"][-2] <= 2 * bw * (nranks - 1) * 2 * ED
    # one dense row: (nearly) everything would travel -> the whole x is gathered
    assert volumes["one dense row"][-1] == n * ED


@pytest.mark.parametrize("fault", ["column index", "rows", "value type"])
def test_sharded_csr_creation_fails_on_every_rank_together(ctx, fault):
    """lk_linop_csr_create_sharded is COLLECTIVE in its errors too: ONE rank's row block is unusable (a column index out of range:
    caught by the library's validation; the wrong number of rows / values of the wrong type: caught by the python wrapper) and
    EVERY rank comes back with an error -- the faulty rank with its own message, the others naming the rank that failed -- instead of
    waiting for it in the metadata exchange.  The ranks then create the operator properly: nothing was left half-built."""
    import scipy.sparse as sp
    from oracle import oracle as ora
    nn, nranks, bad = 600, 3, 1
    L = sp.diags([1.0, -2.0, 1.0], [-1, 0, 1], shape=(nn, nn)).tocsr()
    x_full = np.empty(nn); ora.fill_counter(x_full, 5)

    def body(rank, c, row0, nl):
        blk = L[row0:row0 + nl, :].copy()
        if rank == bad and fault == "column index":
            blk = (blk.indptr.copy(), np.where(np.arange(blk.nnz) == 7, nn + 3, blk.indices).astype(np.int32), blk.data.copy())
        elif rank == bad and fault == "rows":
            blk = L[row0:row0 + nl - 1, :].copy()
        elif rank == bad and fault == "value type":
            blk = blk.astype(np.float32)
        try:
            lk.csr_linop_gpu(blk, c, n_global=nn)
            msg = None
        except Exception as exc:  # noqa: BLE001
            msg = f"{type(exc).__name__}: {exc}"
        A = lk.csr_linop_gpu(L[row0:row0 + nl, :], c, n_global=nn)          # the communicator is still usable, on every rank
        v = lk.dense_vector_gpu.from_array(x_full[row0:row0 + nl], c)
        w = v.zeros_like()
        A.apply_matvec(v, w)
        return msg, w.to_array()

    res, _ = _sharded_with_allgather(nn, nranks, body)
    msgs = [r[0] for r in res]
    assert all(m is not None for m in msgs), msgs
    want = {"column index": "out of range", "rows": "not this rank's block", "value type": "float64 / complex128"}[fault]
    assert want in msgs[bad], msgs
    for r in range(nranks):
        if r != bad:
            assert f"rank {bad} failed" in msgs[r], msgs
    assert np.abs(np.concatenate([r[1] for r in res]) - L @ x_full).max() <= 1e-13 * np.abs(x_full).max() * 4


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("p", [2, 6])
def test_sharded_block_arnoldi_matches_single_context(ctx, dtype, p):
    """lk_arnoldi_block on a row-sharded context (two ranks as threads, emulated all-reduce): the asynchronous block batch -- the block
    Gram-Schmidt on the vector units (p = 2) and on the matrix cores (p = 6), the p-column qr on the three-sweep schedule (a sharded
    context never takes the single launch) -- against the single-context factorisation: same H on every rank, to rounding the single rank's."""
    n, kdim, nranks = 120_001, 6, 2
    ncol = (kdim + 1) * p

    def dvals(row0, nl):
        g = (row0 + np.arange(nl)) / n
        d = 1.0 + g
        return (d * np.exp(0.3j * g)).astype(dtype) if np.dtype(dtype).kind == "c" else d

    def start(X, c):
        for j in range(p):
            X[j].rand(False, seed=70 + j)                          # global vectors, partition independent
        R = np.zeros((p, p), dtype=dtype, order="F")
        assert lk.qr(X[:p], R) == 0

    def body(rank, c, row0, nl):
        X = lk.krylov_basis_gpu(nl, ncol, dtype, c)
        start(X, c)
        H = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
        before = c.resident_stats()[0]
        info = lk.arnoldi(lk.diag_linop_gpu(dvals(row0, nl), c), X, H, blksize=p)
        assert c.resident_stats()[0] == before                     # no single launch on a sharded context
        return info, H, X.download()

    res, grp = _sharded(n, nranks, body)
    X1 = lk.krylov_basis_gpu(n, ncol, dtype, ctx)
    start(X1, ctx)
    H1 = np.zeros((ncol, kdim * p), dtype=dtype, order="F")
    assert lk.arnoldi(lk.diag_linop_gpu(dvals(0, n), ctx), X1, H1, blksize=p) == 0
    for info, H, _ in res:
        assert info == 0
        assert np.array_equal(H, res[0][1])
        for j in range(kdim * p):
            assert np.abs(H[:, j] - H1[:, j]).max() <= 1e-12 * np.abs(H1[:, j]).max(), j
    Xs = np.concatenate([r[2] for r in res], axis=0)
    assert np.abs(Xs.conj().T @ Xs - np.eye(ncol)).max() <= 1e-12
    d = dvals(0, n)
    assert np.abs(d[:, None] * Xs[:, :kdim * p] - Xs @ res[0][1]).max() <= 1e-12 * np.abs(d).max()
