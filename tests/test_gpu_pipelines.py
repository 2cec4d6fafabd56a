"""The asynchronous step loops inside the engine (lk_arnoldi / lk_lanczos / lk_bidiag: every step enqueued without a host round trip, a
device-side stop flag for breakdowns) and the host-side pipelines built on them (eigs / eighs / svds: device segments beside the per-step
small eigenproblems on host threads): each returns bit for bit what the reference's step-by-step loop returns
(src/Krylov/arnoldi.fypp:34-73, lanczos.fypp:7-64, golub_kahan.fypp:7-64, IterativeSolvers.fypp:1059-1100)."""
import ctypes as C
import os

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, seeded
from tests._tol import assert_close, assert_columns_close

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0], ids=["single_launch", "three_sweeps"])
def ctx(request, ctx):
    """Every test of this file that takes the shared context runs on BOTH schedules of the Gram-Schmidt step: the single persistent
    launch for cache-resident panels (csrc/lk_resident.hip.h, the default) and the three sweeps -- at these sizes the single launch
    would otherwise take every panel and the sweeps would lose their small-size coverage.  (Full-size cases never fit the caches:
    once is enough.)"""
    if request.param == 0 and "full_size" in request.node.name:
        pytest.skip("the panel does not fit the caches: the three-sweep schedule ran in the other parametrisation")
    ctx.set_tuning("resident", request.param)
    yield ctx
    ctx.set_tuning("resident", 1)


@pytest.mark.parametrize("dtype", KINDS)
def test_async_arnoldi_equals_the_step_by_step_schedule(dtype):
    """lk_arnoldi enqueues all steps with a device-side breakdown flag (one synchronisation per call).  Same kernels,
    same order, same inputs as the one-round-trip-per-step schedule: H and the basis must be bit-identical, with and
    without a breakdown, and the columns beyond a breakdown must stay untouched (arnoldi.fypp:58-71)."""
    c = lk.Context(device=0)
    n, m = 250_003, 40
    g = np.arange(n) / n
    d = (1.0 + g).astype(dtype) if np.dtype(dtype).kind == "f" else ((1.0 + g) * np.exp(1j * g)).astype(dtype)
    out = {}
    for mode in (0, 1):
        c.set_tuning("async_arnoldi", mode)
        X = lk.krylov_basis_gpu(n, m + 1, dtype, c)
        X[0].rand(True, seed=7)
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        assert lk.arnoldi(lk.diag_linop_gpu(d, c), X, H) == 0
        out[mode] = (H.tobytes(), X.download().tobytes())
    assert out[0] == out[1]
    # invariant subspace after 6 steps: operator with 6 distinct eigenvalues
    d6 = (1.0 + (np.arange(n) % 6)).astype(dtype)
    res = {}
    for mode in (0, 1):
        c.set_tuning("async_arnoldi", mode)
        X = lk.krylov_basis_gpu(n, m + 1, dtype, c)
        X[0].rand(True, seed=9)
        marker = seeded(n, dtype, 123)
        for j in range(7, m + 1):
            X.upload(marker.reshape(-1, 1), j)            # whatever sits beyond the breakdown must survive
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        info = lk.arnoldi(lk.diag_linop_gpu(d6, c), X, H, tol=1e-10)
        res[mode] = (info, H.tobytes(), X.download().tobytes())
        assert info == 6
        assert np.array_equal(X.download(m, 1)[:, 0], marker) and np.array_equal(X.download(7, 1)[:, 0], marker)
    assert res[0] == res[1]
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_arnoldi_delivered_in_segments_is_the_same_factorisation(dtype):
    """lk_arnoldi_segments (round 5): the same steps, enqueued as one batch, with the columns of H reported segment by segment while the
    device runs on.  H and the basis are bit-identical to lk_arnoldi; every step is reported exactly once, in order, in the ranges asked for;
    a progress function that asks to stop ends the call with at most 24 more steps run and nothing more reported; a breakdown inside a
    segment gives the same info and H, reports nothing beyond it and leaves the columns beyond it untouched."""
    c = lk.Context(device=0)
    n, m = 120_007, 100
    g = np.arange(n) / n
    d = (1.0 + g).astype(dtype) if np.dtype(dtype).kind == "f" else ((1.0 + g) * np.exp(1j * g)).astype(dtype)
    A = lk.diag_linop_gpu(d, c)
    X0 = lk.krylov_basis_gpu(n, m + 1, dtype, c); X0[0].rand(True, seed=7)
    H0 = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.arnoldi(A, X0, H0) == 0
    ref = (H0.tobytes(), X0.download().tobytes())
    for segs in ([16, 32, 48, 64, 80, 88, 92, 96, 98, 99, 100], [50], [1, 2, 3, 99], []):
        X = lk.krylov_basis_gpu(n, m + 1, dtype, c); X[0].rand(True, seed=7)
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        seen, snapshots = [], []

        def progress(kfirst, klast, H=H, seen=seen, snapshots=snapshots):
            seen.append((kfirst, klast))
            snapshots.append(H[:klast + 1, kfirst - 1:klast].copy())       # the reported columns are final when reported
            return False
        assert lk.arnoldi(A, X, H, _segments=segs, _progress=progress) == 0
        assert (H.tobytes(), X.download().tobytes()) == ref, segs
        ends = [b for _a, b in seen]
        assert [a for a, _b in seen] == [1] + [b + 1 for b in ends[:-1]] and ends[-1] == m, (segs, seen)
        assert set(s_ for s_ in segs if s_ < m) <= set(ends), (segs, seen)
        for (a, b), snap in zip(seen, snapshots):
            assert np.array_equal(snap, H0[:b + 1, a - 1:b])
    # a restart range (kstart > 1) with segments
    X = lk.krylov_basis_gpu(n, m + 1, dtype, c); X[0].rand(True, seed=7)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.arnoldi(A, X, H, kstart=1, kend=37) == 0
    seen = []
    assert lk.arnoldi(A, X, H, kstart=38, kend=m, _segments=[40, 70, 100], _progress=lambda a, b: seen.append((a, b)) and False) == 0
    assert (H.tobytes(), X.download().tobytes()) == ref and seen == [(38, 40), (41, 70), (71, 100)]
    # stop on request: nothing reported after the request, at most 24 steps run beyond the last reported one
    X = lk.krylov_basis_gpu(n, m + 1, dtype, c); X[0].rand(True, seed=7)
    mark = seeded(n, dtype, 321)
    for j in range(45, m + 1):
        X.upload(mark.reshape(-1, 1), j)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    seen = []

    def stop_at_20(kfirst, klast):
        seen.append((kfirst, klast))
        return klast >= 20
    assert lk.arnoldi(A, X, H, _segments=list(range(4, m + 1, 4)), _progress=stop_at_20) == 0
    assert seen[-1] == (17, 20) and np.array_equal(H[:21, :20], H0[:21, :20]) and not H[:, 20:].any()
    Xg = X.download()
    assert np.array_equal(Xg[:, :21], X0.download()[:, :21])
    assert np.array_equal(Xg[:, 45], mark) and np.array_equal(Xg[:, m], mark)           # 20 + 24 steps at most touched columns <= 44
    # breakdown inside a segment: invariant subspace after 6 steps
    d6 = (1.0 + (np.arange(n) % 6)).astype(dtype)
    res = {}
    for use_segments in (False, True):
        X = lk.krylov_basis_gpu(n, m + 1, dtype, c); X[0].rand(True, seed=9)
        for j in range(7, m + 1):
            X.upload(mark.reshape(-1, 1), j)
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        seen = []
        kw = dict(_segments=[4, 8, 16, 64], _progress=lambda a, b, seen=seen: seen.append((a, b)) and False) if use_segments else {}
        info = lk.arnoldi(lk.diag_linop_gpu(d6, c), X, H, tol=1e-10, **kw)
        res[use_segments] = (info, H.tobytes(), X.download().tobytes())
        assert info == 6 and np.array_equal(X.download(7, 1)[:, 0], mark) and np.array_equal(X.download(m, 1)[:, 0], mark)
        if use_segments:
            assert seen == [(1, 4), (5, 6)], seen
    assert res[False] == res[True]
    c.close()


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_pipelined_eigs_cycle_equals_the_step_by_step_one(ctx, dtype):
    """eigs' whole-cycle pipeline (one asynchronous lk_arnoldi per Krylov-Schur cycle, the per-step geev tests afterwards
    on several host threads) returns exactly what the reference's step / geev / step / geev loop returns: same number of
    Arnoldi steps, same eigenvalues and residuals bit for bit, same eigenvectors -- converging mid-cycle (early stop:
    the work arrays are put back into the reference's state before the restart) and after restarts."""
    n, nev = 4_000, 4
    rng = np.random.default_rng(11)
    d = np.r_[np.array([3.0, 2.6, 2.2, 1.9, 1.7]), 1.0 + 0.4 * rng.random(n - 5)]
    if np.dtype(dtype).kind == "c":
        d = d * np.exp(0.2j * rng.random(n))
    A = lk.diag_linop_gpu(d.astype(dtype), ctx)
    out = {}
    for kdim, maxr, tag in ((40, 60, "early stop inside the first cycle"), (12, 60, "several restarts"),
                            (12, 1, "restarts exhausted: the final eig of the restarted H is computed ahead, beside the last tests"),
                            (37, 0, "one cycle, tapered segments (16, 5 | 8, 4, 2, 1, 1)")):
        for pipe in (False, True):
            V = lk.krylov_basis_gpu(n, nev, dtype, ctx)
            x0 = lk.dense_vector_gpu(n, dtype, ctx); x0.rand(False, seed=3)
            vals, res, info = lk.eigs(A, V, x0=x0, kdim=kdim, tolerance=1e-10 if maxr == 60 else 1e-15, max_restarts=maxr, pipelined=pipe)
            out[(kdim, maxr, pipe)] = (vals, res, info, V.download())
        (v0, r0, i0, X0), (v1, r1, i1, X1) = out[(kdim, maxr, False)], out[(kdim, maxr, True)]
        assert i0 == i1, tag
        assert np.array_equal(v0, v1) and np.array_equal(r0, r1), tag
        assert np.array_equal(X0, X1), tag
        if maxr == 60:
            assert np.abs(np.sort(np.abs(v1))[::-1] - np.sort(np.abs(d))[::-1][:nev]).max() <= 1e-8, tag
    assert out[(40, 60, True)][2] < 40 and out[(12, 60, True)][2] > 12
    assert 12 < out[(12, 1, True)][2] <= 24                                     # two cycles, tolerance out of reach: no early stop
    assert 16 < out[(37, 0, True)][2] <= 37                                    # (real kind: the leading pairs reach a residual of exactly 0 just before the cycle ends)


@pytest.mark.parametrize("dtype", KINDS)
def test_lanczos_tridiagonal_matches_the_oracle(ctx, dtype):
    """lanczos_tridiagonalization (lanczos.fypp:7-64) on a dense symmetric / Hermitian operator: every entry of T
    against the oracle's restatement, normwise 1e-12 per column."""
    n, m = 3001, 40
    rng = np.random.default_rng(11)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    if np.dtype(dtype).kind == "c":
        A = A + 1j * rng.standard_normal((n, n)) / np.sqrt(n)
    A = np.asfortranarray(((A + A.conj().T) / 2 + np.diag(np.linspace(1.0, 3.0, n))).astype(dtype))
    x0 = seeded(n, dtype, 21); x0 /= np.linalg.norm(x0)
    X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X.upload(x0.reshape(-1, 1), 0)
    T = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.lanczos(lk.dense_linop_gpu(A, ctx), X, T) == 0
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F"); Xo[:, 0] = x0
    To = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.lanczos(ora.DenseOp(A), Xo, To) == 0
    assert_columns_close(T, To, f"lanczos dense 3001 x 3001 {np.dtype(dtype)}")
    # the diagonal-operator variant isolates the path (no gemv): 1e-12
    d = (1.0 + np.arange(n) / n).astype(dtype)
    X.upload(np.zeros((n, m + 1), dtype=dtype)); X.upload(x0.reshape(-1, 1), 0)
    T[...] = 0
    assert lk.lanczos(lk.diag_linop_gpu(d, ctx), X, T) == 0
    Xo[...] = 0; Xo[:, 0] = x0; To[...] = 0
    assert ora.lanczos(ora.DiagOp(d), Xo, To) == 0
    for j in range(m):
        assert np.abs(T[:, j] - To[:, j]).max() <= 1e-12 * np.abs(To[:, j]).max()
    G = lk.Gram(X[:m + 1])
    assert np.abs(G - np.eye(m + 1)).max() < 1e-12


@pytest.mark.parametrize("dtype", KINDS)
def test_fused_lanczos_breakdown_restart_ranges_and_the_per_object_loop(ctx, dtype):
    """lk_lanczos (all steps of a call enqueued asynchronously) against the oracle and against the per-object loop of the
    mirror (the reference's own sequence of dot / axpby / double_gram_schmidt_step / norm / scal calls):
    breakdown -- a diagonal operator with three distinct values spans a 3-dimensional Krylov space: info = 3, T(4, 3) below
    tol, X(4) left unscaled (lanczos.fypp:32-36); continued ranges kstart..kend equal the one-shot run; a caller's tolerance
    below atol_dp resumes past the device-side stop."""
    n, m = 20_003, 24
    x0 = seeded(n, dtype, 5); x0 /= np.linalg.norm(x0)
    # (a) breakdown
    d3 = np.choose(np.arange(n) % 3, [1.0, 2.0, 3.5]).astype(dtype)
    X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X.upload(x0.reshape(-1, 1), 0)
    T = np.zeros((m + 1, m), dtype=dtype, order="F")
    info = lk.lanczos(lk.diag_linop_gpu(d3, ctx), X, T)
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F"); Xo[:, 0] = x0
    To = np.zeros((m + 1, m), dtype=dtype, order="F")
    info_o = ora.lanczos(ora.DiagOp(d3), Xo, To)
    assert info == info_o == 3
    assert np.abs(T[:3, :3] - To[:3, :3]).max() <= 1e-12 * np.abs(To[:3, :3]).max()
    assert abs(T[3, 2]) < 1e-12 and not T[:, 3:].any()
    # (b) ranges and the per-object loop
    d = (1.0 + np.arange(n) / n).astype(dtype)
    A = lk.diag_linop_gpu(d, ctx)
    Xo[...] = 0; Xo[:, 0] = x0; To[...] = 0
    assert ora.lanczos(ora.DiagOp(d), Xo, To) == 0
    X.upload(np.zeros((n, m + 1), dtype=dtype)); X.upload(x0.reshape(-1, 1), 0); T[...] = 0
    assert lk.lanczos(A, X, T, kstart=1, kend=7) == 0
    assert lk.lanczos(A, X, T, kstart=8, kend=8) == 0
    assert lk.lanczos(A, X, T, kstart=9, kend=m, tol=1e-300) == 0        # tol < atol_dp: same result, resumable path
    for j in range(m):
        assert np.abs(T[:, j] - To[:, j]).max() <= 1e-12 * np.abs(To[:, j]).max()
    assert np.abs(X.download() - Xo).max() <= 1e-10                      # Krylov vectors (conditioning grows with the step)

    class per_object(lk.abstract_linop):                                 # not an engine operator: the mirror's python loop runs
        def matvec(self, vec_in, vec_out):
            A.matvec(vec_in, vec_out)
    X2 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X2.upload(x0.reshape(-1, 1), 0)
    T2 = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.lanczos(per_object(), X2, T2) == 0
    for j in range(m):
        assert np.abs(T[:, j] - T2[:, j]).max() <= 1e-12 * np.abs(To[:, j]).max()


@pytest.mark.parametrize("dtype", KINDS)
def test_pipelined_eighs_equals_the_step_by_step_one(ctx, dtype):
    """eighs with the Lanczos steps enqueued in asynchronous device segments and the per-step eigh tests on host threads
    (`pipelined=True`) against the reference's alternation of one step and one eigh (`pipelined=False`): same step count,
    eigenvalues, residuals and eigenvectors bit for bit -- including an early stop in the middle of a segment -- and both
    against the oracle."""
    n, nev, kdim = 30_011, 4, 60
    d = np.r_[np.linspace(1.0, 2.0, n - nev), 3.0 + 0.5 * np.arange(nev)].astype(dtype)     # nev separated leading eigenvalues
    x0 = seeded(n, dtype, 9)
    out = []
    for pipe in (False, True):
        X = lk.krylov_basis_gpu(n, nev, dtype, ctx)
        vals, res, info = lk.eighs(lk.diag_linop_gpu(d, ctx), X, x0=lk.dense_vector_gpu.from_array(x0, ctx), kdim=kdim,
                                   tolerance=1e-10, pipelined=pipe)
        out.append((vals, res, info, X.download()))
    (v0, r0, i0, X0), (v1, r1, i1, X1) = out
    assert i0 == i1 and 5 < i0 < kdim                                     # converged before kdim: the pipeline stopped early
    assert np.array_equal(v0, v1) and np.array_equal(r0, r1) and np.array_equal(X0, X1)
    vo, ro, Xo, info_o = ora.eighs(ora.DiagOp(d), x0.copy(), nev, kdim=kdim, tolerance=1e-10)
    assert info_o == i0
    assert_close(v0, vo, f"pipelined eighs values vs oracle {np.dtype(dtype)}")
    assert np.abs(v0 - (3.0 + 0.5 * np.arange(nev))[::-1]).max() <= 1e-9


@pytest.mark.parametrize("dtype", KINDS)
def test_fused_bidiagonalization_against_the_oracle_breakdown_and_ranges(ctx, dtype):
    """lk_bidiag (golub_kahan.fypp:7-64, every step of a call enqueued asynchronously, stop flag per half step) against the
    oracle: a dense non-normal operator (B, both bases), continued ranges, and breakdowns in the right half of a step (alpha
    below tol, the left half of that step must not run) -- a rank-2 operator at step 3, a start vector in the kernel of A^H at
    step 1 (info = 1, nothing of U(2) touched)."""
    n, m = 1501, 20
    rng = np.random.default_rng(3)
    cplx = np.dtype(dtype).kind == "c"
    G = rng.standard_normal((n, n)) / np.sqrt(n) + (1j * rng.standard_normal((n, n)) / np.sqrt(n) if cplx else 0)
    G = np.asfortranarray(G.astype(dtype))
    u0 = seeded(n, dtype, 4); u0 /= np.linalg.norm(u0)
    A = lk.dense_linop_gpu(G, ctx)
    U = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); U.upload(u0.reshape(-1, 1), 0)
    V = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
    B = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.bidiagonalization(A, U, V, B, kstart=1, kend=6) == 0
    assert lk.bidiagonalization(A, U, V, B, kstart=7, kend=7) == 0
    assert lk.bidiagonalization(A, U, V, B, kstart=8, kend=m) == 0
    Uo = np.zeros((n, m + 1), dtype=dtype, order="F"); Uo[:, 0] = u0
    Vo = np.zeros((n, m + 1), dtype=dtype, order="F")
    Bo = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.bidiagonalization(ora.DenseOp(G), ora.DenseOp(np.asfortranarray(G.conj().T)), Uo, Vo, Bo) == 0
    assert_columns_close(B, Bo, f"fused bidiagonalization in ranges {np.dtype(dtype)}")
    Ud, Vd = U.download(), V.download()
    assert np.abs(Ud.conj().T @ Ud - np.eye(m + 1)).max() < 1e-12 and np.abs(Vd[:, :m].conj().T @ Vd[:, :m] - np.eye(m)).max() < 1e-12
    assert_close(G @ Vd[:, :m], Ud @ B, f"fused bidiagonalization relation {np.dtype(dtype)}", scale=np.abs(B).max())   # A V = U B
    # rank 2: the Krylov space of A A^H on u0 has dimension 2
    a, b, c_, e = (seeded(n, dtype, s_) for s_ in (31, 32, 33, 34))
    R = np.asfortranarray((np.outer(a, b.conj()) + np.outer(c_, e.conj())).astype(dtype) / n)
    U.upload(np.zeros((n, m + 1), dtype=dtype)); U.upload(u0.reshape(-1, 1), 0); V.upload(np.zeros((n, m + 1), dtype=dtype)); B[...] = 0
    info = lk.bidiagonalization(lk.dense_linop_gpu(R, ctx), U, V, B, tol=1e-10)
    Uo[...] = 0; Uo[:, 0] = u0; Vo[...] = 0; Bo[...] = 0
    info_o = ora.bidiagonalization(ora.DenseOp(R), ora.DenseOp(np.asfortranarray(R.conj().T)), Uo, Vo, Bo, tol=1e-10)
    # v1, v2 span range(A^H) = span(b, e); u2, u3 use up what span(a, c) adds to u0: V(3) = A^H U(3) has nothing left
    assert info == info_o == 3
    assert_close(B[:3, :2], Bo[:3, :2], f"rank-2 bidiagonalization {np.dtype(dtype)}", scale=np.abs(Bo).max())
    assert abs(B[2, 2]) < 1e-10
    assert not B[3:, :].any() and not B[:, 3:].any()
    assert not U.download(3, m - 2).any() and not V.download(3, m - 2).any()          # nothing beyond the breakdown was touched
    # u0 in the kernel of A^H: alpha = 0 at step 1
    P = np.asfortranarray((np.outer(a, b.conj()) / n).astype(dtype))
    w = u0 - a * (np.vdot(a, u0) / np.vdot(a, a)); w /= np.linalg.norm(w)              # w orthogonal to a: P^H w = 0
    U.upload(np.zeros((n, m + 1), dtype=dtype)); U.upload(w.reshape(-1, 1), 0); V.upload(np.zeros((n, m + 1), dtype=dtype)); B[...] = 0
    assert lk.bidiagonalization(lk.dense_linop_gpu(P, ctx), U, V, B, tol=1e-10) == 1
    assert abs(B[0, 0]) < 1e-10 and not B[1:, :].any() and not U.download(1, m).any()
    # the left half: u0 = a / |a| makes U(2) = A V(1) a multiple of U(1): beta below tol at step 1, V(1) normalised, U(2) not
    ua = a / np.linalg.norm(a)
    U.upload(np.zeros((n, m + 1), dtype=dtype)); U.upload(ua.reshape(-1, 1), 0); V.upload(np.zeros((n, m + 1), dtype=dtype)); B[...] = 0
    info = lk.bidiagonalization(lk.dense_linop_gpu(P, ctx), U, V, B, tol=1e-10)
    Uo[...] = 0; Uo[:, 0] = ua; Vo[...] = 0; Bo[...] = 0
    assert info == ora.bidiagonalization(ora.DenseOp(P), ora.DenseOp(np.asfortranarray(P.conj().T)), Uo, Vo, Bo, tol=1e-10) == 1
    assert abs(B[0, 0] - Bo[0, 0]) <= 1e-12 * abs(Bo[0, 0]) and abs(B[1, 0]) < 1e-10 and not B[:, 1:].any()
    assert abs(np.linalg.norm(V.download(0, 1)) - 1.0) < 1e-14 and np.linalg.norm(U.download(1, 1)) < 1e-10 and not V.download(1, m).any()


@pytest.mark.parametrize("dtype", KINDS)
def test_pipelined_svds_equals_the_step_by_step_one(ctx, dtype):
    """svds with the Golub-Kahan steps in asynchronous device segments and the per-step svd tests on host threads against the
    reference's alternation of one step and one svd: same step count, singular values, residuals and vectors bit for bit
    (early stop inside a segment), and against the oracle."""
    n, nsv, kdim = 3001, 3, 48
    rng = np.random.default_rng(12)
    cplx = np.dtype(dtype).kind == "c"
    G = rng.standard_normal((n, n)) / np.sqrt(n) + (1j * rng.standard_normal((n, n)) / np.sqrt(n) if cplx else 0)
    G[:3, :3] += np.diag([9.0, 7.0, 5.0])
    G = np.asfortranarray(G.astype(dtype))
    u0 = seeded(n, dtype, 8)
    A = lk.dense_linop_gpu(G, ctx)
    out = []
    for pipe in (False, True):
        U = lk.krylov_basis_gpu(n, nsv, dtype, ctx); V = lk.krylov_basis_gpu(n, nsv, dtype, ctx)
        S, res, info = lk.svds(A, U, V, u0=lk.dense_vector_gpu.from_array(u0, ctx), kdim=kdim, tolerance=1e-10, pipelined=pipe)
        out.append((S, res, info, U.download(), V.download()))
    a, b = out
    assert a[2] == b[2] and 3 < a[2] < kdim
    assert all(np.array_equal(x, y) for x, y in zip(a, b) if isinstance(x, np.ndarray))
    So, ro, Uo, Vo, info_o = ora.svds(ora.DenseOp(G), ora.DenseOp(np.asfortranarray(G.conj().T)), u0.copy(), nsv, kdim=kdim, tolerance=1e-10)
    assert info_o == a[2]
    assert_close(a[0], So, f"pipelined svds singular values vs oracle {np.dtype(dtype)}", scale=So[0])
    assert np.abs(a[0] - np.linalg.svd(G, compute_uv=False)[:nsv]).max() <= 1e-9


@pytest.mark.parametrize("dtype", KINDS)
def test_anticipated_first_pass_is_result_neutral_and_disarms_when_unused(dtype):
    """Lazy mode anticipates the first Gram-Schmidt pass of the next Arnoldi step (the norm of column j + 1 runs the dot sweep over
    the columns before it, once `norm(column j)` followed by those dots has been seen): same H as without the anticipation to 1e-12
    (the norm comes out of another kernel), and a prediction nobody uses costs one sweep and switches it off."""
    n, m = 50_003, 24
    g = np.arange(n) / n
    d = (1.0 + g).astype(dtype) if np.dtype(dtype).kind == "f" else ((1.0 + g) * np.exp(1j * g)).astype(dtype)
    out = {}
    for spec in (1, 0):
        c = lk.Context(device=0)
        c.set_tuning("lazy", 1); c.set_tuning("lazy_speculate", spec)
        A = lk.diag_linop_gpu(d, c)

        class pyop(lk.abstract_linop):
            def matvec(self, vi, vo): A.matvec(vi, vo)
        B = lk.krylov_basis_gpu(n, m + 3, dtype, c)
        B[0].rand(True, seed=7)
        X = [B[j] for j in range(m + 1)]
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        assert lk.arnoldi(pyop(), X, H) == 0
        st = c.lazy_speculation_stats()
        assert st == ((m - 2, 0) if spec else (0, 0))
        if spec:
            # break the pattern: the norm of the next column is asked for, its dots are not; the one after that is not anticipated
            B[m + 1].rand(False, seed=90); B[m + 2].rand(False, seed=91)
            n1 = B[m + 1].norm()                                   # anticipated (column m + 1 follows column m): m - 1 sweeps so far
            assert c.lazy_speculation_stats() == (m - 1, 0)
            n2 = B[m + 2].norm()                                   # the previous prediction went unused: disarmed, plain norm
            assert c.lazy_speculation_stats() == (m - 1, 1)
            ref1, ref2 = np.linalg.norm(B.download(m + 1, 1)), np.linalg.norm(B.download(m + 2, 1))
            assert abs(n1 - ref1) <= 1e-13 * ref1 and abs(n2 - ref2) <= 1e-13 * ref2
        out[spec] = H
        del B
        c.close()
    for j in range(m):
        assert np.abs(out[1][:, j] - out[0][:, j]).max() <= 1e-12 * np.abs(out[0][:, j]).max()


@pytest.mark.parametrize("dtype", KINDS)
def test_gmres_inner_cycle_as_one_engine_call_equals_the_step_by_step_one(ctx, dtype):
    """gmres' inner cycle through lk_arnoldi_segments (round 5: the kdim steps enqueued as one batch, the Givens rotations and the residual
    test run on the columns of H as they arrive, a converged residual stops the factorisation) against the one-round-trip-per-step loop:
    same info, same residual history, same solution -- converging inside a cycle (the steps the device ran beyond the stop are not used),
    after restarts, and not at all (maxiter exhausted); against the oracle's gmres as well."""
    from lightkrylov_amd import solvers
    n = 30_011
    rng = np.random.default_rng(4)
    d = (2.0 + np.arange(n) % 23 + rng.random(n)).astype(dtype)
    if np.dtype(dtype).kind == "c":
        d = d * np.exp(0.3j * rng.random(n))
    bvec = seeded(n, dtype, 17)
    out = {}
    for kdim, maxiter, rtol, tag in ((40, 5, 1e-10, "converges inside the first cycle"), (6, 30, 1e-10, "converges after restarts"),
                                     (5, 2, 1e-14, "does not converge")):
        for fused in (False, True):
            solvers._GMRES_FUSED = fused
            try:
                A = lk.diag_linop_gpu(d, ctx)
                x = lk.dense_vector_gpu(n, dtype, ctx); x.zero()
                meta = lk.gmres_dp_metadata()
                info = lk.gmres(A, lk.dense_vector_gpu.from_array(bvec, ctx), x, rtol=rtol, atol=0.0,
                                options=lk.gmres_dp_opts(kdim=kdim, maxiter=maxiter), meta=meta)
            finally:
                solvers._GMRES_FUSED = True
            out[fused] = (info, np.array(meta.res), x.to_array(), A.matvec_counter)
        (i0, r0, x0, c0), (i1, r1, x1, c1) = out[False], out[True]
        assert i0 == i1 and len(r0) == len(r1) and c0 == c1, tag
        assert np.abs(r1 - r0).max() <= 1e-13 * r0[0] and np.abs(x1 - x0).max() <= 1e-13 * np.abs(x0).max(), tag
        xo = np.zeros(n, dtype=dtype)
        info_o, res_o = ora.gmres(ora.DiagOp(d), bvec, xo, rtol=rtol, atol=0.0, kdim=kdim, maxiter=maxiter)
        assert info_o == i1 and np.abs(np.array(res_o) - r1).max() <= 1e-12 * res_o[0] and np.abs(xo - x1).max() <= 1e-12 * np.abs(xo).max(), tag
    assert out[True][0] < 0                                    # the last configuration ran out of iterations
