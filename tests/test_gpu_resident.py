"""Single-launch Gram-Schmidt step (csrc/lk_resident.hip.h) against the oracle and against the three-sweep schedule:
double_gram_schmidt_step (gram_schmidt.fypp:12-57) + qr_no_pivoting's norm and scale (qr.fypp:135-165) in ONE persistent
kernel for panels that fit the memory-side cache.  Tolerance: north_star's 1e-12, normwise per column."""
import numpy as np
import pytest

import lightkrylov_amd as lk
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, orthonormal_basis, seeded

pytestmark = pytest.mark.gpu
RTOL = 1e-12

# every size of test_gpu_parity.DGS_CASES the single launch takes (k <= 128), + the tile-shape boundaries of its launcher
# (16 columns per wave: WC = 1 | 2 | 4 | 8 at k = 16 | 32 | 64 | 128) and ragged / tiny row counts (fewer tiles than CUs)
CASES = [(1000, 1), (1001, 2), (999, 3), (4096, 15), (4097, 16), (4099, 17), (10_000, 31), (10_001, 33), (30_000, 64),
         (30_011, 100), (30_011, 127), (30_011, 128), (129, 64), (65, 8), (1, 1), (300_007, 32), (2, 1), (127, 5), (128, 16),
         (257, 32), (1025, 65), (175_003, 48), (600_001, 8), (400_003, 24)]


def fits_onchip(n, k, dtype, num_cu=256):
    """the launcher's rule (lk_engine.hip, dgs_resident_launch): some shape -- 16 / 8 / 4 columns per wave, 2 / 5 / 9 tiles per block --
    whose row tiles fit one block per CU (which of the fitting shapes runs does not matter here)"""
    rows = 1 if dtype is np.complex128 else 2
    for kc, rt in ((16, 2), (8, 5), (4, 9)):
        if k > kc * 8:
            continue
        wc = 1
        while wc < -(-k // kc):
            wc *= 2
        tile = (8 // wc) * 64 * rows
        if -(-n // tile) <= rt * num_cu:
            return True
    return False


@pytest.fixture()
def rctx(ctx):
    """the shared context with the single launch ON for everything it can take; restored afterwards"""
    ctx.set_tuning("resident", 1)
    ctx.set_tuning("resident_max_mb", 192)
    yield ctx
    ctx.set_tuning("resident", 1)
    ctx.set_tuning("resident_max_mb", 192)
    ctx.set_tuning("resident_spin_ms", 2000)
    ctx.set_tuning("resident_onchip", 1)
    ctx.set_tuning("resident_rev", 1)


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("n,k", CASES)
def test_single_launch_step_against_oracle_and_three_sweeps(rctx, dtype, n, k):
    ctx = rctx
    k = min(k, n)
    Q = orthonormal_basis(n, k, dtype, 3)
    y = seeded(n, dtype, 77)
    B = lk.krylov_basis_gpu(n, k + 1, dtype, ctx)
    B.upload(Q, 0)
    yo = y.copy()
    ho, info_o = ora.double_gram_schmidt_step(yo, Q)
    ynorm = np.linalg.norm(y)
    res = {}
    for route in (1, 0):
        ctx.set_tuning("resident", route)
        B.upload(y.reshape(-1, 1), k)
        beta = np.zeros(k, dtype=dtype)
        before = ctx.resident_stats()
        info = lk.double_gram_schmidt_step(B[k], B[:k], if_chk_orthonormal=False, beta=beta)
        after = ctx.resident_stats()
        assert after[0] - before[0] == route, "the route the knob asks for is the route that ran"
        assert after[1] == before[1]
        assert info == info_o
        yg = B.download(k, 1)[:, 0]
        assert np.abs(beta - ho).max() <= RTOL * ynorm
        assert np.abs(yg - yo).max() <= RTOL * ynorm
        if k < n:
            assert np.abs(Q.conj().T @ yg).max() <= 1e-13 * ynorm
        res[route] = (beta, yg)
    assert np.abs(res[1][0] - res[0][0]).max() <= 1e-13 * ynorm        # the two schedules agree to rounding
    assert np.abs(res[1][1] - res[0][1]).max() <= 1e-13 * ynorm


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("n,k", [(1000, 1), (999, 3), (4097, 16), (10_001, 33), (30_011, 128), (129, 64), (300_007, 32), (175_003, 48),
                                 (600_001, 8), (400_003, 24), (90_001, 100)])
def test_register_resident_and_cache_resident_kernels_agree(rctx, dtype, n, k):
    """dgs_onchip (the panel stays in registers) against dgs_resident (three walks through the caches) and the oracle; the stats say
    which one ran.  Every shape of the on-chip kernel is hit: 16 / 8 / 4 columns per wave x 1..8 wave-columns, ragged last tiles."""
    ctx = rctx
    Q = orthonormal_basis(n, k, dtype, 5)
    y = seeded(n, dtype, 9)
    yo = y.copy()
    ho, _ = ora.double_gram_schmidt_step(yo, Q)
    ynorm = np.linalg.norm(y)
    B = lk.krylov_basis_gpu(n, k + 1, dtype, ctx)
    B.upload(Q, 0)
    outs = []
    for onchip in (1, 0):
        ctx.set_tuning("resident_onchip", onchip)
        B.upload(y.reshape(-1, 1), k)
        beta = np.zeros(k, dtype=dtype)
        before = ctx.resident_stats()
        lk.double_gram_schmidt_step(B[k], B[:k], False, beta=beta)
        after = ctx.resident_stats()
        assert after[0] - before[0] == 1 and after[2] - before[2] == (onchip if fits_onchip(n, k, dtype) else 0)
        yg = B.download(k, 1)[:, 0]
        assert np.abs(beta - ho).max() <= RTOL * ynorm
        assert np.abs(yg - yo).max() <= RTOL * ynorm
        outs.append((beta, yg))
    assert np.abs(outs[0][0] - outs[1][0]).max() <= 1e-13 * ynorm
    assert np.abs(outs[0][1] - outs[1][1]).max() <= 1e-13 * ynorm
    # walking phase 2 forwards instead of backwards changes the order a block adds its tiles in: rounding only
    ctx.set_tuning("resident_rev", 0)
    B.upload(y.reshape(-1, 1), k)
    beta = np.zeros(k, dtype=dtype)
    lk.double_gram_schmidt_step(B[k], B[:k], False, beta=beta)
    assert np.abs(beta - outs[1][0]).max() <= 1e-13 * ynorm


def test_single_launch_results_are_reproducible_bit_for_bit(rctx):
    """fixed summation order at every level: two runs of the same step give the same bits (both kernels, both kinds)"""
    ctx = rctx
    for dtype in KINDS:
        n, k = 200_003, 40
        Q = orthonormal_basis(n, k, dtype, 5)
        y = seeded(n, dtype, 9)
        B = lk.krylov_basis_gpu(n, k + 1, dtype, ctx)
        B.upload(Q, 0)
        for onchip in (1, 0):
            ctx.set_tuning("resident_onchip", onchip)
            runs = []
            for _ in range(3):
                B.upload(y.reshape(-1, 1), k)
                beta = np.zeros(k, dtype=dtype)
                lk.double_gram_schmidt_step(B[k], B[:k], False, beta=beta)
                runs.append((beta, B.download(k, 1)[:, 0]))
            for r in runs[1:]:
                assert np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1])


def test_dispatch_boundary_follows_the_panel_size(rctx):
    ctx = rctx
    n, k = 100_000, 20                       # panel = 21 columns x 0.8 MB = 16.02 MB
    Q = orthonormal_basis(n, k, np.float64, 1)
    B = lk.krylov_basis_gpu(n, k + 1, np.float64, ctx)
    B.upload(Q, 0)
    y = seeded(n, np.float64, 2)
    for mb, want in ((17, 1), (16, 0), (0, 0), (192, 1)):
        ctx.set_tuning("resident_max_mb", mb)
        B.upload(y.reshape(-1, 1), k)
        before = ctx.resident_stats()[0]
        lk.double_gram_schmidt_step(B[k], B[:k], False)
        assert ctx.resident_stats()[0] - before == want
    # beyond 128 columns: always the three-sweep schedule
    n2, k2 = 5003, 129
    Q2 = orthonormal_basis(n2, k2, np.float64, 1)
    B2 = lk.krylov_basis_gpu(n2, k2 + 1, np.float64, ctx)
    B2.upload(Q2, 0)
    B2.upload(seeded(n2, np.float64, 2).reshape(-1, 1), k2)
    before = ctx.resident_stats()[0]
    lk.double_gram_schmidt_step(B2[k2], B2[:k2], False)
    assert ctx.resident_stats()[0] == before


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("n,m", [(1000, 8), (100_000, 64), (175_000, 128), (1_000_000, 32)])
def test_arnoldi_on_the_single_launch_against_oracle(rctx, dtype, n, m):
    ctx = rctx
    d = (1.0 + np.arange(n) / n).astype(dtype)
    if dtype is np.complex128:
        d = d * np.exp(0.3j * np.arange(n) / n)
    x0 = seeded(n, dtype, 7)
    x0 /= np.linalg.norm(x0)
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F")
    Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), dtype=dtype, order="F")
    info_o = ora.arnoldi(ora.DiagOp(d), Xo, Ho)
    Hs = {}
    for route in (1, 0):
        ctx.set_tuning("resident", route)
        X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
        X.upload(x0.reshape(-1, 1), 0)
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        before = ctx.resident_stats()
        info = lk.arnoldi(lk.diag_linop_gpu(d, ctx), X, H)
        after = ctx.resident_stats()
        assert info == info_o == 0
        # every step whose panel fits "resident_max_mb" (192 MB): the batched ones and the last, host-synchronous one
        fit = sum(1 for k in range(1, m + 1) if n * (k + 1) * np.dtype(dtype).itemsize <= 192 * 2**20)
        assert after[0] - before[0] == (fit if route else 0)
        for j in range(m):
            assert np.abs(H[:, j] - Ho[:, j]).max() <= RTOL * np.abs(Ho[:, j]).max()
        Xg = X.download()
        assert np.abs(Xg.conj().T @ Xg - np.eye(m + 1)).max() <= 1e-12
        Hs[route] = H
    assert np.abs(Hs[1] - Hs[0]).max() <= 1e-13 * np.abs(Ho).max()


def test_breakdown_stops_the_batch_on_the_single_launch(rctx):
    """an invariant subspace after 3 steps (arnoldi.fypp:58-71): info = 3, the basis beyond it untouched"""
    ctx = rctx
    n, m = 50_000, 10
    d = np.ones(n)
    d[: n // 3] = 2.0
    d[n // 3: 2 * n // 3] = 3.0                    # three distinct eigenvalues: the Krylov space has dimension 3
    x0 = seeded(n, np.float64, 4)
    x0 /= np.linalg.norm(x0)
    Xo = np.zeros((n, m + 1), order="F")
    Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), order="F")
    info_o = ora.arnoldi(ora.DiagOp(d), Xo, Ho, tol=1e-10)
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    X.upload(x0.reshape(-1, 1), 0)
    H = np.zeros((m + 1, m), order="F")
    before = ctx.resident_stats()[0]
    info = lk.arnoldi(lk.diag_linop_gpu(d, ctx), X, H, tol=1e-10)
    assert ctx.resident_stats()[0] > before
    assert info == info_o == 3
    assert np.abs(H[:4, :3] - Ho[:4, :3]).max() <= 1e-10
    assert np.abs(X.download()[:, 5:]).max() == 0.0


def test_a_launch_that_cannot_get_the_chip_gives_up_cleanly_and_the_step_still_runs():
    """resident_spin_ms = 0: the first grid-wide wait of the launch times out at once (what happens when another persistent
    kernel holds the CUs) -- nothing has been written, the three-sweep schedule redoes the step, the context stops trying."""
    n, m = 300_000, 12
    d = 1.0 + np.arange(n) / n
    x0 = seeded(n, np.float64, 7)
    x0 /= np.linalg.norm(x0)
    Xo = np.zeros((n, m + 1), order="F")
    Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), order="F")
    ora.arnoldi(ora.DiagOp(d), Xo, Ho)
    for sync_first in (False, True):
        ctx = lk.Context(device=0)
        try:
            ctx.set_tuning("resident_spin_ms", 0)
            X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
            X.upload(x0.reshape(-1, 1), 0)
            if sync_first:                                     # the host-synchronous entry gives up first
                Y = lk.krylov_basis_gpu(n, 2, np.float64, ctx)
                Y.upload(seeded(n, np.float64, 3).reshape(-1, 1), 1)
                Y.upload(x0.reshape(-1, 1), 0)
                yo = seeded(n, np.float64, 3)
                ho, _ = ora.double_gram_schmidt_step(yo, x0.reshape(-1, 1).copy(order="F"))
                beta = np.zeros(1)
                lk.double_gram_schmidt_step(Y[1], Y[:1], False, beta=beta)
                assert abs(beta[0] - ho[0]) <= RTOL * np.linalg.norm(yo) + 1e-12
                assert ctx.resident_stats()[:2] == (1, 1)
            H = np.zeros((m + 1, m), order="F")
            assert lk.arnoldi(lk.diag_linop_gpu(d, ctx), X, H) == 0
            st = ctx.resident_stats()
            assert st[1] == 1, st                               # gave up exactly once; never tried again
            for j in range(m):
                assert np.abs(H[:, j] - Ho[:, j]).max() <= RTOL * np.abs(Ho[:, j]).max()
            # and a fresh opt-in works again once the chip is free
            ctx.set_tuning("resident_spin_ms", 2000)
            ctx.set_tuning("resident", 1)
            X.upload(x0.reshape(-1, 1), 0)
            H2 = np.zeros((m + 1, m), order="F")
            assert lk.arnoldi(lk.diag_linop_gpu(d, ctx), X, H2) == 0
            assert ctx.resident_stats()[0] > st[0] and ctx.resident_stats()[1] == 1
            assert np.abs(H2 - H).max() <= 1e-13 * np.abs(H).max()
        finally:
            ctx.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_lanczos_and_bidiagonalization_on_the_single_launch_and_through_a_give_up(dtype):
    """lk_lanczos / lk_bidiag (lanczos.fypp:7-64, golub_kahan.fypp:7-64) take the single launch per step too; a launch that gives up
    (resident_spin_ms = 0) stops the batch, the step runs again on the three-sweep schedule: T and B against the oracle either way."""
    n, m = 60_001, 24
    d = (1.0 + np.arange(n) / n)
    x0 = seeded(n, dtype, 21)
    x0 /= np.linalg.norm(x0)
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F")
    Xo[:, 0] = x0
    To = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.lanczos(ora.DiagOp(d.astype(dtype)), Xo, To) == 0
    dz = d.astype(dtype) * (np.exp(0.4j * np.arange(n) / n) if dtype is np.complex128 else 1.0)
    Uo = np.zeros((n, m + 1), dtype=dtype, order="F")
    Uo[:, 0] = x0
    Vo = np.zeros((n, m + 1), dtype=dtype, order="F")
    Bo = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.bidiagonalization(ora.DiagOp(dz), ora.DiagOp(dz.conj()), Uo, Vo, Bo) == 0
    for spin in (2000, 0):
        ctx = lk.Context(device=0)
        try:
            ctx.set_tuning("resident_spin_ms", spin)
            X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
            X.upload(x0.reshape(-1, 1), 0)
            T = np.zeros((m + 1, m), dtype=dtype, order="F")
            assert lk.lanczos(lk.diag_linop_gpu(d.astype(dtype), ctx), X, T) == 0
            st = ctx.resident_stats()
            assert (st[1] == 0 and st[0] == m) if spin else st[1] == 1, st      # (the batch had enqueued every step before the host saw the give-up)
            for j in range(m):
                assert np.abs(T[:, j] - To[:, j]).max() <= RTOL * np.abs(To[:, j]).max()
            ctx.set_tuning("resident", 1)                         # (re-arm after the give-up)
            U = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
            U.upload(x0.reshape(-1, 1), 0)
            V = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
            B = np.zeros((m + 1, m), dtype=dtype, order="F")
            before = ctx.resident_stats()
            assert lk.bidiagonalization(lk.diag_linop_gpu(dz, ctx), U, V, B) == 0
            after = ctx.resident_stats()
            # 2 m half steps, the first right half has no basis to orthogonalise against
            assert (after[0] - before[0], after[1] - before[1]) == (2 * m - 1, 0 if spin else 1), (before, after)
            for j in range(m):
                assert np.abs(B[:, j] - Bo[:, j]).max() <= RTOL * np.abs(Bo[:, j]).max()
            Ud = U.download()
            assert np.abs(Ud.conj().T @ Ud - np.eye(m + 1)).max() <= 1e-12
        finally:
            ctx.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_a_thousand_single_launches_of_random_shapes_are_reproducible_and_agree_with_the_three_sweeps(rctx, dtype):
    """The hand-off protocol of the grid-wide sums under load: 150 random (rows, columns) shapes -- ragged tiles, fewer tiles than CUs, every
    kernel shape, uneven work per block --, each run three times back to back on the single launch and once on the three sweeps, inside an
    ASYNCHRONOUS Arnoldi batch and as host-synchronous steps.  A stale or torn granule would show as a run that is not bit-identical to its
    repetition (the sums are in a fixed order) or as a step that disagrees with the three-sweep result beyond rounding."""
    ctx = rctx
    rng = np.random.default_rng(2026)
    launches = 0
    for case in range(150):
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20_000), rng.integers(20_000, 400_000)]))
        k = int(min(n, rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 32, 33, 48, 64, 65, 100, 127, 128])))
        Q = orthonormal_basis(n, k, dtype, 1000 + case) if n * k <= 4_000_000 else None
        B = lk.krylov_basis_gpu(n, k + 1, dtype, ctx)
        if Q is not None:
            B.upload(Q, 0)
        else:
            for j in range(k):
                B[j].rand(True, seed=3000 + 131 * case + j)           # (large cases: merely normalised columns -- any X will do here)
        y = seeded(n, dtype, 5000 + case)
        runs = []
        for route in (1, 1, 1, 0):
            ctx.set_tuning("resident", route)
            B.upload(y.reshape(-1, 1), k)
            beta = np.zeros(k, dtype=dtype)
            norms = []
            lk.double_gram_schmidt_step(B[k], B[:k], False, beta, _normalize=True, _norms=norms)
            runs.append((beta, B.download(k, 1)[:, 0], np.array(norms)))
            launches += route
        for r in runs[1:3]:
            assert all(np.array_equal(a, b) for a, b in zip(r, runs[0])), (case, n, k)
        scale = np.linalg.norm(y)
        assert np.abs(runs[0][0] - runs[3][0]).max() <= 1e-13 * scale, (case, n, k)
        assert np.abs(runs[0][2] - runs[3][2]).max() <= 1e-13 * scale, (case, n, k)
        if runs[3][2][2] > 1e-8 * scale:                                # (the direction of a numerically zero remainder is noise on both schedules)
            assert np.abs(runs[0][1] - runs[3][1]).max() <= 1e-10 * max(1.0, scale / runs[3][2][2]), (case, n, k)
        del B
    # the same under back-to-back launches with nothing in between: asynchronous factorisations of several sizes, three times each
    for n, m in ((257, 40), (5_003, 128), (90_001, 100), (250_007, 64)):
        d = (1.0 + np.arange(n) / n).astype(dtype)
        Hs = []
        for rep in range(3):
            ctx.set_tuning("resident", 1)
            X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
            X[0].rand(True, seed=7)
            H = np.zeros((m + 1, m), dtype=dtype, order="F")
            assert lk.arnoldi(lk.diag_linop_gpu(d, ctx), X, H) == 0
            Hs.append(H)
            launches += m
        assert np.array_equal(Hs[0], Hs[1]) and np.array_equal(Hs[0], Hs[2]), (n, m)
    assert launches > 1000 and ctx.resident_stats()[1] == 0


@pytest.mark.parametrize("rearm", [True, False], ids=["rearmed_every_round", "pause_and_retry"])
def test_two_contexts_launching_persistent_kernels_at_once_never_hang_and_stay_correct(rearm):
    """Two contexts (two streams, two host threads) run single-launch factorisations at the same time: each kernel wants every CU, so the
    two can split the chip and wait for blocks that cannot become resident -- the situation the bounded first wait exists for.  With a
    20 ms bound (2 s by default) a launch that starves gives up, its step runs on the three sweeps, and every Hessenberg matrix is still
    the oracle's; nothing hangs.  `rearm`: the contexts re-arm the single launch before every factorisation so that the collision recurs
    as often as possible; without it the engine's own policy runs -- pause for 16 steps, doubling with every give-up, then try again --
    and single launches must have been enqueued AFTER a give-up (the pause ends) while give-ups stay rare."""
    import threading
    n, m, rounds = 300_000, 24, 25
    d = 1.0 + np.arange(n) / n
    x0 = seeded(n, np.float64, 7)
    x0 /= np.linalg.norm(x0)
    Xo = np.zeros((n, m + 1), order="F")
    Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), order="F")
    assert ora.arnoldi(ora.DiagOp(d), Xo, Ho) == 0
    errs, gave_up, launched = [], [0, 0], [0, 0]
    start = threading.Barrier(2)

    def worker(t):
        try:
            ctx = lk.Context(device=0, use_torch_stream=False)
            ctx.set_tuning("resident_spin_ms", 20)
            A = lk.diag_linop_gpu(d, ctx)
            X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
            start.wait(timeout=60)
            after_first_give_up = None
            for _ in range(rounds):
                if rearm:
                    ctx.set_tuning("resident", 1)
                elif after_first_give_up is None and ctx.resident_stats()[1] > 0:
                    after_first_give_up = ctx.resident_stats()[0]
                X.upload(x0.reshape(-1, 1), 0)
                H = np.zeros((m + 1, m), order="F")
                assert lk.arnoldi(A, X, H) == 0
                for j in range(m):
                    assert np.abs(H[:, j] - Ho[:, j]).max() <= RTOL * np.abs(Ho[:, j]).max(), (t, j)
            st = ctx.resident_stats()
            launched[t], gave_up[t] = st[0], st[1]
            if not rearm and after_first_give_up is not None:
                assert st[0] > after_first_give_up, "the single launch never came back after its pause"
                assert st[1] <= 8, st                         # 16 + 32 + ... steps of pause: a handful of give-ups in 600 steps at most
            ctx.close()
        except BaseException as exc:  # noqa: BLE001
            errs.append(exc)
            start.abort()

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not any(t.is_alive() for t in ts), "a worker is still running: a wait did not end"
    assert not errs, errs
    assert min(launched) > 0
    print(f"\n  two contexts at once ({'re-armed every round' if rearm else 'pause and retry'}): single launches {launched}, gave up {gave_up}")
