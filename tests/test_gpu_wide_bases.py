"""Bases of 129..512 (and more) columns: `kdim = (size(X) - p) / p` has no cap in the reference (src/Krylov/arnoldi.fypp:26).  The fused sweeps
on wide register tiles and lane-split blocks (one pass over X per sweep, 3k+4 columns per DGS), the asynchronous Arnoldi / Lanczos /
Golub-Kahan pipelines beyond 128 columns, the per-object lazy path and the pipelined eighs cycle there."""
import ctypes as C
import os

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, seeded, orthonormal_basis
from tests._tol import assert_close, assert_columns_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("n,k", [(20_011, 129), (20_011, 200), (16_384, 256), (9001, 257), (9001, 384), (8192, 512),
                                 (6007, 511), (600, 512), (131, 130), (4099, 513), (3001, 700), (2500, 1100)])
def test_wide_dgs_against_oracle_and_traffic(dtype, n, k):
    """double_gram_schmidt_step against 129..512 (and, as column panels of 512, up to 1100) basis columns vs the oracle:
    coefficients and vector normwise 1e-12, orthogonality 1e-13, and -- from the library's own byte accounting -- exactly three
    sweep launches per DGS for k <= 512 (the panel schedule it replaces took 3 + 2 (npanels - 1))."""
    c = lk.Context(device=0)
    Q = orthonormal_basis(n, k, dtype, 3)
    y = seeded(n, dtype, 77)
    B = lk.krylov_basis_gpu(n, k + 1, dtype, c)
    B.upload(Q, 0)
    B.upload(y.reshape(-1, 1), k)
    beta = np.zeros(k, dtype=dtype)
    c.profile_reset(); c.profile_enable(True)
    info = lk.double_gram_schmidt_step(B[k], B[:k], if_chk_orthonormal=False, beta=beta)
    c.sync()
    launches = [c.profile_get(f"dgs_sweep{i}")[0] for i in (1, 2, 3)]
    by = sum(c.profile_get(f"dgs_sweep{i}")[2] for i in (1, 2, 3))
    c.profile_enable(False)
    yo = y.copy()
    ho, info_o = ora.double_gram_schmidt_step(yo, Q)
    assert info == info_o
    ynorm = np.linalg.norm(y)
    assert np.abs(beta - ho).max() <= 1e-12 * ynorm
    yg = B.download(k, 1)[:, 0]
    assert np.abs(yg - yo).max() <= 1e-12 * ynorm
    assert np.abs(Q.conj().T @ yg).max() <= 1e-13 * ynorm
    if k <= 512:
        s = np.dtype(dtype).itemsize
        assert launches == [1, 1, 1]
        assert by == pytest.approx(s * n * (3 * k + 5))                 # priced on the algorithmic 3k+5 columns
    # single pass (orthogonalize_against_basis) on the same shapes
    B.upload(y.reshape(-1, 1), k)
    b1 = np.zeros(k, dtype=dtype)
    lk.orthogonalize_against_basis(B[k], B[:k], if_chk_orthonormal=False, beta=b1)
    y1 = y.copy()
    h1, _ = ora.orthogonalize_against_basis(y1, Q)
    assert np.abs(b1 - h1).max() <= 1e-12 * ynorm
    assert np.abs(B.download(k, 1)[:, 0] - y1).max() <= 1e-12 * ynorm
    del B
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_wide_sweep_knobs_are_result_invariant(dtype):
    """Store policy, y' recomputation and the sweep-1 kernel choice change no result bit on a wide basis either (the lane-split
    sweep 3 must re-form y' in exactly sweep 2's summation order)."""
    n, k = 7001, 300
    Q = orthonormal_basis(n, k, dtype, 5)
    y = seeded(n, dtype, 9)
    ref = None
    for knobs in (dict(), dict(store_policy=0), dict(store_policy=1), dict(recompute_update=0), dict(dot_colwise=0)):
        c = lk.Context(device=0)
        for kk, v in knobs.items():
            c.set_tuning(kk, v)
        B = lk.krylov_basis_gpu(n, k + 1, dtype, c)
        B.upload(Q, 0); B.upload(y.reshape(-1, 1), k)
        beta = np.zeros(k, dtype=dtype)
        lk.double_gram_schmidt_step(B[k], B[:k], if_chk_orthonormal=False, beta=beta)
        got = (beta.tobytes(), B.download(k, 1).tobytes())
        if "dot_colwise" in knobs or "recompute_update" in knobs:
            # another kernel for sweep 1 / a stored y': same results to rounding
            b0 = np.frombuffer(ref[0], dtype=dtype)
            assert np.abs(beta - b0).max() <= 1e-13 * np.linalg.norm(y)
        else:
            if ref is None:
                ref = got
            assert got == ref
        del B
        c.close()


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("k,wide_regs", [(140, 0), (200, 0), (200, 2), (256, 2), (257, 2), (300, 2), (384, 2), (300, 1)])
def test_sweep3_with_two_column_groups_per_wave_reforms_the_same_bits(dtype, k, wide_regs):
    """Round 4, "wide_s3": where sweep 2 of a DGS runs lane-split (two lane groups per wave), sweep 3 holds both column groups of a
    wave-column in ONE wave's registers on tiles twice as tall.  It must re-form y' = y - X h1 exactly as sweep 2 summed it (or the
    rounding of y' would escape the second projection), and it applies the second set in the same grouping too: the vector that
    comes out is bit-identical to the lane-split sweep 3's, the coefficients are untouched (sweeps 1 and 2 do not change)."""
    n = 5003
    Q = orthonormal_basis(n, k, dtype, 11)
    y = seeded(n, dtype, 12)
    out = []
    for s3 in (0, 1):
        c = lk.Context(device=0)
        c.set_tuning("wide_regs", wide_regs)
        c.set_tuning("wide_s3", s3)
        B = lk.krylov_basis_gpu(n, k + 1, dtype, c)
        B.upload(Q, 0); B.upload(y.reshape(-1, 1), k)
        beta = np.zeros(k, dtype=dtype)
        assert lk.double_gram_schmidt_step(B[k], B[:k], if_chk_orthonormal=False, beta=beta) == 0
        out.append((beta.copy(), B.download(k, 1)[:, 0].copy()))
        del B
        c.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1]), f"max difference {np.abs(out[0][1] - out[1][1]).max():.2e}"
    yo = y.copy()
    ho, _ = ora.double_gram_schmidt_step(yo, Q)
    assert np.abs(out[1][0] - ho).max() <= 1e-12 * np.linalg.norm(y) and np.abs(out[1][1] - yo).max() <= 1e-12 * np.linalg.norm(y)


@pytest.mark.parametrize("dtype", KINDS)
def test_arnoldi_with_256_basis_columns_is_one_asynchronous_batch(dtype):
    """kdim = 256: every step runs as three fused sweeps inside ONE asynchronous batch (one host synchronisation per call);
    H against the oracle column by column, orthonormality, and bit-identity with the one-round-trip-per-step schedule."""
    n, m = 12_007, 256
    g = np.arange(n) / n
    d = (1.0 + g).astype(dtype) if np.dtype(dtype).kind == "f" else ((1.0 + g) * np.exp(1j * g)).astype(dtype)
    x0 = seeded(n, dtype, 7); x0 /= np.linalg.norm(x0)
    out = {}
    for mode in (1, 0):
        c = lk.Context(device=0)
        c.set_tuning("async_arnoldi", mode)
        X = lk.krylov_basis_gpu(n, m + 1, dtype, c); X.upload(x0.reshape(-1, 1), 0)
        H = np.zeros((m + 1, m), dtype=dtype, order="F")
        c.profile_reset(); c.profile_enable(True)
        assert lk.arnoldi(lk.diag_linop_gpu(d, c), X, H) == 0
        c.sync()
        cnt = [c.profile_get(f"dgs_sweep{i}")[0] for i in (1, 2, 3)]
        single = c.profile_get("dgs_sweep_resident")[0]
        c.profile_enable(False)
        # one persistent launch per step while the basis has <= 128 columns (this panel fits the caches), three sweeps per step beyond,
        # whatever the width
        assert single == 128 and cnt == [m - single] * 3
        out[mode] = (H.copy(), X.download())
        if mode == 1:
            G = lk.Gram(X[:128]); G2 = lk.innerprod(X[:128], X[128:m + 1])
            assert np.abs(G - np.eye(128)).max() <= 1e-12 and np.abs(G2).max() <= 1e-12
        del X
        c.close()
    assert out[0][0].tobytes() == out[1][0].tobytes() and out[0][1].tobytes() == out[1][1].tobytes()
    H = out[1][0]
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F"); Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.arnoldi(ora.DiagOp(d), Xo, Ho) == 0
    for j in range(m):
        assert np.abs(H[:, j] - Ho[:, j]).max() <= 1e-12 * np.abs(Ho[:, j]).max()


def test_arnoldi_breakdown_beyond_128_columns_leaves_the_rest_untouched():
    """Invariant subspace at step 150 of a 200-step call: info = 150 and the columns beyond stay as they were (arnoldi.fypp:58-71)
    -- the device-side stop flag of the asynchronous batch on the wide kernels.  Operator: diag(w^i), w = exp(2 pi i / 150), and
    a constant start vector with n = 60 * 150 rows: the Krylov vectors A^k x0 are columns of a DFT, i.e. orthogonal (every
    sub-diagonal entry is 1 to rounding) and A^150 x0 = x0 (a clean breakdown, unlike clustered real spectra whose Krylov
    basis is so ill-conditioned that the breakdown residual never falls below any useful tolerance)."""
    c = lk.Context(device=0)
    r, m = 150, 200
    n = 60 * r
    d = np.exp(2j * np.pi * (np.arange(n) % r) / r)
    X = lk.krylov_basis_gpu(n, m + 1, np.complex128, c)
    X.upload((np.ones(n, dtype=np.complex128) / np.sqrt(n)).reshape(-1, 1), 0)
    marker = seeded(n, np.complex128, 123)
    for j in range(r + 1, m + 1):
        X.upload(marker.reshape(-1, 1), j)
    H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    info = lk.arnoldi(lk.diag_linop_gpu(d, c), X, H, tol=1e-10)
    assert info == r
    assert np.abs(np.abs(np.diag(H, -1)[:r - 1]) - 1.0).max() <= 1e-12 and abs(H[r, r - 1]) < 1e-10
    assert np.array_equal(X.download(m, 1)[:, 0], marker) and np.array_equal(X.download(r + 1, 1)[:, 0], marker)
    G = lk.Gram(X[:r])
    assert np.abs(G - np.eye(r)).max() <= 1e-12
    del X
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_lanczos_and_bidiagonalization_beyond_128_columns(ctx, dtype):
    """lk_lanczos / lk_bidiag with kend = 200: the whole call is one asynchronous batch; T and B against the oracle."""
    n, m = 6007, 200
    d = (1.0 + np.arange(n) / n).astype(dtype)
    x0 = seeded(n, dtype, 21); x0 /= np.linalg.norm(x0)
    X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X.upload(x0.reshape(-1, 1), 0)
    T = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.lanczos(lk.diag_linop_gpu(d, ctx), X, T) == 0
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F"); Xo[:, 0] = x0
    To = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.lanczos(ora.DiagOp(d), Xo, To) == 0
    for j in range(m):
        assert np.abs(T[:, j] - To[:, j]).max() <= 1e-12 * np.abs(To[:, j]).max()
    # Golub-Kahan on a non-normal diagonal-times-shift operator is not available among the engine operators; the diagonal one
    # (complex: non-Hermitian) exercises both bases
    g = np.arange(n) / n
    dz = (1.0 + g).astype(dtype) if np.dtype(dtype).kind == "f" else ((1.0 + g) * np.exp(1j * g)).astype(dtype)
    A = lk.diag_linop_gpu(dz, ctx)
    U = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); U.upload(x0.reshape(-1, 1), 0)
    V = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
    B = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.bidiagonalization(A, U, V, B) == 0
    Uo = np.zeros((n, m + 1), dtype=dtype, order="F"); Uo[:, 0] = x0
    Vo = np.zeros((n, m + 1), dtype=dtype, order="F")
    Bo = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.bidiagonalization(ora.DiagOp(dz), ora.DiagOp(dz.conj()), Uo, Vo, Bo) == 0
    for j in range(m):
        assert np.abs(B[:, j] - Bo[:, j]).max() <= 1e-12 * np.abs(Bo[:, j]).max()


def test_arnoldi_beyond_the_fused_width_and_restarted_ranges(ctx):
    """kdim = 140 > 128: steps 1..128 run as one asynchronous batch, the rest through the wide (unfused) schedule; and a
    factorisation continued with kstart > 1 (what krylov_schur restarts do) equals the one-shot run.  H against the oracle."""
    n, m = 20_011, 140
    d = 1.0 + np.arange(n) / n
    x0 = seeded(n, np.float64, 7); x0 /= np.linalg.norm(x0)
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx); X.upload(x0.reshape(-1, 1), 0)
    H = np.zeros((m + 1, m), order="F")
    A = lk.diag_linop_gpu(d, ctx)
    assert lk.arnoldi(A, X, H) == 0
    Xo = np.zeros((n, m + 1), order="F"); Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), order="F")
    assert ora.arnoldi(ora.DiagOp(d), Xo, Ho) == 0
    for j in range(m):
        assert np.abs(H[:, j] - Ho[:, j]).max() <= 1e-12 * np.abs(Ho[:, j]).max()
    G = lk.Gram(X[:m + 1])
    assert np.abs(G - np.eye(m + 1)).max() <= 1e-12
    # the same factorisation in three pieces
    X2 = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx); X2.upload(x0.reshape(-1, 1), 0)
    H2 = np.zeros((m + 1, m), order="F")
    assert lk.arnoldi(A, X2, H2, kstart=1, kend=50) == 0
    assert lk.arnoldi(A, X2, H2, kstart=51, kend=51) == 0            # a single step takes the synchronous path
    assert lk.arnoldi(A, X2, H2, kstart=52, kend=m) == 0
    assert H2.tobytes() == H.tobytes() and X2.download().tobytes() == X.download().tobytes()


def test_per_object_arnoldi_in_lazy_mode_beyond_128_columns(ctx):
    """The per-object (type-bound-procedure) schedule an unchanged LightKrylov drives, lazy mode, kdim = 200: the batched dot
    sweeps and the fused update + dot sweeps cover up to 512 columns each (one sweep per Gram-Schmidt pass at every step, not
    one per 128 columns); H equals the fused lk_arnoldi factorisation to 1e-12 per column."""
    n, m = 20_011, 200
    d = 1.0 + np.arange(n) / n
    c = lk.Context(device=0)
    c.set_tuning("lazy", 1)
    A = lk.diag_linop_gpu(d, c)

    class pyop(lk.abstract_linop):                            # python operator => the python (reference) step loop
        def matvec(self, vi, vo): A.matvec(vi, vo)
    B = lk.krylov_basis_gpu(n, m + 1, np.float64, c)
    B[0].rand(True, seed=7)
    X = [B[j] for j in range(m + 1)]
    H = np.zeros((m + 1, m), order="F")
    assert lk.arnoldi(pyop(), X, H) == 0
    fs, ls = c.lazy_fusion_stats(), c.lazy_stats()
    assert fs[0] == 2 * m and fs[1] == 0 and fs[3] == 0       # two fused update + dot sweeps per step, nothing materialised
    assert ls[1] == m - 1                                     # one batched dot sweep per step from the second on (one column is a plain dot)
    # from the third step on the norm that opens the first pass runs that sweep itself (one kernel + one synchronisation less)
    assert c.lazy_speculation_stats() == (m - 2, 0)
    X1 = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    X1[0].rand(True, seed=7)
    H1 = np.zeros((m + 1, m), order="F")
    assert lk.arnoldi(lk.diag_linop_gpu(d, ctx), X1, H1) == 0
    for j in range(m):
        assert np.abs(H[:, j] - H1[:, j]).max() <= 1e-12 * np.abs(H1[:, j]).max()
    del B
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_pipelined_eighs_beyond_128_lanczos_steps(ctx, dtype):
    """eighs with kdim = 160: the Lanczos steps beyond 128 basis columns run in the asynchronous device segments too (lk_lanczos
    takes up to 512 columns now), bit-identical to the reference's alternation of one step and one eigh."""
    n, nev, kdim = 12_007, 3, 160
    d = np.linspace(1.0, 2.0, n).astype(dtype)                      # a dense spectrum: no Ritz pair converges to 1e-14 in 160 steps
    x0 = seeded(n, dtype, 9)
    out = []
    for pipe in (False, True):
        X = lk.krylov_basis_gpu(n, nev, dtype, ctx)
        vals, res, info = lk.eighs(lk.diag_linop_gpu(d, ctx), X, x0=lk.dense_vector_gpu.from_array(x0, ctx), kdim=kdim,
                                   tolerance=1e-14, pipelined=pipe)
        out.append((vals, res, info, X.download()))
    (v0, r0, i0, X0), (v1, r1, i1, X1) = out
    assert i0 == i1 == kdim
    assert np.array_equal(v0, v1) and np.array_equal(r0, r1) and np.array_equal(X0, X1)
    assert (np.diff(v0) <= 0).all() and 1.99 < v0[0] <= 2.0 + 1e-12


@pytest.mark.parametrize("dtype", KINDS)
def test_lanczos_and_bidiagonalization_beyond_512_columns_through_the_c_entries(ctx, dtype):
    """lk_lanczos / lk_bidiag have no cap on the basis any more (round 5; the reference has none, lanczos.fypp:20, golub_kahan.fypp:18):
    kend = 530 in ONE call -- the first 512 steps as the asynchronous batch, the rest one host round trip each -- and a restart range that
    starts beyond 512; T and B against the oracle column by column, a breakdown beyond 512 reported like one below."""
    n, m = 1500, 530
    rng = np.random.default_rng(8)
    d = (1.0 + rng.random(n) * 3.0).astype(dtype)                           # (a spread spectrum: 530 steps without a breakdown)
    x0 = seeded(n, dtype, 21); x0 /= np.linalg.norm(x0)
    A = lk.diag_linop_gpu(d, ctx)
    X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X.upload(x0.reshape(-1, 1), 0)
    T = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.lanczos(A, X, T) == 0
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F"); Xo[:, 0] = x0
    To = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.lanczos(ora.DiagOp(d), Xo, To) == 0
    for j in range(m):
        assert np.abs(T[:, j] - To[:, j]).max() <= 1e-12 * np.abs(To[:, j]).max(), j
    Xg = X.download()
    assert np.abs(Xg.conj().T @ Xg - np.eye(m + 1)).max() <= 1e-12
    # the same factorisation in two calls, the second one starting beyond 512 columns
    X2 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X2.upload(x0.reshape(-1, 1), 0)
    T2 = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.lanczos(A, X2, T2, kstart=1, kend=520) == 0 and lk.lanczos(A, X2, T2, kstart=521, kend=m) == 0
    assert np.array_equal(T2, T) and np.array_equal(X2.download(), Xg)
    # Golub-Kahan, both bases beyond 512 columns
    g = np.arange(n) / n
    dz = (1.0 + 3.0 * rng.random(n)).astype(dtype) if np.dtype(dtype).kind == "f" else ((1.0 + 3.0 * rng.random(n)) * np.exp(1j * g)).astype(dtype)
    Az = lk.diag_linop_gpu(dz, ctx)
    U = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); U.upload(x0.reshape(-1, 1), 0)
    V = lk.krylov_basis_gpu(n, m + 1, dtype, ctx)
    B = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.bidiagonalization(Az, U, V, B) == 0
    Uo = np.zeros((n, m + 1), dtype=dtype, order="F"); Uo[:, 0] = x0
    Vo = np.zeros((n, m + 1), dtype=dtype, order="F")
    Bo = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.bidiagonalization(ora.DiagOp(dz), ora.DiagOp(dz.conj()), Uo, Vo, Bo) == 0
    for j in range(m):
        assert np.abs(B[:, j] - Bo[:, j]).max() <= 1e-12 * np.abs(Bo[:, j]).max(), j
    # a stop beyond 512 columns (the tolerance is set above the next beta: info = that step, T(k+1, k) = beta, the vector left unscaled, :32-36)
    X3 = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X3.upload(Xg[:, :521], 0)
    T3 = T.copy(order="F")
    T3[:, 520:] = 0
    info = lk.lanczos(A, X3, T3, kstart=521, kend=m, tol=10.0)
    assert info == 521 and abs(T3[521, 520] - T[521, 520]) <= 1e-12 * abs(T[521, 520]) and not T3[:, 521:].any()
    y = X3.download(521, 1)[:, 0]
    assert abs(np.linalg.norm(y) - abs(T[521, 520])) <= 1e-12 * abs(T[521, 520])           # not normalised
