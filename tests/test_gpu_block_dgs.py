"""Block Gram-Schmidt, innerprod_matrix and Gram with several right-hand sides (all through the
C ABI): DGS_basis_against_basis (src/Krylov/gram_schmidt.fypp:59-105) as a panel x panel schedule -- up to four columns of Y per pass
over X on the VALU kernels, five or more (and every block against more than 128 basis columns) on the FP64 matrix cores, three / four
passes per group of 32, column panels of X beyond 128 columns (round 5) --, X^H Y / X^H X on the matrix cores (AbstractVectors.fypp:645-695)
with the complex kind's three-product kernels, and the block Arnoldi built on them (src/Krylov/arnoldi.fypp:34-56)."""
import ctypes as C
import os

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, seeded, basis, orthonormal_basis
from tests._tol import assert_close, assert_columns_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("n,k,p", [(5003, 3, 3), (5003, 64, 4), (4099, 65, 4), (3001, 100, 7), (2500, 128, 4), (130, 17, 5),
                                   (70_001, 128, 3)])
def test_block_dgs_four_columns_per_pass(ctx, dtype, n, k, p):
    """DGS_basis_against_basis (gram_schmidt.fypp:59-105) and innerprod_matrix with the multi-right-hand-side dot sweep
    (up to 4 columns of Y per pass, column panels of 64 beyond k = 64): coefficients and vectors against the oracle's
    per-column double_gram_schmidt_step, innerprod against one dot per entry."""
    Q, _ = np.linalg.qr(basis(n, k, dtype, 70))
    Q = np.asfortranarray(Q)
    Y = basis(n, p, dtype, 300)
    B = lk.krylov_basis_gpu(n, k + p, dtype, ctx)
    B.upload(Q, 0); B.upload(Y, k)
    M = lk.innerprod(B[:k], B[k:k + p])
    Mo = ora.innerprod(Q, Y)
    assert np.abs(M - Mo).max() <= 1e-12 * np.linalg.norm(Y, axis=0).max()
    beta = np.zeros((k, p), dtype=dtype, order="F")
    assert lk.double_gram_schmidt_step(B[k:k + p], B[:k], False, beta) == 0
    Yg = B.download(k, p)
    for j in range(p):
        yo = Y[:, j].copy()
        ho, _ = ora.double_gram_schmidt_step(yo, Q)
        assert np.abs(beta[:, j] - ho).max() <= 1e-12 * np.linalg.norm(Y[:, j])
        assert np.abs(Yg[:, j] - yo).max() <= 1e-12 * np.linalg.norm(Y[:, j])
    assert np.abs(Q.conj().T @ Yg).max() <= 1e-12 * np.linalg.norm(Y, axis=0).max()


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("n,k,p", [(1, 1, 5), (33, 5, 5), (4099, 17, 6), (5003, 128, 16), (3001, 100, 33), (2051, 64, 21),
                                   (777, 130, 9), (70_001, 128, 32), (1500, 31, 129)])
def test_many_right_hand_sides_on_the_matrix_cores(dtype, mfma, n, k, p):
    """innerprod_matrix, Gram (AbstractVectors.fypp:645-695) and DGS_basis_against_basis (gram_schmidt.fypp:59-105) with
    5+ right-hand sides: X^H Y by panel_xhy_mfma (one pass over X per 128 x 128 block; `xhy_mfma` = 1) and by the VALU
    schedule (4 right-hand sides per pass; = 0), both against the oracle's one-dot-per-entry / per-column restatement at 1e-12
    normwise.  Ragged shapes: k, p not multiples of 16, beyond 128, odd and tiny n (k > n makes X rank deficient: innerprod
    and Gram only)."""
    c = lk.Context(device=0)
    c.set_tuning("xhy_mfma", mfma)
    try:
        X = np.asfortranarray(np.linalg.qr(basis(n, k, dtype, 70))[0]) if n >= k else basis(n, k, dtype, 70)
        Y = basis(n, p, dtype, 300)
        B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(X)
        Z = lk.krylov_basis_gpu(n, p, dtype, c); Z.upload(Y)
        ny = np.linalg.norm(Y, axis=0).max() * max(1.0, np.linalg.norm(X, axis=0).max())
        M = lk.innerprod(B, Z)
        assert np.abs(M - ora.innerprod(X, Y)).max() <= 1e-12 * ny
        G = lk.Gram(B)
        assert np.abs(G - ora.gram(X)).max() <= 1e-12 * max(1.0, np.linalg.norm(X, axis=0).max() ** 2)
        if n > k and k <= 128:                                   # (n <= k: nothing is left of Y after the projection)
            beta = np.zeros((k, p), dtype=dtype, order="F")
            assert lk.double_gram_schmidt_step(Z, B, False, beta) == 0
            Yg = Z.download()
            for j in range(p):
                yo = Y[:, j].copy()
                ho, _ = ora.double_gram_schmidt_step(yo, X)
                assert np.abs(beta[:, j] - ho).max() <= 1e-12 * np.linalg.norm(Y[:, j])
                assert np.abs(Yg[:, j] - yo).max() <= 1e-12 * np.linalg.norm(Y[:, j])
        del B, Z
    finally:
        c.close()


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("k,p", [(7, 2), (40, 2), (64, 4), (128, 2), (100, 3), (33, 5)])
def test_block_dgs_fused_schedule_equals_the_four_pass_one(dtype, k, p):
    """DGS_basis_against_basis (gram_schmidt.fypp:59-105) through lk_dgs_block: the fused three-pass schedule
    (panel_sweep_p) and the four-pass one (dots / update / dots / update) return the same coefficients and leave the same
    vectors to 1e-12, and both match the oracle's per-column double Gram-Schmidt; ragged rows, odd group sizes."""
    n = 7001
    Q = np.asfortranarray(np.linalg.qr(np.column_stack([seeded(n, dtype, 5 + j) for j in range(k)]))[0])
    Y = np.asfortranarray(np.column_stack([seeded(n, dtype, 200 + j) for j in range(p)]))
    out = []
    for fused in (0, 1):
        c = lk.Context(device=0)
        c.set_tuning("block_fused", fused)
        B = lk.krylov_basis_gpu(n, k + p, dtype, c)
        B.upload(Q, 0); B.upload(Y, k)
        beta = np.zeros((k, p), dtype=dtype, order="F")
        info = lk.double_gram_schmidt_step(B[k:k + p], B[:k], False, beta)
        out.append((info, beta.copy(), B.download(k, p)))
        del B
        c.close()
    (i0, b0, y0), (i1, b1, y1) = out
    assert i0 == i1 == 0
    scale = max(np.linalg.norm(Y[:, j]) for j in range(p))
    assert np.abs(b0 - b1).max() <= 1e-12 * scale and np.abs(y0 - y1).max() <= 1e-12 * scale
    for j in range(p):
        yo = Y[:, j].copy()
        ho, _ = ora.double_gram_schmidt_step(yo, Q)
        assert np.abs(b1[:, j] - ho).max() <= 1e-12 * scale and np.abs(y1[:, j] - yo).max() <= 1e-12 * scale


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("n,k,p", [(70_001, 128, 32), (5003, 100, 17), (4099, 17, 6), (3001, 64, 8), (2051, 33, 5), (777, 128, 33),
                                   (63, 16, 32), (20_000, 96, 70)])
def test_block_dgs_three_pass_schedule_on_the_matrix_cores(dtype, fused, n, k, p):
    """DGS_basis_against_basis (gram_schmidt.fypp:59-105) with >= 5 right-hand sides: H1 = X^H Y | Y' = Y - X H1 AND H2 = X^H Y' in
    one fused pass (panel_xhy_upd_mfma) | Y'' = Y' - X H2 -- three passes over X per group of 32 columns (`block_fused` = 1: real
    kind, 2: both kinds) against the four-pass schedule (0); every column against the oracle's double Gram-Schmidt."""
    c = lk.Context(device=0)
    c.set_tuning("block_fused", fused)
    Q = orthonormal_basis(n, k, dtype, 5)
    Y = basis(n, p, dtype, 200)
    B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(Q)
    Z = lk.krylov_basis_gpu(n, p, dtype, c); Z.upload(Y)
    beta = np.zeros((k, p), dtype=dtype, order="F")
    c.profile_reset(); c.profile_enable(True)
    assert lk.double_gram_schmidt_step(Z, B, False, beta) == 0
    c.sync()
    n_fused = c.profile_get("xhy_upd_mfma")[0]
    c.profile_enable(False)
    groups = (p + 31) // 32
    assert n_fused == (groups if (fused == 2 or (fused == 1 and np.dtype(dtype).kind == "f")) else 0)
    Yg = Z.download()
    for j in range(p):
        yo = Y[:, j].copy()
        ho, _ = ora.double_gram_schmidt_step(yo, Q)
        assert np.abs(beta[:, j] - ho).max() <= 1e-12 * np.linalg.norm(Y[:, j])
        assert np.abs(Yg[:, j] - yo).max() <= 1e-12 * np.linalg.norm(Y[:, j])
    assert np.abs(Q.conj().T @ Yg).max() <= 1e-13 * np.linalg.norm(Y, axis=0).max()
    del B, Z
    c.close()


@pytest.mark.parametrize("n,k,p", [(1_300_003, 128, 32), (700_001, 128, 17), (400_000, 120, 24), (300_017, 100, 32), (200_003, 64, 32), (90_001, 40, 20), (60_000, 16, 32),
                                   (50_011, 7, 18), (65, 33, 32), (33, 5, 17), (4099, 128, 32)])
def test_block_dgs_fused_pass_with_row_owner_waves(n, k, p):
    """DGS_basis_against_basis (gram_schmidt.fypp:59-105), real kind, 17..32 right-hand sides: the fused pass by panel_xhy_upd_rs ("upd_rs" = 1: a team of four waves per
    tile, each wave the owner of 16 rows x 16 right-hand sides, coefficients in registers, Y' handed from the update's accumulators straight to the dot products, tiles by
    LDS-DMA into two stages per team, one barrier per iteration) on panels of MANY tiles per block -- both teams, both stages, an odd number of tiles (one team idle in the
    last iteration), a ragged last tile or none, basis widths that are not a multiple of 16 (the last column block partly beyond the panel), fewer than 32 right-hand
    sides (the last ones beyond the panel), panels shorter than two tiles -- every column against the oracle, against the kernel behind "upd_rs" = 0, bit-identical from
    call to call."""
    c = lk.Context(device=0)
    Q = orthonormal_basis(n, k, np.float64, 5)
    Y = basis(n, p, np.float64, 200)
    B = lk.krylov_basis_gpu(n, k, np.float64, c); B.upload(Q)
    out = {}
    for rs in (1, 0, 1):
        c.set_tuning("upd_rs", rs)
        Z = lk.krylov_basis_gpu(n, p, np.float64, c); Z.upload(Y)
        beta = np.zeros((k, p), order="F")
        c.profile_reset(); c.profile_enable(True)
        assert lk.double_gram_schmidt_step(Z, B, False, beta) == 0
        c.sync()
        assert c.profile_get("xhy_upd_mfma")[0] == 1
        c.profile_enable(False)
        Yg = Z.download()
        if rs in out:
            assert np.array_equal(out[rs][0], beta) and np.array_equal(out[rs][1], Yg)       # fixed order of the sums
        out[rs] = (beta, Yg)
        del Z
    beta, Yg = out[1]
    cols = range(p) if n <= 400_000 else (0, 15, 16, p - 1)                                  # (the scalar oracle takes a while on the long panels)
    for j in cols:
        yo = Y[:, j].copy()
        ho, _ = ora.double_gram_schmidt_step(yo, Q)
        assert np.abs(beta[:, j] - ho).max() <= 1e-12 * np.linalg.norm(Y[:, j])
        assert np.abs(Yg[:, j] - yo).max() <= 1e-12 * np.linalg.norm(Y[:, j])
    assert np.abs(out[1][0] - out[0][0]).max() <= 1e-13 * np.linalg.norm(Y, axis=0).max()
    assert np.abs(out[1][1] - out[0][1]).max() <= 1e-13 * np.linalg.norm(Y, axis=0).max()
    assert np.abs(Q.T @ Yg).max() <= 1e-13 * np.linalg.norm(Y, axis=0).max()
    del B
    c.close()


def test_block_dgs_fused_pass_with_row_owner_waves_many_times_over_for_races():
    """panel_xhy_upd_rs orders its LDS-DMA stages by a vmcnt(0) wait and one raw barrier per iteration: an early read or an early overwrite of a stage would show as a
    rare wrong tile, so the same block step is taken 100 times from the same Y and must give the same bits every time."""
    c = lk.Context(device=0)
    n, k, p = 600_011, 128, 32
    Q = orthonormal_basis(n, k, np.float64, 9)
    Y = basis(n, p, np.float64, 300)
    B = lk.krylov_basis_gpu(n, k, np.float64, c); B.upload(Q)
    Z = lk.krylov_basis_gpu(n, p, np.float64, c)
    first = None
    for _ in range(100):
        Z.upload(Y)
        beta = np.zeros((k, p), order="F")
        assert lk.double_gram_schmidt_step(Z, B, False, beta) == 0
        got = (beta, Z.download())
        if first is None:
            first = got
            assert np.abs(Q.T @ got[1]).max() <= 1e-13 * np.linalg.norm(Y, axis=0).max()
        else:
            assert np.array_equal(first[0], got[0]) and np.array_equal(first[1], got[1])
    c.close()


@pytest.mark.parametrize("n,k,p", [(4099, 7, 5), (20_003, 64, 16), (9001, 128, 32), (7001, 100, 17), (255, 128, 32), (12_289, 33, 31)])
def test_complex_innerprod_with_three_real_products_per_complex_one(n, k, p):
    """Round 4: X^H Y with <= 32 right-hand sides, complex kind, on separate real / imaginary planes with P1 = Xr^T Yr, P2 = Xi^T Yi,
    P3 = (Xr + Xi)^T (Yi - Yr), Re = P1 + P2, Im = P3 + P1 - P2 (conj on X as in dotc, AbstractVectors.fypp:550) -- against numpy and
    against the four-product kernel, normwise with the scale of the cancelling terms; the block Gram-Schmidt built on it against the oracle."""
    dtype = np.complex128
    X, Y = basis(n, k, dtype, 41), basis(n, p, dtype, 800)
    ref = X.conj().T @ Y
    scale = (np.abs(X.real) + np.abs(X.imag)).T @ (np.abs(Y.real) + np.abs(Y.imag))
    out = []
    for three in (1, 0):
        c = lk.Context(device=0)
        c.set_tuning("gemm_3m", three)
        Bx = lk.krylov_basis_gpu(n, k, dtype, c); Bx.upload(X)
        By = lk.krylov_basis_gpu(n, p, dtype, c); By.upload(Y)
        out.append(np.array(lk.innerprod(Bx, By)))
        if three:
            Q = orthonormal_basis(n, k, dtype, 43) if n > k else None
            if Q is not None:
                Bx.upload(Q)
                beta = np.zeros((k, p), dtype=dtype, order="F")
                assert lk.double_gram_schmidt_step(By, Bx, if_chk_orthonormal=False, beta=beta) == 0
                Yo = Y.copy(order="F")
                ho = np.zeros((k, p), dtype=dtype, order="F")
                for j in range(p):
                    yj = np.ascontiguousarray(Yo[:, j])
                    ho[:, j], _ = ora.double_gram_schmidt_step(yj, Q)
                    Yo[:, j] = yj
                ynorm = np.linalg.norm(Y, axis=0).max()
                assert np.abs(beta - ho).max() <= 1e-12 * ynorm and np.abs(By.download() - Yo).max() <= 1e-12 * ynorm
        del Bx, By
        c.close()
    assert (np.abs(out[0] - ref) <= 1e-14 * scale).all()
    assert (np.abs(out[0] - out[1]) <= 1e-14 * scale).all()


@pytest.mark.parametrize("n,k", [(4099, 33), (20_003, 64), (9001, 100), (12_289, 128), (255, 128), (31, 48)])
def test_complex_gram_matrix_with_three_real_products_per_complex_one(n, k):
    """Round 4: Gram (AbstractVectors.fypp:645-657) of a complex basis beyond 32 columns -- upper tiles dealt to the waves, P1 = Xr^T Xr,
    P2 = Xi^T Xi, P3 = (Xr + Xi)^T (Xi - Xr), Re = P1 + P2, Im = P3 + P1 - P2 -- against numpy (upper triangle; the reference mirrors it
    WITHOUT conjugation) and against the four-product kernel, with the scale of the cancelling terms."""
    dtype = np.complex128
    X = basis(n, k, dtype, 51)
    ref = X.conj().T @ X
    scale = (np.abs(X.real) + np.abs(X.imag)).T @ (np.abs(X.real) + np.abs(X.imag))
    out = []
    for three in (1, 0):
        c = lk.Context(device=0)
        c.set_tuning("gemm_3m", three)
        Bx = lk.krylov_basis_gpu(n, k, dtype, c); Bx.upload(X)
        out.append(np.array(lk.Gram(Bx)))
        del Bx
        c.close()
    iu = np.triu_indices(k)
    assert (np.abs(out[0][iu] - ref[iu]) <= 1e-14 * scale[iu]).all()
    assert (np.abs(out[0] - out[1]) <= 1e-14 * scale).all()
    assert np.array_equal(out[0], out[0].T)                              # mirrored without conjugation, like the reference


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("fused", [0, 1, 2])
@pytest.mark.parametrize("n,k,p", [(20_011, 129, 32), (9001, 256, 8), (7001, 300, 4), (5003, 512, 33), (4001, 200, 2), (3001, 385, 5),
                                   (2000, 257, 64), (640, 384, 16)])
def test_block_dgs_on_column_panels_beyond_128_columns(dtype, fused, n, k, p):
    """Every column of Y against the oracle's block double Gram-Schmidt (coefficients and vectors normwise 1e-12, orthogonality
    1e-13), for 129..512 basis columns and 2..64 right-hand sides, on the fused (last panel: update + coefficients in one pass) and
    the unfused schedule -- and the launch counts say it IS the panel schedule: per group of <= 32 columns of Y, npanels coefficient
    products for H1 + (npanels - 1 | npanels) for H2, one fused update + product (or none), and no single-vector sweep at all."""
    c = lk.Context(device=0)
    c.set_tuning("block_fused", fused)
    Q = orthonormal_basis(n, k, dtype, 5)
    Y = basis(n, p, dtype, 200)
    B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(Q)
    Z = lk.krylov_basis_gpu(n, p, dtype, c); Z.upload(Y)
    beta = np.zeros((k, p), dtype=dtype, order="F")
    c.profile_reset(); c.profile_enable(True)
    assert lk.double_gram_schmidt_step(Z, B, False, beta) == 0
    c.sync()
    n_dots, n_fused, n_sweeps = c.profile_get("xhy_mfma")[0], c.profile_get("xhy_upd_mfma")[0], c.profile_get("dgs_sweep*")[0]
    c.profile_enable(False)
    groups, npan = (p + 31) // 32, (k + 127) // 128
    is_fused = fused == 2 or (fused == 1 and np.dtype(dtype).kind == "f")
    assert n_sweeps == 0 and n_fused == (groups if is_fused else 0)
    assert n_dots == groups * (2 * npan - (1 if is_fused else 0))
    Yo = Y.copy(order="F")
    Ho, info_o = ora.double_gram_schmidt_step_block(Yo, Q)
    assert info_o == 0
    Yg = Z.download()
    scale = np.linalg.norm(Y, axis=0)
    for j in range(p):
        assert np.abs(beta[:, j] - Ho[:, j]).max() <= 1e-12 * scale[j]
        assert np.abs(Yg[:, j] - Yo[:, j]).max() <= 1e-12 * scale[j]
    assert np.abs(Q.conj().T @ Yg).max() <= 1e-13 * scale.max()
    del B, Z
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("p,steps", [(4, 64), (8, 48)])
def test_block_arnoldi_beyond_128_columns_against_the_oracle(dtype, p, steps):
    """Block Arnoldi with blksize = 4 x 64 steps (basis to 260 columns) and 8 x 48 steps (to 392) on a diagonal operator with a
    well-separated spectrum: H against the oracle's block Arnoldi (arnoldi.fypp:20-73 restated on the oracle's primitives) column by
    column at 1e-12, the Arnoldi relation and orthonormality at 1e-12, and no single-vector sweep against more than 128 columns --
    the panel schedule carries every step."""
    n = 6007
    rng = np.random.default_rng(5)
    d = (1.0 + np.arange(n) / n).astype(dtype)
    if np.dtype(dtype).kind == "c":
        d = d * np.exp(1j * np.arange(n) / n)
    Q0 = orthonormal_basis(n, p, dtype, 70)
    c = lk.Context(device=0)
    X = lk.krylov_basis_gpu(n, (steps + 1) * p, dtype, c)
    X.upload(Q0, 0)
    H = np.zeros(((steps + 1) * p, steps * p), dtype=dtype, order="F")
    c.profile_reset(); c.profile_enable(True)
    assert lk.arnoldi(lk.diag_linop_gpu(d, c), X, H, blksize=p) == 0
    c.sync()
    n_fused_or_dots = c.profile_get("xhy_mfma")[0]
    c.profile_enable(False)
    assert n_fused_or_dots > 0
    Xo = np.zeros((n, (steps + 1) * p), dtype=dtype, order="F"); Xo[:, :p] = Q0
    Ho = np.zeros_like(H)
    assert ora.arnoldi_block(ora.DiagOp(d), Xo, Ho, p) == 0
    for j in range(steps * p):
        assert np.abs(H[:, j] - Ho[:, j]).max() <= 1e-12 * np.abs(Ho[:, j]).max(), j
    Xg = X.download()
    m = steps * p
    assert np.abs(d[:, None] * Xg[:, :m] - Xg @ H).max() <= 1e-12 * np.abs(d).max()
    assert np.abs(Xg.conj().T @ Xg - np.eye(m + p)).max() <= 1e-12
    del X
    c.close()
    _ = rng


@pytest.mark.parametrize("k", [3, 8, 40, 128, 200])
def test_complex_gram_diagonal_has_an_imaginary_part_of_exactly_zero(ctx, k):
    """conj(x) . x has an imaginary part of exactly zero in the reference's dotc (every term is re*im - im*re); the matrix-core kernels --
    three real products per complex one, or the doubled real problem -- would leave O(eps |x|^2) there.  lk_gram and lk_innerprod(X, X)
    return exactly zero on the diagonal whatever kernel served them (round-4 advisor), and real, positive squared norms."""
    n = 20_011
    X = basis(n, k, np.complex128, 900)
    B = lk.krylov_basis_gpu(n, k, np.complex128, ctx); B.upload(X)
    G = lk.Gram(B)
    assert np.all(np.diag(G).imag == 0.0) and np.all(np.diag(G).real > 0)
    M = lk.innerprod(B, B)
    assert np.all(np.diag(M).imag == 0.0)
    ref = X.conj().T @ X
    assert np.abs(np.triu(G) - np.triu(ref)).max() <= 1e-12 * np.abs(ref).max()
