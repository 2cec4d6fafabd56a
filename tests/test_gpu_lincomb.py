"""linear_combination (AbstractVectors.fypp:571-643): the tall-skinny product X B in one pass -- krylov_schur's basis update and the
eigenvector reconstruction (BaseKrylov.fypp:816-824, IterativeSolvers.fypp:1127-1132) on the FP64 matrix cores, narrow products
(q = 1..4: the GMRES solution update, gmres.fypp:200-214) on the streaming kernel, the complex kind with three real products per complex one."""
import ctypes as C
import os

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora
from tests._gpu_helpers import KINDS, seeded, basis
from tests._tol import assert_close, assert_columns_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("n,k,q", [(1, 1, 1), (130, 3, 2), (4097, 128, 64), (2051, 200, 70), (1023, 33, 17),
                                   (777, 129, 65), (5000, 16, 48), (3000, 7, 16)])
def test_linear_combination_matrix_shapes(ctx, dtype, n, k, q):
    """linear_combination_matrix (AbstractVectors.fypp:605-643): Y = X C for every split of the output columns
    over the kernel's column groups (q <= 16, 32, 64, > 64), k beyond one register chunk, ragged row tiles."""
    X, Cm = basis(n, k, dtype, 10), basis(k, q, dtype, 900)
    Bx = lk.krylov_basis_gpu(n, k, dtype, ctx); Bx.upload(X)
    Y = lk.linear_combination(Bx, Cm).download()
    ref = X @ Cm
    scale = np.abs(X).max() * np.abs(Cm).max() * k
    assert np.abs(Y - ref).max() <= 4e-15 * scale
    for j in {0, q - 1, q // 2}:                                       # and against the reference's k-axpby schedule
        r = ora.linear_combination(X, np.ascontiguousarray(Cm[:, j]))
        assert np.abs(Y[:, j] - r).max() <= 1e-13 * max(np.abs(r).max(), 1e-300) * max(1, k) ** 0.5


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("mfma_min", [0, 1, 2, 100])
@pytest.mark.parametrize("n,k,q", [(1, 1, 1), (4099, 128, 1), (20_011, 64, 1), (2051, 200, 1), (777, 300, 2), (4097, 128, 2),
                                   (4097, 17, 3), (5003, 128, 4), (5003, 33, 5), (1023, 64, 8), (3001, 128, 9)])
def test_narrow_linear_combinations_on_every_kernel_choice(dtype, mfma_min, n, k, q):
    """linear_combination with 1..9 output columns through the streaming kernel (q accumulators per lane) and through the
    matrix-core kernel (`gemm_mfma_min` moves the crossover): each output column against the oracle's loop of axpbys."""
    c = lk.Context(device=0)
    c.set_tuning("gemm_mfma_min", mfma_min)
    X = basis(n, k, dtype, 10)
    Cm = basis(k, q, dtype, 900)
    Bx = lk.krylov_basis_gpu(n, k, dtype, c); Bx.upload(X)
    Yg = lk.linear_combination(Bx, Cm if q > 1 else np.ascontiguousarray(Cm[:, 0]))
    Yh = Yg.download() if q > 1 else Yg.to_array().reshape(-1, 1)
    for j in range(q):
        ref = ora.linear_combination(X, np.ascontiguousarray(Cm[:, j]))
        assert np.abs(Yh[:, j] - ref).max() <= 1e-13 * np.abs(ref).max() * max(1, k) ** 0.5
    del Bx, Yg
    c.close()


@pytest.mark.parametrize("n,k,q", [(4099, 5, 9), (20_003, 64, 16), (20_003, 128, 17), (9001, 128, 32), (9001, 100, 33), (7001, 128, 48),
                                   (12_289, 128, 64), (5003, 200, 70), (255, 130, 64)])
def test_complex_product_with_three_real_products_per_complex_one(n, k, q):
    """Round 4, "gemm_3m": the complex tall-skinny product on the matrix cores as P1 = Xr Cr, P2 = Xi Ci, P3 = (Xr + Xi)(Cr + Ci),
    Re = P1 - P2, Im = P3 - P1 - P2 (6 flop per complex multiplication instead of 8; the four-product kernel already runs at the
    matrix pipe's sustained rate).  Every output column against the oracle's loop of axpbys (AbstractVectors.fypp:605-643) and against
    the four-product kernel, normwise -- the imaginary part carries the cancellation of P3 - P1 - P2, bounded by eps * sum (|xr| +
    |xi|)(|cr| + |ci|), which is what the bound below is scaled with; groups of 16 outputs, ragged rows and columns, k in chunks."""
    dtype = np.complex128
    X = basis(n, k, dtype, 31)
    Cm = basis(k, q, dtype, 700)
    scale = (np.abs(X.real) + np.abs(X.imag)).max(axis=0) @ (np.abs(Cm.real) + np.abs(Cm.imag))      # per output column
    out = []
    for three in (1, 0):
        c = lk.Context(device=0)
        c.set_tuning("gemm_3m", three)
        Bx = lk.krylov_basis_gpu(n, k, dtype, c); Bx.upload(X)
        Yg = lk.linear_combination(Bx, Cm)
        out.append(Yg.download())
        del Bx, Yg
        c.close()
    for j in range(q):
        ref = ora.linear_combination(X, np.ascontiguousarray(Cm[:, j]))
        assert np.abs(out[0][:, j] - ref).max() <= 1e-14 * scale[j]
        assert np.abs(out[0][:, j] - out[1][:, j]).max() <= 1e-14 * scale[j]


def test_gmres_update_uses_the_streaming_kernel(ctx):
    """The GMRES solution update dx = V(:, :k) y (gmres.fypp:200-201) is a q = 1 product: priced at k + 1 columns and run by
    the one-accumulator kernel (same profile tag, one launch)."""
    n, k = 100_003, 30
    Bx = lk.krylov_basis_gpu(n, k, np.float64, ctx); Bx.upload(basis(n, k, np.float64, 4))
    v = seeded(k, np.float64, 8)
    ctx.profile_reset(); ctx.profile_enable(True)
    y = lk.linear_combination(Bx, v)
    ctx.sync()
    cnt, _ms, by = ctx.profile_get("lincomb")
    ctx.profile_enable(False)
    assert cnt == 1 and by == pytest.approx(8.0 * n * (k + 1))
    ref = ora.linear_combination(Bx.download(), v)
    assert np.abs(y.to_array() - ref).max() <= 1e-13 * np.abs(ref).max() * k ** 0.5
