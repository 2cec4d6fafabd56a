"""Block Gram-Schmidt on panels against bases WIDER than 128 columns (round 5; all through the C ABI).

The reference routine is size-generic -- DGS_basis_against_basis, src/Krylov/gram_schmidt.fypp:59-105, called by the block Arnoldi
with `blksize` (src/Krylov/arnoldi.fypp:34-56): a block Arnoldi with p = 4 holds more than 128 basis columns after 32 steps.  The
engine keeps the panel x panel schedule there (coefficients and updates on the FP64 matrix cores, column panels of X of <= 128
columns, 4 k - |last panel| columns of X per group of <= 32 columns of Y) instead of falling back to one three-sweep DGS per column.
"""
import numpy as np
import pytest

import lightkrylov_amd as lk
from oracle import oracle as ora

pytestmark = pytest.mark.gpu
KINDS = [np.float64, np.complex128]


def basis(n, k, dtype, seed):
    X = np.empty((n, k), dtype=dtype, order="F")
    for j in range(k):
        ora.fill_counter(X[:, j], seed + j)
    return X


def orthonormal_basis(n, k, dtype, seed):
    Q, _ = np.linalg.qr(basis(n, k, dtype, seed))
    return np.asfortranarray(Q)


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
