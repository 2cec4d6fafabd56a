"""Tuning knobs (lk_set_tuning) choose kernels and cache policies, never results: the store policies of the sweeps, both kernels of sweep 1,
and the lazy path's batching rules (a dot batch stops at the columns ever written; the reference's Gram loop costs one pass)."""
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
def test_store_policy_knobs_are_result_invariant(dtype):
    """The cache policy of the y'' store (plain / nt / sc1 / sc0 sc1) and the lane-split store only change HOW the same
    bytes are written: coefficients and vector must be bit-identical."""
    c = lk.Context(device=0)
    n, k = 300_007, 37
    Xh, yh = basis(n, k, dtype, 50), seeded(n, dtype, 99)
    Q, _ = np.linalg.qr(Xh)
    B = lk.krylov_basis_gpu(n, k + 1, dtype, c)
    out = []
    for pol, split in ((0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (1, 1)):
        c.set_tuning("store_policy", pol); c.set_tuning("store_split", split)
        B.upload(np.asfortranarray(Q)); B.upload(yh.reshape(-1, 1), k)
        h = np.zeros(k, dtype=dtype)
        lk.double_gram_schmidt_step(B[k], B[:k], False, h)
        out.append((h.tobytes(), B.download(k, 1).tobytes()))
    assert all(o == out[0] for o in out[1:])
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("cfg", [dict(dot_colwise=0), dict(dot_colwise=1, cw_u=4), dict(dot_colwise=1, cw_u=8),
                                 dict(dot_colwise=1, cw_u=8, cw_grid_mult=1)])
def test_both_sweep1_kernels_match_the_oracle(dtype, cfg):
    """DGS sweep 1 / innerprod by either kernel -- all columns per tile (panel_sweep<DOT>) or one column at a time
    (panel_dot_cw, 4 or 8 loads per lane and column, more tiles than blocks) -- against the oracle's innerprod
    (AbstractVectors.fypp:659-695), normwise 1e-12; ragged sizes around the tile sizes (512 / 1024 / 2048 / 4096 rows)."""
    c = lk.Context(device=0)
    for key, val in cfg.items():
        c.set_tuning(key, val)
    rng = np.random.default_rng(3)
    try:
        for n, k in [(1, 1), (2, 1), (511, 3), (1025, 17), (2047, 128), (4097, 33), (1_000_003, 8), (300_001, 128)]:
            A = rng.standard_normal((n, k + 1)) + (1j * rng.standard_normal((n, k + 1)) if np.dtype(dtype).kind == "c" else 0)
            A = np.asfortranarray(A.astype(dtype))
            B = lk.krylov_basis_gpu(n, k + 1, dtype, c)
            B.upload(A)
            got = np.asarray(lk.innerprod(B[:k], B[k]))
            want = ora.innerprod(A[:, :k], A[:, k])
            scale = np.linalg.norm(A[:, k]) * np.linalg.norm(A[:, :k], axis=0).max()
            assert np.abs(got - want).max() <= 1e-12 * scale, (n, k, cfg)
            del B
    finally:
        c.close()


def test_lazy_dot_batch_stops_at_the_columns_ever_written(ctx):
    """A slab-like panel with 160 columns of which 5 hold vectors: X(i)%dot(y) with y in ANOTHER panel must sweep 5
    columns, not 128 (the batch is capped by the panel's high-water mark)."""
    n = 200_001
    c = lk.Context(device=0)
    c.set_tuning("lazy", 1)
    P = lk.krylov_basis_gpu(n, 160, np.float64, c)
    Xh = basis(n, 5, np.float64, 7)
    P.upload(Xh, 0)
    y = lk.dense_vector_gpu.from_array(seeded(n, np.float64, 70), c)
    c.profile_reset(); c.profile_enable(True)
    got = [P[i].dot(y) for i in range(5)]
    c.sync()
    cnt, _ms, by = c.profile_get("dgs_sweep1")
    c.profile_enable(False)
    hits, sweeps, _q, _f = c.lazy_stats()
    assert (sweeps, hits) == (1, 4) and cnt == 1
    assert by == pytest.approx(8.0 * n * (5 + 1))            # 5 columns + y, not 128 + 1
    ref = ora.innerprod(Xh, y.to_array())
    assert np.abs(np.array(got) - ref).max() <= 1e-12 * np.linalg.norm(Xh[:, 0]) * np.linalg.norm(y.to_array())
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_lazy_gram_loop_of_the_reference_costs_one_pass(dtype):
    """gram_matrix through the per-object calls an unchanged LightKrylov makes (AbstractVectors.fypp:651-656:
    G(i,j) = X(i)%dot(X(j)), j = i..k; G(j,i) = G(i,j)) on a lazy context: the second call of the run computes X^H X on the
    matrix cores, the other k(k+1)/2 - 2 are served from it -- against the oracle's Gram and an eager context; a write in
    between invalidates."""
    n, k = 20_011, 40
    X = basis(n, k, dtype, 21)
    Go = ora.gram(X)
    res = {}
    for lazy in (0, 1):
        c = lk.Context(device=0)
        c.set_tuning("lazy", lazy)
        B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(X)
        G = np.zeros((k, k), dtype=dtype)
        for i in range(k):
            for j in range(i, k):
                G[i, j] = B[i].dot(B[j]); G[j, i] = G[i, j]
        hits, sweeps, _q, _f = c.lazy_stats()
        assert np.abs(G - Go).max() <= 1e-12 * np.linalg.norm(X, axis=0).max() ** 2
        if lazy:
            assert sweeps == 1 and hits >= k * (k + 1) // 2 - 2
            B[3].scal(2.0)                                           # a write: the memo must not survive it
            assert abs(B[3].dot(B[3]) - 4.0 * Go[3, 3]) <= 1e-12 * abs(Go[3, 3]) * 4
            assert abs(B[2].dot(B[3]) - 2.0 * Go[2, 3]) <= 1e-12 * np.linalg.norm(X[:, 2]) * np.linalg.norm(X[:, 3]) * 2
        res[lazy] = G
        del B
        c.close()
    assert np.abs(res[0] - res[1]).max() <= 1e-12 * np.linalg.norm(X, axis=0).max() ** 2


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("cfg", [dict(), dict(xhy_db=0), dict(xhy_db=2), dict(gram_rs=0), dict(gram_rs=2), dict(gram_rs=5),
                                 dict(gemm_roll=0), dict(gemm_roll=2)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "defaults")
def test_matrix_core_kernel_variants_agree_with_the_oracle(dtype, cfg):
    """The matrix-core kernel variants behind lk_set_tuning (double-buffered tiles, dealt Gram tiles, rolling prefetch) are schedules of
    the SAME sums: Gram (AbstractVectors.fypp:645-657), innerprod_matrix (:670-695), the block
    DGS (gram_schmidt.fypp:59-105) and linear_combination (AbstractVectors.fypp:596-642) against the oracle at ragged sizes."""
    c = lk.Context(device=0)
    for key, val in cfg.items():
        c.set_tuning(key, val)
    n = 10_037
    for k, p in ((128, 40), (120, 6), (72, 64), (48, 9)):
        X, Y = basis(n, k, dtype, 7), basis(n, p, dtype, 400)
        Bx = lk.krylov_basis_gpu(n, k, dtype, c); Bx.upload(X)
        By = lk.krylov_basis_gpu(n, p, dtype, c); By.upload(Y)
        scale = np.linalg.norm(X, axis=0).max() * max(np.linalg.norm(X, axis=0).max(), np.linalg.norm(Y, axis=0).max())
        assert np.abs(lk.Gram(Bx) - ora.gram(X)).max() <= 1e-13 * scale
        assert np.abs(lk.innerprod(Bx, By) - X.conj().T @ Y).max() <= 1e-13 * scale
        # X <- X Z (the restart update) on a copy
        rng = np.random.default_rng(5)
        Z = rng.standard_normal((k, p)) + (1j * rng.standard_normal((k, p)) if dtype == np.complex128 else 0)
        Z = np.asfortranarray(Z.astype(dtype))
        Bz = lk.linear_combination(Bx, Z)
        ref = X @ Z
        assert np.abs(Bz.download(0, p) - ref).max() <= 1e-13 * np.abs(X).max() * np.abs(Z).sum(axis=0).max() * 4
        # block DGS of Y against an orthonormal X
        Q = np.asfortranarray(np.linalg.qr(X)[0])
        Bx.upload(Q)
        beta = np.zeros((k, p), dtype=dtype, order="F")
        assert lk.double_gram_schmidt_step(By, Bx, False, beta) == 0
        Yo = Y.copy(order="F")
        ho = np.zeros((k, p), dtype=dtype, order="F")
        for j in range(p):
            hj, _ = ora.double_gram_schmidt_step(Yo[:, j], Q)
            ho[:, j] = hj
        assert np.abs(beta - ho).max() <= 1e-12 * np.linalg.norm(Y, axis=0).max()
        assert_columns_close(By.download(0, p), Yo, f"block DGS k={k} p={p} {cfg}")
        del Bx, By, Bz
    c.close()


@pytest.mark.parametrize("k,q", [(128, 64), (64, 32), (128, 16), (64, 64)])
def test_rolling_prefetch_complex_product_over_many_tiles_per_block(k, q):
    """The same for the complex three-product kernel (panel_gemm_mfma3m<NG, NR, ROLL>; 1, 2 and 4 output groups): ring against batch schedule bit for
    bit on panels of several tiles per block with a ragged last tile, sampled rows against numpy."""
    c = lk.Context(device=0)
    n = 300_007
    X = lk.krylov_basis_gpu(n, k, np.complex128, c)
    for j in range(k):
        X[j].rand(True, seed=300 + j)
    rng = np.random.default_rng(12)
    Z = np.asfortranarray(rng.standard_normal((k, q)) + 1j * rng.standard_normal((k, q)))
    out = {}
    for roll in (0, 1):
        c.set_tuning("gemm_roll", roll)
        Y = lk.linear_combination(X, Z)
        out[roll] = Y.download(0, q)
        del Y
    assert np.array_equal(out[0], out[1])
    rows = np.r_[0:300, 131_000:131_400, n - 400:n]
    Xh = X.download(0, k)[rows]
    assert np.abs(out[1][rows] - Xh @ Z).max() <= 1e-13 * np.abs(Xh).max() * np.abs(Z).sum(axis=0).max() * 8
    c.close()


@pytest.mark.parametrize("k,q", [(128, 64), (64, 48), (48, 64), (128, 33)])
def test_rolling_prefetch_product_over_many_tiles_per_block(k, q):
    """linear_combination (AbstractVectors.fypp:596-642; the restart update of BaseKrylov.fypp:816-824) with "gemm_roll": panels long enough
    that every block runs SEVERAL 512-row tiles, so the ring of k-steps carries the next tile's first columns across the epilogue -- the same
    MFMAs in the same order as the batch schedule: bit-identical results, and the sampled rows agree with numpy; a ragged last tile and a
    basis width that is not a multiple of 16 take the guarded path."""
    c = lk.Context(device=0)
    n = 1_200_007
    X = lk.krylov_basis_gpu(n, k, np.float64, c)
    for j in range(k):
        X[j].rand(True, seed=100 + j)
    rng = np.random.default_rng(11)
    Z = np.asfortranarray(rng.standard_normal((k, q)))
    out = {}
    for roll in (0, 2):
        c.set_tuning("gemm_roll", roll)
        Y = lk.linear_combination(X, Z)
        out[roll] = Y.download(0, q)
        del Y
    assert np.array_equal(out[0], out[2])
    rows = np.r_[0:600, 511_900:512_700, n - 700:n]
    Xh = X.download(0, k)[rows]
    ref = Xh @ Z
    assert np.abs(out[2][rows] - ref).max() <= 1e-13 * np.abs(Xh).max() * np.abs(Z).sum(axis=0).max() * 4
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
@pytest.mark.parametrize("n,k", [(200_003, 128), (70_001, 113), (16_400, 128), (300_017, 97), (150_000, 96), (90_001, 81), (200_003, 64), (70_001, 49), (33, 56),
                                 (120_007, 48), (50_011, 33), (1_000_001, 40), (250_005, 80), (100_000, 72), (17, 70), (500_003, 32), (300_001, 17), (200_000, 8),
                                 (100_001, 5), (2_000_003, 16)])
def test_row_split_gram_kernel_over_many_tiles_per_block(dtype, n, k):
    """gram_matrix (AbstractVectors.fypp:645-657) of 5..128 real columns by panel_gram_rs and of 33..80 complex columns by panel_gram_rs3m (rows of the staged tile dealt
    to the waves, tiles staged by LDS-DMA into a ring of three to five buffers behind counted waits, operands by inline-asm LDS reads) on panels long enough that every
    block runs MANY tiles -- every buffer of the ring, the prefetch running off the end of the panel, a ragged last tile or none, widths that are not a multiple of 16
    (the last column block partly beyond the panel), every number of column blocks 3..8, panels of one tile and a bit -- against the oracle, against the kernel behind
    "gram_rs" = 0, the same bits from call to call, and from a grid of a different size only up to rounding (the partial sums change)."""
    c = lk.Context(device=0)
    X = basis(n, k, dtype, 31)
    B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(X)
    ref = ora.gram(X)
    scale = np.linalg.norm(X, axis=0).max() ** 2
    out = {}
    for rs in (1, 3, 0):
        c.set_tuning("gram_rs", rs)
        out[rs] = lk.Gram(B)
        assert np.abs(out[rs] - ref).max() <= 1e-13 * scale, rs
        assert np.array_equal(out[rs], out[rs].T)                    # (the reference mirrors the upper triangle without conjugating)
    c.set_tuning("gram_rs", 1)
    assert np.array_equal(lk.Gram(B), out[1])                        # fixed order of the sums: the same bits again
    assert np.abs(out[1] - out[0]).max() <= 1e-13 * scale
    c.close()


def test_row_split_gram_kernel_many_times_over_for_races():
    """The LDS-DMA ring of panel_gram_rs / panel_gram_rs3m is ordered by counted vmcnt waits and one raw barrier per tile, its operand reads by the kernel's own lgkmcnt
    wait: an early read would show as a rare wrong tile, so the same Gram matrix is taken 200 times at several widths (two and one blocks per CU, both kinds) and must come
    out bit-identical every time, and right."""
    c = lk.Context(device=0)
    for n, k, dtype in ((400_003, 96, np.float64), (250_001, 128, np.float64), (600_000, 48, np.float64), (300_007, 48, np.complex128), (200_001, 80, np.complex128)):
        X = basis(n, k, dtype, 5)
        B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(X)
        first = lk.Gram(B)
        assert np.abs(first - ora.gram(X)).max() <= 1e-13 * np.linalg.norm(X, axis=0).max() ** 2
        for _ in range(200):
            assert np.array_equal(lk.Gram(B), first)
        del B
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_block_dgs_updates_on_the_ring_over_many_tiles_per_block(dtype):
    """DGS_basis_against_basis (gram_schmidt.fypp:59-105), k = 128, p = 32 / 64, on a panel of several tiles per block: the ACCUMULATING products Y -= X H (complex:
    panel_gemm_mfma3m on the ring, the next tile's loads of X in flight under the read-modify-write of Y; real p = 64: panel_gemm_mfma<..., ROLL>) give the same
    bits as the batch schedule, and Y comes out orthogonal to X."""
    n, k = 150_001, 128
    res = {}
    for roll in (0, 1):
        c = lk.Context(device=0)
        c.set_tuning("gemm_roll", roll)
        B = lk.krylov_basis_gpu(n, k + 64, dtype, c)
        for j in range(k + 64):
            B[j].rand(True, seed=700 + j)
        R = np.zeros((k, k), dtype=dtype, order="F")
        assert lk.qr(B[:k], R) == 0                                  # (qr.fypp:116-167: the basis orthonormal, as DGS_basis_against_basis expects)
        out = []
        for p0, p in ((k, 32), (k, 64)):
            beta = np.zeros((k, p), dtype=dtype, order="F")
            assert lk.double_gram_schmidt_step(B[p0:p0 + p], B[:k], False, beta) == 0
            out.append((beta.copy(), B.download(p0, p)))
            assert np.abs(lk.innerprod(B[:k], B[p0:p0 + p])).max() <= 1e-13 * np.sqrt(n)
        res[roll] = out
        c.close()
    for (b0, y0), (b1, y1) in zip(res[0], res[1]):
        assert np.array_equal(b0, b1) and np.array_equal(y0, y1)


@pytest.mark.parametrize("dtype", KINDS)
def test_gram_across_the_kernel_dispatch_boundaries(dtype):
    """gram_matrix (AbstractVectors.fypp:645-657) at every width and length where lk_gram changes kernels or kernel instances (32 | 33, 48 | 49, 64 | 65, 80 | 81, 96 | 97, 112 | 113, 128 | 129 columns;
    panels of 1, 2, 3 rows, one row short of / at / beyond a tile, a few tiles) against numpy."""
    c = lk.Context(device=0)
    for n in (1, 2, 3, 31, 32, 33, 63, 64, 65, 1000, 4097):
        Xall = basis(n, 130, dtype, 5)
        for k in (5, 32, 33, 47, 48, 49, 50, 63, 64, 65, 80, 81, 96, 97, 111, 112, 113, 127, 128, 129):
            X = np.asfortranarray(Xall[:, :k])
            B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(X)
            G = lk.Gram(B)
            ref = ora.gram(X)
            scale = max(np.linalg.norm(X, axis=0).max() ** 2, 1e-300)
            assert np.abs(G - ref).max() <= 1e-13 * scale, (n, k)
            del B
    c.close()


@pytest.mark.parametrize("dtype", KINDS)
def test_lincomb_across_the_kernel_dispatch_boundaries(dtype):
    """linear_combination (AbstractVectors.fypp:596-642) at the widths and lengths where lk_lincomb changes kernels or schedules (VALU | MFMA, 1 / 2 / 4 output groups, batch
    schedule | ring for k = 64 / 128, a basis wider than one 128-column chunk; panels shorter than a tile, one row short of / beyond a tile) against numpy."""
    c = lk.Context(device=0)
    rng = np.random.default_rng(3)
    for n in (1, 7, 255, 256, 257, 511, 512, 513, 3000):
        Xall = basis(n, 130, dtype, 9)
        for k in (4, 63, 64, 65, 127, 128, 129, 130):
            X = np.asfortranarray(Xall[:, :k])
            B = lk.krylov_basis_gpu(n, k, dtype, c); B.upload(X)
            for q in (1, 4, 5, 9, 16, 17, 32, 33, 48, 64, 65):
                Z = rng.standard_normal((k, q)) + (1j * rng.standard_normal((k, q)) if dtype == np.complex128 else 0)
                Z = np.asfortranarray(Z.astype(dtype))
                Y = lk.linear_combination(B, Z)
                got = Y.download(0, q)
                del Y
                tol = 1e-13 * max(np.abs(X).max(), 1e-300) * np.abs(Z).sum(axis=0).max() * 8
                assert np.abs(got - X @ Z).max() <= tol, (n, k, q)
            del B
    c.close()


def test_the_shipped_library_has_no_key_that_gives_wrong_results():
    """Round-5 review: the phase-timing switches ("xhy_debug", "upd_debug": parts of a kernel OFF, wrong results) and the variants measured
    slower and never defaulted ("mfma_4x4", "xhy_tr32") are not reachable through the ABI of the library build() produces."""
    c = lk.Context(device=0)
    try:
        for key in ("xhy_debug", "upd_debug", "mfma_4x4", "xhy_tr32"):
            with pytest.raises(Exception, match="unknown key"):
                c.set_tuning(key, 1)
    finally:
        c.close()
