"""GPU tests added in round 3 (all through the C ABI).

  * bases of 129..512 columns on the lane-split fused sweeps (one pass over X per sweep, 3k+4 columns per DGS) and the
    asynchronous Arnoldi / Lanczos / Golub-Kahan pipelines beyond 128 columns -- reference: `kdim = (size(X) - p) / p` has no
    cap (src/Krylov/arnoldi.fypp:26);
  * narrow tall-skinny products (q = 1..4: the GMRES solution update, gmres.fypp:200-214, and every X * v of
    linear_combination, AbstractVectors.fypp:571-643) on the streaming kernel with q accumulators per lane;
  * the column pool's slab geometry on row-sharded contexts and its generation counter;
  * the operator's own time inside the asynchronous batch (`matvec` profile tag).
"""
import ctypes as C

import numpy as np
import pytest

import lightkrylov_amd as lk
from lightkrylov_amd import _capi
from oracle import oracle as ora

pytestmark = pytest.mark.gpu
KINDS = [np.float64, np.complex128]


def seeded(n, dtype, seed):
    x = np.empty(n, dtype=dtype)
    ora.fill_counter(x, seed)
    return x


def basis(n, k, dtype, seed):
    X = np.empty((n, k), dtype=dtype, order="F")
    for j in range(k):
        ora.fill_counter(X[:, j], seed + j)
    return X


def orthonormal_basis(n, k, dtype, seed):
    Q, _ = np.linalg.qr(basis(n, k, dtype, seed))
    return np.asfortranarray(Q)


# ----------------------------------------------------------------------------- wide bases: the fused sweeps
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
        c.profile_enable(False)
        assert cnt == [m, m, m]                                             # three launches per step, whatever the width
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


# ----------------------------------------------------------------------------- narrow tall-skinny products
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


# ----------------------------------------------------------------------------- column pool
def _pool_fns(ctx):
    lib = _capi.load()

    def acquire(dtype, n, tag):
        slab, col = C.c_void_p(), C.c_int()
        _capi.check(lib.lk_pool_acquire(ctx._h, dtype, n, C.c_uint64(tag), C.byref(slab), C.byref(col)))
        return slab.value, col.value

    def info(slab, col):
        t, g = C.c_uint64(), C.c_uint64()
        _capi.check(lib.lk_pool_column_info(ctx._h, C.c_void_p(slab), col, C.byref(t), C.byref(g)))
        return t.value, g.value

    return lib, acquire, info


def test_pool_generation_counter_exposes_stale_bit_copies():
    """A column's generation goes up every time the pool hands it out: first use, re-use by the same owner tag (an object
    re-created at a dead one's address), re-use after a release.  A handle that remembers the generation it was bound at --
    the Fortran plugin's does -- can tell that its column now belongs to something else."""
    c = lk.Context(device=0)
    lib, acquire, info = _pool_fns(c)
    c.set_tuning("pool_slab_cols", 8)
    a = acquire(_capi.LK_F64, 1000, 0x1000)
    ga = info(*a)[1]
    assert info(*a) == (0x1000, ga) and ga >= 1
    b = acquire(_capi.LK_F64, 1000, 0x2000)
    gb = info(*b)[1]
    assert info(*b)[0] == 0x2000 and gb > ga                                          # ONE counter per context: never the same value twice
    assert acquire(_capi.LK_F64, 1000, 0x1000) == a and info(*a)[1] > gb             # same address again: same column, a later generation
    _capi.check(lib.lk_pool_release(c._h, C.c_void_p(b[0]), b[1]))
    assert info(*b) == (0, gb)
    g3 = info(*a)[1]
    assert acquire(_capi.LK_F64, 1000, 0x3000) == b and info(*b)[0] == 0x3000 and info(*b)[1] > g3   # released column handed to another owner
    assert info(a[0], 7) == (0, 0) and info(0xdead0, 0) == (0, 0)                     # never carved / not a slab: safe to ask
    # a stale handle from before lk_pool_release_all must not match whatever a NEW slab hands out, even when that slab lands on
    # the freed one's heap address and the column index is the same: generations do not restart
    seen = {info(*a)[1], info(*b)[1]}
    _capi.check(lib.lk_pool_release_all(c._h))
    a2 = acquire(_capi.LK_F64, 1000, 0x1000)
    assert info(*a2)[1] not in seen and info(*a2)[1] > max(seen)
    _capi.check(lib.lk_pool_release_all(c._h))
    c.close()


def test_pool_slab_geometry_is_rank_independent_on_sharded_contexts():
    """On a single-rank context a slab shrinks to a quarter of the free memory; on a row-sharded one it must not (free memory
    differs between ranks; unequal slabs would desynchronise the lazy path's batched all-reduces): there a slab has exactly
    pool_slab_cols columns, or the acquisition fails."""
    import torch
    free_b, _total = torch.cuda.mem_get_info(0)
    want = 2048
    n = int(0.3 * free_b / (8.0 * want))                    # `want` columns = 0.3 of the free memory: more than a quarter, and it fits
    lib = _capi.load()

    def slab_cols(ctx):
        slab, col = C.c_void_p(), C.c_int()
        _capi.check(lib.lk_pool_acquire(ctx._h, _capi.LK_F64, n, C.c_uint64(0x1000), C.byref(slab), C.byref(col)))
        nc = C.c_int()
        _capi.check(lib.lk_basis_info(slab, None, None, C.byref(nc), None, None))
        _capi.check(lib.lk_pool_release_all(ctx._h))
        return nc.value

    single = lk.Context(device=0)
    single.set_tuning("pool_slab_cols", want)
    got1 = slab_cols(single)
    single.close()
    assert got1 < want                                      # single rank: memory-derived
    cb = _capi.ALLREDUCE_FN(lambda _u, _p, _n, _s: 0)       # a 2-rank context (the reduction itself is not exercised here)
    sharded = lk.Context(device=0)
    _capi.check(lib.lk_set_allreduce(sharded._h, cb, None, 2, 0))
    sharded.set_tuning("pool_slab_cols", want)
    got2 = slab_cols(sharded)
    assert got2 == want                                     # sharded: exactly the configured geometry
    # ... and an impossible geometry fails loudly instead of shrinking
    sharded.set_tuning("pool_slab_cols", 4096)
    big = int(free_b / (8.0 * 4096) * 1.5)
    slab, col = C.c_void_p(), C.c_int()
    rc = lib.lk_pool_acquire(sharded._h, _capi.LK_F64, big, C.c_uint64(0x2000), C.byref(slab), C.byref(col))
    assert rc != 0 and b"pool_slab_cols" in lib.lk_last_error()
    _capi.check(lib.lk_set_allreduce(sharded._h, _capi.ALLREDUCE_FN(), None, 1, 0))
    sharded.close()


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


# ----------------------------------------------------------------------------- measurement hygiene
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


# ----------------------------------------------------------------------------- block DGS with many right-hand sides: three passes
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
