"""Pins the CPU oracle against every known-answer test and invariant the reference's own tests hold
for the hot path.  The reference's inputs are unseeded random numbers, so there are no
stored vectors to compare with; what its tests assert are (i) analytic spectra, (ii) factorisation /
orthonormality invariants at rtol_dp, (iii) vector-space axioms at 1e-14.  Each is restated here.
"""
import numpy as np
import pytest

from oracle import oracle as ora

N = 128  # test_size, src/Utilities/TestUtils.fypp:18


def seeded(n, dtype, seed):
    x = np.empty(n, dtype=dtype)
    ora.fill_counter(x, seed)
    return x


# The reference's own analytic known answers are the oracle's pin.  The reference asserts them at rtol_dp = 3.2e-8; every matrix
# here is NORMAL (condition number 1 for every eigenvalue), so the pin is made at north_star's tolerance instead.
KAT_RTOL = 1e-12


def test_constants():
    assert ora.ATOL_DP == 1e-15 and abs(ora.RTOL_DP - 3.1622776601683794e-08) < 1e-22   # src/Constants.f90:33-37


def test_eigs_complex_known_answer():
    """test/TestIterativeSolvers.fypp:176-185, 201-203: A(i,i)=n, A(i,i+1)=i*sqrt(i(n-i)), A(i+1,i)=-A(i,i+1)
    => eigenvalues 2(n-i+1)-1 = 255, 253, ..., 1.  The reference asserts rtol_dp; the pin is made at north_star's 1e-12, ELEMENTWISE
    relative (A = n I + i B with B real antisymmetric is Hermitian, hence normal: every eigenvalue has condition number 1;
    measured 1.0e-13 on the eigenvalue 1 of a matrix of norm 255)."""
    A = np.zeros((N, N), dtype=np.complex128)
    for i in range(1, N + 1):
        A[i - 1, i - 1] = N
        if i < N:
            A[i - 1, i] = 1j * np.sqrt(1.0 * i * (N - i))
            A[i, i - 1] = -A[i - 1, i]
    w, res, V, niter = ora.eigs(ora.DenseOp(A), seeded(N, np.complex128, 3), nev=N, tolerance=ora.ATOL_DP)
    true = np.array([2 * (N - i + 1) - 1 for i in range(1, N + 1)], dtype=float)
    assert np.max(np.abs(w - true) / np.abs(true)) < KAT_RTOL
    # eigenvector check of the same test (:205-216): ||A V - V diag(w)|| (rtol_dp there; 1e-12 ||A|| here, measured 2.6e-14)
    assert np.linalg.norm(A @ V - V * w[None, :]) < KAT_RTOL * np.abs(true).max()


def test_eigs_real_toeplitz_known_answer():
    """test/TestIterativeSolvers.fypp:164-174, 193-199: a on the diagonal, +b / -b off-diagonals
    => eigenvalues a +- 2 b cos(k pi/(n+1)) i (a I + antisymmetric: normal; pinned at 1e-12 elementwise, measured 4.8e-15)."""
    a_, b_ = 0.37, 0.61
    A = a_ * np.eye(N) + b_ * np.eye(N, k=1) - b_ * np.eye(N, k=-1)
    w, res, V, niter = ora.eigs(ora.DenseOp(A), seeded(N, np.float64, 4), nev=N, tolerance=ora.ATOL_DP)
    true = np.zeros(N, dtype=complex)
    k = 1
    for i in range(0, N, 2):
        true[i] = a_ + 2j * b_ * np.cos(k * np.pi / (N + 1))
        true[i + 1] = a_ - 2j * b_ * np.cos(k * np.pi / (N + 1))
        k += 1
    assert np.max(np.abs(w - true) / np.abs(true)) < KAT_RTOL


def test_lanczos_spd_toeplitz_known_answer():
    """test/TestIterativeSolvers.fypp:254-280 (eighs): SPD Toeplitz => a + 2|b| cos(i pi/(n+1)); pinned at 1e-12 elementwise
    (measured 1.5e-15)."""
    a, b = 2.5, 0.8
    A = a * np.eye(N) + b * (np.eye(N, k=1) + np.eye(N, k=-1))
    X = np.zeros((N, N + 1), order="F")
    x0 = seeded(N, np.float64, 5)
    X[:, 0] = x0 / np.linalg.norm(x0)
    T = np.zeros((N + 1, N), order="F")
    info = ora.lanczos(ora.DenseOp(A), X, T)
    k = info if info > 0 else N
    lam = np.sort(np.linalg.eigvalsh((T[:k, :k] + T[:k, :k].T) / 2))[::-1]
    true = np.array([a + 2 * abs(b) * np.cos(i * np.pi / (N + 1)) for i in range(1, N + 1)])
    assert k == N and np.max(np.abs(lam - true) / np.abs(true)) < KAT_RTOL


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_arnoldi_factorisation_and_orthonormality(dtype):
    """test/TestKrylov.fypp:194-242: A X = X+ H+ and X^H X = I, asserted at rtol_dp (we hold 1e-13)."""
    rng = np.random.default_rng(0)
    A = rng.standard_normal((N, N)).astype(dtype)
    if np.dtype(dtype).kind == "c":
        A = A + 1j * rng.standard_normal((N, N))
    m = 64
    X = np.zeros((N, m + 1), dtype=dtype, order="F")
    x0 = seeded(N, dtype, 6)
    X[:, 0] = x0 / np.linalg.norm(x0)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.arnoldi(ora.DenseOp(A), X, H) == 0
    assert np.abs(A @ X[:, :m] - X @ H).max() < 1e-12
    assert np.abs(X.conj().T @ X - np.eye(m + 1)).max() < 1e-13
    assert np.abs(np.tril(H, -2)).max() == 0.0


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_block_arnoldi_factorisation_and_orthonormality(dtype):
    """test/TestKrylov.fypp:244-296, the reference's own sizes: p = 2, kdim = test_size / 2 (the basis then has p (kdim + 1) = 130
    columns in a 128-dimensional space: the last block is colinear and re-drawn, qr.fypp:146-162) -- A X(:, :p kdim) = X H and the Gram
    matrix of X(:, :p kdim) = I, asserted by the reference at rtol_dp (we hold 1e-12); H block upper Hessenberg."""
    p, kdim = 2, N // 2
    rng = np.random.default_rng(1)
    A = rng.standard_normal((N, N)).astype(dtype)
    if np.dtype(dtype).kind == "c":
        A = A + 1j * rng.standard_normal((N, N))
    A = np.asfortranarray(A / np.sqrt(N))
    X = np.zeros((N, p * (kdim + 1)), dtype=dtype, order="F")
    for j in range(p):
        X[:, j] = seeded(N, dtype, 40 + j)
    R0 = np.zeros((p, p), dtype=dtype, order="F")
    assert ora.qr_no_pivoting(X[:, :p], R0) == 0                       # initialize_krylov_subspace orthonormalises X0 (utilities.fypp)
    H = np.zeros((p * (kdim + 1), p * kdim), dtype=dtype, order="F")
    info = ora.arnoldi_block(ora.DenseOp(A), X, H, p)
    assert info in (0, p * kdim)                                       # the 65th block has no room left: breakdown at the last step is legal
    m = p * kdim
    assert np.abs(A @ X[:, :m] - X @ H).max() < 1e-12
    assert np.abs(X[:, :m].conj().T @ X[:, :m] - np.eye(m)).max() < 1e-12
    assert np.abs(np.tril(H, -(p + 1))).max() == 0.0


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_vector_axioms_and_dense_vector_ops(dtype):
    """test/TestVectors.fypp:50-179: norm/add/sub/dot/scal against array formulas (rtol_dp there) and
    the axiom harness of AbstractVectors.fypp:733-927 (tolerance 1e-14)."""
    rng = np.random.default_rng(2)
    cplx = np.dtype(dtype).kind == "c"
    tol = 1e-14

    def rnd():
        v = rng.standard_normal(N)
        return (v + 1j * rng.standard_normal(N)).astype(dtype) if cplx else v

    for _ in range(100):
        x, y, z = rnd(), rnd(), rnd()
        a, b = (rng.standard_normal() + (1j * rng.standard_normal() if cplx else 0)), rng.standard_normal()
        assert abs(ora.norm(x) - np.linalg.norm(x)) < tol * np.linalg.norm(x) * 10
        assert abs(ora.dot(x, y) - np.vdot(x, y)) < tol * np.linalg.norm(x) * np.linalg.norm(y) * 10
        # commutativity / associativity of addition
        u = x.copy(); ora.axpby(1.0, y, 1.0, u)
        v = y.copy(); ora.axpby(1.0, x, 1.0, v)
        assert np.abs(u - v).max() < tol
        u2 = u.copy(); ora.axpby(1.0, z, 1.0, u2)
        v2 = y.copy(); ora.axpby(1.0, z, 1.0, v2); ora.axpby(1.0, x, 1.0, v2)
        assert np.abs(u2 - v2).max() < tol * 10
        # additive identity and inverse
        w = x.copy(); ora.axpby(-1.0, x, 1.0, w)
        assert np.abs(w).max() == 0.0
        # scalar compatibility and distributivity
        s1 = x.copy(); ora.scal(s1, a); ora.scal(s1, b)
        s2 = x.copy(); ora.scal(s2, a * b)
        assert np.abs(s1 - s2).max() < tol * 10
        d1 = x.copy(); ora.axpby(1.0, y, 1.0, d1); ora.scal(d1, a)
        d2 = x.copy(); ora.scal(d2, a); ora.axpby(a, y, 1.0, d2)
        assert np.abs(d1 - d2).max() < tol * 10
        # inner product: conjugate symmetry, linearity in the second argument (conj on self)
        assert abs(ora.dot(x, y) - np.conj(ora.dot(y, x))) < tol * 100
        lin = y.copy(); ora.scal(lin, a)
        assert abs(ora.dot(x, lin) - a * ora.dot(x, y)) < tol * 1000


def test_dgs_matches_two_projections_and_reports_zero_vector():
    """gram_schmidt.fypp: h = h1 + h2 with info from the second pass; zero vector => info = 1."""
    Q, _ = np.linalg.qr(np.random.default_rng(3).standard_normal((N, 10)))
    Q = np.asfortranarray(Q)
    y = seeded(N, np.float64, 9)
    y0 = y.copy()
    h, info = ora.double_gram_schmidt_step(y, Q)
    assert info == 0
    assert np.abs(Q.T @ y).max() < 1e-15 * np.linalg.norm(y0) * 10
    assert np.abs(h - Q.T @ y0).max() < 1e-14 * np.linalg.norm(y0)
    z = np.zeros(N)
    _, info = ora.double_gram_schmidt_step(z, Q)
    assert info == 1


def test_qr_no_pivoting_with_engine_streams_for_colinear_columns():
    """qr.fypp:146-162: a colinear column is re-drawn, orthogonalised against the columns before it and normalised.  With `column_seed`
    the oracle draws from the counter stream the engine uses for that column, so GPU tests can compare the re-drawn columns entry by
    entry; here: the factorisation still holds on the independent columns, Q comes out orthonormal, R(j, j) = 0 marks the re-draws,
    `info` is the first of them, and the default streams give the same R on the independent columns (both kinds)."""
    import numpy as np
    from oracle import oracle as ora
    for dtype in (np.float64, np.complex128):
        n, p = 257, 6
        Y = np.empty((n, p), dtype=dtype, order="F")
        for j in range(p):
            ora.fill_counter(Y[:, j], 40 + j)
        Y[:, 2] = 3.0 * Y[:, 0]
        Y[:, 4] = Y[:, 1] - Y[:, 3]
        Q, R = Y.copy(order="F"), np.zeros((p, p), dtype=dtype, order="F")
        assert ora.qr_no_pivoting(Q, R, tol=1e-10, column_seed=lambda j: 0x5EED + j + 1) == 3
        assert R[2, 2] == 0 and R[4, 4] == 0
        assert np.abs(Q.conj().T @ Q - np.eye(p)).max() <= 1e-12
        keep = [0, 1, 3, 5]
        assert np.abs(Y[:, keep] - Q @ R[:, keep]).max() <= 1e-12 * np.abs(Y).max() * p
        col = np.empty(n, dtype=dtype)
        ora.fill_counter(col, 0x5EED + 2 + 1)                       # the raw re-draw of column 2 ...
        h = Q[:, :2].conj().T @ col                                 # ... orthogonalised against columns 0, 1 and normalised is Q(:, 2)
        w = col - Q[:, :2] @ h
        assert np.abs(Q[:, 2] - w / np.linalg.norm(w)).max() <= 1e-12
        Q2, R2 = Y.copy(order="F"), np.zeros((p, p), dtype=dtype, order="F")
        assert ora.qr_no_pivoting(Q2, R2, tol=1e-10) == 3
        assert np.array_equal(R2[:, :2], R[:, :2])                  # (columns in front of the first re-draw do not depend on the stream)
