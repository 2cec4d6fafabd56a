"""Host logic of lightkrylov_amd without a GPU: the generic abstract_vector path of
double_gram_schmidt_step / qr / arnoldi / lanczos / gmres / eigs / krylov_schur, driven with the
test-only oracle-backed vector type, must reproduce the oracle's own restatement of the reference
drivers (same arithmetic underneath, so agreement is to rounding of the host LAPACK calls)."""
import os

import numpy as np
import pytest

import lightkrylov_amd as lk
from oracle import oracle as ora
from tests._oracle_vector import oracle_dense_linop, oracle_diag_linop, oracle_lap5_linop, oracle_vector


def seeded(n, dtype, seed):
    x = np.empty(n, dtype=dtype)
    ora.fill_counter(x, seed)
    return x


def list_basis(n, ncols, dtype):
    return [oracle_vector(np.zeros(n, dtype=dtype)) for _ in range(ncols)]


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_generic_arnoldi_equals_oracle(dtype):
    n, m = 500, 20
    d = (1.0 + np.arange(n) / n).astype(dtype)
    x0 = seeded(n, dtype, 7); x0 /= np.linalg.norm(x0)
    X = list_basis(n, m + 1, dtype); X[0].data[:] = x0
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    A = oracle_diag_linop(d)
    assert lk.arnoldi(A, X, H) == 0 and A.matvec_counter == m
    Xo = np.zeros((n, m + 1), dtype=dtype, order="F"); Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert ora.arnoldi(ora.DiagOp(d), Xo, Ho) == 0
    assert np.array_equal(H, Ho)                               # same arithmetic, same order => bit-exact
    assert all(np.array_equal(X[j].data, Xo[:, j]) for j in range(m + 1))


def test_generic_block_arnoldi_invariants():
    """test/TestKrylov.fypp:244-296 (blksize = 2)."""
    n, kdim, p = 128, 10, 2
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, n))
    X = list_basis(n, (kdim + 1) * p, np.float64)
    Q, _ = np.linalg.qr(rng.standard_normal((n, p)))
    for i in range(p):
        X[i].data[:] = Q[:, i]
    H = np.zeros(((kdim + 1) * p, kdim * p), order="F")
    assert lk.arnoldi(oracle_dense_linop(A), X, H, blksize=p) == 0
    Xm = np.stack([x.data for x in X], axis=1)
    assert np.abs(A @ Xm[:, :kdim * p] - Xm @ H).max() < 1e-12
    assert np.abs(Xm.T @ Xm - np.eye((kdim + 1) * p)).max() < 1e-13


def test_dgs_checks_and_beta_shape():
    n, k = 200, 6
    Q, _ = np.linalg.qr(np.random.default_rng(1).standard_normal((n, k)))
    X = [oracle_vector(Q[:, j].copy()) for j in range(k)]
    y = oracle_vector(seeded(n, np.float64, 3))
    yo = y.data.copy()
    beta = np.zeros(k)
    assert lk.double_gram_schmidt_step(y, X, beta=beta) == 0           # default orthonormality check passes
    ho, _ = ora.double_gram_schmidt_step(yo, np.asfortranarray(Q))
    assert np.array_equal(beta, ho) and np.array_equal(y.data, yo)
    with pytest.raises(ValueError):
        lk.double_gram_schmidt_step(y, X, False, beta=np.zeros(k + 1))
    X[0].scal(3.0)
    with pytest.raises(RuntimeError, match="not orthonormal"):
        lk.double_gram_schmidt_step(y, X)
    z = oracle_vector(np.zeros(n))
    X[0].scal(1.0 / 3.0)
    assert lk.double_gram_schmidt_step(z, X, False) == 1


def test_qr_rank_deficient_column_is_flagged():
    """qr.fypp:146-162: a colinear column sets info = j, R(j,j) = 0 and is replaced by a random direction."""
    n = 100
    a = seeded(n, np.float64, 1)
    Q = [oracle_vector(a.copy()), oracle_vector(2.0 * a), oracle_vector(seeded(n, np.float64, 2))]
    R = np.zeros((3, 3), order="F")
    assert lk.qr(Q, R) == 2 and R[1, 1] == 0.0
    Qm = np.stack([q.data for q in Q], axis=1)
    assert np.abs(Qm.T @ Qm - np.eye(3)).max() < 1e-13


def test_generic_lanczos_equals_oracle():
    n = 64
    A = 2.5 * np.eye(n) + 0.8 * (np.eye(n, k=1) + np.eye(n, k=-1))
    x0 = seeded(n, np.float64, 5); x0 /= np.linalg.norm(x0)
    X = list_basis(n, n + 1, np.float64); X[0].data[:] = x0
    T = np.zeros((n + 1, n), order="F")
    info = lk.lanczos(oracle_dense_linop(A), X, T)
    Xo = np.zeros((n, n + 1), order="F"); Xo[:, 0] = x0
    To = np.zeros((n + 1, n), order="F")
    assert info == ora.lanczos(ora.DenseOp(A), Xo, To)
    assert np.array_equal(T, To)


def test_generic_gmres_equals_oracle():
    N = 24
    n = N * N
    b = seeded(n, np.float64, 11)
    x = oracle_vector(np.zeros(n))
    meta = lk.gmres_dp_metadata()
    A = oracle_lap5_linop(N)
    info = lk.gmres(A, oracle_vector(b.copy()), x, rtol=1e-8, options=lk.gmres_dp_opts(kdim=30, maxiter=2), meta=meta)
    xo = np.zeros(n)
    info_o, res_o = ora.gmres(ora.Lap5Op(N), b, xo, rtol=1e-8, kdim=30, maxiter=2)
    assert info == info_o and meta.n_iter == abs(info)
    np.testing.assert_allclose(meta.res, res_o, rtol=1e-12, atol=0)
    np.testing.assert_allclose(x.data, xo, rtol=1e-12, atol=1e-300)
    # info sign convention (gmres.fypp:234-238): converged => +n_iter
    x2 = oracle_vector(np.zeros(n))
    assert lk.gmres(A, oracle_vector(b.copy()), x2, rtol=1e-6, options=lk.gmres_dp_opts(kdim=60, maxiter=20)) > 0
    r = np.empty(n); ora.Lap5Op(N).matvec(x2.data, r)
    assert np.linalg.norm(r - b) < 1e-6 * np.linalg.norm(b) * 1.01


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_generic_eigs_equals_oracle_and_known_answer(dtype):
    n = 128
    if np.dtype(dtype).kind == "c":
        A = np.zeros((n, n), dtype=dtype)
        for i in range(1, n + 1):
            A[i - 1, i - 1] = n
            if i < n:
                A[i - 1, i] = 1j * np.sqrt(1.0 * i * (n - i)); A[i, i - 1] = -A[i - 1, i]
    else:
        A = 0.37 * np.eye(n) + 0.61 * np.eye(n, k=1) - 0.61 * np.eye(n, k=-1)
    nev, kdim = 6, 40
    x0 = seeded(n, dtype, 3)
    X = list_basis(n, nev, dtype)
    vals, res, info = lk.eigs(oracle_dense_linop(A), X, x0=oracle_vector(x0.copy()), kdim=kdim, tolerance=1e-10)
    vo, ro, Vo, info_o = ora.eigs(ora.DenseOp(A), x0, nev, kdim, 1e-10)
    assert info == info_o
    np.testing.assert_allclose(vals, vo, rtol=1e-12)
    np.testing.assert_allclose(res, ro, rtol=1e-6, atol=1e-14)
    V = np.stack([x.data for x in X], axis=1)
    np.testing.assert_allclose(V, Vo, rtol=1e-9, atol=1e-12)
    exact = np.linalg.eigvals(A)
    for lam in vals:
        assert np.abs(exact - lam).min() < 1e-9 * np.abs(exact).max()


def test_krylov_schur_residual():
    """test/TestKrylov.fypp:298-347: after the restart A X(:n) = X(:n+1) H(:n+1, :n) still holds."""
    n, kdim = 128, 32
    rng = np.random.default_rng(4)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    X = list_basis(n, kdim + 1, np.float64)
    x0 = seeded(n, np.float64, 8); X[0].data[:] = x0 / np.linalg.norm(x0)
    H = np.zeros((kdim + 1, kdim), order="F")
    op = oracle_dense_linop(A)
    assert lk.arnoldi(op, X, H) == 0
    nsel = lk.krylov_schur(X, H, lambda lam: np.abs(lam) > np.median(np.abs(lam)))
    assert 0 < nsel < kdim
    Xm = np.stack([x.data for x in X], axis=1)
    assert np.abs(A @ Xm[:, :nsel] - Xm[:, :nsel + 1] @ H[:nsel + 1, :nsel]).max() < 1e-12
    assert np.abs(Xm[:, :nsel + 1].T @ Xm[:, :nsel + 1] - np.eye(nsel + 1)).max() < 1e-12
    assert not Xm[:, nsel + 1:].any()


def test_row_partition_covers_every_row_once():
    for n in (0, 1, 7, 10, 10**8, 10**8 + 3):
        for P in (1, 2, 3, 4, 8):
            blocks = [lk.row_partition(n, P, r) for r in range(P)]
            assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == n
            for (r0, nl), (r1, _) in zip(blocks, blocks[1:]):
                assert r0 + nl == r1 and nl % 2 == 0


def test_generic_bidiagonalization_equals_oracle_and_strang_known_answer():
    """test/TestKrylov.fypp:365-429 (A V = U B, orthonormal bases) and the Strang-matrix singular values
    2 - 2 cos(i pi/(n+1)) of test/TestIterativeSolvers.fypp:444-452, 479-485 (svds' known answer)."""
    n = 64
    A = 2.0 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    u0 = seeded(n, np.float64, 21); u0 /= np.linalg.norm(u0)
    U = list_basis(n, n + 1, np.float64); U[0].data[:] = u0
    V = list_basis(n, n + 1, np.float64)
    B = np.zeros((n + 1, n), order="F")
    info = lk.bidiagonalization(oracle_dense_linop(A), U, V, B)
    Uo = np.zeros((n, n + 1), order="F"); Uo[:, 0] = u0
    Vo = np.zeros((n, n + 1), order="F")
    Bo = np.zeros((n + 1, n), order="F")
    assert info == ora.bidiagonalization(ora.DenseOp(A), ora.DenseOp(A.T.copy()), Uo, Vo, Bo)
    k = info if info > 0 else n
    np.testing.assert_allclose(B[:k, :k], Bo[:k, :k], rtol=1e-12, atol=1e-14)
    sv = np.sort(np.linalg.svd(B[:k, :k], compute_uv=False))[::-1]
    true = np.sort(np.array([2.0 - 2.0 * np.cos(i * np.pi / (n + 1)) for i in range(1, n + 1)]))[::-1]
    assert k == n and np.abs(sv - true).max() < lk.rtol_dp
    Um = np.stack([u.data for u in U[:k]], axis=1)
    Vm = np.stack([v.data for v in V[:k]], axis=1)
    assert np.abs(Um.T @ Um - np.eye(k)).max() < 1e-12 and np.abs(Vm.T @ Vm - np.eye(k)).max() < 1e-12


def test_on_disk_outputs_follow_the_reference_layout(tmp_path, monkeypatch):
    """write_results / save_eigenspectrum (IterativeSolvers.fypp:881-963): header, Fortran E16.9 columns,
    sort by residual, n x 3 npy layout read back by example/ginzburg_landau/eigenplots.py."""
    vals = np.array([1.5 - 0.25j, -2.0 + 0.0j, 0.001 + 3.0j])
    res = np.array([1e-3, 1e-12, 5e-9])
    f = tmp_path / "eigs_output.txt"
    r2 = res.copy()
    lk.write_results(str(f), vals, r2, 1e-8)
    lines = f.read_text().splitlines()
    assert lines[0] == "  Iter                Re                Im           modulus          residual  conv"
    assert lines[1] == "     3  -0.200000000E+01   0.000000000E+00   0.200000000E+01   0.100000000E-11     T"
    assert lines[2].endswith("0.500000000E-08     T") and lines[3].endswith("0.100000000E-02     F")
    assert np.array_equal(r2, np.sort(res))                    # the reference sorts `res` in place
    npy = tmp_path / "spectrum.npy"
    lk.save_eigenspectrum(vals, res, str(npy))
    arr = np.load(npy)
    assert arr.shape == (3, 3) and np.array_equal(arr[:, 0], vals.real) and np.array_equal(arr[:, 2], res)
    lk.save_eigenspectrum(vals.real, res, str(npy))
    assert np.load(npy).shape == (3, 2)
    # eigs(write_intermediate=True) writes eigs_output.txt in the working directory every step
    monkeypatch.chdir(tmp_path)
    n = 64
    A = 0.37 * np.eye(n) + 0.61 * np.eye(n, k=1) - 0.61 * np.eye(n, k=-1)
    X = list_basis(n, 2, np.float64)
    lk.eigs(oracle_dense_linop(A), X, x0=oracle_vector(seeded(n, np.float64, 3)), kdim=20, tolerance=1e-8,
            write_intermediate=True)
    out = (tmp_path / "eigs_output.txt").read_text().splitlines()
    assert out[0].split() == ["Iter", "Re", "Im", "modulus", "residual", "conv"] and len(out) >= 3


def test_verify_vector_axioms_harness_accepts_a_good_type_and_rejects_a_broken_one():
    """AbstractVectors.fypp:733-927 / test/TestVectors.fypp:50-60 (test_size = 128)."""
    good = oracle_vector(np.zeros(128))
    assert lk.verify_vector_axioms(good, ntrials=20)
    assert lk.verify_vector_axioms(oracle_vector(np.zeros(128, dtype=np.complex128)), ntrials=20)

    class broken(oracle_vector):                       # axpby that forgets beta
        def zeros_like(self):
            return broken(np.zeros_like(self.data))

        def axpby(self, alpha, vec, beta):
            self.data[:] = alpha * vec.data + self.data
    assert not lk.verify_vector_axioms(broken(np.zeros(128)), ntrials=5)


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_composite_linops(dtype):
    """Id / scaled / axpby / adjoint operators against matrix formulas (test/TestLinops.fypp:186-420)."""
    n = 40
    rng = np.random.default_rng(6)
    cplx = np.dtype(dtype).kind == "c"
    mk = lambda: (rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)).astype(dtype)  # noqa: E731
    A, B = mk(), mk()
    x = seeded(n, dtype, 1)
    vin = oracle_vector(x.copy())
    out = oracle_vector(np.zeros(n, dtype=dtype))
    opA, opB = oracle_dense_linop(A), oracle_dense_linop(B)
    sigma, al, be = (0.7 - 0.2j, 1.5 + 0.5j, -0.25j) if cplx else (0.7, 1.5, -0.25)

    lk.Id().apply_matvec(vin, out); np.testing.assert_array_equal(out.data, x)
    S = lk.scaled_linop(opA, sigma)
    S.apply_matvec(vin, out); np.testing.assert_allclose(out.data, sigma * (A @ x), rtol=1e-13)
    S.apply_rmatvec(vin, out); np.testing.assert_allclose(out.data, sigma * (A.conj().T @ x), rtol=1e-13)
    Cop = lk.axpby_linop(opA, opB, al, be, transA=False, transB=True)
    Cop.apply_matvec(vin, out); np.testing.assert_allclose(out.data, al * (A @ x) + be * (B.conj().T @ x), rtol=1e-12)
    Cop.apply_rmatvec(vin, out); np.testing.assert_allclose(out.data, al * (A.conj().T @ x) + be * (B @ x), rtol=1e-12)
    T = lk.adjoint_linop(opA)
    T.apply_matvec(vin, out); np.testing.assert_allclose(out.data, A.conj().T @ x, rtol=1e-13)
    T.apply_rmatvec(vin, out); np.testing.assert_allclose(out.data, A @ x, rtol=1e-13)
    assert opA.matvec_counter == 3 and opA.rmatvec_counter == 3 and T.matvec_counter == 1


def test_grid_partition_covers_the_grid_in_whole_lines():
    """Row-sharded 5-point Laplacian: every rank owns whole grid lines, contiguous, together exactly N of them."""
    for N in (1, 7, 64, 4096):
        for nranks in (1, 2, 3, 8):
            if nranks > N:
                continue
            parts = [lk.grid_partition(N, nranks, r) for r in range(nranks)]
            assert parts[0][0] == 0 and sum(nj for _j0, nj in parts) == N
            for (j0, nj), (j1, _n1) in zip(parts, parts[1:]):
                assert nj >= 1 and j0 + nj == j1


def test_threadable_gesdd_is_bit_identical_to_scipys_svd():
    """_hostlapack.gesdd = LAPACK gesdd (jobz 'A', workspace query) through ctypes: the bits of scipy.linalg.svd on the lower
    bidiagonal matrices svds decomposes, from one thread or from eight at once."""
    from concurrent.futures import ThreadPoolExecutor
    from scipy.linalg import svd
    from lightkrylov_amd import _hostlapack as hl
    rng = np.random.default_rng(8)
    Br = np.tril(np.triu(rng.standard_normal((41, 40)), -1))
    Bz = Br + 1j * np.tril(np.triu(rng.standard_normal((41, 40)), -1))

    def check(k):
        ok = True
        for B in (Br, Bz):
            u, s_, vh = hl.gesdd(np.asfortranarray(B[:k, :k]))
            u2, s2, vh2 = svd(B[:k, :k])
            ok = ok and np.array_equal(u, u2) and np.array_equal(s_, s2) and np.array_equal(vh, vh2)
        return ok

    assert all(check(k) for k in range(1, 41))
    with hl.blas_threads(1), ThreadPoolExecutor(8) as pool:
        assert all(pool.map(check, list(range(1, 41)) * 2))


def test_threadable_syev_is_bit_identical_to_scipys_wrapper():
    """_hostlapack.syev = LAPACK syev / heev on the UPPER triangle (stdlib's eigh, EIGHS/eighs.fypp:87) through ctypes: the
    bits of scipy.linalg.eigh(lower=False, driver="ev"), from one thread or from eight at once; the lower triangle is not read."""
    from concurrent.futures import ThreadPoolExecutor
    from scipy.linalg import eigh
    from lightkrylov_amd import _hostlapack as hl
    rng = np.random.default_rng(6)
    Tr = rng.standard_normal((40, 40)); Tr = (Tr + Tr.T) / 2
    Tz = rng.standard_normal((40, 40)) + 1j * rng.standard_normal((40, 40)); Tz = (Tz + Tz.conj().T) / 2

    def check(k):
        ok = True
        for T in (Tr, Tz):
            w, v = hl.syev(T[:k, :k])
            w2, v2 = eigh(T[:k, :k], lower=False, driver="ev")
            junk = np.triu(T[:k, :k]) + np.tril(np.random.default_rng(k).standard_normal((k, k)), -1)   # garbage below the diagonal
            w3, v3 = hl.syev(junk.astype(T.dtype))
            ok = ok and np.array_equal(w, w2) and np.array_equal(v, v2) and np.array_equal(w, w3) and np.array_equal(v, v3)
        return ok

    assert all(check(k) for k in range(1, 41))
    with hl.blas_threads(1), ThreadPoolExecutor(8) as pool:
        assert all(pool.map(check, list(range(1, 41)) * 2))


def test_threadable_geev_is_bit_identical_to_scipys_wrapper():
    """lightkrylov_amd._hostlapack.geev calls the same OpenBLAS routine scipy.linalg.lapack.{d,z}geev calls, with the
    same workspace size, outside the interpreter lock: same bits, from one thread or from eight at once."""
    from concurrent.futures import ThreadPoolExecutor
    from scipy.linalg import lapack
    from lightkrylov_amd import _hostlapack as hl
    rng = np.random.default_rng(5)
    Hr = np.asfortranarray(np.triu(rng.standard_normal((41, 40)), -1))
    Hz = np.asfortranarray(np.triu(rng.standard_normal((41, 40)) + 1j * rng.standard_normal((41, 40)), -1))

    def check(k):
        vr, vals = hl.geev(Hz[:k, :k])
        w, _vl, v2, info = lapack.zgeev(np.asfortranarray(Hz[:k, :k]), compute_vl=0, compute_vr=1)
        ok = info == 0 and np.array_equal(vals, w) and np.array_equal(vr, v2)
        vr, vals = hl.geev(Hr[:k, :k])
        wr, wi, _vl, v2, info = lapack.dgeev(np.asfortranarray(Hr[:k, :k]), compute_vl=0, compute_vr=1)
        return ok and info == 0 and np.array_equal(vals, wr + 1j * wi) and np.array_equal(vr, v2)

    assert all(check(k) for k in range(1, 41))
    with hl.blas_threads(1), ThreadPoolExecutor(8) as pool:
        assert all(pool.map(check, list(range(1, 41)) * 3))


def test_threadable_gees_is_bit_identical_to_scipys_wrapper():
    """_hostlapack.gees = LAPACK gees (jobvs 'V', sort 'N', scipy's default lwork = 3 n) through ctypes: the bits of
    scipy.linalg.lapack.{d,z}gees -- T, Z and the eigenvalues -- from one thread or from eight at once.  eigs starts the restart's Schur
    factorisation with it beside the last Ritz tests of a cycle (the scipy wrapper takes a python callback and holds the interpreter
    lock for the whole call)."""
    from concurrent.futures import ThreadPoolExecutor
    from scipy.linalg import lapack
    from lightkrylov_amd import _hostlapack as hl
    rng = np.random.default_rng(6)
    Hr = np.asfortranarray(np.triu(rng.standard_normal((40, 40)), -1))
    Hz = np.asfortranarray(np.triu(rng.standard_normal((40, 40)) + 1j * rng.standard_normal((40, 40)), -1))

    def check(k):
        T, Z, w = hl.gees(Hz[:k, :k])
        T0, _s, w0, Z0, _wk, info = lapack.zgees(lambda *a: False, np.asfortranarray(Hz[:k, :k]), sort_t=0)
        ok = info == 0 and np.array_equal(T, T0) and np.array_equal(Z, Z0) and np.array_equal(w, w0)
        T, Z, w = hl.gees(Hr[:k, :k])
        T0, _s, wr, wi, Z0, _wk, info = lapack.dgees(lambda *a: False, np.asfortranarray(Hr[:k, :k]), sort_t=0)
        return ok and info == 0 and np.array_equal(T, T0) and np.array_equal(Z, Z0) and np.array_equal(w, wr + 1j * wi)

    assert all(check(k) for k in range(1, 41))
    with hl.blas_threads(1), ThreadPoolExecutor(8) as pool:
        assert all(pool.map(check, list(range(1, 41)) * 2))


def test_oracle_cg_eighs_svds_known_answers():
    """The oracle's restatements of the three thin solver loops against numpy: cg solves an SPD system, eighs finds the
    leading eigenvalues, svds the leading singular values (the reference's own tests check the same invariants:
    test/TestIterativeSolvers.fypp, cg / eighs / svds sections)."""
    from oracle import oracle as ora
    rng = np.random.default_rng(0)
    n = 300
    M = rng.standard_normal((n, n)) / np.sqrt(n)
    A = np.asfortranarray(M.T @ M + np.eye(n))
    A[:3, :3] += np.diag([8.0, 6.0, 4.0])
    b = rng.standard_normal(n)
    x = np.zeros(n)
    info, res = ora.cg(ora.DenseOp(A), b, x, rtol=1e-10)
    assert info > 0 and np.abs(A @ x - b).max() <= 1e-8 and (np.diff(np.log(res[::5])) < 0).all()
    vals, r, X, k = ora.eighs(ora.DenseOp(A), b.copy(), 3, kdim=40, tolerance=1e-10)
    assert np.abs(vals - np.sort(np.linalg.eigvalsh(A))[::-1][:3]).max() <= 1e-9 and (r < 1e-10).all()
    assert np.abs(X.T @ X - np.eye(3)).max() <= 1e-10
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    G[:3, :3] += np.diag([9.0, 7.0, 5.0])
    G = np.asfortranarray(G)
    S, r, U, V, k = ora.svds(ora.DenseOp(G), ora.DenseOp(np.asfortranarray(G.T)), b.copy(), 3, kdim=40, tolerance=1e-10)
    assert np.abs(S - np.linalg.svd(G, compute_uv=False)[:3]).max() <= 1e-9
    assert np.abs(G @ V - U * S).max() <= 1e-8


def test_bench_helpers_csr_rows_self_launch_and_traffic_record(monkeypatch):
    """bench.py's host-side pieces: the rows of the 5-point Laplacian it hands to the row-sharded CSR operator equal scipy's
    matrix; `--gpus N` without a launcher's environment composes a torch.distributed.run child (loopback rendezvous, NCCL_ALGO
    pinned); and the committed PMC traffic record describes THIS build's kernel sources (bench.py refuses any other)."""
    import json
    import subprocess
    import sys
    import scipy.sparse as sp
    import bench
    N = 9
    T = sp.diags([-1.0, 4.0, -1.0], [-1, 0, 1], shape=(N, N))
    L = ((sp.kron(sp.identity(N), T) + sp.kron(sp.diags([-1.0, -1.0], [-1, 1], shape=(N, N)), sp.identity(N))) * float((N + 1) ** 2)).tocsr()
    for row0, nl in ((0, N * N), (18, 27), (72, 9)):
        rp, ci, v = bench._laplacian_csr_rows(N, row0, nl)
        assert rp[0] == 0 and ci.dtype == np.int32 and (np.diff(ci[rp[0]:rp[1]]) > 0).all()
        assert abs(sp.csr_matrix((v, ci, rp), shape=(nl, N * N)) - L[row0:row0 + nl]).max() == 0.0
    seen = {}

    def fake_run(cmd, env=None, cwd=None):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("NCCL_ALGO", raising=False)
    assert bench._self_launch(4) == 7                                   # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    # the launcher's own store picks and holds the port (--standalone); the rendezvous stays on loopback
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["NCCL_ALGO"] == "Ring" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one record per rows-per-rank: the metric's N = 1 / 2 / 4 / 8 shards, each measured on THIS build's kernel sources
    doc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")))
    by_n = {r["n_local"]: r for r in doc["records"]}
    assert set(by_n) >= {100_000_000, 50_000_000, 25_000_000, 12_500_000}
    for n_local, rec in by_n.items():
        assert rec["kernel_source_sha256"] == bench.kernel_source_hash(), (
            f"profiles/pmc_traffic.json (n_local = {n_local}) was measured on other kernel sources: re-run tools/run_profiles.sh + tools/make_profiles.py")
        assert 0.98 <= rec["traffic_over_algorithmic"] <= 1.02
        traffic, src = bench.load_traffic_record(n_local, 128, "f64")
        assert traffic == rec["hbm_bytes_per_launch"] and src["kernel_source_sha256"] == rec["kernel_source_sha256"]
    assert bench.load_traffic_record(12345, 128, "f64")[0] is None           # no record: null, with the reason
    # the scaling model printed next to `value`: T1 / N + what does not shrink
    p1, p8 = bench.predicted_iters_per_s(100_000_000, 128, 1, "diag", "f64"), bench.predicted_iters_per_s(100_000_000, 128, 8, "diag", "f64")
    assert p1["predicted_it_s"] > 35 and 6.5 * p1["predicted_it_s"] < p8["predicted_it_s"] < 8 * p1["predicted_it_s"]
    assert bench.predicted_iters_per_s(10_000_000, 64, 2, "diag", "f64") is None


def test_bench_adopts_the_launchers_world_size_and_ignores_leaked_test_hooks():
    """`torchrun --nproc-per-node N bench.py` without --gpus runs on the launcher's size (only an EXPLICIT disagreeing --gpus is refused);
    the LK_TEST_* fault-injection variables act only with the opt-in LK_BENCH_TEST_HOOKS=1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # no --gpus under a launcher's environment: no refusal, the run proceeds to the engine (which this container cannot start: no GPU)
    env = dict(os.environ, WORLD_SIZE="4", RANK="1", LOCAL_RANK="0", LK_TEST_HANG_RANK="1", LK_BENCH_WATCHDOG="30")
    env.pop("LK_BENCH_TEST_HOOKS", None)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert "refusing to run" not in out.stderr and "bench.py[rank 1/4" in out.stderr
    # (the leaked LK_TEST_HANG_RANK did not put the rank to sleep: it got past "creating the engine context" or failed before, never the watchdog)
    assert "Timeout (" not in out.stderr
    src = open(os.path.join(root, "bench.py")).read()
    assert 'os.environ.get("LK_TEST_' not in src            # every LK_TEST_* hook goes through hook(), i.e. behind LK_BENCH_TEST_HOOKS=1


def test_bench_watchdog_dumps_every_stack_and_exits_nonzero():
    """bench.py's progress markers + watchdog (N > 1 robustness): a rank that stops moving writes where it was (its last phase
    marker), the Python stack of every thread, and leaves with status 1 -- a fresh exit -- within the watchdog's period."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import bench, threading, time\n"
            "threading.Thread(target=time.sleep, args=(60,), daemon=True).start()\n"
            "p = bench.Progress(1.0, 3, 8)\n"
            "p.phase('rendezvous')\n"
            "p.tick()\n"
            "p.phase('stuck here')\n"
            "time.sleep(60)\n")
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=50)
    assert out.returncode == 1 and time.time() - t0 < 30
    assert "bench.py[rank 3/8" in out.stderr and "phase: stuck here" in out.stderr
    assert "Timeout (0:00:01)!" in out.stderr and out.stderr.count("Thread 0x") >= 2 and 'File "<string>", line 7' in out.stderr
    # ... and a run that finishes cancels it
    ok = subprocess.run([sys.executable, "-c", "import bench, time\np = bench.Progress(1.0)\np.phase('a')\np.done()\ntime.sleep(2.5)\n"],
                        cwd=root, capture_output=True, text=True, timeout=50)
    assert ok.returncode == 0 and "Timeout" not in ok.stderr


def test_bench_refuses_a_launcher_environment_that_disagrees_with_gpus():
    """--gpus N under a launcher whose WORLD_SIZE differs (a stale export, a scheduler's variables) is refused with status 2 on
    every rank -- never a line for a job the caller did not ask for."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="1", LOCAL_RANK="1")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "--gpus 8 but WORLD_SIZE=4" in out.stderr and not out.stdout.strip()


def test_eigs_segments_taper_towards_the_end_of_a_cycle():
    """The device segments of the pipelined eigs cycle: full segments of 16 steps, the last 16 steps as 8, 4, 2, 1, 1 -- every step
    exactly once, in order, whatever kstart / kdim (a restarted cycle starts mid-way; a short cycle is all taper)."""
    from lightkrylov_amd.solvers import _tapered_segments
    assert _tapered_segments(1, 128)[-5:] == [(113, 120), (121, 124), (125, 126), (127, 127), (128, 128)]
    assert _tapered_segments(1, 128)[:4] == [(1, 64), (65, 80), (81, 96), (97, 112)]            # the first half in one segment
    assert _tapered_segments(1, 30) == [(1, 14), (15, 22), (23, 26), (27, 28), (29, 29), (30, 30)]
    for kstart in (1, 2, 7, 65, 120, 128):
        for kdim in (1, 2, 5, 16, 17, 31, 33, 128, 200):
            if kstart > kdim:
                continue
            segs = _tapered_segments(kstart, kdim)
            steps = [k for a, b in segs for k in range(a, b + 1)]
            assert steps == list(range(kstart, kdim + 1)), (kstart, kdim, segs)
            assert segs[-1][0] == segs[-1][1] == kdim and all(b - a + 1 <= max(16, (kdim - kstart + 1) // 2) for a, b in segs)


def test_bench_all_core_baseline_leg_runs_bound_in_a_child_process():
    """cpu_baseline's all-core leg: a child process started with OMP_PROC_BIND=spread / OMP_PLACES=cores (the binding is read when the
    OpenMP runtime starts), parallel first touch of the basis, thread scan, the timed sample -- its JSON says how the team was bound."""
    import bench
    leg = bench._fused_leg_in_child(200_000, 4, 100_000_000, 128)
    assert leg["value"] and leg["value"] > 0 and leg["cores"] >= 1, leg
    assert leg["binding"]["OMP_PROC_BIND"] == "spread" and leg["binding"]["OMP_PLACES"] == "cores"
    assert leg["binding"]["distinct_cpus_of_the_team"] == min(leg["cores"], os.cpu_count())       # every thread on a CPU of its own
    assert set(leg["thread_scan_GBps"]) == set(leg["thread_scan_seconds"]) and str(leg["cores"]) in leg["thread_scan_seconds"]
