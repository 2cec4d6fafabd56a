"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them (regression pin) and the reference-run values recorded in
SURVEY.md Appendix A.  GPU: the HIP engine reproduces them at the north_star tolerance (1e-12)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as ora
from tests._tol import assert_close, assert_columns_close, assert_ritz_close
from tests.golden.make_golden import GL_REF, arnoldi_diag, cfg1_matrix, diag_values, gl_reference_size, seeded

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    out = []
    for f in sorted(glob.glob(os.path.join(G, "arnoldi_diag_*.npz"))):
        z = np.load(f)
        out.append((os.path.basename(f), int(z["n"]), int(z["m"]), np.complex128 if "cdp" in f else np.float64))
    return out


def test_oracle_reproduces_the_reference_run_recorded_by_the_survey():
    """the reference's own arnoldi (amdflang build made while surveying) gave
    H(1,1), H(2,1), H(m+1,m) for n=1000, m=8, d_i = 1+(i-1)/n, x0_i = sin(i)/||.||.  All 17 digits."""
    z = np.load(os.path.join(G, "survey_reference_run_n1000_m8.npz"))
    n, m = 1000, 8
    d = 1.0 + np.arange(n) / n
    x0 = np.sin(np.arange(1, n + 1, dtype=float))
    x0 /= np.sqrt(np.sum(x0 ** 2))
    X = np.zeros((n, m + 1), order="F"); X[:, 0] = x0
    H = np.zeros((m + 1, m), order="F")
    assert ora.arnoldi(ora.DiagOp(d), X, H) == 0
    assert H[0, 0] == float(z["H11"]) and H[1, 0] == float(z["H21"]) and H[m, m - 1] == float(z["Hlast"])


@pytest.mark.parametrize("name,n,m,dtype", [c for c in _cases() if c[1] <= 20_011])
def test_oracle_regression_against_fixtures(name, n, m, dtype):
    z = np.load(os.path.join(G, name))
    got = arnoldi_diag(n, m, dtype)
    assert got["info"] == int(z["info"])
    # bit for bit on the host that made the fixtures; on another CPU numpy's own BLAS may normalise x0 an ulp differently (its nrm2 kernel is
    # chosen per architecture), which 128 steps carry into the last columns: there every column still agrees normwise far inside 1e-12
    if not np.array_equal(got["H"], z["H"]):
        err = max(np.abs(got["H"][:, j] - z["H"][:, j]).max() / np.abs(z["H"][:, j]).max() for j in range(m))
        assert err <= 1e-13, err


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,m,dtype", _cases())
def test_engine_reproduces_arnoldi_fixtures(ctx, name, n, m, dtype):
    import lightkrylov_amd as lk
    z = np.load(os.path.join(G, name))
    d = diag_values(n, dtype)
    x0 = seeded(n, dtype, int(z["seed"]))
    x0 /= np.linalg.norm(x0)
    X = lk.krylov_basis_gpu(n, m + 1, dtype, ctx); X.upload(x0.reshape(-1, 1), 0)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")
    assert lk.arnoldi(lk.diag_linop_gpu(d, ctx), X, H) == int(z["info"])
    assert_columns_close(H, z["H"], f"fixture {name}")                                    # Hessenberg, normwise per column, 1e-12
    # Ritz values: 1e-12 * kappa_i * ||H||, kappa_i = the eigenvalue's condition number computed from H (tests/_tol.py)
    assert_ritz_close(np.linalg.eigvals(H[:m, :m]), z["ritz"], H[:m, :m], f"fixture {name}")


@pytest.mark.gpu
def test_engine_reproduces_cfg1_fixture(ctx):
    import lightkrylov_amd as lk
    z = np.load(os.path.join(G, "cfg1_dense1000_m30_rdp.npz"))
    A, x0 = cfg1_matrix()
    op = lk.dense_linop_gpu(A, ctx)
    X = lk.krylov_basis_gpu(1000, 31, np.float64, ctx); X.upload(x0.reshape(-1, 1), 0)
    H = np.zeros((31, 30), order="F")
    assert lk.arnoldi(op, X, H) == int(z["info"])
    assert_columns_close(H, z["H"], "fixture cfg1 (dense 1000 x 1000, m = 30)")
    V = lk.krylov_basis_gpu(1000, 4, np.float64, ctx)
    vals, res, niter = lk.eigs(op, V, x0=lk.dense_vector_gpu.from_array(x0, ctx), kdim=30, tolerance=1e-10)
    assert niter == int(z["eig_niter"])
    # converged eigenvalues of A: conditioning taken from A itself (kappa_i = 1 / |y_i^H x_i| of the 1000 x 1000 matrix)
    assert_ritz_close(vals, z["eig_vals"], A, "fixture cfg1 eigs(nev = 4, kdim = 30)")


def test_oracle_reproduces_the_ginzburg_landau_fixture():
    """SURVEY 8(c) fixture (5): the reference example's own size (nx = 512, nev = 8, kdim = 16, tau = 0.01)."""
    z = np.load(os.path.join(G, "gl_nx512_kdim16_cdp.npz"))
    got = gl_reference_size()
    assert np.array_equal(got["H"], z["H"]) and got["info"] == int(z["info"]) and got["eig_niter"] == int(z["eig_niter"])
    assert np.array_equal(got["eig_vals"], z["eig_vals"])
    assert z["H"].shape == (17, 16) and len(z["eig_vals"]) == 8
    # the 8 leading eigenvalues of the propagator map back to growth rates log(lambda) / tau with decreasing real part
    lam = np.log(z["eig_vals"]) / GL_REF["tau"]
    assert (np.diff(lam.real) < 0).all()


@pytest.mark.gpu
def test_engine_reproduces_the_ginzburg_landau_fixture(ctx):
    """The engine on the reference example's own configuration (Ginzburg_Landau.f90:23-33, main.f90:20-66): H(:17, :16) of the
    first factorisation at 1e-12 per column, its Ritz values and the 8 eigenvalues eigs returns at 1e-12 * kappa."""
    import lightkrylov_amd as lk
    z = np.load(os.path.join(G, "gl_nx512_kdim16_cdp.npz"))
    g = GL_REF
    A = lk.ginzburg_landau_linop_gpu(g["n"], ctx, tau=g["tau"], nsub=g["nsub"], nu=g["nu"], gamma=g["gamma"], mu_0=0.38, c_mu=0.2,
                                     mu_2=g["mu2"], dx=g["dx"])
    assert abs(A.params["mu_c"] - g["mu_c"]) < 1e-16
    x0 = seeded(g["n"], np.complex128, g["seed"])
    m = g["kdim"]
    X = lk.krylov_basis_gpu(g["n"], m + 1, np.complex128, ctx); X.upload((x0 / np.linalg.norm(x0)).reshape(-1, 1), 0)
    H = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    assert lk.arnoldi(A, X, H) == int(z["info"])
    assert_columns_close(H, z["H"], "fixture GL nx = 512, kdim = 16")
    assert_ritz_close(np.linalg.eigvals(H[:m, :m]), z["ritz"], H[:m, :m], "fixture GL nx = 512, kdim = 16")
    V = lk.krylov_basis_gpu(g["n"], g["nev"], np.complex128, ctx)
    vals, res, niter = lk.eigs(A, V, x0=lk.dense_vector_gpu.from_array(x0, ctx), kdim=m)
    assert niter == int(z["eig_niter"])
    # conditioning of the converged eigenvalues from the operator's own matrix (512 columns of the propagator, from the oracle)
    Ao = ora.GLOp(g["n"], g["dx"], g["tau"], g["nsub"], g["nu"], g["gamma"], g["mu_c"], g["mu2"])
    P = np.stack([Ao.apply(e) for e in np.eye(g["n"], dtype=np.complex128)], axis=1)
    assert_ritz_close(vals, z["eig_vals"], P, "fixture GL nx = 512 eigs(nev = 8, kdim = 16)")


@pytest.mark.gpu
def test_engine_reproduces_gmres_fixture(ctx):
    import lightkrylov_amd as lk
    z = np.load(os.path.join(G, "gmres_poisson64_k30.npz"))
    N = 64
    b = seeded(N * N, np.float64, 11)
    x = lk.dense_vector_gpu(N * N, np.float64, ctx)
    meta = lk.gmres_dp_metadata()
    info = lk.gmres(lk.laplacian2d_linop_gpu(N, ctx), lk.dense_vector_gpu.from_array(b, ctx), x, rtol=1e-8,
                    options=lk.gmres_dp_opts(kdim=30, maxiter=2), meta=meta)
    assert info == int(z["info"]) and len(meta.res) == len(z["res"])
    assert_close(np.array(meta.res), z["res"], "gmres Poisson 64 fixture: residual history", scale=z["res"][0])
    xa = x.to_array()
    assert_close(np.linalg.norm(xa), float(z["x_norm"]), "gmres Poisson 64 fixture: |x|")
    assert_close(xa[:64], z["x_head"], "gmres Poisson 64 fixture: head of x", scale=np.abs(xa).max())
