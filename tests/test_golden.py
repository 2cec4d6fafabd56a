"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them (regression pin) and the reference-run values recorded in
SURVEY.md Appendix A.  GPU: the HIP engine reproduces them at the north_star tolerance (1e-12)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as ora
from tests.golden.make_golden import arnoldi_diag, cfg1_matrix, diag_values, seeded

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
    assert np.array_equal(got["H"], z["H"]) and got["info"] == int(z["info"])


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
    Ho = z["H"]
    for j in range(m):
        assert np.abs(H[:, j] - Ho[:, j]).max() <= 1e-12 * np.abs(Ho[:, j]).max()        # Hessenberg, normwise per column
    ritz = np.sort_complex(np.linalg.eigvals(H[:m, :m]))
    assert np.abs(ritz - z["ritz"]).max() <= 1e-12 * np.abs(z["ritz"]).max() * (10 if m > 100 else 1)   # Ritz values


@pytest.mark.gpu
def test_engine_reproduces_cfg1_fixture(ctx):
    import lightkrylov_amd as lk
    z = np.load(os.path.join(G, "cfg1_dense1000_m30_rdp.npz"))
    A, x0 = cfg1_matrix()
    op = lk.dense_linop_gpu(A, ctx)
    X = lk.krylov_basis_gpu(1000, 31, np.float64, ctx); X.upload(x0.reshape(-1, 1), 0)
    H = np.zeros((31, 30), order="F")
    assert lk.arnoldi(op, X, H) == int(z["info"])
    for j in range(30):
        assert np.abs(H[:, j] - z["H"][:, j]).max() <= 1e-11 * np.abs(z["H"][:, j]).max()
    V = lk.krylov_basis_gpu(1000, 4, np.float64, ctx)
    vals, res, niter = lk.eigs(op, V, x0=lk.dense_vector_gpu.from_array(x0, ctx), kdim=30, tolerance=1e-10)
    assert niter == int(z["eig_niter"])
    assert np.abs(vals - z["eig_vals"]).max() <= 1e-10 * np.abs(z["eig_vals"]).max()


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
    assert np.abs(np.array(meta.res) - z["res"]).max() <= 1e-10 * z["res"][0]
    xa = x.to_array()
    assert abs(np.linalg.norm(xa) - float(z["x_norm"])) <= 1e-10 * float(z["x_norm"])
    assert np.abs(xa[:64] - z["x_head"]).max() <= 1e-10 * np.abs(z["x_head"]).max()
