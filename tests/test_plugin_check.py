"""The LightKrylov type-extension plugin (fortran/dense_vector_gpu.f90) compiles and links against the reference's
real module interfaces, and the reference's own arnoldi / gmres run unchanged through it (host mock of the C ABI):
tools/check_plugin.sh.  Build-container only -- needs /root/reference and amdflang; skipped elsewhere (the GPU box
has no reference tree).  Not a parity pin: see the script's header."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("LK_PLUGIN_CHECK_DIR", "/tmp/lk_plugin_check")
SOURCES = ["fortran/dense_vector_gpu.f90", "fortran/lk_hip_iso_c.f90", "tools/check_plugin.sh",
           "tools/plugin_check/driver.f90", "examples/fortran/gmres_dense.f90", "tools/plugin_check/mock_abi.c", "tools/plugin_check/gen_stdlib_stubs.py",
           "include/lightkrylov_hip.h"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src") or not os.path.exists("/opt/rocm/bin/amdflang"),
                    reason="needs the reference tree and amdflang (build container only)")
def test_plugin_compiles_links_and_runs_the_reference_solvers():
    stamp = os.path.join(OUT, "ok.stamp")
    newest = max(os.path.getmtime(os.path.join(ROOT, s)) for s in SOURCES)
    if os.path.exists(stamp) and os.path.getmtime(stamp) >= newest:
        return                                        # unchanged since the last green run (the check takes ~2 min)
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_plugin.sh")], capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "plugin_driver: all checks passed" in out.stdout and "FAIL" not in out.stdout
    open(stamp, "w").write(out.stdout)


def _read_cases(path):
    """ref_runs.txt: 'case <name> <ints...>' lines, each followed by full-precision numbers (one real or one
    (re, im) pair per line)."""
    cases, name = {}, None
    for line in open(path):
        t = line.split()
        if t and t[0] == "case":
            name = t[1]
            cases[name] = ([int(v) for v in t[2:]], [])
        elif t:
            cases[name][1].append([float(v) for v in t])
    return cases


@pytest.mark.skipif(not os.path.isdir("/root/reference/src") or not os.path.exists("/opt/rocm/bin/amdflang"),
                    reason="needs the reference tree and amdflang (build container only)")
def test_oracle_follows_runs_of_the_reference_control_flow():
    """The driver also runs the REFERENCE's own arnoldi / double_gram_schmidt_step / gmres on its own dense_vector
    and dumps inputs + outputs (ref_runs.txt).  The oracle, fed the same inputs, must reproduce them to rounding
    (the stand-in BLAS-1 loops are the textbook ones the oracle restates, so the Arnoldi / DGS cases agree to the
    last few ulps).  Execution evidence that the restatement follows the reference's control flow -- indicative
    only, because the build uses stand-ins for fortran-lang/stdlib; it is NOT counted as an oracle pin."""
    import numpy as np
    from oracle import oracle as ora
    test_plugin_compiles_links_and_runs_the_reference_solvers()
    path = os.path.join(OUT, "ref_runs.txt")
    assert os.path.exists(path)
    cases = _read_cases(path)

    # arnoldi, real diagonal operator
    (n, m, info), rows = cases["arnoldi_rdp_diag"]
    v = np.array(rows)[:, 0]
    d, x0, H = v[:n], v[n:2 * n], v[2 * n:].reshape(m + 1, m, order="F")
    X = np.zeros((n, m + 1), order="F"); X[:, 0] = x0
    Ho = np.zeros((m + 1, m), order="F")
    assert ora.arnoldi(ora.DiagOp(d), X, Ho) == info == 0
    err = np.abs(Ho - H).max() / np.abs(H).max()
    assert err <= 4e-16, err

    # double_gram_schmidt_step against that basis
    (n, k, info), rows = cases["dgs_rdp"]
    v = np.array(rows)[:, 0]
    y, Xr = v[:n].copy(), np.asfortranarray(v[n:n + n * k].reshape(n, k, order="F"))
    beta, yout = v[n + n * k:n + n * k + k], v[n + n * k + k:]
    assert np.abs(Xr - X[:, :k]).max() <= 4e-16            # the oracle's basis IS the reference's basis
    h, oinfo = ora.double_gram_schmidt_step(y, Xr)
    assert oinfo == info
    assert np.array_equal(h, beta) and np.array_equal(y, yout)          # same inputs: bit for bit

    # arnoldi, complex dense operator
    (n, m, info), rows = cases["arnoldi_cdp_dense"]
    v = np.array(rows); v = v[:, 0] + 1j * v[:, 1]
    A, x0, H = v[:n * n].reshape(n, n, order="F"), v[n * n:n * n + n], v[n * n + n:].reshape(m + 1, m, order="F")
    X = np.zeros((n, m + 1), dtype=complex, order="F"); X[:, 0] = x0
    Ho = np.zeros((m + 1, m), dtype=complex, order="F")
    assert ora.arnoldi(ora.DenseOp(A), X, Ho) == info == 0
    err = np.abs(Ho - H).max() / np.abs(H).max()
    assert err <= 1e-14, err

    # gmres(10), maxiter = 2, real dense operator: info, residual history, solution
    (n, kdim, info, nres), rows = cases["gmres_rdp_dense"]
    v = np.array(rows)[:, 0]
    A, b = v[:n * n].reshape(n, n, order="F"), v[n * n:n * n + n]
    x, res = v[n * n + n:n * n + 2 * n], v[n * n + 2 * n:]
    assert res.size == nres
    xo = np.zeros(n)
    oinfo, ores = ora.gmres(ora.DenseOp(A), b.copy(), xo, rtol=1e-10, atol=1e-14, kdim=kdim, maxiter=2)
    assert oinfo == info, (oinfo, info)
    assert ores.size == res.size
    assert np.abs(ores - res).max() <= 1e-12 * res[0]
    assert np.abs(xo - x).max() <= 1e-12 * np.abs(x).max()
