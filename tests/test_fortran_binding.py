"""The ISO_C_BINDING layer (fortran/lk_hip_iso_c.f90) drives the engine from a Fortran host program
(fortran/test_iso_c.f90, built by __graft_entry__.build() with amdflang): same case the survey ran
through the reference's own arnoldi -- compare with those recorded values."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fortran", "test_iso_c")


def test_interface_module_declares_the_c_abi():
    """Every bind(C) name in the Fortran interface module is a symbol of the C header."""
    import re
    f90 = open(os.path.join(ROOT, "fortran", "lk_hip_iso_c.f90")).read()
    hdr = open(os.path.join(ROOT, "include", "lightkrylov_hip.h")).read()
    names = set(re.findall(r'bind\(C, name="(lk_[a-z0-9_]+)"\)', f90))
    assert len(names) >= 25
    for nme in names:
        assert re.search(rf"\b{nme}\s*\(", hdr), f"{nme} is not declared in the C header"


@pytest.mark.gpu
def test_fortran_host_program_runs_arnoldi_on_the_gpu():
    if not os.path.exists(EXE):
        pytest.fail("fortran/test_iso_c not built: run __graft_entry__.build()")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "lightkrylov_amd"), "/opt/rocm/lib", "/opt/rocm/lib/llvm/lib",
                                              env.get("LD_LIBRARY_PATH", "")])
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    vals = {ln.split()[0]: float(ln.split()[1]) for ln in out.stdout.splitlines() if len(ln.split()) == 2}
    from oracle import oracle as ora
    n, m = 1000, 8
    z = np.load(os.path.join(ROOT, "tests", "golden", "survey_reference_run_n1000_m8.npz"))
    assert vals["info"] == 0
    for key, ref in (("H11", float(z["H11"])), ("H21", float(z["H21"])), ("Hlast", float(z["Hlast"]))):
        assert abs(vals[key] - ref) <= 1e-12 * abs(ref), (key, vals[key], ref)
    assert vals["orth"] < 1e-13
    assert abs(vals["norm_last"] - 1.0) < 1e-14
    # lk_arnoldi_segments with a Fortran bind(C) progress function: three reports (1..3, 4..6, 7..8), in order, the same H bit for bit
    assert (vals["seg_info"], vals["seg_calls"], vals["seg_last"], vals["seg_in_order"]) == (0, 3, 8, 1) and vals["seg_H_diff"] == 0.0
    # the per-object (type-bound-procedure) schedule driven from Fortran in lazy mode, compared with the fused single-pass call.
    # m = 8 dots: one batched sweep + 7 memo hits; the 8 axpbys are queued onto a virtual proj; y%sub(proj) + y%norm()
    # = one fused sweep (the norm is the 8th memo hit); nothing was flushed as a plain panel update, proj never written
    assert (vals["lazy_sweeps"], vals["lazy_hits"], vals["lazy_queued"], vals["lazy_flushes"]) == (1, 8, 8, 0)
    assert (vals["lazy_fused_sweeps"], vals["lazy_temporaries_written"]) == (1, 0)
    assert vals["lazy_h_err"] < 1e-13 and vals["lazy_y_err"] < 1e-13
    # complex(dp) pass: the same case through the oracle
    from oracle import oracle as ora
    n, m = 1000, 8
    i = np.arange(1, n + 1, dtype=np.float64)
    dz = (1.0 + (i - 1) / n) + 0.25j * np.sin(i)
    x0 = np.sin(i) + 1j * np.cos(2 * i)
    x0 /= np.sqrt(np.sum(np.abs(x0) ** 2))
    Xo = np.zeros((n, m + 1), dtype=np.complex128, order="F"); Xo[:, 0] = x0
    Ho = np.zeros((m + 1, m), dtype=np.complex128, order="F")
    assert ora.arnoldi(ora.DiagOp(dz.astype(np.complex128)), Xo, Ho) == 0 and vals["z_info"] == 0
    got11, got12 = complex(vals["z_H11_re"], vals["z_H11_im"]), complex(vals["z_H12_re"], vals["z_H12_im"])
    assert abs(got11 - Ho[0, 0]) <= 1e-12 * abs(Ho[0, 0])           # libm sin/cos of the two hosts may differ by an ulp
    assert abs(got12 - Ho[0, 1]) <= 1e-12 * np.abs(Ho[:, 1]).max()
    assert abs(vals["z_Hlast"] - Ho[m, m - 1].real) <= 1e-12 * abs(Ho[m, m - 1])
    # round 6: lk_lanczos, lk_bidiag, lk_qr and lk_arnoldi_block from the Fortran host, every entry against the oracle at 1e-12 (normwise per column)
    i = np.arange(1, n + 1, dtype=np.float64)
    dr = 1.0 + (i - 1) / n
    xr = np.sin(i)
    xr /= np.sqrt(np.sum(xr ** 2))
    def entries(prefix, rows, cols):
        M = np.zeros((rows, cols))
        for a in range(rows):
            for b in range(cols):
                M[a, b] = vals[f"{prefix}_{a + 1}_{b + 1}"]
        return M
    Xo = np.zeros((n, m + 1), order="F"); Xo[:, 0] = xr
    To = np.zeros((m + 1, m), order="F")
    assert ora.lanczos(ora.DiagOp(dr), Xo, To) == 0 and vals["lz_info"] == 0
    T = entries("lz_T", m + 1, m)
    Uo = np.zeros((n, m + 1), order="F"); Uo[:, 0] = xr
    Vo = np.zeros((n, m + 1), order="F")
    Bo = np.zeros((m + 1, m), order="F")
    assert ora.bidiagonalization(ora.DiagOp(dr), ora.DiagOp(dr), Uo, Vo, Bo) == 0 and vals["bd_info"] == 0
    B = entries("bd_B", m + 1, m)
    for j in range(m):
        assert np.abs(T[:, j] - To[:, j]).max() <= 1e-12 * np.abs(To[:, j]).max(), ("lanczos", j)
        assert np.abs(B[:, j] - Bo[:, j]).max() <= 1e-12 * np.abs(Bo[:, j]).max(), ("bidiag", j)
    p, kd = 2, m // 2
    Y = np.asfortranarray(np.stack([xr, np.cos(i)], axis=1))
    Ro = np.zeros((p, p), order="F")
    assert ora.qr_no_pivoting(Y, Ro) == 0 and vals["qr_info"] == 0
    for key, ref in (("qr_R11", Ro[0, 0]), ("qr_R12", Ro[0, 1]), ("qr_R22", Ro[1, 1])):
        assert abs(vals[key] - ref) <= 1e-12 * np.abs(Ro).max(), key
    Xb = np.zeros((n, p * (kd + 1)), order="F"); Xb[:, :p] = Y
    Hbo = np.zeros((p * (kd + 1), p * kd), order="F")
    assert ora.arnoldi_block(ora.DiagOp(dr), Xb, Hbo, p) == 0 and vals["bk_info"] == 0
    Hb = entries("bk_H", p * (kd + 1), p * kd)
    for j in range(p * kd):
        assert np.abs(Hb[:, j] - Hbo[:, j]).max() <= 1e-12 * np.abs(Hbo[:, j]).max(), ("block arnoldi", j)
    # column pool driven like the LightKrylov plugin: consecutive columns in one slab, and 200 emulated Gram-Schmidt
    # passes with recurring temporaries carve nothing new
    assert vals["pool_consecutive"] == 1 and vals["pool_slabs"] == 1
    assert vals["pool_carved_before"] == m + 1 and vals["pool_carved_after"] == m + 3
    assert vals["pool_reused"] >= 198 and vals["pool_orth_resid"] < 1e-10
