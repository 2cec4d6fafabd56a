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
           "tools/plugin_check/driver.f90", "tools/plugin_check/mock_abi.c", "tools/plugin_check/gen_stdlib_stubs.py",
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
