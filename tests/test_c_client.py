"""A plain-C program (tests/c/test_abi.c, compiled with gcc against include/lightkrylov_hip.h) uses the
engine through the C ABI alone.  CPU: it compiles and links.  GPU: it runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "test_abi.c")
EXE = os.path.join(ROOT, "tests", "c", "test_abi")
LIBDIR = os.path.join(ROOT, "lightkrylov_amd")


def _build():
    cmd = ["gcc", "-O1", "-std=c11", "-Wall", "-o", EXE, SRC, f"-L{LIBDIR}", "-llightkrylov_hip", "-L/opt/rocm/lib",
           "-lamdhip64", "-lm", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return EXE


def test_c_client_compiles_and_links_against_the_header():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C client ok" in out.stdout, out.stdout + out.stderr
