import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build the native pieces if a fresh checkout has not run __graft_entry__.build() yet (hipcc cross-compiles
    # without a GPU; seconds).  Never masks a build failure: the tests that need the artefacts fail loudly.
    import subprocess
    lib = os.path.join(ROOT, "lightkrylov_amd", "liblightkrylov_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.call(["make", "-C", os.path.join(ROOT, "lightkrylov_amd", "csrc"), "-s"])
    if not os.path.exists(os.path.join(ROOT, "fortran", "test_iso_c")) and os.path.exists("/opt/rocm/bin/amdflang") \
            and os.path.exists(lib):
        subprocess.call(["make", "-C", os.path.join(ROOT, "fortran"), "-s"])


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.device_count() > 0 and torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container (GPU tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def ctx():
    import lightkrylov_amd as lk
    c = lk.Context(device=0)
    yield c
    c.close()
