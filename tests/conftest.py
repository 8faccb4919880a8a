import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU parity case")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """Compile the plain-C oracle (and, where /root/reference exists, oracle/_ref) once per session."""
    so = os.path.join(ROOT, "oracle", "libgoicp_oracle.so")
    src = os.path.join(ROOT, "oracle", "goicp_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libgoicp_oracle.so"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    yield


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_first():
    """On a GPU box: let torch bring its (bundled) HIP runtime up before the engine library's runtime has created streams and
    parked gigabytes of device memory in the same process; tests that hand torch buffers to the engine then find it initialised."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield


def have_ref():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgoicp_ref.so"))
