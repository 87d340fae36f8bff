import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    """One library context on GPU 0 for the whole GPU test session."""
    import strugatzki_amd as S
    c = S.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["auto", "spectral"])
def screen_kernel_env(request, monkeypatch):
    """Runs a test twice: with the library's own choice of screening kernel (the matrix cores for the short targets
    most tests use) and with the spectral kernel forced through the environment, which child processes inherit."""
    monkeypatch.delenv("SGZ_FFT_N", raising=False)
    if request.param == "spectral":
        monkeypatch.setenv("SGZ_SCREEN_KERNEL", "spectral")
    else:
        monkeypatch.delenv("SGZ_SCREEN_KERNEL", raising=False)
    return request.param
