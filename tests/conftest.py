import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    binding.build()
    return binding.lib()


@pytest.fixture(params=["windows of 256", "windows of 1024"])
def window_form(request, monkeypatch):
    """The fused kernel walks windows of 256 paths on wavefronts of up to 2^20 paths and of 1024 paths above
    (csrc/epsm_grad_scatter.hip, launch): the parity tests, whose sizes are all on the small side, run both forms."""
    monkeypatch.setenv("EPSM_SMALL_WAVEFRONT", str(1 << 20) if request.param == "windows of 256" else "0")
    return request.param
