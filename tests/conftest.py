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
def window_form(request):
    """The accumulating kernel has two window forms (csrc/epsm_backward_cp.hip, launch): wavefronts of up to 2^20 paths are
    cut into windows of 128 .. 1024 paths and flushed into replicas, larger ones walk windows of 2048 paths.  The parity
    tests, whose sizes are all on the small side, run both (the ids keep the sizes of the round they were written in)."""
    from epsm_mitsuba3_amd import _lib
    with _lib.options(small_wavefront_paths=(1 << 20) if request.param == "windows of 256" else 0):
        yield request.param
