import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multi_gpu: needs at least two MI355X on one node (skips itself otherwise)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on first use."""
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first(request):
    """On a GPU box, let torch initialise its (bundled) HIP runtime before the library's first call: the order bench.py
    and the rank processes use.  Initialising it late, after a few hundred library calls, once came back with "no GPUs
    found" in a partial run of the suite (test_allreduce_callback_plumbing_single_gpu)."""
    if request.config.getoption("-m") and "not gpu" in request.config.getoption("-m"):
        return
    import torch
    if torch.cuda.device_count() > 0 and torch.cuda.is_available():
        torch.cuda.init()
