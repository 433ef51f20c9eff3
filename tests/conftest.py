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
