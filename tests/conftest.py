import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure), compiled on demand with oracle/Makefile."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly when it is missing (no fallback)."""
    from msk144cudecoder_amd import hipdecoder
    hipdecoder.load_library()
    return hipdecoder
