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


_PARITY_REPORT = {}


@pytest.fixture
def parity_report():
    """Collects the per-test parity report dicts (near-ties, marginal counts with their verification, max |dLLR|); the session
    writes them to gpurun_out/parity_report.json so that every run - the driver's too - leaves the counts behind."""
    def add(name, report):
        _PARITY_REPORT[name] = report
    return add


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY_REPORT:
        return
    import json
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
            json.dump(_PARITY_REPORT, f, indent=1, sort_keys=True, default=float)
    except OSError:
        pass
