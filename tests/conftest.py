import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (oracle/r2_oracle.c via ctypes) -- the checker, never the thing under test."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from r2_gaussian_amd import _lib
    _lib.lib()   # fail loudly if the HIP extension is missing: there is no fallback to test
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """Parity margins of every value check of the session (tests/helpers.py PARITY_LOG) -> gpurun_out/parity_report.json."""
    try:
        from tests import helpers as Hh
        if Hh.PARITY_LOG:
            import json
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_report.json"), "w") as f:
                json.dump(Hh.PARITY_LOG, f, indent=1)
    except Exception:   # reporting must never turn a green run red
        pass
