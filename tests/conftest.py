import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the full-length form of a stress test (2000 graph replays, 600 forwards); the default run takes the "
                                       "sampled form (200 / 100) of the same test.  Enable with --runslow or TUTEL_AMD_RUN_SLOW=1")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="also run the full-length stress tests (marker `slow`)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if not (config.getoption("--runslow") or os.environ.get("TUTEL_AMD_RUN_SLOW") == "1"):
        slow = pytest.mark.skip(reason="full-length stress: --runslow / TUTEL_AMD_RUN_SLOW=1 (the sampled form of the same test runs by default)")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(slow)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, see oracle/moe_oracle.py)."""
    from oracle import moe_oracle
    moe_oracle._lib()
    return moe_oracle
