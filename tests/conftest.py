import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # the product has no fallback: build the HIP library (and the oracle) first if a fresh checkout lacks it
    if not os.path.exists(os.path.join(REPO, "cbinfer_amd", "libcbinfer_hip.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/cb_oracle.c)."""
    from oracle import cb_oracle
    cb_oracle.lib()
    return cb_oracle
