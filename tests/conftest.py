import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress must fail, not hang the box: give every GPU test a
    time limit when pytest-timeout is available (it is in this image)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(480))     # the first import of torch on a cold box alone can take 1-2 min


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): oracle/ccp_oracle.c behind ctypes."""
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference headers; only where /root/reference existed at build time."""
    import oracle
    try:
        return oracle.Ref()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not built (no /root/reference here)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
