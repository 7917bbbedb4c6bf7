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
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu():
    """The product library on device 0.  GPU tests must never pass on a fallback: if the library
    or the device is missing this raises instead of skipping."""
    import blaze_amd

    L = blaze_amd.lib()
    n = L.blz_device_count()
    assert n >= 1, "no HIP device visible: -m gpu tests need the MI355X box"
    return L
