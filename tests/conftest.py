import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on device 0.  No CPU fallback: a missing library or GPU
    is an error for tests marked gpu."""
    import libh264_amd
    eng = libh264_amd.Engine(0)
    yield eng
    eng.close()
