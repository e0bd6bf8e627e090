import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENES = os.path.join(ROOT, "tests", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    import rtxamd
    return rtxamd


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu(rt):
    """GPU tests fail loudly (never skip, never fall back) when no device is visible."""
    n = rt.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests must run on the GPU box"
    return n
