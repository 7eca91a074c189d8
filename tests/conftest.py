import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from oatk_amd import _lib
        return os.path.exists(_lib.LIB_PATH) and _lib.load().oatk_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip():
    """one device context for the whole session; GPU tests fail (not skip) if it cannot be created"""
    from oatk_amd import HipSyncasm
    ctx = HipSyncasm(0)
    yield ctx
    ctx.close()
