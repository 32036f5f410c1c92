import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def _gpu_present():
    try:
        from numpywren_amd.device import hip_available
        return hip_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def host_store(monkeypatch):
    """Keep BigMatrix tiles in host memory (the spill tier) so storage logic runs without a GPU."""
    monkeypatch.setenv("NUMPYWREN_AMD_STORE", "host")
    from numpywren_amd import matrix
    matrix.OBJECTS.clear()
    yield
    matrix.OBJECTS.clear()


@pytest.fixture
def oracle_backend(monkeypatch):
    """Inject the NumPy/oracle checker backend (tests/oracle_backend.py) under the host logic."""
    monkeypatch.setenv("NUMPYWREN_AMD_STORE", "hbm")
    from numpywren_amd import device, matrix
    from oracle_backend import OracleBackend
    be = OracleBackend()
    device.set_backend(be)
    matrix.OBJECTS.clear()
    yield be
    device.set_backend(None)
    matrix.OBJECTS.clear()


@pytest.fixture
def hbm_store(monkeypatch):
    monkeypatch.setenv("NUMPYWREN_AMD_STORE", "hbm")
    from numpywren_amd import matrix
    matrix.OBJECTS.clear()
    yield
    matrix.OBJECTS.clear()
