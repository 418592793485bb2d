import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def native():
    """The ctypes binding; GPU tests fail (not skip) when the library is missing."""
    from kzg_snark_amd import _native
    _native.lib()
    return _native
