import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def lib():
    """ctypes handle of libbigkrls_hip.so; the GPU tests call through this C ABI."""
    from bigkrls_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def ctx():
    import bigkrls_amd as bk
    return bk.Context(0)
