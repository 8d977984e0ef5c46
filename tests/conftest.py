import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/*.c) -- the checker, never the thing under test."""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def hip():
    """the product package (clarabel.rs_amd), loaded through its ctypes C-ABI binding"""
    import __graft_entry__ as g
    return g.load_package()


def has_gpu():
    try:
        import __graft_entry__ as g
        return g.load_package().device_count() > 0
    except Exception:
        return False
