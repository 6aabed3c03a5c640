import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _have_gpu():
    try:
        import ctypes as C
        from aether_primitives_amd import _lib
        n = C.c_int(0)
        return _lib.load().aeth_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    """One context for the whole GPU session.  Fails (does not skip) when the HIP
    library is missing; skips only when there is no device (CPU container)."""
    import aether_primitives_amd as ap
    if not _have_gpu():
        pytest.skip("no GPU visible")
    c = ap.Context(0)
    yield c
    c.close()
