import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# load the HIP library before torch so that it binds the system ROCm runtime
try:
    from lamp_amd._capi import lib as _lib
    _lib.load()
except Exception:  # the not-gpu ABI test reports a missing library explicitly
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import ctypes as C
        from lamp_amd._capi import lib
        n = C.c_int(0)
        lib.lamp_has_gpu(C.byref(n))
        return n.value == 1
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _has_gpu():
        pytest.fail("no MI355X visible: the product path has no CPU fallback (run -m 'not gpu' on CPU boxes)")
    return 0
