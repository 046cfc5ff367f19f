import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# a clean checkout has no shared library (it is git-ignored): build it once, as `__graft_entry__.build()` does
if not os.path.exists(os.path.join(ROOT, "lamp_amd", "lib", "liblamp_hip.so")) and not os.environ.get("LAMP_TESTS_NO_BUILD"):
    try:
        import __graft_entry__
        __graft_entry__.build()
    except Exception as e:  # the ABI test reports the missing library with the reason
        sys.stderr.write(f"conftest: building liblamp_hip.so failed: {e}\n")

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


def pytest_generate_tests(metafunc):
    """LAMP_SOAK=N (scripts/soak.sh): every collected test N times (an extra, ignored parameter, as pytest-repeat does it)"""
    n = int(os.environ.get("LAMP_SOAK", "0") or 0)
    if n > 1:
        metafunc.fixturenames.append("_lamp_soak_rep")
        metafunc.parametrize("_lamp_soak_rep", range(n), indirect=True, ids=[f"soak{i}" for i in range(n)])


@pytest.fixture
def _lamp_soak_rep(request):
    return request.param


def pytest_collection_modifyitems(config, items):
    """LAMP_SOAK: the whole list shuffled (seed LAMP_SOAK_SEED) - the hunt for the order- or timing-dependent failure of VERDICT r3 (weak 1):
    spin-wait grids, deferred reductions, cross-stream allocator reuse, graph capture."""
    if int(os.environ.get("LAMP_SOAK", "0") or 0) <= 1:
        return
    import random
    random.Random(int(os.environ.get("LAMP_SOAK_SEED", "1"))).shuffle(items)
