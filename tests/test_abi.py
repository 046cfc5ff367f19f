"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports
every symbol include/*.h declares (no compute calls here)."""
import ctypes as C
import os

import pytest

from lamp_amd import _capi


def test_library_exists_and_loads():
    assert os.path.exists(_capi.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = _capi.lib.load()
    assert dll is not None
    assert b"lamp_hip" in dll.lamp_version()


def test_every_declared_symbol_is_exported():
    _capi.lib.load()
    assert len(_capi.lib.decls) > 150
    assert _capi.lib.missing == [], f"declared in include/*.h but not exported: {_capi.lib.missing}"


def test_header_parser_maps_every_signature():
    for name, (res, args, raw) in _capi.lib.decls.items():
        assert len(args) == len(raw), name


def test_errors_surface_as_exceptions_without_gpu():
    # a call that fails before touching the GPU: null handle
    with pytest.raises(_capi.LampError):
        n = C.c_int()
        _capi.lib.lamp_tensor_ndim(None, C.byref(n))


def test_host_tensors_compute_on_the_host_and_gpu_only_operators_need_a_gpu():
    import numpy as np
    from lamp_amd import sten as S
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    t = S.STen.from_numpy(a, device=S.CPU)
    assert t.shape == [3, 4] and t.device == S.CPU
    assert np.array_equal(t.to_numpy(), a)
    assert np.array_equal(t.transpose(0, 1).to_numpy(), a.T)          # views + host strided copy work
    assert np.array_equal(t.castToDouble().to_numpy(), a.astype(np.float64))
    assert np.array_equal(t.relu().to_numpy(), np.maximum(a, 0))        # lamp's CPU device: element-wise where the tensor lives
    # everything else is a GPU kernel: all-host arguments are staged through the current GPU (tests/test_host_staging.py, -m gpu) -
    # on a box without one that fails loudly, it does not fall back to anything
    from tests.conftest import _has_gpu
    if not _has_gpu():
        with pytest.raises(_capi.LampError, match="no usable MI355X|no CPU fallback|no ROCm-capable"):
            t.logSoftMax(1)
