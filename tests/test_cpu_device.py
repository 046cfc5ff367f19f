"""lamp's CPU device (device.scala:138 `case object CPU`): tensors that live in host memory.

The library is a GPU backend - convolutions, norms, attention, optimisers exist only as HIP kernels and refuse host tensors loudly.  What
lamp does with CPU tensors around the hot path is small: it builds them from JVM arrays, views / casts / copies them, does scalar and
element-wise arithmetic on a few (loss accumulators, class weights, index lists), sums them, and ships them to the device (call-site
audit: DESIGN.md section 9).  Those operations run where the tensor lives, with the functors the GPU kernels use - checked here against
numpy on a box without a GPU."""
import ctypes as C

import numpy as np
import pytest

from lamp_amd._capi import lib, LampError, handle_array
from lamp_amd import sten as S

lib.load()


def H(a, dtype=None):
    return S.STen.from_numpy(np.asarray(a), S.CPU, dtype)


def test_host_elementwise_broadcast_and_scalars():
    a, b = np.arange(6, dtype=np.float64).reshape(2, 3) - 2.0, np.array([[1.0, 2.0, 4.0]])
    A, B = H(a), H(b)
    assert A.device == S.CPU and (A + B).device == S.CPU
    for got, want in (((A + B), a + b), ((A - B), a - b), ((A * B), a * b), ((A / B), a / b), ((A * 2.5), a * 2.5), ((A + 1.0), a + 1.0),
                      (A.exp(), np.exp(a)), (A.relu(), np.maximum(a, 0)), (A.sigmoid(), 1 / (1 + np.exp(-a))), (A.tanh(), np.tanh(a))):
        assert np.allclose(got.to_numpy(), want, rtol=1e-15, atol=0), (got.to_numpy(), want)
    Z = H(np.zeros((2, 3)))
    o = C.c_void_p(); lib.lamp_lt(C.byref(o), A, Z)
    assert np.array_equal(S.STen(o).to_numpy().astype(bool), a < 0)
    # in place and out forms
    acc, loss = H(np.zeros(1)), H(np.array([3.5]))
    lib.lamp_add_(acc, loss, 2.0)                              # acc += 2 * loss: the epoch-loss accumulator on a CPU model
    assert acc.to_numpy()[0] == 7.0
    i = H(np.arange(5, dtype=np.int64))
    assert np.array_equal((i * 3).to_numpy(), np.arange(5) * 3) and (i * 3).dtype == S.I64
    f = H(a.astype(np.float32))
    assert (f + f).dtype == S.F32 and np.array_equal((f + f).to_numpy(), (a + a).astype(np.float32))


def test_host_reductions_views_and_mm():
    a = (np.arange(24, dtype=np.float64).reshape(2, 3, 4) * 7 % 11) - 5
    A = H(a)
    assert np.allclose(A.sum([1], False).to_numpy(), a.sum(1)) and np.allclose(A.sum([0, 2], True).to_numpy(), a.sum((0, 2), keepdims=True))
    assert np.allclose(A.sum().to_numpy(), a.sum())
    o = C.c_void_p(); lib.lamp_mean_dims(C.byref(o), A, (C.c_int64 * 1)(2), 1, 0)
    assert np.allclose(S.STen(o).to_numpy(), a.mean(2))
    o = C.c_void_p(); lib.lamp_norm2_dims(C.byref(o), A, (C.c_int64 * 2)(1, 2), 2, 0)
    assert np.allclose(S.STen(o).to_numpy(), np.sqrt((a * a).sum((1, 2))))
    o = C.c_void_p(); lib.lamp_max_all(C.byref(o), A)
    assert S.STen(o).to_numpy() == a.max()
    t = A.transpose(1, 2)                                       # views are free on either device
    assert np.allclose((t + t).to_numpy(), 2 * a.transpose(0, 2, 1))
    x, w = np.arange(12, dtype=np.float64).reshape(3, 4), np.arange(8, dtype=np.float64).reshape(4, 2) - 3
    assert np.allclose(H(x).mm(H(w)).to_numpy(), x @ w)
    assert np.allclose(H(x.astype(np.float32)).mm(H(w.astype(np.float32))).to_numpy(), (x @ w).astype(np.float32))
    li = H(np.array([4, 5, 6], dtype=np.int64)).sum()
    assert li.dtype == S.I64 and li.to_numpy() == 15


def test_gpu_only_operators_on_host_tensors_need_a_gpu():
    """Operators that exist only as kernels stage all-host arguments through the current GPU (parity on host tensors:
    tests/test_host_staging.py, -m gpu).  Without a GPU that fails loudly - there is no CPU implementation to fall back to."""
    from tests.conftest import _has_gpu
    if _has_gpu():
        pytest.skip("covered on the GPU by tests/test_host_staging.py")
    a = H(np.ones((2, 3, 8, 8), dtype=np.float32))
    w = H(np.ones((4, 3, 3, 3), dtype=np.float32))
    o = C.c_void_p()
    from lamp_amd._capi import i64_array
    for call in (lambda: lib.lamp_convolution(C.byref(o), a, w, None, i64_array([1, 1]), i64_array([1, 1]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1),
                 lambda: lib.lamp_log_softmax(C.byref(o), a, 1)):
        with pytest.raises(LampError, match="no usable MI355X|no CPU fallback|no ROCm-capable"):
            call()


def test_host_index_select_gathers_a_minibatch():
    """BatchStream.minibatchesFromFull gathers rows of the host-resident data set into its pinned buffer (BatchStream.scala:540-573)"""
    import numpy as np
    from lamp_amd import sten as S
    from lamp_amd import _capi
    a = np.arange(5 * 3 * 4, dtype=np.float32).reshape(5, 3, 4)
    idx = np.array([4, 0, 0, 2], dtype=np.int64)
    t, ix = S.STen.from_numpy(a, device=S.CPU), S.STen.from_numpy(idx, device=S.CPU)
    assert np.array_equal(t.indexSelect(0, ix).to_numpy(), a[idx])
    ix1 = S.STen.from_numpy(np.array([2, 1], dtype=np.int64), device=S.CPU)
    assert np.array_equal(t.indexSelect(1, ix1).to_numpy(), a[:, [2, 1]])
    assert np.array_equal(t.transpose(0, 1).indexSelect(0, ix1).to_numpy(), a.transpose(1, 0, 2)[[2, 1]])
    bad = S.STen.from_numpy(np.array([5], dtype=np.int64), device=S.CPU)
    with pytest.raises(_capi.LampError, match="out of range"):
        t.indexSelect(0, bad)
