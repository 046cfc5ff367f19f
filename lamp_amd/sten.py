"""STen - host-side mirror of lamp's tensor facade over the C ABI.

Reference: lamp-sten/src/main/scala/lamp/STen.scala (class STen :845-1900, companion :15-676).
Method names follow the Scala ones (camelCase kept where lamp has it) so the parity tests read
like the reference's own tests.  Every method is ONE call into liblamp_hip.so; nothing here
computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from ._capi import lib, LampError, i64_array, f64_array, handle_array

# scalar type bytes (STen.scala:726-731)
U8, I8, I16, I32, I64, F16, F32, F64, BOOL, BF16 = 0, 1, 2, 3, 4, 5, 6, 7, 11, 15
_NP = {U8: np.uint8, I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64, F32: np.float32, F64: np.float64,
       BOOL: np.bool_, BF16: np.uint16, F16: np.float16}
_FROM_NP = {np.dtype(np.uint8): U8, np.dtype(np.int8): I8, np.dtype(np.int16): I16, np.dtype(np.int32): I32, np.dtype(np.int64): I64, np.dtype(np.float32): F32,
            np.dtype(np.float64): F64, np.dtype(np.bool_): BOOL, np.dtype(np.float16): F16}

CPU = -1
INT64_MIN = -(2 ** 63)


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """round-to-nearest-even f32 -> bf16 bit patterns (uint16)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def _out():
    return C.c_void_p()


class STen:
    """One owned handle to a lamp_tensor. Released when garbage collected or by release()."""

    __slots__ = ("h", "__weakref__")

    def __init__(self, handle):
        if isinstance(handle, C.c_void_p):
            handle = handle.value
        if not handle:
            raise LampError("null tensor handle")
        self.h = handle

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def _as_parameter_(self):
        # lets an STen be passed straight to a C-ABI call; the object (and so the handle) stays
        # alive for the duration of the call because ctypes holds the argument tuple
        return C.c_void_p(self.h)

    def release(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            lib.lamp_tensor_release(h)

    # ---- factories (STen.scala:42-330) --------------------------------------------------------
    @staticmethod
    def from_numpy(a: np.ndarray, device: int = 0, dtype: Optional[int] = None) -> "STen":
        a = np.asarray(a)
        if dtype is None:
            dtype = _FROM_NP[a.dtype]
        if dtype == BF16:
            host = f32_to_bf16_bits(a.astype(np.float32))
        else:
            host = np.ascontiguousarray(a.astype(_NP[dtype], copy=False))
        o = _out()
        lib.lamp_empty(C.byref(o), i64_array(a.shape), a.ndim, dtype, device)
        t = STen(o)
        lib.lamp_copy_from_host(t.h, host.ctypes.data_as(C.c_void_p), host.nbytes)
        return t

    @staticmethod
    def zeros(shape, dtype=F32, device=0):
        o = _out(); lib.lamp_zeros(C.byref(o), i64_array(shape), len(shape), dtype, device); return STen(o)

    @staticmethod
    def ones(shape, dtype=F32, device=0):
        o = _out(); lib.lamp_ones(C.byref(o), i64_array(shape), len(shape), dtype, device); return STen(o)

    @staticmethod
    def full(shape, value, dtype=F32, device=0):
        o = _out(); lib.lamp_full(C.byref(o), i64_array(shape), len(shape), float(value), dtype, device); return STen(o)

    @staticmethod
    def scalarDouble(value, dtype=F64, device=0):
        o = _out(); lib.lamp_scalar_tensor(C.byref(o), float(value), dtype, device); return STen(o)

    @staticmethod
    def scalarLong(value, dtype=I64, device=0):
        o = _out(); lib.lamp_scalar_tensor_l(C.byref(o), int(value), dtype, device); return STen(o)

    @staticmethod
    def arange(start, end, step=1, dtype=I64, device=0):
        o = _out(); lib.lamp_arange(C.byref(o), float(start), float(end), float(step), dtype, device); return STen(o)

    @staticmethod
    def normal(mean, std, shape, dtype=F32, device=0):
        o = _out(); lib.lamp_normal(C.byref(o), float(mean), float(std), i64_array(shape), len(shape), dtype, device); return STen(o)

    @staticmethod
    def rand(shape, dtype=F32, device=0):
        o = _out(); lib.lamp_rand(C.byref(o), i64_array(shape), len(shape), dtype, device); return STen(o)

    @staticmethod
    def randn(shape, dtype=F32, device=0):
        o = _out(); lib.lamp_randn(C.byref(o), i64_array(shape), len(shape), dtype, device); return STen(o)

    @staticmethod
    def randint(low, high, shape, dtype=I64, device=0):
        o = _out(); lib.lamp_randint(C.byref(o), int(low), int(high), i64_array(shape), len(shape), dtype, device); return STen(o)

    @staticmethod
    def cat(tensors: Sequence["STen"], dim: int):
        o = _out(); lib.lamp_cat(C.byref(o), handle_array([t.h for t in tensors]), len(tensors), dim); return STen(o)

    @staticmethod
    def stack(tensors: Sequence["STen"], dim: int):
        o = _out(); lib.lamp_stack(C.byref(o), handle_array([t.h for t in tensors]), len(tensors), dim); return STen(o)

    @staticmethod
    def where(cond: "STen", a: "STen", b: "STen"):
        o = _out(); lib.lamp_where(C.byref(o), cond.h, a.h, b.h); return STen(o)

    # ---- metadata -----------------------------------------------------------------------------
    @property
    def shape(self):
        n = C.c_int(); lib.lamp_tensor_ndim(self.h, C.byref(n))
        buf = (C.c_int64 * 8)(); lib.lamp_tensor_sizes(self.h, buf)
        return [int(buf[i]) for i in range(n.value)]

    sizes = shape

    @property
    def strides(self):
        n = C.c_int(); lib.lamp_tensor_ndim(self.h, C.byref(n))
        buf = (C.c_int64 * 8)(); lib.lamp_tensor_strides(self.h, buf)
        return [int(buf[i]) for i in range(n.value)]

    @property
    def numel(self):
        n = C.c_int64(); lib.lamp_tensor_numel(self.h, C.byref(n)); return n.value

    @property
    def scalarTypeByte(self):
        n = C.c_int(); lib.lamp_tensor_scalar_type(self.h, C.byref(n)); return n.value

    dtype = scalarTypeByte

    @property
    def device(self):
        n = C.c_int(); lib.lamp_tensor_device(self.h, C.byref(n)); return n.value

    @property
    def data_ptr(self):
        p = C.c_void_p(); lib.lamp_tensor_data_ptr(self.h, C.byref(p)); return p.value

    @property
    def storage_id(self):
        p = C.c_uint64(); lib.lamp_tensor_storage_id(self.h, C.byref(p)); return p.value

    def is_contiguous(self):
        n = C.c_int(); lib.lamp_tensor_is_contiguous(self.h, C.byref(n)); return bool(n.value)

    def __repr__(self):
        return f"STen(shape={self.shape}, dtype={self.dtype}, device={self.device})"

    # ---- host transfer (TensorHelpers.scala) --------------------------------------------------
    def to_numpy(self) -> np.ndarray:
        """dense row-major copy on the host; bf16 comes back as float32."""
        dt = self.dtype
        arr = np.empty(self.shape, dtype=_NP[dt])
        c = self if self.is_contiguous() else self.contiguous()
        lib.lamp_copy_to_host(c.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes)
        if dt == BF16:
            return bf16_bits_to_f32(arr).reshape(self.shape)
        return arr

    toMat = to_numpy

    def toDoubleArray(self):
        return self.to_numpy().astype(np.float64).reshape(-1)

    def toLongArray(self):
        return self.to_numpy().astype(np.int64).reshape(-1)

    def item(self) -> float:
        d = C.c_double(); lib.lamp_item(self.h, C.byref(d)); return d.value

    # ---- copies / casts ------------------------------------------------------------------------
    def _u(self, fn, *args):
        o = _out(); getattr(lib, fn)(C.byref(o), self.h, *args); return STen(o)

    def cloneTensor(self): return self._u("lamp_clone")
    clone = cloneTensor
    def contiguous(self): return self._u("lamp_contiguous")
    def castToType(self, dtype): return self._u("lamp_cast", dtype)
    def castToFloat(self): return self.castToType(F32)
    def castToDouble(self): return self.castToType(F64)
    def castToLong(self): return self.castToType(I64)
    def castToBF16(self): return self.castToType(BF16)
    def to(self, dtype=None, device=None, non_blocking=True, copy=False):
        return self._u("lamp_to", self.dtype if dtype is None else dtype, self.device if device is None else device,
                       int(non_blocking), int(copy))
    def cpu(self): return self.to(device=CPU)
    def pin(self):
        """STen.pin (STen.scala `pin`): a copy of a host tensor in page-locked memory (what the host-resident minibatch stream gathers from)"""
        return self._u("lamp_pin_memory")
    def copyFrom(self, src: "STen", nonBlocking=True):
        lib.lamp_copy_(self.h, src.h, int(nonBlocking))
    def zero_(self): lib.lamp_zero_(self.h)
    def fill_(self, v): lib.lamp_fill_(self.h, float(v))
    def zerosLike(self): return self._u("lamp_zeros_like")
    def onesLike(self): return self._u("lamp_ones_like")

    # ---- views (STen.scala:956-971,1374-1380,1472-1491,1740-1775) -----------------------------
    def view(self, *dims): return self._u("lamp_view", i64_array(dims), len(dims))
    def reshape(self, *dims): return self._u("lamp_reshape", i64_array(dims), len(dims))
    def flatten(self, startDim=0, endDim=-1): return self._u("lamp_flatten", startDim, endDim)
    def transpose(self, d0, d1): return self._u("lamp_transpose", d0, d1)
    @property
    def t(self): return self._u("lamp_t")
    def select(self, dim, index): return self._u("lamp_select", dim, index)
    def slice(self, dim, start, end, step=1): return self._u("lamp_slice", dim, start, end, step)
    def narrow(self, dim, start, length): return self._u("lamp_narrow", dim, start, length)
    def expand(self, shape): return self._u("lamp_expand", i64_array(shape), len(shape))
    def expandAs(self, other): return self._u("lamp_expand_as", other.h)
    def squeeze(self, dim=None): return self._u("lamp_squeeze", INT64_MIN if dim is None else dim)
    def unsqueeze(self, dim): return self._u("lamp_unsqueeze", dim)
    def unbroadcast(self, sizes): return self._u("lamp_unbroadcast", i64_array(sizes), len(sizes))

    # ---- arithmetic (STen.scala:1110-1217) ----------------------------------------------------
    def _bin(self, fn_t, fn_s, other, *extra):
        if isinstance(other, STen):
            return self._u(fn_t, other.h, *extra)
        return self._u(fn_s, float(other), *extra)

    def __add__(self, o): return self._bin("lamp_add", "lamp_add_scalar", o, 1.0)
    def __sub__(self, o): return self._bin("lamp_sub", "lamp_sub_scalar", o, 1.0)
    def __mul__(self, o): return self._bin("lamp_mul", "lamp_mul_scalar", o)
    def __truediv__(self, o): return self._bin("lamp_div", "lamp_div_scalar", o)
    def add(self, o, alpha): return self._bin("lamp_add", "lamp_add_scalar", o, float(alpha))
    def sub(self, o, alpha): return self._bin("lamp_sub", "lamp_sub_scalar", o, float(alpha))
    def __iadd__(self, o):
        if isinstance(o, STen): lib.lamp_add_(self.h, o.h, 1.0)
        else: lib.lamp_add_scalar_(self.h, float(o), 1.0)
        return self
    def __isub__(self, o):
        if isinstance(o, STen): lib.lamp_sub_(self.h, o.h, 1.0)
        else: lib.lamp_add_scalar_(self.h, -float(o), 1.0)
        return self
    def __imul__(self, o):
        if isinstance(o, STen): lib.lamp_mul_(self.h, o.h)
        else: lib.lamp_mul_scalar_(self.h, float(o))
        return self
    def __itruediv__(self, o):
        if isinstance(o, STen): lib.lamp_div_(self.h, o.h)
        else: lib.lamp_mul_scalar_(self.h, 1.0 / float(o))
        return self
    def __neg__(self): return self._u("lamp_neg")

    @staticmethod
    def addOut(out, a, b, alpha): lib.lamp_add_out(out.h, a.h, b.h, float(alpha))
    @staticmethod
    def subOut(out, a, b, alpha): lib.lamp_sub_out(out.h, a.h, b.h, float(alpha))
    @staticmethod
    def mulOut(out, a, b): lib.lamp_mul_out(out.h, a.h, b.h)
    @staticmethod
    def divOut(out, a, b): lib.lamp_div_out(out.h, a.h, b.h)
    @staticmethod
    def addcmulOut(out, self_, t1, t2, alpha): lib.lamp_addcmul_out(out.h, self_.h, t1.h, t2.h, float(alpha))
    @staticmethod
    def addcdivOut(out, self_, t1, t2, alpha): lib.lamp_addcdiv_out(out.h, self_.h, t1.h, t2.h, float(alpha))
    def addcmulSelf(self, t1, t2, alpha): lib.lamp_addcmul_out(self.h, self.h, t1.h, t2.h, float(alpha))

    def max(self, other: "STen"): return self._u("lamp_maximum", other.h)
    def min(self, other: "STen"): return self._u("lamp_minimum", other.h)
    def pow(self, e):
        return self._u("lamp_pow_tensor", e.h) if isinstance(e, STen) else self._u("lamp_pow_scalar", float(e))
    def maskedFill(self, mask, v): return self._u("lamp_masked_fill", mask.h, float(v))

    def _cmp(self, name, o):
        return self._u(f"lamp_{name}", o.h) if isinstance(o, STen) else self._u(f"lamp_{name}_scalar", float(o))
    def lt(self, o): return self._cmp("lt", o)
    def le(self, o): return self._cmp("le", o)
    def gt(self, o): return self._cmp("gt", o)
    def ge(self, o): return self._cmp("ge", o)
    def equ(self, o): return self._cmp("eq", o)
    def ne(self, o): return self._cmp("ne", o)
    def logicalNot(self): return self._u("lamp_logical_not")

    # unary
    def relu(self): return self._u("lamp_relu")
    def relu_(self): lib.lamp_relu_(self.h)
    def leakyRelu(self, slope): return self._u("lamp_leaky_relu", float(slope))
    def gelu(self): return self._u("lamp_gelu")
    def sigmoid(self): return self._u("lamp_sigmoid")
    def tanh(self): return self._u("lamp_tanh")
    def hardSwish(self): return self._u("lamp_hardswish")
    def softplus(self, beta, threshold): return self._u("lamp_softplus", float(beta), float(threshold))
    def exp(self): return self._u("lamp_exp")
    def exp_(self): lib.lamp_exp_(self.h)
    def log(self): return self._u("lamp_log")
    def log1p(self): return self._u("lamp_log1p")
    def sqrt(self): return self._u("lamp_sqrt")
    def sqrt_(self): lib.lamp_sqrt_(self.h)
    def square(self): return self._u("lamp_square")
    def reciprocal(self): return self._u("lamp_reciprocal")
    def reciprocal_(self): lib.lamp_reciprocal_(self.h)
    def neg(self): return self._u("lamp_neg")
    def abs(self): return self._u("lamp_abs")
    def sign(self): return self._u("lamp_sign")
    def sin(self): return self._u("lamp_sin")
    def cos(self): return self._u("lamp_cos")
    def tan(self): return self._u("lamp_tan")
    def atan(self): return self._u("lamp_atan")

    # ---- reductions ----------------------------------------------------------------------------
    def sum(self, dim=None, keepDim=False):
        if dim is None: return self._u("lamp_sum_all")
        dim = [dim] if isinstance(dim, int) else list(dim)
        return self._u("lamp_sum_dims", i64_array(dim), len(dim), int(keepDim))
    def rowSum(self): return self.sum(1, True)
    def colSum(self): return self.sum(0, True)
    def mean(self, dim=None, keepDim=False):
        if dim is None: return self._u("lamp_mean_all")
        dim = [dim] if isinstance(dim, int) else list(dim)
        return self._u("lamp_mean_dims", i64_array(dim), len(dim), int(keepDim))
    def norm2(self, dim, keepDim):
        dim = [dim] if isinstance(dim, int) else list(dim)
        return self._u("lamp_norm2_dims", i64_array(dim), len(dim), int(keepDim))
    def varAndMean(self, dim, unbiased, keepDim):
        dim = [dim] if isinstance(dim, int) else list(dim)
        v, m = _out(), _out()
        lib.lamp_var_mean_dims(C.byref(v), C.byref(m), self.h, i64_array(dim), len(dim), int(unbiased), int(keepDim))
        return STen(v), STen(m)
    def maxAll(self): return self._u("lamp_max_all")
    def minAll(self): return self._u("lamp_min_all")
    def argmax(self, dim, keepDim): return self._u("lamp_argmax", dim, int(keepDim))

    # ---- GEMM (STen.scala:1146,1220-1240) -------------------------------------------------------
    def mm(self, o): return self._u("lamp_mm", o.h)
    def bmm(self, o): return self._u("lamp_bmm", o.h)
    def matmul(self, o): return self._u("lamp_matmul", o.h)
    def addmm(self, m1, m2, beta, alpha): return self._u("lamp_addmm", m1.h, m2.h, float(beta), float(alpha))
    def baddbmm(self, b1, b2, beta, alpha): return self._u("lamp_baddbmm", b1.h, b2.h, float(beta), float(alpha))
    @staticmethod
    def mmOut(out, a, b): lib.lamp_mm_out(out.h, a.h, b.h)
    @staticmethod
    def bmmOut(out, a, b): lib.lamp_bmm_out(out.h, a.h, b.h)
    @staticmethod
    def addmmOut(out, self_, a, b, beta, alpha): lib.lamp_addmm_out(out.h, self_.h, a.h, b.h, float(beta), float(alpha))
    @staticmethod
    def addmm_out_transposed1(out, self_, a, b, beta, alpha):
        lib.lamp_addmm_out_transposed1(out.h, self_.h, a.h, b.h, float(beta), float(alpha))
    @staticmethod
    def addmm_out_transposed2(out, self_, a, b, beta, alpha):
        lib.lamp_addmm_out_transposed2(out.h, self_.h, a.h, b.h, float(beta), float(alpha))
    @staticmethod
    def baddbmm_out_transposed1(out, self_, a, b, beta, alpha):
        lib.lamp_baddbmm_out_transposed1(out.h, self_.h, a.h, b.h, float(beta), float(alpha))
    @staticmethod
    def baddbmm_out_transposed2(out, self_, a, b, beta, alpha):
        lib.lamp_baddbmm_out_transposed2(out.h, self_.h, a.h, b.h, float(beta), float(alpha))

    # ---- softmax / index ------------------------------------------------------------------------
    def logSoftMax(self, dim): return self._u("lamp_log_softmax", dim)
    def softmax(self, dim): return self._u("lamp_softmax", dim)
    def indexSelect(self, dim, index): return self._u("lamp_index_select", dim, index.h)
    def indexAdd(self, dim, index, source): return self._u("lamp_index_add", dim, index.h, source.h)
    def maskedSelect(self, mask): return self._u("lamp_masked_select", mask.h)
    def repeatInterleave(self, repeats, dim): return self._u("lamp_repeat_interleave", int(repeats), dim)
    def oneHot(self, n): return self._u("lamp_one_hot", n)
    def topk(self, k, dim, largest, sorted_):
        v, i = _out(), _out()
        lib.lamp_topk(C.byref(v), C.byref(i), self.h, k, dim, int(largest), int(sorted_))
        return STen(v), STen(i)
    def dropout_(self, p, training): lib.lamp_dropout_(self.h, float(p), int(training))

    # ---- sorting / overwriting scatters / triangles (STen.scala:1037-1068, 1412-1423, 1551-1557, 1592, 1715-1726, 1761, 1883-1886) ----
    def sort(self, dim, descending):
        v, i = _out(), _out()
        lib.lamp_sort(C.byref(v), C.byref(i), self.h, dim, int(descending))
        return STen(v), STen(i)
    def argsort(self, stable, dim, descending): return self._u("lamp_argsort", int(stable), dim, int(descending))
    def median(self, dim, keepDim):
        v, i = _out(), _out()
        lib.lamp_median_dim(C.byref(v), C.byref(i), self.h, dim, int(keepDim))
        return STen(v), STen(i)
    def unique(self, sorted_=True, returnInverse=True, returnCounts=True):
        v, i, c = _out(), _out(), _out()
        lib.lamp_unique(C.byref(v), C.byref(i), C.byref(c), self.h)
        return STen(v), STen(i), STen(c)
    def bincount(self, weights=None, minLength=0): return self._u("lamp_bincount", weights.h if weights is not None else None, int(minLength))
    def scatter(self, dim, index, source):
        if isinstance(source, STen):
            return self._u("lamp_scatter", dim, index.h, source.h)
        return self._u("lamp_scatter_value", dim, index.h, float(source))
    def indexPut(self, indices, values, accumulate):
        return self._u("lamp_index_put", handle_array([t.h for t in indices]), len(indices), values.h, int(accumulate))
    def put(self, index, values, accumulate): return self._u("lamp_put", index.h, values.h, int(accumulate))
    def indexCopy(self, dim, index, source): return self._u("lamp_index_copy", dim, index.h, source.h)
    def tril(self, diagonal=0): return self._u("lamp_tril", int(diagonal))
    def triu(self, diagonal=0): return self._u("lamp_triu", int(diagonal))
    def tril_(self, diagonal=0): lib.lamp_tril_out(self.h, self.h, int(diagonal))
    def diagonalView(self, offset, dim1, dim2): return self._u("lamp_diagonal", int(offset), int(dim1), int(dim2))
    def trace(self): return self._u("lamp_trace")

    @staticmethod
    def randperm(n, dtype=I64, device=0):
        o = _out(); lib.lamp_randperm(C.byref(o), int(n), dtype, device); return STen(o)

    @staticmethod
    def multinomial(probs, numSamples, replacement):
        o = _out(); lib.lamp_multinomial(C.byref(o), probs.h, int(numSamples), int(replacement)); return STen(o)


def synchronize():
    lib.lamp_device_synchronize()


def live_tensor_count() -> int:
    n = C.c_int64(); lib.lamp_live_tensor_count(C.byref(n)); return n.value


# ---- Device.toBatched / BufferPair (lamp-sten/src/main/scala/lamp/device.scala:48-114, :236-249) --------------------------------
class BufferPair:
    """a (pinned) host source buffer and a device destination buffer of `size` elements"""

    def __init__(self, source: STen, destination: STen):
        self.source, self.destination = source, destination

    @staticmethod
    def allocate(size: int, device: int, dtype: int = F32) -> "BufferPair":
        o = _out(); lib.lamp_empty(C.byref(o), i64_array([size]), 1, dtype, CPU)
        host = STen(o)
        p = _out(); lib.lamp_pin_memory(C.byref(p), host)
        d = _out(); lib.lamp_empty(C.byref(d), i64_array([size]), 1, dtype, device)
        return BufferPair(STen(p), STen(d))


def toBatched(tensors: Sequence[STen], buffers: BufferPair) -> list:
    """Device.toBatched: the tensors (one dtype, on the source buffer's device) are concatenated into the host buffer, cross the bus
    in ONE copy and are split into clones on the destination device - the same ATen call sequence as the reference (view, slice,
    cat_out, copyFrom, narrow, view, clone)."""
    views = [t.view(-1) for t in tensors]
    sizes = [v.numel for v in views]
    total = sum(sizes)
    host_slice = buffers.source.slice(0, 0, total)
    lib.lamp_cat_out(host_slice, handle_array([v.h for v in views]), len(views), 0)
    buffers.destination.slice(0, 0, total).copyFrom(host_slice, True)
    dev = buffers.destination.slice(0, 0, total)
    out, off = [], 0
    for t, n in zip(tensors, sizes):
        out.append(dev.narrow(0, off, n).view(*t.shape).cloneTensor())
        off += n
    return out
