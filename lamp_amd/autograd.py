"""Variable / Op mirror of lamp.autograd over the host C ABI (include/lamp_host.h).

Reference: lamp-core/src/main/scala/lamp/autograd/autograd.scala:176-486 (Variable and its
methods), ops.scala (one case class per op).  Each method below builds ONE op node by calling
lamp_op_apply with the Scala case-class name; the graph, the backward closures and every kernel
launch live in liblamp_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from ._capi import lib, i64_array, f64_array, handle_array
from .sten import STen


class Variable:
    __slots__ = ("h", "__weakref__")

    def __init__(self, handle):
        if isinstance(handle, C.c_void_p):
            handle = handle.value
        self.h = handle

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                lib.lamp_var_release(h)
            except Exception:
                pass

    @property
    def _as_parameter_(self):
        return C.c_void_p(self.h)

    # -- accessors ---------------------------------------------------------------------------------
    @property
    def value(self) -> STen:
        o = C.c_void_p(); lib.lamp_var_value(self.h, C.byref(o)); return STen(o)

    @property
    def partialDerivative(self) -> Optional[STen]:
        o = C.c_void_p(); lib.lamp_var_grad(self.h, C.byref(o)); return STen(o) if o.value else None

    grad = partialDerivative

    @property
    def needsGrad(self) -> bool:
        n = C.c_int(); lib.lamp_var_needs_grad(self.h, C.byref(n)); return bool(n.value)

    @property
    def shape(self):
        return self.value.shape

    def zeroGrad(self): lib.lamp_var_zero_grad(self.h)
    def backprop(self): lib.lamp_var_backprop(self.h)

    @property
    def wengert_size(self):
        n = C.c_int64(); lib.lamp_var_wengert_size(self.h, C.byref(n)); return n.value

    # -- ops (autograd.scala:296-486) ----------------------------------------------------------------
    def transpose(self, d1=0, d2=1): return apply_op("Transpose", [self], i=[d1, d2])
    @property
    def t(self): return apply_op("Transpose", [self], i=[0, 1])
    def view(self, shape): return apply_op("View", [self], i=list(shape))
    def reshape(self, shape): return apply_op("Reshape", [self], i=list(shape))
    def flatten(self, startDim=0, endDim=-1): return apply_op("Flatten", [self], i=[startDim, endDim])
    def flattenLastDimensions(self, dims): return self.flatten(len(self.shape) - dims, -1)
    def cat(self, other, dim): return apply_op("Concatenate", [self, other], i=[dim])
    def __add__(self, o): return apply_op("Add", [self, o]) if isinstance(o, Variable) else apply_op("ConstAdd", [self], d=[o])
    def __sub__(self, o): return apply_op("Minus", [self, o])
    def __mul__(self, o): return apply_op("Mult", [self, o]) if isinstance(o, Variable) else apply_op("ConstMult", [self], d=[o])
    def __truediv__(self, o): return apply_op("Div", [self, o])
    def mm(self, o): return apply_op("MatMul", [self, o])
    def bmm(self, o): return apply_op("BatchedMatMul", [self, o])
    def relu(self): return apply_op("Relu", [self])
    def leakyRelu(self, slope): return apply_op("LeakyRelu", [self], d=[slope])
    def gelu(self): return apply_op("Gelu", [self])
    def sigmoid(self): return apply_op("Sigmoid", [self])
    def hardSwish(self): return apply_op("HardSwish", [self])
    def tanh(self): return apply_op("Tanh", [self])
    def softplus(self, beta, threshold): return apply_op("Softplus", [self], d=[beta, threshold])
    def exp(self): return apply_op("Exp", [self])
    def log(self): return apply_op("Log", [self])
    def log1p(self): return apply_op("Log1p", [self])
    def sin(self): return apply_op("Sin", [self])
    def cos(self): return apply_op("Cos", [self])
    def pow(self, e): return apply_op("PowConst", [self], d=[e])
    def dropout(self, prob, train): return apply_op("Dropout", [self], d=[prob], i=[int(train)])
    def sum(self, dim=(), keepDim=False): return apply_op("Sum", [self], i=[int(keepDim)] + list(dim))
    def rowSum(self): return self.sum([1], True)
    def colSum(self): return self.sum([0], True)
    def mean(self, dim, keepDim=True): return apply_op("Mean", [self], i=[int(keepDim)] + list(dim))
    def norm2(self, dim, keepDim=False): return apply_op("Norm2", [self], i=[int(keepDim)] + list(dim))
    def logSoftMax(self, dim): return apply_op("LogSoftMax", [self], i=[dim])
    def indexSelect(self, dim, index): return apply_op("IndexSelect", [self, index], i=[dim])
    def euclideanDistance(self, b, dim): return apply_op("EuclideanDistance", [self, b], i=[dim])
    def scaledDotProductAttention(self, key, value, isCausal=False, attentionBias: Optional[STen] = None):
        """ScaledDotProductAttention(query, key, value, attentionBias, isCausal) - ops.scala:2342-2390."""
        return apply_op("ScaledDotProductAttention", [self, key, value], tensors=[attentionBias] if attentionBias is not None else [], i=[int(isCausal)])
    def nllLoss(self, target: STen, weights: STen, reduction=1, ignore=-100):
        return apply_op("NllLoss", [self], tensors=[target, weights], i=[reduction, ignore])
    def mseLoss(self, target: STen, reduction=1): return apply_op("MseLoss", [self], tensors=[target], i=[reduction])
    # the rest of Variable's methods (autograd.scala:296-486) over ops2.cpp
    def select(self, dim, index): return apply_op("Select", [self], i=[dim, index])
    def slice(self, dim, start, end, step): return apply_op("Slice", [self], i=[dim, start, end, step])
    def assign(self, other): return apply_op("Assign", [self, other])
    def maskFill(self, mask, fill): return apply_op("MaskFill", [self], tensors=[mask.value], d=[fill])
    def maskSelect(self, mask): return apply_op("MaskSelect", [self, mask])
    def cast(self, scalarTypeByte): return apply_op("CastToPrecision", [self], i=[scalarTypeByte])
    def scatterAdd(self, index, dim, maxIndex): return apply_op("ScatterAdd", [self, index], i=[dim, maxIndex])
    def indexAdd(self, index, dim, maxIndex): return apply_op("IndexAdd", [self, index], i=[dim, maxIndex])
    def indexAddFromSource(self, index, dim, source): return apply_op("IndexAddToTarget", [self, source, index], i=[dim])
    def indexFill(self, index, dim, fillValue): return apply_op("IndexFill", [self, index], d=[fillValue], i=[dim])
    def expandAs(self, other: STen): return apply_op("ExpandAs", [self], tensors=[other])
    def expand(self, shape): return apply_op("Expand", [self], i=list(shape))
    def diag(self, diagonal=0): return apply_op("Diag", [self], i=[diagonal])
    def cross(self, other, dim): return apply_op("Cross", [self, other], i=[dim])
    def argmax(self, dim, keepDim=False): return apply_op("ArgMax", [self], i=[dim, 1 if keepDim else 0])
    def oneHot(self, numClasses): return apply_op("OneHot", [self], i=[numClasses])
    def eqWhere(self, b): return apply_op("EqWhere", [self], i=[int(b)])
    def tan(self): return apply_op("Tan", [self])
    def atan(self): return apply_op("ArcTan", [self])
    def powv(self, exponent): return apply_op("Pow", [self, exponent])
    def minimum(self, other): return apply_op("ElementWiseMinimum", [self, other])
    def maximum(self, other): return apply_op("ElementWiseMaximum", [self, other])
    def crossEntropy(self, other): return (self * other).rowSum() * -1.0            # autograd.scala:391-392
    def squaredFrobenius(self): return apply_op("SquaredFrobeniusMatrixNorm", [self])
    def variance(self, dim): return apply_op("Variance", [self], i=list(dim))
    def repeatInterleave(self, repeats, dim): return apply_op("RepeatInterleave", [self, repeats], i=[dim])
    def smoothL1Loss(self, target: STen, reduction=1, beta=1.0): return apply_op("SmoothL1Loss", [self], tensors=[target], d=[beta], i=[reduction])
    def binaryCrossEntropyWithLogitsLoss(self, target: STen, posWeights: Optional[STen] = None, reduction=1):
        return apply_op("BinaryCrossEntropyWithLogitsLoss", [self], tensors=[target] + ([posWeights] if posWeights is not None else []), i=[reduction])


def const(t: STen) -> Variable:
    o = C.c_void_p(); lib.lamp_var_const(C.byref(o), t.h); return Variable(o)


def param(t: STen) -> Variable:
    o = C.c_void_p(); lib.lamp_var_param(C.byref(o), t.h); return Variable(o)


def apply_op(name: str, variables: Sequence[Optional[Variable]], tensors: Sequence[STen] = (), d: Sequence[float] = (),
             i: Sequence[int] = ()) -> Variable:
    o = C.c_void_p()
    vs = handle_array([v.h if v is not None else None for v in variables])
    ts = handle_array([t.h for t in tensors])
    lib.lamp_op_apply(C.byref(o), name.encode(), vs, len(variables), ts, len(tensors), f64_array([float(x) for x in d]), len(d),
                      i64_array([int(x) for x in i]), len(i))
    return Variable(o)


# explicit constructors mirroring the case classes that take many arguments
def PackedSelfAttention(x, wQuery, wKeys, wValues, numHeads, isCausal=True):
    """self-attention's three projections + the fused attention as one node (csrc/host/ops.cpp packed_self_attention): (batch, sequence, heads x d)"""
    return apply_op("PackedSelfAttention", [x, wQuery, wKeys, wValues], i=[int(numHeads), int(isCausal)])


def Convolution(input, weight, bias, stride, padding, dilation, transposed, outputPadding, groups):
    ns = len(stride)
    return apply_op("Convolution", [input, weight, bias],
                    i=[ns] + list(stride) + list(padding) + list(dilation) + [int(transposed)] + list(outputPadding) + [groups])


def AvgPool2D(input, kernelSize, stride, padding): return apply_op("AvgPool2D", [input], i=[kernelSize, stride, padding])
def MaxPool2D(input, kernelSize, stride, padding, dilation): return apply_op("MaxPool2D", [input], i=[kernelSize, stride, padding, dilation])


def BatchNorm(input, weight, bias, runningMean: STen, runningVar: STen, training, momentum, eps):
    return apply_op("BatchNorm", [input, weight, bias], tensors=[runningMean, runningVar], d=[momentum, eps], i=[int(training)])


def BatchNorm2D(input, weight, bias, runningMean: STen, runningVar: STen, training, momentum, eps):
    return apply_op("BatchNorm2D", [input, weight, bias], tensors=[runningMean, runningVar], d=[momentum, eps], i=[int(training)])


def LayerNormOp(input, weight, bias, normalizedShape, eps):
    return apply_op("LayerNormOp", [input, weight, bias], d=[eps], i=list(normalizedShape))


def CappedShiftedNegativeExponential(a, shift): return apply_op("CappedShiftedNegativeExponential", [a], d=[shift])
def Stack(inputs, dim): return apply_op("Stack", list(inputs), i=[dim])
def Where(condition: STen, trueBranch, falseBranch): return apply_op("Where", [trueBranch, falseBranch], tensors=[condition])
def WeightNorm(v, g, dim): return apply_op("WeightNorm", [v, g], i=[dim])
def MaxPool1D(input, kernelSize, stride=1, padding=0, dilation=1): return apply_op("MaxPool1D", [input], i=[kernelSize, stride, padding, dilation])
def Embedding(input, weight): return apply_op("Embedding", [input, weight])
