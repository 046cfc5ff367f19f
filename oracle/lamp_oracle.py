"""CPU oracle for the lamp hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module;
the product (lamp_amd/) never does - it fails loudly when the HIP library is missing.

What it is
----------
A restatement, op for op, of what the reference executes on its CPU path: lamp's autograd /
nn / optimizer logic (Scala, reference paths below) sequencing ATen operators.  The arithmetic
of the reference lives in a third-party dependency that is NOT in /root/reference:
libtorch 2.5.1 reached through io.github.pityka:aten-scala-core:0.0.0+119-7231a9c7
(build.sbt:125, .github/workflows/ci.yml:14).  Here the same ATen operators are called through
`torch.ops.aten.*` on CPU tensors (torch 2.10 in this image; CPU kernels differ from 2.5.1 at
ulp level only, integer outputs are identical).

Pinned: tests/test_oracle_kat.py checks this module against the known-answer values hard-coded
in the reference's own tests (lamp-core/src/test/scala/lamp/autograd/autograd.test.scala,
nn/nn.test.scala, nn/adamw.test.scala, nn/sgd.test.scala, lamp-knn knn.test.scala,
lamp-umap umap.test.scala) - those constants are transcribed in tests/golden/reference_kats.json.
Unpinned parts are named in DESIGN.md (fused SDPA, bf16, UMAP layout trajectory).

Each class/function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

aten = torch.ops.aten


# =================================================================================================
# autograd core  (lamp-core/src/main/scala/lamp/autograd/autograd.scala:63-282, 488-518)
# =================================================================================================
class Variable:
    """Graph node = (op, value, pre-zeroed grad)  - autograd.scala:88-96, 176-282."""

    def __init__(self, value: torch.Tensor, op: Optional["Op"] = None, needs_grad: bool = True):
        self.value = value
        self.op = op
        # Variable.apply allocates a zeros_like grad buffer for every op output (autograd.scala:89-96)
        self.grad: Optional[torch.Tensor] = torch.zeros_like(value) if needs_grad else None

    @property
    def needsGrad(self):
        return self.grad is not None

    @property
    def shape(self):
        return list(self.value.shape)

    def zeroGrad(self):
        if self.grad is not None:
            self.grad.zero_()

    def backprop(self):
        """autograd.scala:264-282: seed fill_(1); walk the Wengert list root -> leaves."""
        if self.grad is None:
            return
        self.grad.fill_(1.0)
        for v in topological_sort(self):
            if v.op is not None:
                for (inp, fn) in v.op.params:
                    if inp.needsGrad:
                        fn(v.grad, inp.grad)

    # convenience mirrors of Variable's methods (autograd.scala:296-486)
    def mm(self, o): return MatMul(self, o).value
    def bmm(self, o): return BatchedMatMul(self, o).value
    def __add__(self, o): return (Add(self, o) if isinstance(o, Variable) else ConstAdd(self, o)).value
    def __sub__(self, o): return Minus(self, o).value
    def __mul__(self, o): return (Mult(self, o) if isinstance(o, Variable) else ConstMult(self, o)).value
    def __truediv__(self, o): return Div(self, o).value
    def relu(self): return Relu(self).value
    def gelu(self): return Gelu(self).value
    def sigmoid(self): return Sigmoid(self).value
    def tanh(self): return Tanh(self).value
    def exp(self): return Exp(self).value
    def log(self): return Log(self).value
    def log1p(self): return Log1p(self).value
    def sum(self, dim=None, keepDim=False): return Sum(self, dim or [], keepDim).value
    def mean(self, dim, keepDim=True): return Mean(self, dim, keepDim).value
    def logSoftMax(self, dim): return LogSoftMax(self, dim).value
    def view(self, shape): return View(self, shape).value
    def flatten(self, start, end=-1): return Flatten(self, start, end).value
    def transpose(self, a, b): return Transpose(self, a, b).value
    def indexSelect(self, dim, index): return IndexSelect(self, dim, index).value
    def euclideanDistance(self, b, dim): return EuclideanDistance(self, b, dim).value
    def nllLoss(self, target, weights, reduction=1, ignore=-100): return NllLoss(self, target, weights, reduction, ignore).value


def const(t: torch.Tensor) -> Variable:   # autograd/package.scala:60-68
    return Variable(t, None, needs_grad=False)


def param(t: torch.Tensor) -> Variable:   # autograd/package.scala:70-78
    return Variable(t, None, needs_grad=True)


def topological_sort(root: Variable) -> List[Variable]:
    """autograd.scala:490-518: DFS, children before parents, result root-first."""
    order: List[Variable] = []
    marks = set()

    def visit(n: Variable):
        if id(n) in marks:
            return
        if n.op is not None:
            for (c, _) in n.op.params:
                visit(c)
        marks.add(id(n))
        order.insert(0, n)

    visit(root)
    return order


def unbroadcast(p: torch.Tensor, sizes: Sequence[int]) -> torch.Tensor:
    """TensorHelpers.unbroadcast (lamp-sten/src/main/scala/lamp/TensorHelpers.scala:7-41)."""
    sizes = list(sizes)
    if list(p.shape) == sizes:
        return p
    lead = p.dim() - len(sizes)
    dims = [i for i in range(p.dim()) if i < lead or (sizes[i - lead] == 1 and p.shape[i] != 1)]
    s = aten.sum.dim_IntList(p, dims, True)
    return aten._unsafe_view(s, sizes)


class Op:
    value: Variable
    params: List[Tuple[Variable, object]]


# ---- shape ops (ops.scala:15-49, 1827-1843) -------------------------------------------------------
class Transpose(Op):
    def __init__(self, a, d1=0, d2=1):
        self.params = [(a, lambda p, out: out.add_(p.transpose(d1, d2)))]
        self.value = Variable(a.value.transpose(d1, d2), self)


class View(Op):
    def __init__(self, a, shape):
        self.params = [(a, lambda p, out: out.add_(p.reshape(out.shape)))]
        self.value = Variable(a.value.view(*shape), self)


class Flatten(Op):
    def __init__(self, a, start, end):
        self.params = [(a, lambda p, out: out.add_(p.reshape(out.shape)))]
        self.value = Variable(aten.flatten.using_ints(a.value, start, end), self)


# ---- arithmetic (ops.scala:511-621) ---------------------------------------------------------------
class Add(Op):
    def __init__(self, a, b):
        self.params = [(a, lambda p, out: out.add_(unbroadcast(p, a.shape))),
                       (b, lambda p, out: out.add_(unbroadcast(p, b.shape)))]
        self.value = Variable(aten.add.Tensor(a.value, b.value), self)


class ConstAdd(Op):
    def __init__(self, a, b: float):
        self.params = [(a, lambda p, out: out.add_(unbroadcast(p, a.shape)))]
        self.value = Variable(aten.add.Scalar(a.value, b), self)


class Minus(Op):
    def __init__(self, a, b):
        self.params = [(a, lambda p, out: out.add_(unbroadcast(p, a.shape))),
                       (b, lambda p, out: out.sub_(unbroadcast(p, b.shape)))]
        self.value = Variable(aten.sub.Tensor(a.value, b.value), self)


class ConstMult(Op):
    def __init__(self, a, b: float):
        self.params = [(a, lambda p, out: out.add_(unbroadcast(aten.mul.Scalar(p, b), a.shape)))]
        self.value = Variable(aten.mul.Scalar(a.value, b), self)


class Mult(Op):
    def __init__(self, a, b):
        self.params = [(a, lambda p, out: out.add_(unbroadcast(p * b.value, a.shape))),
                       (b, lambda p, out: out.add_(unbroadcast(p * a.value, b.shape)))]
        self.value = Variable(aten.mul.Tensor(a.value, b.value), self)


class Div(Op):
    def __init__(self, a, b):
        def db(p, out):
            tmp = self.value.value / b.value
            tmp = tmp * p
            out.sub_(unbroadcast(tmp, b.shape))
        self.params = [(a, lambda p, out: out.add_(unbroadcast(p / b.value, a.shape))), (b, db)]
        self.value = Variable(aten.div.Tensor(a.value, b.value), self)


class Sum(Op):   # ops.scala:623-630
    def __init__(self, a, dim, keepDim):
        self.params = [(a, lambda p, out: out.add_(p))]
        v = aten.sum.default(a.value) if len(dim) == 0 else aten.sum.dim_IntList(a.value, list(dim), keepDim)
        self.value = Variable(v, self)


class Mean(Op):  # ops.scala:1034-1054
    def __init__(self, a, dim, keepDim):
        n = 1
        for d in dim:
            n *= a.shape[d]
        self.params = [(a, lambda p, out: out.add_(p, alpha=1.0 / n))]
        self.value = Variable(aten.mean.dim(a.value, list(dim), keepDim), self)


class Norm2(Op):  # ops.scala:632-645
    def __init__(self, a, dim, keepDim):
        def da(p, out):
            pa = p * a.value
            pa = pa / self.value.value
            out.add_(pa)
        self.params = [(a, da)]
        self.value = Variable(aten.norm.ScalarOpt_dim(a.value, 2.0, list(dim), keepDim), self)


# ---- GEMM (ops.scala:665-724) ---------------------------------------------------------------------
class MatMul(Op):
    def __init__(self, a, b):
        # dA += p . B^T (Tensor.addmm_out_transposed2) ; dB += A^T . p (addmm_out_transposed1), beta = alpha = 1
        self.params = [(a, lambda p, out: out.copy_(aten.addmm(out, p, b.value.t(), beta=1, alpha=1))),
                       (b, lambda p, out: out.copy_(aten.addmm(out, a.value.t(), p, beta=1, alpha=1)))]
        self.value = Variable(aten.mm(a.value, b.value), self)


class BatchedMatMul(Op):
    def __init__(self, a, b):
        self.params = [(a, lambda p, out: out.copy_(aten.baddbmm(out, p, b.value.transpose(1, 2), beta=1, alpha=1))),
                       (b, lambda p, out: out.copy_(aten.baddbmm(out, a.value.transpose(1, 2), p, beta=1, alpha=1)))]
        self.value = Variable(aten.bmm(a.value, b.value), self)


# ---- element-wise functions (ops.scala:754-1032) --------------------------------------------------
class Exp(Op):
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.addcmul_(p, self.value.value, value=1.0))]
        self.value = Variable(aten.exp(a.value), self)


class Log(Op):
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.addcmul_(p, aten.reciprocal(a.value), value=1.0))]
        self.value = Variable(aten.log(a.value), self)


class Log1p(Op):
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.addcmul_(p, aten.reciprocal(a.value + 1.0), value=1.0))]
        self.value = Variable(aten.log1p(a.value), self)


class Tanh(Op):
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.add_(aten.tanh_backward(p, self.value.value)))]
        self.value = Variable(aten.tanh(a.value), self)


class PowConst(Op):
    def __init__(self, a, e: float):
        self.params = [(a, lambda p, out: out.addcmul_(p, aten.pow.Tensor_Scalar(a.value, e - 1), value=e))]
        self.value = Variable(aten.pow.Tensor_Scalar(a.value, e), self)


class Relu(Op):
    """ops.scala:918-935: dX += p * where(a < 0, 0, 1)  - the gradient at a == 0 is 1."""
    def __init__(self, a):
        def da(p, out):
            pred = aten.lt.Scalar(a.value, 0.0)
            ones = torch.ones(1, dtype=a.value.dtype)
            zeros = torch.zeros(1, dtype=a.value.dtype)
            tmp = aten.where.self(pred, zeros, ones)
            out.addcmul_(p, tmp, value=1.0)
        self.params = [(a, da)]
        self.value = Variable(aten.relu(a.value), self)


class LeakyRelu(Op):
    def __init__(self, a, slope):
        def da(p, out):
            pred = aten.lt.Scalar(a.value, 0.0)
            ones = torch.ones(1, dtype=a.value.dtype)
            s = torch.zeros(1, dtype=a.value.dtype) + slope
            out.addcmul_(p, aten.where.self(pred, s, ones), value=1.0)
        self.params = [(a, da)]
        self.value = Variable(aten.leaky_relu(a.value, slope), self)


class Gelu(Op):
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.add_(aten.gelu_backward(p, a.value)))]
        self.value = Variable(aten.gelu(a.value), self)


class Sigmoid(Op):
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.add_(aten.sigmoid_backward(p, self.value.value)))]
        self.value = Variable(aten.sigmoid(a.value), self)


class HardSwish(Op):
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.add_(aten.hardswish_backward(p, a.value)))]
        self.value = Variable(aten.hardswish(a.value), self)


class LogSoftMax(Op):   # ops.scala:955-975
    def __init__(self, a, dim):
        self.params = [(a, lambda p, out: out.add_(aten._log_softmax_backward_data(p, self.value.value, dim, a.value.dtype)))]
        self.value = Variable(aten._log_softmax(a.value, dim, False), self)


class Dropout(Op):      # ops.scala:1079-1100 (only p <= 0 is parity-relevant)
    def __init__(self, a, prob, train):
        assert prob <= 0.0, "the oracle only restates the p = 0 branch"
        self.params = [(a, lambda p, out: out.add_(p))]
        self.value = Variable(a.value, self)


class NllLoss(Op):      # ops.scala:1249-1304
    def __init__(self, inp, target, weights, reduction=1, ignore=-100):
        assert inp.value.dim() == 2 and target.dim() == 1
        v, total_weight = aten.nll_loss_forward(inp.value, target, weights, reduction, ignore)
        self.params = [(inp, lambda p, out: out.add_(
            aten.nll_loss_backward(p, inp.value, target, weights, reduction, ignore, total_weight)))]
        self.value = Variable(v, self)


class MseLoss(Op):      # ops.scala:1176-1206
    def __init__(self, inp, target, reduction=1):
        tv = target.view(inp.value.shape)
        self.params = [(inp, lambda p, out: out.add_(aten.mse_loss_backward(p, inp.value, tv, reduction)))]
        self.value = Variable(aten.mse_loss(inp.value, tv, reduction), self)


# ---- index / distance ops used by UMAP (ops.scala:179-197, 725-786) -------------------------------
class IndexSelect(Op):
    def __init__(self, inp, dim, index: Variable):
        # `val tmp = out.indexAdd(dim, index, p); out += tmp`  (yes: out ends up as 2*out + scatter(p);
        # out is a zeroed buffer whenever there is a single consumer)
        def da(p, out):
            tmp = aten.index_add(out, dim, index.value, p)
            out.add_(tmp)
        self.params = [(inp, da)]
        self.value = Variable(aten.index_select(inp.value, dim, index.value), self)


class EuclideanDistance(Op):
    def __init__(self, a, b, dim):
        self.diff = a.value - b.value
        self.norm = aten.norm.ScalarOpt_dim(self.diff, 2.0, [dim], True)
        self.params = [(a, lambda p, out: out.addcmul_(p, self.diff / self.norm, value=1.0)),
                       (b, lambda p, out: out.addcmul_(p, self.diff / self.norm, value=-1.0))]
        self.value = Variable(self.norm, self)


class CappedShiftedNegativeExponential(Op):
    def __init__(self, a, shift):
        pred = aten.le.Scalar(a.value, shift)
        ones = torch.ones(1, dtype=a.value.dtype)
        above = aten.sub.Tensor(aten.scalar_tensor(shift, dtype=a.value.dtype), a.value)
        above = aten.exp(above)
        result = aten.where.self(pred, ones, above)

        def da(p, out):
            zeros = torch.zeros(1, dtype=a.value.dtype)
            out.addcmul_(p, aten.where.self(pred, zeros, result * -1.0), value=1.0)
        self.params = [(a, da)]
        self.value = Variable(result, self)


# ---- convolution / pooling (ops.scala:1547-1825) --------------------------------------------------
class Convolution(Op):
    def __init__(self, inp, weight, bias, stride, padding, dilation, transposed, outputPadding, groups):
        args = (stride, padding, dilation, transposed, outputPadding, groups)
        bs = list(bias.value.shape)

        def back(mask, idx):
            def f(p, out):
                r = aten.convolution_backward(p, inp.value, weight.value, bs, *args, mask)
                out.add_(r[idx])
            return f
        self.params = [(inp, back([True, False, False], 0)), (weight, back([False, True, False], 1)),
                       (bias, back([False, False, True], 2))]
        self.value = Variable(aten.convolution(inp.value, weight.value, bias.value, *args), self)


class AvgPool2D(Op):
    def __init__(self, inp, kernelSize, stride, padding):
        a = ([kernelSize], [stride], [padding], False, True, None)
        self.params = [(inp, lambda p, out: out.add_(aten.avg_pool2d_backward(p, inp.value, *a)))]
        self.value = Variable(aten.avg_pool2d(inp.value, *a), self)


class MaxPool2D(Op):
    def __init__(self, inp, kernelSize, stride, padding, dilation):
        a = ([kernelSize], [stride], [padding], [dilation], False)
        out, mask = aten.max_pool2d_with_indices(inp.value, *a)
        self.mask = mask
        self.params = [(inp, lambda p, o: o.add_(aten.max_pool2d_with_indices_backward(p, inp.value, *a, mask)))]
        self.value = Variable(out, self)


# ---- normalisation (ops.scala:1846-2140) ----------------------------------------------------------
class BatchNorm(Op):
    """1-D variant: flattens dims 1.. (ops.scala:1858) then native_batch_norm."""
    def __init__(self, inp, weight, bias, runningMean, runningVar, training, momentum, eps):
        x = aten.flatten.using_ints(inp.value, 1, inp.value.dim() - 1)
        out, save_mean, save_invstd = aten.native_batch_norm(x, weight.value, bias.value, runningMean, runningVar,
                                                             training, momentum, eps)

        def back(mask, idx):
            def f(p, o):
                fp = aten.flatten.using_ints(p, 1, p.dim() - 1)
                r = aten.native_batch_norm_backward(fp, x, weight.value, runningMean, runningVar, save_mean,
                                                    save_invstd, training, eps, mask)
                o.add_(r[idx].reshape(o.shape))
            return f

        def dbias(p, o):
            fp = aten.flatten.using_ints(p, 1, p.dim() - 1)
            o.add_(unbroadcast(fp, o.shape))
        self.params = [(inp, back([True, False, False], 0)), (weight, back([False, True, False], 1)), (bias, dbias)]
        self.value = Variable(out.reshape(inp.value.shape), self)


class BatchNorm2D(Op):
    def __init__(self, inp, weight, bias, runningMean, runningVar, training, momentum, eps):
        x = inp.value
        out, save_mean, save_invstd = aten.native_batch_norm(x, weight.value, bias.value, runningMean, runningVar,
                                                             training, momentum, eps)

        def back(mask, idx):
            def f(p, o):
                r = aten.native_batch_norm_backward(p, x, weight.value, runningMean, runningVar, save_mean,
                                                    save_invstd, training, eps, mask)
                o.add_(r[idx].reshape(o.shape))
            return f

        def dbias(p, o):
            tgt = list(o.shape) + [1] * (p.dim() - 2)
            o.add_(unbroadcast(p, tgt).reshape(o.shape))
        self.params = [(inp, back([True, False, False], 0)), (weight, back([False, True, False], 1)), (bias, dbias)]
        self.value = Variable(out, self)


class LayerNormOp(Op):   # ops.scala:1956-2032
    def __init__(self, inp, weight: Optional[Variable], bias: Optional[Variable], normalizedShape, eps):
        w = weight.value if weight is not None else None
        b = bias.value if bias is not None else None
        out, mean, rstd = aten.native_layer_norm(inp.value, normalizedShape, w, b, eps)

        def back(mask, idx):
            def f(p, o):
                r = aten.native_layer_norm_backward(p, inp.value, normalizedShape, mean, rstd, w, b, mask)
                o.add_(r[idx])
            return f
        self.params = [(inp, back([True, False, False], 0))]
        if weight is not None:
            self.params.append((weight, back([False, True, False], 1)))
        if bias is not None:
            self.params.append((bias, back([False, False, True], 2)))
        self.value = Variable(out, self)


# =================================================================================================
# nn modules (lamp-core/src/main/scala/lamp/nn/*.scala) - deterministic init supplied by callers
# =================================================================================================
class Module:
    def state(self) -> List[Variable]:   # all tensors incl. consts, in lamp's order
        raise NotImplementedError

    def parameters(self) -> List[Variable]:   # Module.scala:290-296: the ones with needsGrad
        return [v for v in self.state() if v.needsGrad]

    def forward(self, x: Variable) -> Variable:
        raise NotImplementedError

    def gradients(self, loss: Variable, zeroGrad=True):   # Module.scala:300-314
        if zeroGrad:
            for p in self.parameters():
                p.zeroGrad()
        loss.backprop()
        return [p.grad for p in self.parameters()]


class Linear(Module):      # nn/Linear.scala:7-67 - x.mm(W) then bias[1,out] + v
    def __init__(self, weights: Variable, bias: Optional[Variable]):
        self.weights, self.bias = weights, bias

    def state(self): return [self.weights] + ([self.bias] if self.bias is not None else [])

    def forward(self, x):
        v = x.mm(self.weights)
        return (self.bias + v) if self.bias is not None else v


class Conv2D(Module):      # nn/Conv2D.scala:8-83 - bias is always a tensor (const zeros when bias=false)
    def __init__(self, weights, bias, stride=1, padding=0, dilation=1, groups=1):
        self.weights, self.bias, self.stride, self.padding, self.dilation, self.groups = weights, bias, stride, padding, dilation, groups

    def state(self): return [self.weights, self.bias]

    def forward(self, x):
        return Convolution(x, self.weights, self.bias, [self.stride] * 2, [self.padding] * 2, [self.dilation] * 2,
                           False, [0, 0], self.groups).value


class BatchNormModule(Module):   # nn/BatchNorm.scala:7-88
    def __init__(self, weight, bias, runningMean, runningVar, training=True, momentum=0.1, eps=1e-5):
        self.weight, self.bias, self.runningMean, self.runningVar = weight, bias, runningMean, runningVar
        self.training, self.momentum, self.eps = training, momentum, eps

    def state(self): return [self.weight, self.bias, self.runningMean, self.runningVar]

    def forward(self, x):
        return BatchNorm(x, self.weight, self.bias, self.runningMean.value, self.runningVar.value, self.training,
                         self.momentum, self.eps).value


class BatchNorm2DModule(BatchNormModule):   # nn/BatchNorm2D.scala:8-70
    def forward(self, x):
        return BatchNorm2D(x, self.weight, self.bias, self.runningMean.value, self.runningVar.value, self.training,
                           self.momentum, self.eps).value


class Fun(Module):
    def __init__(self, f): self.f = f
    def state(self): return []
    def forward(self, x): return self.f(x)


class Sequential(Module):
    def __init__(self, *mods): self.mods = list(mods)
    def state(self): return [s for m in self.mods for s in m.state()]

    def forward(self, x):
        for m in self.mods:
            x = m.forward(x)
        return x


class Residual(Module):    # example-cifar100/src/main/scala/lamp/example/cifar/cnn.scala:11-21
    def __init__(self, right: Module, left: Optional[Module]): self.right, self.left = right, left
    def state(self): return self.right.state() + (self.left.state() if self.left else [])

    def forward(self, x):
        r = self.right.forward(x)
        l = self.left.forward(x) if self.left else x
        return r + l


# ---- deterministic (closed form, no RNG) initialisation shared with the HIP side ------------------
def closed_form(n: int, salt: int = 0, scale: float = 1.0, dtype=torch.float64) -> torch.Tensor:
    """x[i] = (((i + salt) * 7919) mod 1009) / 1009 - 0.5, scaled."""
    i = torch.arange(n, dtype=torch.int64) + salt
    return (((i * 7919) % 1009).to(torch.float64) / 1009.0 - 0.5).mul(scale).to(dtype)


def make_bn(c, dtype, salt, two_d):
    w = param((closed_form(c, salt, 0.5, dtype) + 1.0))
    b = param(closed_form(c, salt + 17, 0.2, dtype))
    rm = const(torch.zeros(c, dtype=dtype))
    rv = const(torch.zeros(c, dtype=dtype))   # running_var initialised to 0 (BatchNorm2D.scala:62-66)
    return (BatchNorm2DModule if two_d else BatchNormModule)(w, b, rm, rv)


def make_conv(cin, cout, k, dtype, salt, stride=1, padding=0):
    std = math.sqrt(2.0 / (cout + cin))
    w = param(closed_form(cout * cin * k * k, salt, 2.0 * std, dtype).reshape(cout, cin, k, k))
    b = const(torch.zeros(cout, dtype=dtype))
    return Conv2D(w, b, stride=stride, padding=padding)


def residual_make(cin, cout, dtype, stride, salt):   # cnn.scala:33-87 with dropout = 0
    right = Sequential(make_conv(cin, cout, 3, dtype, salt, stride=stride, padding=1), make_bn(cout, dtype, salt + 1, True),
                       Fun(lambda v: v.relu()),
                       make_conv(cout, cout, 3, dtype, salt + 2, stride=1, padding=1), make_bn(cout, dtype, salt + 3, True))
    left = None
    if not (cin == cout and stride == 1):
        left = Sequential(make_conv(cin, cout, 1, dtype, salt + 4, stride=stride, padding=0), make_bn(cout, dtype, salt + 5, True))
    return Sequential(Residual(right, left), Fun(lambda v: v.relu()))


def resnet(num_classes: int, dtype=torch.float32) -> Sequential:
    """Cnn.resnet (cnn.scala:89-137), dropout 0, deterministic closed-form weights."""
    return Sequential(
        make_conv(3, 6, 5, dtype, 100, padding=2),
        Sequential(residual_make(6, 6, dtype, 2, 200), residual_make(6, 16, dtype, 2, 300),
                   residual_make(16, 128, dtype, 1, 400), residual_make(128, num_classes, dtype, 1, 500)),
        Fun(lambda v: AvgPool2D(v, 8, 1, 0).value),
        Fun(lambda v: v.flatten(v.value.dim() - 3)),
        Fun(lambda v: v.logSoftMax(1)))


def mlp(in_, out, hidden: Sequence[int], dtype=torch.float32) -> Sequential:
    """MLP.apply (nn/MLP.scala:40-167) with the defaults: BatchNorm, relu, dropout 0 => Linear without bias."""
    mods = []
    dims = [in_] + list(hidden)
    salt = 1000
    for a, b in zip(dims[:-1], dims[1:]):
        std = math.sqrt(2.0 / (a + b))
        mods.append(Sequential(Linear(param(closed_form(a * b, salt, 2 * std, dtype).reshape(a, b)), None),
                               make_bn(b, dtype, salt + 1, False), Fun(lambda v: v.relu())))
        salt += 10
    a = dims[-1]
    std = math.sqrt(2.0 / (a + out))
    mods.append(Sequential(Linear(param(closed_form(a * out, salt, 2 * std, dtype).reshape(a, out)), None),
                           make_bn(out, dtype, salt + 1, False)))
    return Sequential(*mods)


# =================================================================================================
# optimizers (nn/AdamW.scala:29-177, nn/SGD.scala:19-99, nn/package.scala:72-100)
# =================================================================================================
def gradient_clipping_in_place(gradients: Sequence[Optional[torch.Tensor]], theta: float):
    gs = [g for g in gradients if g is not None]
    one = torch.ones(1, dtype=gs[0].dtype)
    s = torch.zeros(1, dtype=gs[0].dtype)
    for g in gs:
        s += aten.pow.Tensor_Scalar(aten.norm.ScalarOpt_dim(g.view(-1), 2.0, [0], False), 2.0)
    norm = aten.sqrt(s)
    scalar = torch.tensor(theta, dtype=gs[0].dtype) / norm
    s2 = aten.minimum(scalar, one)
    for g in gs:
        g.mul_(s2)


class AdamW:
    def __init__(self, parameters: Sequence[torch.Tensor], weightDecay, learningRate=0.001, beta1=0.9, beta2=0.999,
                 eps=1e-8, clip=None, debias=True, mixedPrecision=False):
        self.parameters = list(parameters)
        self.wd, self.lr, self.b1, self.b2, self.eps, self.clip, self.debias, self.mixed = weightDecay, learningRate, beta1, beta2, eps, clip, debias, mixedPrecision
        up = (lambda t: t.float() if (mixedPrecision and t.dtype in (torch.float16, torch.bfloat16)) else t)
        self.up = up
        self.workingCopy = [up(p).clone() if up(p) is not p else None for p in self.parameters]
        self.mt = [up(torch.zeros_like(p)) for p in self.parameters]
        self.vt = [up(torch.zeros_like(p)) for p in self.parameters]
        self.stepCount = 0

    def step(self, gradients: Sequence[Optional[torch.Tensor]], scheduleFactor: float = 1.0):
        if self.clip is not None:
            gradients_present = [g for g in gradients if g is not None]
            gradient_clipping_in_place(gradients_present, self.clip)
        self.stepCount += 1
        for p_model, g0, mt, vt, wc in zip(self.parameters, gradients, self.mt, self.vt, self.workingCopy):
            if g0 is None:
                continue
            g = self.up(g0)
            mt.mul_(self.b1)
            mt.add_(g, alpha=(1.0 - self.b1))
            vt.mul_(self.b2)
            vt.addcmul_(g, g, value=1 - self.b2)
            denom = aten.sqrt(vt)
            denom.add_(self.eps)
            if self.debias:
                stepParam = scheduleFactor * self.lr * math.sqrt(1 - math.pow(self.b2, float(self.stepCount))) / (
                    1 - math.pow(self.b1, float(self.stepCount)))
            else:
                stepParam = scheduleFactor * self.lr
            stepWd = stepParam * self.wd
            p = wc if wc is not None else p_model
            if self.wd != 0.0:
                p.add_(p, alpha=-1 * stepWd)
            p.addcdiv_(mt, denom, value=-1 * stepParam)
            if wc is not None:
                p_model.copy_(p.to(p_model.dtype))


class SGDW:
    def __init__(self, parameters, learningRate, weightDecay, momentum=None, clip=None):
        self.parameters, self.lr, self.wd, self.momentum, self.clip = list(parameters), learningRate, weightDecay, momentum, clip
        self.velocity = [torch.zeros_like(p) if momentum is not None else None for p in self.parameters]

    def step(self, gradients, scheduleFactor=1.0):
        if self.clip is not None:
            gradient_clipping_in_place([g for g in gradients if g is not None], self.clip)
        for p, g, v in zip(self.parameters, gradients, self.velocity):
            if g is None:
                continue
            if v is None:
                if self.wd != 0.0:
                    p.add_(p, alpha=-1 * self.wd * scheduleFactor)
                p.add_(g, alpha=-1 * self.lr * scheduleFactor)
            else:
                v.mul_(self.momentum)
                v.add_(g, alpha=self.lr * scheduleFactor)
                if self.wd != 0.0:
                    p.add_(p, alpha=-1 * self.wd * scheduleFactor)
                p.add_(v, alpha=-1)


# =================================================================================================
# training step (nn/SupervisedModel.scala:190-211; lamp-data IOLoops.scala:621-658)
# =================================================================================================
def nll_loss(numClasses, classWeights, reduction=1, ignore=-100):   # LossFunctions.scala:39-55
    def f(out: Variable, target: torch.Tensor):
        return out.nllLoss(target, classWeights, reduction, ignore), out.shape[0]
    return f


def training_step(module: Module, loss_fn, x: torch.Tensor, target: torch.Tensor, optimizer, acc: Optional[torch.Tensor] = None):
    """addTotalLossAndReturnGradientsAndNumExamples + optimizer.step: returns (loss value, grads)."""
    out = module.forward(const(x))
    loss, n = loss_fn(out, target)
    grads = module.gradients(loss, zeroGrad=True)
    if acc is not None:
        acc += loss.value * float(n)
    if optimizer is not None:
        optimizer.step(grads, 1.0)
    return loss.value.clone(), grads


# =================================================================================================
# data parallel gradient averaging (lamp-data/.../distributed/package.scala:690-719)
# =================================================================================================
def average_gradients(per_rank_grads: Sequence[Sequence[torch.Tensor]], per_rank_examples: Sequence[int]):
    """g_r *= n_r ; sum over ranks ; / sum n  - what root ends up with after ncclReduce."""
    total = float(sum(per_rank_examples))
    out = []
    for k in range(len(per_rank_grads[0])):
        s = torch.zeros_like(per_rank_grads[0][k])
        for r, gs in enumerate(per_rank_grads):
            s += gs[k] * float(per_rank_examples[r])
        out.append(s / total)
    return out


# =================================================================================================
# kNN (lamp-knn/src/main/scala/lamp/knn/package.scala:21-80)
# =================================================================================================
def squared_euclidean_distance(v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
    outer = aten.mm(v1, v2.t())
    n1 = aten.sum.dim_IntList(v1 * v1, [1], True)
    n2 = aten.sum.dim_IntList(v2 * v2, [1], True)
    return aten.maximum(n1 + n2.t() - outer * 2, torch.zeros(1, dtype=v1.dtype))


def jaccard_distance(v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:    # package.scala:32-44
    outer = aten.mm(v1, v2.t())
    n1 = aten.sum.dim_IntList(v1, [1], True)
    n2 = aten.sum.dim_IntList(v2, [1], True)
    denom = n1 + n2.t() - outer
    sim = outer / denom
    return torch.ones(1, dtype=sim.dtype) - sim


def knn_minibatched(d: torch.Tensor, query: torch.Tensor, k: int, minibatch: int, distance=None) -> torch.Tensor:
    outs = []
    for s in range(0, query.shape[0], minibatch):
        dist = (distance or squared_euclidean_distance)(query[s:s + minibatch], d)
        _, idx = aten.topk(dist, k, 1, False, False)
        outs.append(idx)
    return torch.cat(outs, 0)


# =================================================================================================
# UMAP (lamp-umap/src/main/scala/lamp/umap/umap.scala:14-286)
# =================================================================================================
def _binary_search(target, fun, eps=1e-6):
    lo, hi, mid, it = 0.0, float("inf"), 1.0, 0
    while True:
        if it > 1000:
            return mid
        at = fun(mid)
        if abs(at - target) < eps:
            return mid
        if at > target:
            hi = mid
            mid = (lo + mid) * 0.5
        else:
            lo = mid
            mid = mid * 2 if math.isinf(hi) else (hi + mid) * 0.5
        it += 1


def edge_weights(knn_distances, knn):
    """umap.scala:50-113. knn_distances: [n,k] floats (python lists / arrays), knn: [n,k] ints.
    Returns rows (i, j, b) for i != j in the reference's emission order."""
    n, k = len(knn), len(knn[0])
    rho = [min(d for d in knn_distances[i] if d > 0) for i in range(n)]
    log2k = math.log(k) / math.log(2.0)
    sigma = []
    for i in range(n):
        r = rho[i]
        sigma.append(_binary_search(log2k, lambda s: sum(math.exp((-1 * max(0.0, d - r)) / s) for d in knn_distances[i])))
    rows = []
    for i in range(n):
        for jidx, j in enumerate(knn[i]):
            if i == j:
                continue
            d = knn_distances[i][jidx]
            wij = math.exp((-1 * max(0.0, d - rho[i])) / sigma[i])
            row_j = list(knn[j])
            if i in row_j:
                l = row_j.index(i)
                wji = math.exp((-1 * max(0.0, knn_distances[j][l] - rho[j])) / sigma[j])
            else:
                wji = 0.0
            rows.append((float(i), float(j), wij + wji - wij * wji))
    return rows


def umap_loss(locations: Variable, index1, index2, index3, index4, b: torch.Tensor, minDist=0.0,
              balance=True, repulsionStrength=1.0) -> Variable:
    """umap.scala:132-176 (float64 in the reference)."""
    i1, i2, i3, i4 = const(index1), const(index2), const(index3), const(index4)
    bv = const(b)
    l1, l2 = locations.indexSelect(0, i1), locations.indexSelect(0, i2)
    l3, l4 = locations.indexSelect(0, i3), locations.indexSelect(0, i4)
    n1 = l1.euclideanDistance(l2, 1).view([-1])
    if minDist == 0.0:
        attractions = (n1 * bv).sum() * (-1.0)
    else:
        attractions = (CappedShiftedNegativeExponential(n1, minDist).value.log() * bv).sum()
    n2 = l3.euclideanDistance(l4, 1).view([-1])
    if minDist == 0.0:
        repulsions = ((n2 * (-1.0)).exp() * (-1.0)).log1p().sum()
    else:
        p = CappedShiftedNegativeExponential(n2, minDist).value * (-1.0) + 1e-6
        repulsions = p.log1p().sum()
    if balance:
        return (attractions / bv.sum() + repulsions * (repulsionStrength / l3.shape[0])) * (-1.0)
    return (attractions + repulsions) * (-1.0)
