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


# ---- the rest of the operators the reference's gradient suite exercises (autograd.test.scala) ---------
# Each closure issues the ATen calls of the Scala closure, in its order - quirks included (named where they are).
class Stack(Op):            # ops.scala:64-72
    def __init__(self, a: Sequence[Variable], dim):
        self.params = [(v, (lambda idx: lambda p, out: out.add_(aten.select.int(p, dim, idx)))(i)) for i, v in enumerate(a)]
        self.value = Variable(aten.stack([v.value for v in a], dim), self)


class Concatenate(Op):      # ops.scala:51-62
    def __init__(self, a: Sequence[Variable], dim):
        self.params, start = [], 0
        for v in a:
            end = start + v.value.shape[dim]
            self.params.append((v, (lambda s, e: lambda p, out: out.add_(aten.slice.Tensor(p, dim, s, e, 1)))(start, end)))
            start = end
        self.value = Variable(aten.cat([v.value for v in a], dim), self)


class Reshape(Op):          # ops.scala:40-49
    def __init__(self, a, shape):
        self.params = [(a, lambda p, out: out.add_(aten.reshape(p, list(out.shape))))]
        self.value = Variable(aten.reshape(a.value, list(shape)), self)


class Select(Op):           # ops.scala:74-95
    def __init__(self, a, dim, index):
        def da(p, out):
            tmp = torch.zeros(list(out.shape), dtype=a.value.dtype)
            scalar = torch.tensor(index, dtype=torch.int64)
            pshape = list(p.shape)
            p2 = p.view(pshape[:dim] + [1] + pshape[dim:])
            out.add_(aten.index_add(tmp, dim, scalar.reshape(1), p2))
        self.params = [(a, da)]
        self.value = Variable(aten.select.int(a.value, dim, index), self)


class Slice(Op):            # ops.scala:96-119
    def __init__(self, a, dim, start, end, step):
        def da(p, out):
            tmp = torch.zeros(list(out.shape), dtype=a.value.dtype)
            out.add_(aten.index_add(tmp, dim, torch.arange(start, end, step, dtype=torch.int64), p))
        self.params = [(a, da)]
        self.value = Variable(aten.slice.Tensor(a.value, dim, start, end, step), self)


class MaskFill(Op):         # ops.scala:148-159
    def __init__(self, inp, mask: Variable, fill):
        self.params = [(inp, lambda p, out: out.add_(aten.masked_fill.Scalar(p, mask.value, 0.0)))]
        self.value = Variable(aten.masked_fill.Scalar(inp.value, mask.value, fill), self)


class MaskSelect(Op):       # ops.scala:133-146
    def __init__(self, inp, mask: Variable):
        self.params = [(inp, lambda p, out: out.add_(aten.masked_scatter(torch.zeros_like(out), mask.value, p)))]
        self.value = Variable(aten.masked_select(inp.value, mask.value), self)


class IndexFill(Op):        # ops.scala:160-177
    def __init__(self, inp, dim, index: Variable, fill):
        self.params = [(inp, lambda p, out: out.add_(aten.index_fill.int_Scalar(p, dim, index.value, 0.0)))]
        self.value = Variable(aten.index_fill.int_Scalar(inp.value, dim, index.value, fill), self)


class Where(Op):            # ops.scala:198-229
    def __init__(self, condition: torch.Tensor, t: Variable, f: Variable):
        def dt(p, out):
            out.addcmul_(p, aten.where.self(condition, torch.ones_like(t.value), torch.zeros_like(f.value)), value=1.0)

        def df(p, out):
            out.addcmul_(p, aten.where.self(condition, torch.zeros_like(t.value), torch.ones_like(f.value)), value=1.0)
        self.params = [(t, dt), (f, df)]
        self.value = Variable(aten.where.self(condition, t.value, f.value), self)


class Assign(Op):           # ops.scala:242-249
    def __init__(self, abandon, keep):
        self.params = [(abandon, lambda p, out: None), (keep, lambda p, out: out.add_(p))]
        self.value = Variable(keep.value, self)


class CastToPrecision(Op):  # ops.scala:260-288: the same precision returns the variable itself
    def __init__(self, a, dtype):
        self.params = [(a, lambda p, out: out.add_(p.to(a.value.dtype)))]
        self.value = a if a.value.dtype == dtype else Variable(a.value.to(dtype), self)


class ScatterAdd(Op):       # ops.scala:410-434
    def __init__(self, src, index: Variable, dim, maxIndex):
        assert src.value.shape[dim] == index.value.shape[dim]
        self.params = [(src, lambda p, out: out.add_(aten.gather(p, dim, index.value)))]
        shape = list(src.value.shape)
        shape[dim] = maxIndex
        self.value = Variable(aten.scatter_add(torch.zeros(shape, dtype=src.value.dtype), dim, index.value, src.value), self)


class IndexAdd(Op):         # ops.scala:436-460
    def __init__(self, src, index: Variable, dim, maxIndex):
        self.params = [(src, lambda p, out: out.add_(aten.index_select(p, dim, index.value)))]
        shape = list(src.value.shape)
        shape[dim] = maxIndex
        self.value = Variable(aten.index_add(torch.zeros(shape, dtype=src.value.dtype), dim, index.value, src.value), self)


class IndexAddToTarget(Op):  # ops.scala:462-482
    def __init__(self, target, src, index: Variable, dim):
        self.params = [(src, lambda p, out: out.add_(aten.index_select(p, dim, index.value))), (target, lambda p, out: out.add_(p))]
        self.value = Variable(aten.index_add(target.value, dim, index.value, src.value), self)


class RepeatInterleave(Op):  # ops.scala:484-509: the closure scatters back along dimension 0 whatever `dim` was
    def __init__(self, a, repeats: Variable, dim):
        def da(p, out):
            plain = torch.arange(0, a.value.shape[0], 1, dtype=torch.int64)
            rep = aten.repeat_interleave.self_Tensor(plain, repeats.value, 0)
            out.add_(aten.index_add(torch.zeros_like(out), 0, rep, p))
        self.params = [(a, da)]
        self.value = Variable(aten.repeat_interleave.self_Tensor(a.value, repeats.value, dim), self)


class ExpandAs(Op):         # ops.scala:647-653
    def __init__(self, a, other: torch.Tensor):
        self.params = [(a, lambda p, out: out.add_(unbroadcast(p, a.shape)))]
        self.value = Variable(aten.expand_as(a.value, other), self)


class Sin(Op):              # ops.scala:819-829
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.addcmul_(p, aten.cos(a.value), value=1.0))]
        self.value = Variable(aten.sin(a.value), self)


class Cos(Op):              # ops.scala:830-840
    def __init__(self, a):
        def da(p, out):
            tmp = aten.sin(a.value)
            out.addcmul_(p, tmp, value=-1.0)
        self.params = [(a, da)]
        self.value = Variable(aten.cos(a.value), self)


class Diag(Op):             # ops.scala:333-350
    def __init__(self, a, diagonal):
        def da(p, out):
            out += aten.diag(p, diagonal)
        self.params = [(a, da)]
        self.value = Variable(aten.diag(a.value, diagonal), self)


class Cross(Op):            # ops.scala:581-601 (the backward closures as the reference writes them)
    def __init__(self, a, b, dim):
        def da(p, out):
            out -= p * aten.cross(torch.ones(a.value.shape, dtype=p.dtype), b.value, dim)

        def db(p, out):
            out += p * aten.cross(torch.ones(b.value.shape, dtype=p.dtype), a.value, dim)
        self.params = [(a, da), (b, db)]
        self.value = Variable(aten.cross(a.value, b.value, dim), self)


class _NotDifferentiable(Op):   # ArgMax / OneHot (ops.scala:230-259)
    def __init__(self, a, value, name):
        def da(p, out):
            raise RuntimeError(name + " is not differentiable")
        self.params = [(a, da)]
        self.value = Variable(value, self)


class Tan(Op):              # ops.scala:841-853
    def __init__(self, a):
        def da(p, out):
            tmp1 = aten.pow.Tensor_Scalar(self.value.value, 2.0)
            tmp1 += torch.ones(1, dtype=a.value.dtype)
            out.addcmul_(p, tmp1, value=1.0)
        self.params = [(a, da)]
        self.value = Variable(aten.tan(a.value), self)


class ArcTan(Op):           # ops.scala:864-878
    def __init__(self, a):
        def da(p, out):
            tmp1 = aten.pow.Tensor_Scalar(a.value, 2.0)
            tmp1 += torch.ones(1, dtype=a.value.dtype)
            tmp1.reciprocal_()
            out.addcmul_(p, tmp1, value=1.0)
        self.params = [(a, da)]
        self.value = Variable(aten.atan(a.value), self)


class Pow(Op):              # ops.scala:890-916: the exponent is read as ONE host number; its gradient is p.unbroadcast([out.head or 1, 1]) * SUM(a^e log a)
    def __init__(self, a, exponent: Variable):
        def da(p, out):
            e = float(exponent.value.reshape(-1)[0])
            out.addcmul_(p, aten.pow.Tensor_Scalar(a.value, e - 1), value=e)

        def de(p, out):
            e = float(exponent.value.reshape(-1)[0])
            tmp3 = aten.pow.Tensor_Scalar(a.value, e) * aten.log(a.value)
            p2 = unbroadcast(p, [1 if out.dim() == 0 else out.shape[0], 1])
            out.addcmul_(p2, aten.sum(tmp3), value=1.0)
        self.params = [(a, da), (exponent, de)]
        self.value = Variable(aten.pow.Tensor_Tensor(a.value, exponent.value), self)


class Softplus(Op):         # ops.scala:989-1003
    def __init__(self, a, beta, threshold):
        self.params = [(a, lambda p, out: out.add_(aten.softplus_backward(p, a.value, beta, threshold)))]
        self.value = Variable(aten.softplus(a.value, beta, threshold), self)


class ElementWiseMinMax(Op):  # ops.scala:2287-2340: masked_scatter consumes p IN ORDER (not the elements under the mask)
    def __init__(self, a, b, is_min):
        val = aten.minimum(a.value, b.value) if is_min else aten.maximum(a.value, b.value)
        mask = aten.eq.Tensor(a.value, val)
        maskneg = aten.logical_not(mask)
        self.params = [(a, lambda p, out: out.add_(aten.masked_scatter(torch.zeros_like(out), mask, p))),
                       (b, lambda p, out: out.add_(aten.masked_scatter(torch.zeros_like(out), maskneg, p)))]
        self.value = Variable(val, self)


class Variance(Op):         # ops.scala:1055-1077: 2 / (SUM of the reduced sizes - 1), as written
    def __init__(self, a, dim):
        v, m = aten.var_mean.correction(a.value, dim, correction=1, keepdim=True)
        n = sum(a.value.shape[d] for d in dim) - 1
        self.params = [(a, lambda p, out: out.addcmul_(p, a.value - m, value=2.0 / n))]
        self.value = Variable(v, self)


class SquaredFrobeniusMatrixNorm(Op):  # ops.scala:1369-1383
    def __init__(self, a):
        self.params = [(a, lambda p, out: out.addcmul_(p, a.value, value=2.0))]
        fr = aten.linalg_vector_norm(a.value, 2.0, [-2, -1], False)
        self.value = Variable(aten.pow.Tensor_Scalar(fr, 2.0), self)


class WeightNorm(Op):       # ops.scala:1103-1160 (arXiv 1602.07868 eq. 2 and 3)
    def __init__(self, v, g, dim):
        assert v.value.dim() == 2 and list(g.value.shape) == [1, v.value.shape[1]]
        norm = aten.norm.ScalarOpt_dim(v.value, 2.0, [dim], False)

        def gradg(p):
            tmp1 = aten.sum.dim_IntList(p * v.value, [0], False)
            return tmp1 / norm

        def dv(p, out):
            tmp3 = (g.value / norm) * p
            tmp2 = g.value * gradg(p)
            tmp2 = tmp2 / norm
            tmp2 = tmp2 / norm
            tmp4 = tmp2 * v.value
            out.add_(tmp3 - tmp4)
        self.params = [(v, dv), (g, lambda p, out: out.add_(gradg(p)))]
        self.value = Variable((v.value * g.value) / norm, self)


class SmoothL1Loss(Op):     # ops.scala:1207-1247
    def __init__(self, inp, target, reduction=1, beta=1.0):
        assert inp.value.numel() == target.numel()
        tv = target.view(inp.value.shape)
        self.params = [(inp, lambda p, out: out.add_(aten.smooth_l1_loss_backward(p, inp.value, tv, reduction, beta)))]
        self.value = Variable(aten.smooth_l1_loss(inp.value, tv, reduction, beta), self)


class BinaryCrossEntropyWithLogitsLoss(Op):  # ops.scala:1309-1367
    def __init__(self, inp, target, posWeights=None, reduction=1):
        assert list(inp.value.shape) == list(target.shape)

        def da(p, out):
            if posWeights is not None:
                t = posWeights * target
                t2 = t + 1.0
                t2 -= target
                t2 *= aten.sigmoid(inp.value)
                t2 -= t
            else:
                t2 = aten.sigmoid(inp.value)
                t2 -= target
            t2 *= p
            if reduction == 1:
                t2 *= 1.0 / inp.value.numel()
            out.add_(t2)
        self.params = [(inp, da)]
        self.value = Variable(aten.binary_cross_entropy_with_logits(inp.value, target, None, posWeights, reduction), self)


class MaxPool1D(Op):        # ops.scala:1658-1715
    def __init__(self, inp, k, stride=1, padding=0, dilation=1):
        assert inp.value.dim() == 3
        out, mask = aten.max_pool1d_with_indices(inp.value, [k], [stride], [padding], [dilation], False)

        def da(p, o):
            zeros = torch.zeros_like(o)
            pf, mf, zf = p.flatten(0, 1), mask.flatten(0, 1), zeros.flatten(0, 1)
            added = [aten.index_add(zf[i], 0, mf[i], pf[i]) for i in range(pf.shape[0])]
            o.add_(aten._unsafe_view(aten.cat(added, 0), list(o.shape)))
        self.params = [(inp, da)]
        self.value = Variable(out, self)


class Embedding(Op):        # ops.scala:2141-2185: embedding_backward with padding_idx = 0 (row 0 never receives a gradient)
    def __init__(self, inp: Variable, weight: Variable):
        def dw(p, out):
            out.add_(aten.embedding_backward(p, inp.value, weight.value.shape[0], 0, False, False))
        self.params = [(weight, dw)]
        self.value = Variable(aten.embedding(weight.value, inp.value), self)


def _more_variable_methods():
    V = Variable
    V.t = lambda self: Transpose(self, 0, 1).value
    V.reshape = lambda self, shape: Reshape(self, shape).value
    V.cat = lambda self, other, dim: Concatenate([self, other], dim).value
    V.select = lambda self, dim, index: Select(self, dim, index).value
    V.slice = lambda self, dim, start, end, step: Slice(self, dim, start, end, step).value
    V.assign = lambda self, other: Assign(self, other).value
    V.maskFill = lambda self, mask, fill: MaskFill(self, mask, fill).value
    V.maskSelect = lambda self, mask: MaskSelect(self, mask).value
    V.cast = lambda self, dtype: CastToPrecision(self, dtype).value
    V.scatterAdd = lambda self, index, dim, maxIndex: ScatterAdd(self, index, dim, maxIndex).value
    V.indexAdd = lambda self, index, dim, maxIndex: IndexAdd(self, index, dim, maxIndex).value
    V.indexAddFromSource = lambda self, index, dim, source: IndexAddToTarget(self, source, index, dim).value
    V.indexFill = lambda self, index, dim, fill: IndexFill(self, dim, index, fill).value
    V.expandAs = lambda self, other: ExpandAs(self, other).value
    V.rowSum = lambda self: self.sum([1], True)
    V.colSum = lambda self: self.sum([0], True)
    V.sin = lambda self: Sin(self).value
    V.cos = lambda self: Cos(self).value
    V.diag = lambda self, diagonal=0: Diag(self, diagonal).value
    V.cross = lambda self, other, dim: Cross(self, other, dim).value
    V.argmax = lambda self, dim, keepDim=False: _NotDifferentiable(self, aten.argmax(self.value, dim, keepDim), "Argmax").value
    V.oneHot = lambda self, n: _NotDifferentiable(self, aten.one_hot(self.value, n), "OneHot").value
    V.tan = lambda self: Tan(self).value
    V.atan = lambda self: ArcTan(self).value
    V.pow = lambda self, e: PowConst(self, e).value
    V.powv = lambda self, e: Pow(self, e).value
    V.softplus = lambda self, beta, threshold: Softplus(self, beta, threshold).value
    V.minimum = lambda self, o: ElementWiseMinMax(self, o, True).value
    V.maximum = lambda self, o: ElementWiseMinMax(self, o, False).value
    V.crossEntropy = lambda self, other: (self * other).rowSum() * -1.0          # autograd.scala:391-392
    V.squaredFrobenius = lambda self: SquaredFrobeniusMatrixNorm(self).value
    V.variance = lambda self, dim: Variance(self, dim).value
    V.repeatInterleave = lambda self, repeats, dim: RepeatInterleave(self, repeats, dim).value
    V.mseLoss = lambda self, target, reduction=1: MseLoss(self, target, reduction).value
    V.smoothL1Loss = lambda self, target, reduction=1, beta=1.0: SmoothL1Loss(self, target, reduction, beta).value
    V.binaryCrossEntropyWithLogitsLoss = lambda self, target, posWeights=None, reduction=1: BinaryCrossEntropyWithLogitsLoss(self, target, posWeights, reduction).value
    V.flattenLastDimensions = lambda self, dims: self.flatten(self.value.dim() - dims, -1)


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


def umap_optimize(edge_weights_rows: torch.Tensor, total: int, lr: float, iterations: int, minDist: float, negativeSampleSize: int,
                  randomSeed: int, balance: bool = True, repulsionStrength: float = 1.0, numDim: int = 2, positiveSamples=None, losses=None):
    """Umap.optimize (umap.scala:115-286) restated op for op: float64 locations, per iteration a random subsample of the edges
    (`positiveSamples`), `negativeSampleSize` negatives per sampled edge drawn with randint(0, total - 1) (the last point is never
    drawn) and filtered by ii != jj, the loss above, backprop through the oracle's own ops, AdamW(wd 0, lr, beta2 0.95, clip 1).
    The reference draws the initial layout from the JVM's Cmwc5 generator and the samples from libtorch's: neither stream can be
    reproduced here, so both come from torch's CPU generator seeded with `randomSeed` - what is comparable between implementations
    is the DISTRIBUTION of the trajectory, not a trajectory.  edge_weights_rows: [m, 3] float64 rows (i, j, b)."""
    gen = torch.Generator().manual_seed(int(randomSeed))
    index1 = edge_weights_rows[:, 0].long()
    index2 = edge_weights_rows[:, 1].long()
    b = edge_weights_rows[:, 2].contiguous().double()
    locations = param(torch.rand(total, numDim, dtype=torch.float64, generator=gen))
    opt = AdamW([locations.value], weightDecay=0.0, learningRate=lr, beta1=0.9, beta2=0.95, clip=1.0)      # AdamW.factory defaults (AdamW.scala:12)
    last = 0.0
    for _ in range(int(iterations)):
        if positiveSamples is None:
            i1, i2, bb = index1, index2, b
        else:
            pos = torch.randint(0, index1.shape[0], (min(int(positiveSamples), index1.shape[0]),), generator=gen)
            i1, i2, bb = index1[pos], index2[pos], b[pos]
        ii = i1.repeat_interleave(int(negativeSampleSize), 0)
        jj = torch.randint(0, total - 1, (ii.shape[0],), generator=gen)
        mask = ii != jj
        lossV = umap_loss(locations, i1, i2, ii[mask], jj[mask], bb, minDist, balance, repulsionStrength)
        last = float(lossV.value.reshape(-1)[0])
        if losses is not None:
            losses.append(last)
        locations.zeroGrad()
        lossV.backprop()
        opt.step([locations.grad], 1.0)
    return locations.value, last


_more_variable_methods()
