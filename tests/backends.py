"""The two autograd implementations the KATs run against."""
import numpy as np
import torch

from tests.kats import Backend


class OracleBackend(Backend):
    def __init__(self, dtype=torch.float64):
        from oracle import lamp_oracle as O
        self.O, self.dtype = O, dtype

    def _t(self, a, dtype=None):
        a = np.asarray(a)
        if a.dtype == np.int64:
            return torch.from_numpy(a.copy())
        return torch.tensor(a, dtype=dtype or self.dtype)

    def param(self, a): return self.O.param(self._t(a))
    def const(self, a): return self.O.const(self._t(a))
    def tensor(self, a, dtype=None): return self._t(a, dtype)
    def scalar(self, v): return float(v.value.reshape(-1)[0])
    def grad(self, v): return v.grad.detach().double().numpy().copy()
    def conv(self, x, w, b, stride, padding, dilation, transposed, out_pad, groups):
        return self.O.Convolution(x, w, b, stride, padding, dilation, transposed, out_pad, groups).value
    def batch_norm(self, x, w, b, rm, rv, training, momentum, eps): return self.O.BatchNorm(x, w, b, rm, rv, training, momentum, eps).value
    def batch_norm_2d(self, x, w, b, rm, rv, training, momentum, eps): return self.O.BatchNorm2D(x, w, b, rm, rv, training, momentum, eps).value
    def layer_norm(self, x, w, b, shape, eps): return self.O.LayerNormOp(x, w, b, shape, eps).value
    def avg_pool2d(self, x, k, s, p): return self.O.AvgPool2D(x, k, s, p).value
    def max_pool2d(self, x, k, s, p, d): return self.O.MaxPool2D(x, k, s, p, d).value
    def max_pool1d(self, x, k, s, p, d): return self.O.MaxPool1D(x, k, s, p, d).value
    def stack(self, xs, dim): return self.O.Stack(xs, dim).value
    def where(self, cond, a, b): return self.O.Where(cond, a, b).value
    def weight_norm(self, v, g, dim): return self.O.WeightNorm(v, g, dim).value
    def embedding(self, inp, weight): return self.O.Embedding(inp, weight).value
    def capped_exp(self, x, shift): return self.O.CappedShiftedNegativeExponential(x, shift).value
    def cast_single(self, x): return x.cast(torch.float32)
    def eq_scalar(self, t, v): return torch.ops.aten.eq.Scalar(t, v)
    def const_bool(self, t): return self.O.const(t)
    def param_transposed01(self, a): return self.O.param(self._t(a).transpose(0, 1))
    def param_f32(self, a): return self.O.param(torch.tensor(np.asarray(a), dtype=torch.float32))
    def sdpa(self, q, k, v, causal):
        from oracle import lamp_transformer_oracle as TO
        return TO.ScaledDotProductAttention(q, k, v, causal).value


# give the oracle Variable the few method spellings the KATs use (lamp's names)
def _patch_oracle():
    from oracle import lamp_oracle as O
    V = O.Variable
    V.leakyRelu = lambda self, s: O.LeakyRelu(self, s).value
    V.hardSwish = lambda self: O.HardSwish(self).value
    V.norm2 = lambda self, dim, keepDim=False: O.Norm2(self, dim, keepDim).value
    V.nllLoss = lambda self, target, weights, reduction=1, ignore=-100: O.NllLoss(self, target, weights, reduction, ignore).value


_patch_oracle()


class HipBackend(Backend):
    def __init__(self, dtype=None, device=0):
        from lamp_amd import sten, autograd
        self.S, self.A = sten, autograd
        self.dtype = sten.F64 if dtype is None else dtype
        self.device = device

    def _t(self, a, dtype=None):
        a = np.asarray(a)
        if a.dtype == np.int64:
            return self.S.STen.from_numpy(a, device=self.device)
        return self.S.STen.from_numpy(a.astype(np.float64), device=self.device, dtype=dtype or self.dtype)

    def param(self, a): return self.A.param(self._t(a))
    def const(self, a): return self.A.const(self._t(a))
    def tensor(self, a, dtype=None): return self._t(a, dtype)
    def scalar(self, v): return float(v.value.to_numpy().reshape(-1)[0])
    def grad(self, v): return v.partialDerivative.to_numpy().astype(np.float64)
    def conv(self, x, w, b, stride, padding, dilation, transposed, out_pad, groups):
        return self.A.Convolution(x, w, b, stride, padding, dilation, transposed, out_pad, groups)
    def batch_norm(self, x, w, b, rm, rv, training, momentum, eps): return self.A.BatchNorm(x, w, b, rm, rv, training, momentum, eps)
    def batch_norm_2d(self, x, w, b, rm, rv, training, momentum, eps): return self.A.BatchNorm2D(x, w, b, rm, rv, training, momentum, eps)
    def layer_norm(self, x, w, b, shape, eps): return self.A.LayerNormOp(x, w, b, shape, eps)
    def avg_pool2d(self, x, k, s, p): return self.A.AvgPool2D(x, k, s, p)
    def max_pool2d(self, x, k, s, p, d): return self.A.MaxPool2D(x, k, s, p, d)
    def max_pool1d(self, x, k, s, p, d): return self.A.MaxPool1D(x, k, s, p, d)
    def stack(self, xs, dim): return self.A.Stack(xs, dim)
    def where(self, cond, a, b): return self.A.Where(cond, a, b)
    def weight_norm(self, v, g, dim): return self.A.WeightNorm(v, g, dim)
    def embedding(self, inp, weight): return self.A.Embedding(inp, weight)
    def capped_exp(self, x, shift): return self.A.CappedShiftedNegativeExponential(x, shift)
    def cast_single(self, x): return x.cast(self.S.F32)
    def eq_scalar(self, t, v): return t.equ(v)
    def const_bool(self, t): return self.A.const(t)
    def param_transposed01(self, a): return self.A.param(self._t(a).transpose(0, 1))
    def param_f32(self, a): return self.A.param(self.S.STen.from_numpy(np.asarray(a, dtype=np.float32), device=self.device, dtype=self.S.F32))
    def sdpa(self, q, k, v, causal): return q.scaledDotProductAttention(k, v, causal, None)
