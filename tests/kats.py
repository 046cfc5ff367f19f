"""The reference's own known-answer tests, transcribed once and run against BOTH the oracle
(oracle/lamp_oracle.py, CPU) and the HIP path (lamp_amd, GPU).

Source: lamp-core/src/test/scala/lamp/autograd/autograd.test.scala (line numbers per case in
tests/golden/reference_kats.json).  Every case is `testGradientAndValue(id)(input, expected)`:
the scalar L must equal `expected` to 4 decimals, and the autograd gradient w.r.t. `input` must
equal the central finite difference (eps 1e-6) to 4 decimals, in float64.

Note the fixture gotcha: saddle's Mat(Vec(1,2),Vec(3,4),Vec(5,6)) is built from
COLUMN vectors, so mat2x3 = [[1,3,5],[2,4,6]] and mat2x3_2 = [[-1,3,5],[2,-4,6]].
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
    GOLDEN = json.load(f)

mat2x3 = np.array([[1., 3., 5.], [2., 4., 6.]])
mat3x2 = mat2x3.T.copy()
mat2x3_2 = np.array([[-1., 3., 5.], [2., -4., 6.]])
nd1x2x3x3 = np.arange(18, dtype=np.float64).reshape(1, 2, 3, 3)
nd1x2x2x2 = np.ones((1, 2, 2, 2))
ar18 = np.array([1., 2, 3, 4, 5, 6] * 3)
nd3x2x3 = ar18.reshape(3, 2, 3)
nd3x3x2 = ar18.reshape(3, 3, 2)
ndx1 = np.array([1.0])
ndx2 = np.array([1.0, 1.0])
ndx3 = np.array([1.0, 2.0, 3.0])


class Backend:
    """what a KAT needs from an autograd implementation (implemented for oracle and HIP)."""
    def param(self, a): raise NotImplementedError
    def const(self, a): raise NotImplementedError
    def tensor(self, a, dtype=None): raise NotImplementedError   # plain tensor (targets, running stats)
    def scalar(self, v): raise NotImplementedError               # python float of a 1-element Variable
    def grad(self, v): raise NotImplementedError                 # numpy array
    def conv(self, x, w, b, stride, padding, dilation, transposed, out_pad, groups): raise NotImplementedError
    def batch_norm(self, x, w, b, rm, rv, training, momentum, eps): raise NotImplementedError
    def batch_norm_2d(self, x, w, b, rm, rv, training, momentum, eps): raise NotImplementedError
    def layer_norm(self, x, w, b, shape, eps): raise NotImplementedError
    def avg_pool2d(self, x, k, s, p): raise NotImplementedError
    def max_pool2d(self, x, k, s, p, d): raise NotImplementedError


def _sq_frob(w):
    return (w * w).sum()


# name -> (input array, function(backend, m) -> (L variable, variable whose gradient is checked))
def _cases():
    c = {}
    c["sum"] = (mat2x3, lambda B, m: (lambda x: (x.sum(), x))(B.param(m)))
    c["mm - left"] = (mat2x3, lambda B, m: (lambda x: (x.mm(B.param(mat3x2 * 2)).sum(), x))(B.param(m)))
    c["mm - right"] = (mat2x3, lambda B, m: (lambda x: (B.param(mat3x2 * 2).mm(x).sum(), x))(B.param(m)))
    c["add - left"] = (mat2x3, lambda B, m: (lambda x: ((x + B.param(mat2x3 * 2)).sum(), x))(B.param(m)))
    c["mult - right"] = (mat2x3, lambda B, m: (lambda x: ((B.param(mat2x3 * 2) * x).sum(), x))(B.param(m)))
    c["div - left"] = (mat2x3, lambda B, m: (lambda x: ((x / B.param(mat2x3 * 2)).sum(), x))(B.param(m)))
    c["relu"] = (mat2x3_2, lambda B, m: (lambda x: (x.relu().sum(), x))(B.param(m)))
    c["leakyrelu"] = (mat2x3_2, lambda B, m: (lambda x: (x.leakyRelu(0.5).sum(), x))(B.param(m)))
    c["gelu"] = (mat2x3_2, lambda B, m: (lambda x: (x.gelu().sum(), x))(B.param(m)))
    c["sigmoid"] = (mat2x3_2, lambda B, m: (lambda x: (x.sigmoid().sum(), x))(B.param(m)))
    c["hardswish"] = (mat2x3_2, lambda B, m: (lambda x: (x.hardSwish().sum(), x))(B.param(m + 0.1)))
    c["exp"] = (mat2x3_2, lambda B, m: (lambda x: (x.exp().sum(), x))(B.param(m)))
    c["softmax"] = (mat2x3_2, lambda B, m: (lambda x: (x.logSoftMax(1).sum(), x))(B.param(m)))
    c["mean"] = (mat2x3_2, lambda B, m: (lambda x: (x.mean([0, 1]), x))(B.param(m)))
    c["norm2"] = (mat2x3_2, lambda B, m: (lambda x: (x.norm2([0, 1], True), x))(B.param(m)))
    c["euclidean distance wrt a"] = (mat2x3_2, lambda B, m: (lambda x: (x.euclideanDistance(B.const(mat2x3), 1).sum(), x))(B.param(m)))

    def nll(B, m):
        w = B.param(m)
        data = B.const(mat3x2)
        y = B.tensor(np.array([0, 1, 2], dtype=np.int64))
        cw = B.tensor(np.ones(3))
        L = data.mm(w).logSoftMax(1).nllLoss(y, cw, 2) + _sq_frob(w)
        return L, w
    c["l2 logistic regression loss - nll_loss"] = (mat2x3_2, nll)

    def conv_w(pad):
        def f(B, m):
            x, w, b = B.param(nd1x2x3x3), B.param(m), B.param(np.ones(1))
            return B.conv(x, w, b, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1).sum(), w
        return f

    def conv_x(pad):
        def f(B, m):
            x, w, b = B.param(m), B.param(nd1x2x2x2), B.param(np.ones(1))
            return B.conv(x, w, b, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1).sum(), x
        return f

    def conv_b(B, m):
        x, w, b = B.param(nd1x2x3x3), B.param(nd1x2x2x2), B.param(m)
        return B.conv(x, w, b, [1, 1], [0, 0], [1, 1], False, [0, 0], 1).sum(), b
    c["conv2d - wrt weights"] = (nd1x2x2x2, conv_w(0))
    c["conv2d - wrt input"] = (nd1x2x3x3, conv_x(0))
    c["conv2d - padded - wrt input"] = (nd1x2x3x3, conv_x(1))
    c["conv2d - wrt bias"] = (ndx1, conv_b)
    c["maxpool2d strided"] = (nd1x2x3x3, lambda B, m: (lambda x: (B.max_pool2d(x, 2, 2, 0, 1).sum(), x))(B.param(m)))
    c["maxpool2d strided padded"] = (nd1x2x3x3, lambda B, m: (lambda x: (B.max_pool2d(x, 2, 2, 1, 1).sum(), x))(B.param(m)))
    c["avgpool2d strided padded"] = (nd1x2x3x3, lambda B, m: (lambda x: (B.avg_pool2d(x, 2, 2, 1).sum(), x))(B.param(m)))

    def bn1d_x(B, m):
        x, w, b = B.param(m), B.param(ndx3), B.param(np.zeros(3))
        return B.batch_norm(x, w, b, B.tensor(np.ones(3)), B.tensor(np.ones(3)), True, 0.1, 1e-5).sum(), x

    def bn1d_w(B, m):
        x, w, b = B.param(mat2x3), B.param(m), B.param(np.zeros(3))
        return B.batch_norm(x, w, b, B.tensor(np.ones(3)), B.tensor(np.ones(3)), True, 0.1, 1e-5).sum(), w
    c["batch norm 1d - wrt to input"] = (mat2x3, bn1d_x)
    c["batch norm 1d - wrt to weight"] = (ndx3, bn1d_w)

    def bn2d(which):
        def f(B, m):
            x = B.param(m if which == 0 else nd1x2x3x3)
            w = B.param(m if which == 1 else np.ones(2))
            b = B.param(m if which == 2 else ndx2)
            L = B.batch_norm_2d(x, w, b, B.tensor(np.zeros(2)), B.tensor(np.zeros(2)), True, 0.1, 1e-5).sum()
            return L, (x, w, b)[which]
        return f
    c["BatchNorm2D - wrt to input"] = (nd1x2x3x3, bn2d(0))
    c["BatchNorm2D - wrt to weights"] = (ndx2, bn2d(1))
    c["BatchNorm2D - wrt to bias"] = (ndx2, bn2d(2))

    def ln_x(B, m):
        x, w, b = B.param(m), B.param(ndx3), B.param(np.zeros(3))
        return B.layer_norm(x, w, b, [3], 1e-5).mean([0, 1]), x
    c["layer norm 1d - wrt to input"] = (mat2x3, ln_x)
    c["bmm - wrt left"] = (nd3x2x3, lambda B, m: (lambda x: (x.bmm(B.param(nd3x3x2)).sum(), x))(B.param(m)))
    c["bmm - wrt right"] = (nd3x3x2, lambda B, m: (lambda x: (B.param(nd3x2x3).bmm(x).sum(), x))(B.param(m)))
    return c


CASES = _cases()
EXPECTED = {k: v["expected"] for k, v in GOLDEN["autograd"].items()}
assert set(CASES) == set(EXPECTED), set(CASES) ^ set(EXPECTED)


def run_case(B: Backend, name: str, m=None, backprop=True):
    inp, f = CASES[name]
    L, v = f(B, inp if m is None else m)
    if backprop:
        L.backprop()
    return B.scalar(L), (B.grad(v) if backprop else None)


def finite_difference(B: Backend, name: str, eps=1e-6):
    inp, _ = CASES[name]
    g = np.zeros_like(inp)
    it = np.nditer(inp, flags=["multi_index"])
    for _ in it:
        idx = it.multi_index
        d = np.zeros_like(inp)
        d[idx] = eps
        g[idx] = (run_case(B, name, inp + d, False)[0] - run_case(B, name, inp - d, False)[0]) / (2 * eps)
    return g
