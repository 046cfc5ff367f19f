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
    def max_pool1d(self, x, k, s, p, d): raise NotImplementedError
    def stack(self, xs, dim): raise NotImplementedError
    def where(self, cond, a, b): raise NotImplementedError          # cond: plain bool tensor
    def weight_norm(self, v, g, dim): raise NotImplementedError
    def embedding(self, inp, weight): raise NotImplementedError
    def capped_exp(self, x, shift): raise NotImplementedError
    def cast_single(self, x): raise NotImplementedError             # Variable.cast(SinglePrecision)
    def eq_scalar(self, t, v): raise NotImplementedError            # plain tensor -> bool tensor


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
    _more_cases(c)
    return c


mat1x1 = np.array([[1.0]])
mat3x1 = np.array([[1.0], [2.0], [3.0]])
mat1x3 = mat3x1.T.copy()
mat3x1_2 = np.array([[2.0], [3.0], [4.0]])
mat2x2 = np.array([[4.0, 1.0], [6.0, 2.0]])            # Mat(Vec(4, 1), Vec(6, 2)).T
nd1x2x3 = mat2x3.reshape(1, 2, 3)
nd1x2x3_2 = (mat2x3 * 3).reshape(1, 2, 3)
nd1x2x2 = np.ones((1, 2, 2))
nd1x4x3x3 = np.arange(36, dtype=np.float64).reshape(1, 4, 3, 3)
nd2x2x2x2 = np.ones((2, 2, 2, 2))
ndx6 = np.arange(1.0, 7.0)
ndx18 = ar18.copy()


def _module_cases(c):
    """composite module KATs of nn.test.scala that are stated with deterministic parameters (the others draw theirs from libtorch's
    RNG): `testGradientAndValue(id)(input, module, expected)` - value of module.forward(input).sum to 1e-6, gradients of the module's
    parameters against central differences to 4 decimals.  One case per checked parameter."""
    eye3 = np.eye(3)

    def linear0(which):
        def f(B, m):
            w = B.param(m if which == 0 else np.ones((3, 1)))
            b = B.param(m if which == 1 else np.ones(1))
            return (B.const(mat2x3).mm(w) + b).sum(), (w, b)[which]
        return f
    c["nn Linear 0 - wrt weight"] = (np.ones((3, 1)), linear0(0))
    c["nn Linear 0 - wrt bias"] = (np.ones(1), linear0(1))

    def logistic2(which):
        def f(B, m):
            w = B.param(m if which == 0 else np.ones((2, 3)))
            b = B.param(m if which == 1 else np.ones((1, 3)))
            L = (B.const(mat3x2).mm(w) + b).logSoftMax(1).crossEntropy(B.const(eye3)).sum() + _sq_frob(w) + _sq_frob(b)
            return L, (w, b)[which]
        return f
    c["nn Logistic 2 - wrt weight"] = (np.ones((2, 3)), logistic2(0))
    c["nn Logistic 2 - wrt bias"] = (np.ones((1, 3)), logistic2(1))

    def mlp1(which):
        def f(B, m):
            shapes = [(2, 32), (1, 32), (32, 3), (1, 3)]
            ps = [B.param(m if which == i else np.ones(sh)) for i, sh in enumerate(shapes)]
            h = ((B.const(mat3x2).mm(ps[0]) + ps[1]).logSoftMax(1).gelu()).mm(ps[2]) + ps[3]
            L = h.crossEntropy(B.const(eye3)).sum()
            for q in ps:
                L = L + _sq_frob(q)
            return L, ps[which]
        return f
    c["nn Mlp1 - wrt first weight"] = (np.ones((2, 32)), mlp1(0))
    c["nn Mlp1 - wrt second weight"] = (np.ones((32, 3)), mlp1(2))
    c["nn Mlp1 - wrt second bias"] = (np.ones((1, 3)), mlp1(3))


def _more_cases(c):
    _module_cases(c)
    """the remaining cases of autograd.test.scala that stay inside SURVEY section 8 (linear algebra and sparse tensors are
    not mirrored).  Inputs and operator calls transcribed from the lines given in tests/golden/reference_kats.json."""
    P2 = lambda B: B.param(mat2x3 * 2)

    def un(f): return lambda B, m: (lambda x: (f(B, x), x))(B.param(m))
    c["diag"] = (mat3x1, un(lambda B, x: x.view([-1]).diag(0).sum()))
    c["cross left"] = (mat2x3, un(lambda B, x: x.cross(B.const(mat2x3_2), 1).sum()))
    c["cross right"] = (mat2x3, un(lambda B, x: B.const(mat2x3_2).cross(x, 1).sum()))
    c["colSum"] = (mat2x3, un(lambda B, x: x.colSum().sum()))
    c["rowSum"] = (mat2x3, un(lambda B, x: x.rowSum().sum()))
    c["assign - right"] = (mat2x3, un(lambda B, x: P2(B).assign(x).sum()))
    c["assign - left"] = (mat2x3, un(lambda B, x: x.assign(P2(B)).sum()))
    c["add broadcasted - left"] = (mat1x1, un(lambda B, x: (x + P2(B)).sum()))
    c["add broadcasted - right"] = (mat1x1, un(lambda B, x: (P2(B) + x).sum()))
    c["add - right"] = (mat2x3, un(lambda B, x: (P2(B) + x).sum()))
    c["minus - left"] = (mat2x3, un(lambda B, x: (x - P2(B)).sum()))
    c["minus broadcasted - left"] = (mat1x1, un(lambda B, x: (x - P2(B)).sum()))
    c["minus broadcasted - right"] = (mat1x1, un(lambda B, x: (P2(B) - x).sum()))
    c["minus - right"] = (mat2x3, un(lambda B, x: (P2(B) - x).sum()))
    c["constmult"] = (mat2x3, un(lambda B, x: (x * 2.0).sum()))
    c["cast to float"] = (mat2x3, un(lambda B, x: B.cast_single(x).sum()))
    c["constadd"] = (mat2x3, un(lambda B, x: (x + 2.0).sum()))
    c["mult broadcasted - left"] = (mat1x1, un(lambda B, x: (x * P2(B)).sum()))
    c["mult broadcasted - right"] = (mat1x1, un(lambda B, x: (P2(B) * x).sum()))
    c["div broadcasted - left"] = (mat1x1, un(lambda B, x: (x / P2(B)).sum()))
    c["div - right"] = (mat2x3, un(lambda B, x: (P2(B) / x).sum()))
    c["min - left"] = (mat2x3, un(lambda B, x: x.minimum(P2(B)).sum()))
    c["min - right"] = (mat2x3, un(lambda B, x: P2(B).minimum(x).sum()))
    c["max - left"] = (mat2x3, un(lambda B, x: x.maximum(P2(B)).sum()))
    c["max - right"] = (mat2x3, un(lambda B, x: P2(B).maximum(x).sum()))
    c["crossentropy - left"] = (mat2x3, un(lambda B, x: x.crossEntropy(P2(B)).sum()))
    c["crossentropy - right"] = (mat2x3, un(lambda B, x: P2(B).crossEntropy(x).sum()))
    c["log"] = (mat2x3, un(lambda B, x: x.log().sum()))
    c["log1p"] = (mat2x3, un(lambda B, x: x.log1p().sum()))
    c["softplus"] = (mat2x3, un(lambda B, x: x.softplus(2.0, 0.0).sum()))
    c["sin"] = (mat2x3_2, un(lambda B, x: x.sin().sum()))
    c["cos"] = (mat2x3_2, un(lambda B, x: x.cos().sum()))
    c["tan"] = (mat2x3_2, un(lambda B, x: x.tan().sum()))
    c["atan"] = (mat2x3_2, un(lambda B, x: x.atan().sum()))
    c["capped exp"] = (mat3x1, un(lambda B, x: B.capped_exp(x, 2.5).sum()))
    c["pow"] = (mat2x3_2, un(lambda B, x: x.pow(2.0).sum()))
    c["euclidean distance wrt b"] = (mat2x3_2, un(lambda B, x: B.const(mat2x3).euclideanDistance(x, 1).sum()))
    c["pow  2"] = (mat1x1, un(lambda B, x: B.param(mat2x3_2).powv(x).sum()))
    c["tanh"] = (mat2x3_2, un(lambda B, x: x.tanh().sum()))
    c["where true branch"] = (mat2x3_2, un(lambda B, x: B.where(B.eq_scalar(B.tensor(mat2x3_2), 2.0), x, B.param(mat2x3)).sum()))
    c["where false branch"] = (mat2x3, un(lambda B, x: B.where(B.eq_scalar(B.tensor(mat2x3_2), 2.0), B.param(mat2x3_2), x).sum()))
    c["squaredFrobenius"] = (mat2x3_2, un(lambda B, x: x.squaredFrobenius().sum()))
    c["transpose"] = (mat2x3_2, un(lambda B, x: x.transpose(0, 1).sum()))          # Variable.t = Transpose(this) with the default dims 0, 1
    c["mse loss"] = (mat3x1, un(lambda B, x: x.mseLoss(B.tensor(mat3x1_2.reshape(3))).sum()))
    c["l1 loss"] = (mat3x1, un(lambda B, x: x.smoothL1Loss(B.tensor(mat3x1_2.reshape(3))).sum()))

    def logistic(kind):
        def f(B, m):
            w = B.param(m)
            data = B.const(mat3x2)
            if kind == "ce":
                L = data.mm(w).logSoftMax(1).crossEntropy(B.const(np.eye(3))).sum() + w.squaredFrobenius()
            elif kind in ("bce", "bce mean"):
                y = B.tensor(np.array([[0.0, 0.0], [1.0, 0.5], [0.5, 1.0]]))          # Mat(Vec(0, 1, 0.5), Vec(0, 0.5, 1))
                L = data.mm(w).binaryCrossEntropyWithLogitsLoss(y, B.tensor(np.ones((1, 2))), 2 if kind == "bce" else 1)
            else:
                y = B.tensor(np.array([0, 1, 2], dtype=np.int64))
                L = data.mm(w).logSoftMax(1).nllLoss(y, B.tensor(np.ones(3)), 0).sum() + w.squaredFrobenius()
            return L, w
        return f
    c["l2 logistic regression loss"] = (mat2x3_2, logistic("ce"))
    c["l2 logistic regression loss - bce loss"] = (mat2x2, logistic("bce"))
    c["l2 logistic regression loss - bce loss mean"] = (mat2x2, logistic("bce mean"))
    c["l2 logistic regression loss - nll_loss unreduced"] = (mat2x3_2, logistic("nll none"))
    c["weight norm - wrt g"] = (mat2x3[0:1], lambda B, m: (lambda g: (B.weight_norm(B.param(np.ones((2, 3))), g, 0).sum(), g))(B.param(m)))
    c["weight norm - wrt v"] = (mat2x3, lambda B, m: (lambda v: (B.weight_norm(v, B.param(np.ones((1, 3))), 0).sum(), v))(B.param(m)))

    def mask(B, vals):
        return B.const_bool(B.eq_scalar(B.tensor(np.array(vals, dtype=np.float64).reshape(1, 2, 2)), 1.0))
    c["mask-fill"] = (nd1x2x2, un(lambda B, x: x.maskFill(mask(B, [1, 0, 0, 0]), 2.0).sum()))
    c["mask-select"] = (nd1x2x2, un(lambda B, x: x.flatten(0, -1).maskSelect(mask(B, [1, 0, 0, 1]).flatten(0, -1)).sum()))
    c["index_fill"] = (nd1x2x2, un(lambda B, x: x.indexFill(B.const(np.array([1], dtype=np.int64)), 1, 2.0).sum()))
    c["expand as"] = (nd1x2x2, un(lambda B, x: x.expandAs(B.tensor(np.zeros((2, 2, 2, 2)))).sum()))
    c["scatter sum"] = (mat2x3, un(lambda B, x: x.scatterAdd(B.const(np.array([[0, 0, 1], [0, 1, 1]], dtype=np.int64)), 0, 2).sum()))
    c["variance"] = (mat2x3, un(lambda B, x: x.variance([1]).sum()))
    c["index sum"] = (mat2x3, un(lambda B, x: x.indexAdd(B.const(np.array([1, 1], dtype=np.int64)), 0, 2).sum()))
    idx0 = lambda B: B.const(np.array([0], dtype=np.int64))
    c["index add by target"] = (mat2x3, un(lambda B, x: x.indexAddFromSource(idx0(B), 0, B.param(mat1x3)).sum()))
    c["index add by src"] = (mat1x3, un(lambda B, x: B.param(mat2x3).indexAddFromSource(idx0(B), 0, x).sum()))
    c["repeat interleave"] = (mat2x3, un(lambda B, x: x.repeatInterleave(B.const(np.array([2, 3], dtype=np.int64)), 0).sum()))
    c["index_select"] = (nd1x2x2, un(lambda B, x: x.indexSelect(1, B.const(np.array([1, 1, 1], dtype=np.int64))).sum()))

    def conv1d(wrt, stride, pad):
        def f(B, m):
            x = B.param(m if wrt == "x" else nd1x2x3)
            w = B.param(m if wrt == "w" else nd1x2x2)
            b = B.param(m if wrt == "b" else np.ones(1))
            return B.conv(x, w, b, [stride], [pad], [1], False, [0], 1).sum(), {"x": x, "w": w, "b": b}[wrt]
        return f
    c["conv1d - wrt weights"] = (nd1x2x2, conv1d("w", 1, 0))
    c["conv1d - wrt input"] = (nd1x2x3, conv1d("x", 1, 0))
    c["conv1d - padded - wrt weights"] = (nd1x2x2, conv1d("w", 1, 1))
    c["conv1d -padded - wrt input"] = (nd1x2x3, conv1d("x", 1, 1))
    c["conv1d - stride-2 - wrt weights"] = (nd1x2x2, conv1d("w", 2, 1))
    c["conv1d -stride-2 - wrt input"] = (nd1x2x3, conv1d("x", 2, 1))
    c["conv1d -stride-2 - wrt bias"] = (ndx1, conv1d("b", 2, 1))

    def conv_groups(B, m):
        x, w, b = B.param(nd1x4x3x3), B.param(m), B.param(np.ones(2))
        return B.conv(x, w, b, [1, 1], [0, 0], [1, 1], False, [0, 0], 2).sum(), w
    c["conv2d - wrt weights - groups"] = (nd2x2x2x2, conv_groups)
    c["maxpool1d padded"] = (nd1x2x3, un(lambda B, x: B.max_pool1d(x, 2, 1, 1, 1).sum()))
    c["maxpool1d unpadded"] = (nd1x2x3, un(lambda B, x: B.max_pool1d(x, 2, 1, 0, 1).sum()))
    c["maxpool1d strided"] = (nd1x2x3, un(lambda B, x: B.max_pool1d(x, 2, 2, 0, 1).sum()))

    def bn_nd(which, feat, x0, other_w=1.0, other_b=1.0, rm=1.0):
        def f(B, m):
            x = B.param(m if which == 0 else x0)
            w = B.param(m if which == 1 else np.full(feat, other_w))
            b = B.param(m if which == 2 else np.full(feat, other_b))
            L = B.batch_norm(x, w, b, B.tensor(np.full(feat, rm)), B.tensor(np.full(feat, rm)), True, 0.1, 1e-5).sum()
            return L, (x, w, b)[which]
        return f
    c["batch norm 1d - wrt to bias"] = (ndx3, bn_nd(2, 3, mat2x3, other_w=0.0))
    c["batch norm 2d - wrt to input"] = (nd1x2x3, bn_nd(0, 6, nd1x2x3))
    c["batch norm 2d - wrt to weights"] = (ndx6, bn_nd(1, 6, nd1x2x3))
    c["batch norm 2d - wrt to bias"] = (ndx6, bn_nd(2, 6, nd1x2x3))
    c["batch norm 3d - wrt to input"] = (nd1x2x3x3, bn_nd(0, 18, nd1x2x3x3))
    c["batch norm 3d - wrt to weights"] = (ndx18, bn_nd(1, 18, nd1x2x3x3))
    c["batch norm 3d - wrt to bias"] = (ndx18, bn_nd(2, 18, nd1x2x3x3))

    def ln(which, has_w, has_b, mean=True):
        def f(B, m):
            x = B.param(m if which == 0 else mat2x3)
            w = (B.param(m if which == 1 else (np.zeros(3) if which == 2 else ndx3))) if has_w else None
            b = (B.param(m if which == 2 else np.zeros(3))) if has_b else None
            out = B.layer_norm(x, w, b, [3], 1e-5)
            return (out.mean([0, 1]) if mean else out), (x, w, b)[which]
        return f
    c["layer norm 1d - wrt to input - no scale and no bias"] = (mat2x3, ln(0, False, False))
    c["layer norm 1d - wrt to weight"] = (ndx3, ln(1, True, True))
    c["layer norm 1d - wrt to weight - no bias"] = (ndx3, ln(1, True, False))
    c["layer norm 1d - wrt to bias"] = (ndx3, ln(2, True, True))
    c["layer norm 1d - wrt to bias - no scale"] = (ndx3, ln(2, False, True))
    c["flatten "] = (nd1x2x3x3, un(lambda B, x: x.flattenLastDimensions(3).sum()))
    c["select 0 0 "] = (nd1x2x3x3, un(lambda B, x: x.select(0, 0).sum()))
    c["select 2 1 "] = (nd1x2x3x3, un(lambda B, x: x.select(2, 1).sum()))
    c["slice "] = (nd1x2x3x3, un(lambda B, x: x.slice(2, 1, 3, 1).sum()))
    c["stack 0"] = (nd1x2x3, un(lambda B, x: B.stack([x, x], 0).sum()))
    c["stack 1"] = (nd1x2x3, un(lambda B, x: B.stack([x, x], 1).sum()))
    c["cat 1 "] = (nd1x2x3, un(lambda B, x: x.cat(B.param(nd1x2x3_2), 1).sum()))
    c["cat 2 "] = (nd1x2x3, un(lambda B, x: x.cat(B.param(nd1x2x3_2), 2).sum()))
    c["view 1 "] = (nd1x2x3, un(lambda B, x: x.view([1, 1, 2, 3]).sum()))
    c["reshape 1 "] = (nd1x2x3, un(lambda B, x: x.reshape([1, 1, 2, 3]).sum()))
    c["embedding "] = (mat2x3, lambda B, m: (lambda w: (B.embedding(B.const(np.ones((4, 5), dtype=np.int64)), w).sum(), w))(B.param(m)))

    def tconv(wrt, pad):
        def f(B, m):
            x = B.param(m if wrt == "x" else nd1x2x3x3)
            w = B.param_transposed01(m if wrt == "w" else nd1x2x2x2)
            b = B.param(m if wrt == "b" else np.ones(1))
            return B.conv(x, w, b, [1, 1], [pad, pad], [1, 1], True, [0, 0], 1).sum(), {"x": x, "w": w, "b": b}[wrt]
        return f
    c["tranposed conv2d - wrt input"] = (nd1x2x3x3, tconv("x", 0))
    c["tranposed conv2d - wrt input - padded"] = (nd1x2x3x3, tconv("x", 1))
    c["tranposed conv2d - wrt weight"] = (nd1x2x2x2, tconv("w", 0))
    c["tranposed conv2d - wrt bias"] = (ndx1, tconv("b", 0))


# ---- the reference's fused-attention KATs (autograd.test.scala:219-285; CUDA only there: `testGradientAndValueCudaOnly`) -------------
# q / k / v of shape (1, 8, 1, 8) in FLOAT32: one query and one key per head, so the softmax is 1 and the output is v - the value is
# sum(v) (64, 64, 704 = 64 * 11) and the gradient w.r.t. q and k is exactly 0, w.r.t. v exactly 1.  All-ones K / V cannot discriminate
# between layouts, but they are the only vectors the reference holds for the fused operator, so they are run as written.
SDPA = GOLDEN["sdpa"]


def run_sdpa_case(B: "Backend", name: str, m=None, backprop=True):
    wrt = name[-1]
    m = np.ones(64) if m is None else m
    x = np.asarray(m, dtype=np.float32).reshape(1, 8, 1, 8)
    ones = np.ones((1, 8, 1, 8), dtype=np.float32)
    q = B.param_f32(x + np.float32(0.0) if wrt == "q" else ones)
    k = B.param_f32(x + np.float32(0.0) if wrt == "k" else ones)
    v = B.param_f32(x + np.float32(10.0) if wrt == "v" else ones)
    L = B.sdpa(q, k, v, False).sum()
    if backprop:
        L.backprop()
    return B.scalar(L), (B.grad({"q": q, "k": k, "v": v}[wrt]).reshape(-1) if backprop else None)


def sdpa_finite_difference(B: "Backend", name: str):
    eps = SDPA[name]["eps"]
    g = np.zeros(64)
    for i in range(64):
        d = np.zeros(64); d[i] = eps
        g[i] = (run_sdpa_case(B, name, np.ones(64) + d, False)[0] - run_sdpa_case(B, name, np.ones(64) - d, False)[0]) / (2 * eps)
    return g


CASES = _cases()
EXPECTED = {k: v["expected"] for k, v in GOLDEN["autograd"].items()}
if not os.environ.get("LAMP_KATS_NO_ASSERT"):
    assert set(CASES) == set(EXPECTED), set(CASES) ^ set(EXPECTED)


def run_case(B: Backend, name: str, m=None, backprop=True):
    inp, f = CASES[name]
    L, v = f(B, inp if m is None else m)
    if backprop:
        L.backprop()
    return B.scalar(L), (B.grad(v) if backprop else None)


# the reference's step for the central difference: 1e-6, except where the test passes its own (autograd.test.scala:483 "cast to float": 1e-2,
# a sum in float32 cannot resolve 1e-6)
EPS = {"cast to float": 1e-2}


def finite_difference(B: Backend, name: str, eps=None):
    eps = EPS.get(name, 1e-6) if eps is None else eps
    inp, _ = CASES[name]
    g = np.zeros_like(inp)
    it = np.nditer(inp, flags=["multi_index"])
    for _ in it:
        idx = it.multi_index
        d = np.zeros_like(inp)
        d[idx] = eps
        g[idx] = (run_case(B, name, inp + d, False)[0] - run_case(B, name, inp - d, False)[0]) / (2 * eps)
    return g
