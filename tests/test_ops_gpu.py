"""Op-level parity: every C-ABI compute entry point vs the same ATen operator on CPU.

The oracle side calls torch.ops.aten.* on CPU tensors (the operators the reference dispatches,
see oracle/lamp_oracle.py header); the HIP side goes through liblamp_hip.so.  Inputs are
deterministic closed forms.  Tolerances: bit-exact for index/bool results, <= 1e-5 relative for
f32 forward (BASELINE.json), 1e-12 for f64, bf16 at bf16 resolution (stated per test).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from lamp_amd import sten as S
from lamp_amd._capi import lib, i64_array, handle_array
from tests.util import DTYPES, FWD_TOL, assert_close, closed_form, to_sten, to_torch

pytestmark = pytest.mark.gpu
aten = torch.ops.aten


def _mask3(a, b, c):
    return (C.c_uint8 * 3)(a, b, c)


def _out3():
    return (C.c_void_p * 3)()


def _wrap3(arr):
    return [S.STen(arr[i]) if arr[i] else None for i in range(3)]


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
def test_binary_broadcast_and_inplace(gpu, dt):
    a = closed_form((5, 1, 7), 1, 4.0, dt)
    b = closed_form((3, 7), 50, 3.0, dt) + 1.6
    A, B = to_sten(a), to_sten(b)
    tol = FWD_TOL[dt]
    assert_close(to_torch(A + B), (a + b).float() if dt == torch.bfloat16 else a + b, tol, "add")
    assert_close(to_torch(A - B), (a - b).float() if dt == torch.bfloat16 else a - b, tol, "sub")
    assert_close(to_torch(A * B), (a * b).float() if dt == torch.bfloat16 else a * b, tol, "mul")
    assert_close(to_torch(A / B), (a / b).float() if dt == torch.bfloat16 else a / b, tol, "div")
    assert_close(to_torch(A.add(B, 0.25)), aten.add.Tensor(a, b, alpha=0.25).double(), tol, "add alpha")
    assert_close(to_torch(A * 3.5), aten.mul.Scalar(a, 3.5).double(), tol, "mul scalar")
    assert_close(to_torch(A + 2.0), aten.add.Scalar(a, 2.0).double(), tol, "add scalar")
    # in place on a broadcast rhs
    full = closed_form((3, 7), 9, 1.0, dt)
    F = to_sten(full)
    F += to_sten(closed_form((1, 7), 3, 1.0, dt))
    assert_close(to_torch(F), (full + closed_form((1, 7), 3, 1.0, dt)).double(), tol, "+= broadcast")
    # addcmul / addcdiv out (AdamW building blocks)
    t1, t2 = closed_form((3, 7), 11, 2.0, dt), closed_form((3, 7), 13, 2.0, dt) + 1.5
    O = to_sten(full)
    S.STen.addcmulOut(O, O, to_sten(t1), to_sten(t2), 0.3)
    assert_close(to_torch(O), aten.addcmul(full, t1, t2, value=0.3).double(), tol, "addcmul")
    O = to_sten(full)
    S.STen.addcdivOut(O, O, to_sten(t1), to_sten(t2), -0.7)
    assert_close(to_torch(O), aten.addcdiv(full, t1, t2, value=-0.7).double(), tol, "addcdiv")


@pytest.mark.parametrize("dt", DTYPES)
def test_unary_functions(gpu, dt):
    x = closed_form((4, 33), 7, 6.0, dt)
    X = to_sten(x)
    tol = FWD_TOL[dt]
    pos = x.abs() + 0.25
    P = to_sten(pos)
    cases = [("relu", X.relu(), aten.relu(x)), ("gelu", X.gelu(), aten.gelu(x)), ("sigmoid", X.sigmoid(), aten.sigmoid(x)),
             ("tanh", X.tanh(), aten.tanh(x)), ("hardswish", X.hardSwish(), aten.hardswish(x)),
             ("leaky", X.leakyRelu(0.1), aten.leaky_relu(x, 0.1)), ("softplus", X.softplus(1.0, 20.0), aten.softplus(x, 1.0, 20.0)),
             ("exp", X.exp(), aten.exp(x)), ("log", P.log(), aten.log(pos)), ("log1p", P.log1p(), aten.log1p(pos)),
             ("sqrt", P.sqrt(), aten.sqrt(pos)), ("reciprocal", P.reciprocal(), aten.reciprocal(pos)), ("neg", X.neg(), aten.neg(x)),
             ("abs", X.abs(), aten.abs(x)), ("sign", X.sign(), aten.sign(x)), ("sin", X.sin(), aten.sin(x)), ("cos", X.cos(), aten.cos(x)),
             ("atan", X.atan(), aten.atan(x)), ("pow2", X.pow(2.0), aten.pow.Tensor_Scalar(x, 2.0)),
             ("pow0.5", P.pow(0.5), aten.pow.Tensor_Scalar(pos, 0.5)), ("pow1.7", P.pow(1.7), aten.pow.Tensor_Scalar(pos, 1.7))]
    for name, got, ref in cases:
        assert_close(to_torch(got), ref.double(), tol * (4 if name in ("exp", "pow1.7", "tan") else 1), name)
    g = closed_form((4, 33), 99, 2.0, dt)
    G = to_sten(g)
    o = C.c_void_p(); lib.lamp_gelu_backward(C.byref(o), G, X)
    assert_close(to_torch(S.STen(o)), aten.gelu_backward(g, x).double(), tol, "gelu_backward")
    y = aten.sigmoid(x)
    o = C.c_void_p(); lib.lamp_sigmoid_backward(C.byref(o), G, to_sten(y))
    assert_close(to_torch(S.STen(o)), aten.sigmoid_backward(g, y).double(), tol, "sigmoid_backward")
    y = aten.tanh(x)
    o = C.c_void_p(); lib.lamp_tanh_backward(C.byref(o), G, to_sten(y))
    assert_close(to_torch(S.STen(o)), aten.tanh_backward(g, y).double(), tol, "tanh_backward")
    o = C.c_void_p(); lib.lamp_hardswish_backward(C.byref(o), G, X)
    assert_close(to_torch(S.STen(o)), aten.hardswish_backward(g, x).double(), tol, "hardswish_backward")
    # fused relu backward accumulate == lamp's lt/where/addcmul chain, gradient at x == 0 is 1
    xz = x.clone(); xz[0, :5] = 0
    out0 = closed_form((4, 33), 5, 1.0, dt)
    O = to_sten(out0)
    lib.lamp_relu_backward_accumulate_(O, G, to_sten(xz), 0.0)
    ref = out0.double() + g.double() * torch.where(xz.double() < 0, 0.0, 1.0)
    assert_close(to_torch(O), ref, tol, "relu backward accumulate")


def test_comparisons_where_are_exact(gpu):
    a = closed_form((6, 9), 1, 4.0, torch.float32)
    b = closed_form((6, 9), 4, 4.0, torch.float32)
    A, B = to_sten(a), to_sten(b)
    for name, got, ref in [("lt", A.lt(B), a < b), ("le", A.le(B), a <= b), ("gt", A.gt(0.3), a > 0.3), ("ge", A.ge(B), a >= b),
                           ("eq", A.equ(A), a == a), ("ne", A.ne(B), a != b), ("lt0", A.lt(0.0), a < 0)]:
        assert np.array_equal(got.to_numpy(), ref.numpy()), name
    w = S.STen.where(A.lt(0.0), A, B)
    assert np.array_equal(w.to_numpy(), torch.where(a < 0, a, b).numpy())
    assert np.array_equal(A.maskedFill(A.gt(0.0), -2.0).to_numpy(), a.masked_fill(a > 0, -2.0).numpy())
    i = torch.arange(12).reshape(3, 4)
    I = to_sten(i)
    assert np.array_equal(I.ne(to_sten(torch.full((3, 4), 5))).to_numpy(), (i != 5).numpy())


@pytest.mark.parametrize("dt", DTYPES)
def test_reductions(gpu, dt):
    x = closed_form((6, 5, 8, 3), 3, 2.0, dt)
    X = to_sten(x)
    tol = FWD_TOL[dt] * 4
    xd = x.double()
    assert_close(to_torch(X.sum()), xd.sum(), tol * 8, "sum all")
    for dims, keep in [([0], True), ([3], False), ([0, 2, 3], False), ([1, 3], True), ([0, 2], False), ([1], True), ([0, 1, 2, 3], False)]:
        assert_close(to_torch(X.sum(dims, keep)), xd.sum(dims, keepdim=keep), tol * 4, f"sum {dims}")
        assert_close(to_torch(X.mean(dims, keep)), xd.mean(dims, keepdim=keep), tol * 4, f"mean {dims}")
    assert_close(to_torch(X.norm2([3], True)), torch.linalg.vector_norm(xd, 2, [3], True), tol, "norm2")
    v, m = X.varAndMean([0, 2], True, True)
    rv, rm = torch.var_mean(xd, [0, 2], unbiased=True, keepdim=True)
    assert_close(to_torch(v), rv, tol * 4, "var")
    assert_close(to_torch(m), rm, tol * 4, "mean")
    big = closed_form((3, 70001), 5, 2.0, dt)
    assert_close(to_torch(to_sten(big).sum(1, False)), big.double().sum(1), tol * 8, "long rows")
    assert_close(to_torch(to_sten(big).sum(0, True)), big.double().sum(0, keepdim=True), tol * 4, "many columns")
    assert np.array_equal(X.argmax(1, False).to_numpy(), torch.argmax(x.float(), 1).numpy())
    # unbroadcast (TensorHelpers.scala:7-41)
    p = closed_form((4, 5, 6), 2, 1.0, dt)
    assert_close(to_torch(to_sten(p).unbroadcast([1, 6])), p.double().sum((0, 1)).reshape(1, 6), tol * 4, "unbroadcast")
    assert_close(to_torch(to_sten(p).unbroadcast([5, 1])), p.double().sum((0, 2)).reshape(5, 1), tol * 4, "unbroadcast2")


def test_views_share_storage_and_strided_copy(gpu):
    x = closed_form((4, 6, 5), 0, 1.0, torch.float32)
    X = to_sten(x)
    v = X.transpose(0, 2)
    assert v.storage_id == X.storage_id and v.shape == [5, 6, 4]
    assert np.array_equal(v.to_numpy(), x.transpose(0, 2).numpy())
    assert np.array_equal(X.select(1, 2).to_numpy(), x.select(1, 2).numpy())
    assert np.array_equal(X.slice(2, 1, 5, 2).to_numpy(), x[:, :, 1:5:2].numpy())
    assert np.array_equal(X.narrow(0, 1, 2).to_numpy(), x.narrow(0, 1, 2).numpy())
    assert np.array_equal(X.view(24, 5).to_numpy(), x.view(24, 5).numpy())
    assert np.array_equal(X.view(-1).to_numpy(), x.view(-1).numpy())
    assert np.array_equal(v.reshape(30, 4).to_numpy(), x.transpose(0, 2).reshape(30, 4).numpy())
    assert np.array_equal(X.flatten(1).to_numpy(), x.flatten(1).numpy())
    assert np.array_equal(X.unsqueeze(1).expand([4, 3, 6, 5]).to_numpy(), x.unsqueeze(1).expand(4, 3, 6, 5).numpy())
    assert np.array_equal(S.STen.cat([X, X.slice(0, 0, 2)], 0).to_numpy(), torch.cat([x, x[:2]], 0).numpy())
    assert np.array_equal(S.STen.stack([X, X], 1).to_numpy(), torch.stack([x, x], 1).numpy())
    with pytest.raises(Exception):
        v.view(20, 6)
    # writing through a view is visible in the base
    X.select(0, 0).fill_(7.0)
    assert (X.to_numpy()[0] == 7.0).all()
    assert np.array_equal(X.castToDouble().to_numpy(), X.to_numpy().astype(np.float64))
    assert np.array_equal(to_sten(torch.arange(10)).castToFloat().to_numpy(), np.arange(10, dtype=np.float32))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 3, 3), (64, 64, 64), (130, 70, 45), (256, 384, 192), (1024, 256, 784), (1024, 10, 256), (1, 1, 1),
                                   (2048, 2048, 2048)])   # aligned and large: the 256x128 LDS-DMA kernel, all four operand layouts
def test_gemm_family(gpu, dt, shape):
    M, N, K = shape
    a, b = closed_form((M, K), 1, 2.0, dt), closed_form((K, N), 77, 2.0, dt)
    A, B = to_sten(a), to_sten(b)
    # bf16: 2^-7 per element against the EXACT product of the bf16 operands - f32 accumulation and one rounding stay inside one ulp
    # (was 1.6e-2); the forms that round twice (a rounded product plus a rounded operand) get twice that below
    tol = {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 2.0 ** -7}[dt]
    ref = a.double() @ b.double()
    assert_close(to_torch(A.mm(B)), ref, tol, "mm")
    # column-major operands through transposed views (no copies needed)
    assert_close(to_torch(to_sten(a.t().contiguous()).t.mm(to_sten(b.t().contiguous()).t)), ref, tol, "mm of transposed views")
    out0, p = closed_form((K, N), 5, 1.0, dt), closed_form((M, N), 9, 1.0, dt)
    O, P = to_sten(out0), to_sten(p)
    S.STen.addmm_out_transposed1(O, O, A, P, 1.0, 1.0)          # dB += A^T . p   (ops.scala:680-690)
    assert_close(to_torch(O), out0.double() + a.double().t() @ p.double(), tol * 2, "addmm_out_transposed1")
    out1 = closed_form((M, K), 6, 1.0, dt)
    O1 = to_sten(out1)
    S.STen.addmm_out_transposed2(O1, O1, P, B, 1.0, 1.0)        # dA += p . B^T   (ops.scala:669-679)
    assert_close(to_torch(O1), out1.double() + p.double() @ b.double().t(), tol * 2, "addmm_out_transposed2")
    bias = closed_form((1, N), 4, 1.0, dt)
    assert_close(to_torch(to_sten(bias).addmm(A, B, 0.5, 2.0)), 0.5 * bias.double() + 2.0 * ref, tol * 2, "addmm broadcast self")
    o = C.c_void_p(); lib.lamp_linear_bias(C.byref(o), A, B, to_sten(bias))
    assert_close(to_torch(S.STen(o)), ref + bias.double(), tol * 2, "linear_bias")


@pytest.mark.parametrize("K", [128, 192])
def test_gemm_256_tile_all_layouts(gpu, K):
    """the 256 x 256 LDS-DMA kernel (>= 200 tiles): all four operand layouts, even and odd stage counts"""
    dt, n = torch.bfloat16, 4096
    a, b = closed_form((n, K), 1, 2.0, dt), closed_form((K, n), 77, 2.0, dt)
    A, B = to_sten(a), to_sten(b)
    ref = a.double() @ b.double()
    assert_close(to_torch(A.mm(B)), ref, 2.0 ** -7, "mm")
    assert_close(to_torch(to_sten(a.t().contiguous()).t.mm(to_sten(b.t().contiguous()).t)), ref, 2.0 ** -7, "mm of transposed views")
    at, p = closed_form((K, n), 3, 2.0, dt), closed_form((K, n), 9, 1.0, dt)          # out[n, n] += at^T . p
    o0 = closed_form((n, n), 5, 1.0, dt)
    O = to_sten(o0)
    S.STen.addmm_out_transposed1(O, O, to_sten(at), to_sten(p), 1.0, 1.0)
    assert_close(to_torch(O), o0.double() + at.double().t() @ p.double(), 2.0 ** -7, "addmm_out_transposed1")
    p2, b2 = closed_form((n, K), 11, 1.0, dt), closed_form((n, K), 13, 2.0, dt)        # out[n, n] += p2 . b2^T
    O1 = to_sten(o0)
    S.STen.addmm_out_transposed2(O1, O1, to_sten(p2), to_sten(b2), 1.0, 1.0)
    assert_close(to_torch(O1), o0.double() + p2.double() @ b2.double().t(), 2.0 ** -7, "addmm_out_transposed2")
    bias = closed_form((1, n), 4, 1.0, dt)
    o = C.c_void_p(); lib.lamp_linear_bias(C.byref(o), A, B, to_sten(bias))
    assert_close(to_torch(S.STen(o)), ref + bias.double(), 2.0 ** -7, "linear_bias")


def test_gemm_4096_cube_is_the_benchmarked_linear(gpu):
    """BASELINE.json config 2 at full size: lamp `Linear(4096, 4096, bias)` on x[4096, 4096] in bf16 - y = x.W + b, dW += x^T.p,
    dX += p.W^T with beta = 1 into non-zero accumulators (MatMul's backward closures, ops.scala:665-695) - against f64 on 192
    sampled rows of each result (all 4096 columns), per element at rtol 2^-7: one bf16 rounding of an f32-accumulated sum."""
    dt, n = torch.bfloat16, 4096
    x, w = closed_form((n, n), 1, 2.0, dt), closed_form((n, n), 77, 2.0, dt)
    b, p = closed_form((1, n), 3, 1.0, dt), closed_form((n, n), 9, 1.0, dt)
    dw0, dx0 = closed_form((n, n), 5, 8.0, dt), closed_form((n, n), 6, 8.0, dt)
    X, W, Bv, P = to_sten(x), to_sten(w), to_sten(b), to_sten(p)
    o = C.c_void_p(); lib.lamp_linear_bias(C.byref(o), X, W, Bv)
    Y = S.STen(o)
    dW, dX = to_sten(dw0), to_sten(dx0)
    S.STen.addmm_out_transposed1(dW, dW, X, P, 1.0, 1.0)             # dW += x^T . p
    S.STen.addmm_out_transposed2(dX, dX, P, W, 1.0, 1.0)             # dX += p . W^T
    rows = torch.tensor(sorted(set(((torch.arange(192) * 2654435761) % n).tolist())))
    xd, wd, pd = x.double(), w.double(), p.double()
    y, dw, dx = to_torch(Y), to_torch(dW), to_torch(dX)
    assert_close(y[rows], xd[rows] @ wd + b.double(), 2.0 ** -7, "y = x.W + b")
    assert_close(dw[rows], dw0.double()[rows] + xd[:, rows].t() @ pd, 2.0 ** -7, "dW += x^T.p")
    assert_close(dx[rows], dx0.double()[rows] + pd[rows] @ wd.t(), 2.0 ** -7, "dX += p.W^T")
    # f32 at the same size: <= 1e-5 per element (BASELINE.json's forward bar)
    xf, wf = x.float(), w.float()
    yf = to_torch(to_sten(xf).mm(to_sten(wf)))
    assert_close(yf[rows], xd[rows] @ wd, 1e-5, "f32 x.W")


@pytest.mark.parametrize("M,N,K", [(768, 768, 12288), (256, 768, 6144), (768, 3072, 4096), (512, 256, 2048), (24576, 768, 2048), (22016, 768, 768)])
def test_gemm_split_k(gpu, M, N, K):
    """few output tiles over a long K (weight gradients of a token batch): K split over blockIdx.z, f32 slices summed by the reduce
    kernel - all operand layouts, beta operand, and the same answer as the unsplit kernel up to bf16 rounding of one result.  The two
    tall shapes are 288 / 258 tiles of 256 x 256: the rows of the nearly empty last round run as a second, K-split product."""
    dt = torch.bfloat16
    xt, p = closed_form((K, M), 3, 2.0, dt), closed_form((K, N), 9, 2.0, dt)
    ref = xt.double().t() @ p.double()
    got = to_torch(to_sten(xt).t.mm(to_sten(p)))                       # x^T . p: A m-contiguous, B n-contiguous
    assert_close(got, ref, 8e-3, "x^T . p")
    a, b = closed_form((M, K), 5, 2.0, dt), closed_form((N, K), 6, 2.0, dt)
    assert_close(to_torch(to_sten(a).mm(to_sten(b).t)), a.double() @ b.double().t(), 8e-3, "a . b^T")
    assert_close(to_torch(to_sten(a).mm(to_sten(b.t().contiguous()))), a.double() @ b.double().t(), 8e-3, "a . b")
    o0 = closed_form((M, N), 7, float(K) ** 0.5, dt)
    O = to_sten(o0)
    S.STen.addmm_out_transposed1(O, O, to_sten(xt), to_sten(p), 1.0, 1.0)
    assert_close(to_torch(O), o0.double() + ref, 1.6e-2, "out += x^T . p")


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("M,N,K", [(1024, 256, 784), (784, 256, 1024), (256, 10, 1024), (100, 60, 1000), (64, 64, 130), (1, 300, 4097)])
def test_gemm_f32_split_k(gpu, dt, M, N, K):
    """the f32 / f64 kernel with few output tiles over a long K (config 1's MLP shapes: 64 / 52 / 4 tiles on 256 CUs)
    splits K over blockIdx.z - chunks that are multiples of the 16-deep k-step, a shorter last chunk, K not a multiple of anything -
    and sums the slices in order: north_star's 1e-5 (f32) against f64, the beta operand and transposed layouts included, and the
    same bits on every run."""
    a, b = closed_form((M, K), 1, 2.0, dt), closed_form((K, N), 77, 2.0, dt)
    ref = a.double() @ b.double()
    tol = 1e-5 if dt == torch.float32 else 1e-12
    A, B = to_sten(a), to_sten(b)
    got = A.mm(B).to_numpy()
    assert_close(torch.from_numpy(got), ref, tol, "mm")
    assert np.array_equal(got, A.mm(B).to_numpy()), "fixed summation order: bitwise reproducible"
    assert_close(to_torch(to_sten(a.t().contiguous()).t.mm(to_sten(b.t().contiguous()).t)), ref, tol, "mm of transposed views")
    o0, p = closed_form((K, N), 5, 1.0, dt), closed_form((M, N), 9, 1.0, dt)
    O = to_sten(o0)
    S.STen.addmm_out_transposed1(O, O, A, to_sten(p), 1.0, 1.0)          # K of THIS product is M
    assert_close(to_torch(O), o0.double() + a.double().t() @ p.double(), tol * 2, "out += a^T . p")
    bias = closed_form((1, N), 4, 1.0, dt)
    assert_close(to_torch(to_sten(bias).addmm(A, B, 0.5, 2.0)), 0.5 * bias.double() + 2.0 * ref, tol * 2, "addmm broadcast self")


@pytest.mark.parametrize("M,N,K", [(96, 40, 72), (512, 1024, 768), (4096, 4096, 512), (256, 256, 4096)])
def test_linear_bias_is_bitwise_the_mm_add_chain(gpu, M, N, K):
    """lamp's Linear and the transformer MLP issue x.mm(w) + bias; lamp_linear_bias adds the row vector in the GEMM epilogue after
    rounding the product to bf16, which makes it bit for bit the two-operator chain (every bf16 kernel incl. the split-K reduce)"""
    dt = torch.bfloat16
    x, w, b = closed_form((M, K), 1, 2.0, dt), closed_form((K, N), 2, 2.0, dt), closed_form((1, N), 3, 8.0, dt)
    X, W, B = to_sten(x), to_sten(w), to_sten(b)
    o = C.c_void_p(); lib.lamp_linear_bias(C.byref(o), X, W, B)
    fused = S.STen(o).to_numpy()
    chain = (X.mm(W) + B).to_numpy()
    assert np.array_equal(fused, chain)
    assert_close(torch.from_numpy(fused), x.double() @ w.double() + b.double(), 1.6e-2, "value")


@pytest.mark.parametrize("dt", DTYPES)
def test_mul_add_is_bitwise_the_mult_add_chain(gpu, dt):
    """lamp_mul_add = (a * b) + c with the product rounded first: bit for bit the two elementwise operators it replaces (dense,
    row-vector broadcast and scalar operands)"""
    a, c = closed_form((96, 136), 1, 3.0, dt), closed_form((96, 136), 2, 3.0, dt)
    for b in (closed_form((136,), 3, 2.0, dt), closed_form((96, 136), 4, 2.0, dt), closed_form((1,), 5, 2.0, dt)):
        A, B, Cc = to_sten(a), to_sten(b), to_sten(c)
        o = C.c_void_p(); lib.lamp_mul_add(C.byref(o), A, B, Cc)
        assert np.array_equal(S.STen(o).to_numpy(), ((A * B) + Cc).to_numpy())


@pytest.mark.parametrize("dt", DTYPES)
def test_bmm_family(gpu, dt):
    a, b = closed_form((3, 33, 65), 1, 2.0, dt), closed_form((3, 65, 17), 5, 2.0, dt)
    A, B = to_sten(a), to_sten(b)
    tol = {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 1.6e-2}[dt]
    ref = a.double() @ b.double()
    assert_close(to_torch(A.bmm(B)), ref, tol, "bmm")
    p = closed_form((3, 33, 17), 8, 1.0, dt)
    P = to_sten(p)
    o1 = closed_form((3, 65, 17), 3, 1.0, dt); O1 = to_sten(o1)
    S.STen.baddbmm_out_transposed1(O1, O1, A, P, 1.0, 1.0)
    assert_close(to_torch(O1), o1.double() + a.double().transpose(1, 2) @ p.double(), tol * 2, "baddbmm_t1")
    o2 = closed_form((3, 33, 65), 2, 1.0, dt); O2 = to_sten(o2)
    S.STen.baddbmm_out_transposed2(O2, O2, P, B, 1.0, 1.0)
    assert_close(to_torch(O2), o2.double() + p.double() @ b.double().transpose(1, 2), tol * 2, "baddbmm_t2")
    assert_close(to_torch(A.matmul(to_sten(b[0]))), a.double() @ b[0].double(), tol, "matmul 3d x 2d")


@pytest.mark.parametrize("dt", DTYPES)
def test_log_softmax_and_nll(gpu, dt):
    x = closed_form((37, 100), 3, 8.0, dt)
    X = to_sten(x)
    tol = FWD_TOL[dt]
    y = aten._log_softmax(x, 1, False)
    assert_close(to_torch(X.logSoftMax(1)), y.double(), tol * (1 if dt != torch.bfloat16 else 2), "log_softmax")
    assert_close(to_torch(X.softmax(1)), aten._softmax(x, 1, False).double(), tol, "softmax")
    x3 = closed_form((4, 7, 5), 3, 4.0, dt)
    assert_close(to_torch(to_sten(x3).logSoftMax(1)), aten._log_softmax(x3, 1, False).double(), tol * 2, "log_softmax middle dim")
    g = closed_form((37, 100), 8, 1.0, dt)
    o = C.c_void_p(); lib.lamp_log_softmax_backward_data(C.byref(o), to_sten(g), to_sten(y), 1)
    assert_close(to_torch(S.STen(o)), aten._log_softmax_backward_data(g, y, 1, x.dtype).double(), tol * 4, "log_softmax backward")
    target = (torch.arange(37) * 7) % 100
    target[5] = -100
    w = closed_form((100,), 1, 1.0, dt) + 1.0
    T, W = to_sten(target), to_sten(w)
    for red in (0, 1, 2):
        ref, ref_tw = aten.nll_loss_forward(y, target, w, red, -100)
        o, tw = C.c_void_p(), C.c_void_p()
        lib.lamp_nll_loss_forward(C.byref(o), C.byref(tw), to_sten(y), T, W, red, -100)
        O, TW = S.STen(o), S.STen(tw)
        assert_close(to_torch(O), ref.double(), tol * 4, f"nll forward red={red}")
        if red:
            assert_close(to_torch(TW), ref_tw.double(), tol * 4, "total_weight")
        gy = closed_form(tuple(ref.shape), 2, 1.0, dt) + 1.0
        gi = C.c_void_p()
        lib.lamp_nll_loss_backward(C.byref(gi), to_sten(gy), to_sten(y), T, W, red, -100, to_sten(ref_tw))
        refb = aten.nll_loss_backward(gy, y, target, w, red, -100, ref_tw)
        assert_close(to_torch(S.STen(gi)), refb.double(), tol * 4, f"nll backward red={red}")
    t = closed_form((37, 100), 1, 1.0, dt)
    for red in (0, 1, 2):
        o = C.c_void_p(); lib.lamp_mse_loss(C.byref(o), X, to_sten(t), red)
        assert_close(to_torch(S.STen(o)), aten.mse_loss(x, t, red).double(), tol * 8, "mse")
        gy = torch.ones((), dtype=dt) if red else closed_form((37, 100), 3, 1.0, dt)
        o = C.c_void_p(); lib.lamp_mse_loss_backward(C.byref(o), to_sten(gy), X, to_sten(t), red)
        assert_close(to_torch(S.STen(o)), aten.mse_loss_backward(gy, x, t, red).double(), tol * 4, "mse backward")


def test_nll_target_out_of_range_raises_at_the_next_host_wait(gpu):
    """ATen raises a device assert for a class index outside [0, C) that is not ignore_index (the reference calls
    ATen.nll_loss_forward, ops.scala:1249-1304).  Here the kernel reports it and the next host wait (item / copy to host /
    synchronize) raises a LampError - a label or vocabulary mismatch must not train silently on a subset of the rows."""
    from lamp_amd._capi import LampError
    x = to_sten(aten._log_softmax(closed_form((8, 5), 3, 2.0, torch.float32), 1, False))
    for bad, red in ((5, 1), (-1, 2), (7, 0)):
        target = torch.arange(8) % 5
        target[3] = bad
        o, tw = C.c_void_p(), C.c_void_p()
        lib.lamp_nll_loss_forward(C.byref(o), C.byref(tw), x, to_sten(target), None, red, -100)     # the launch itself succeeds
        with pytest.raises(LampError, match="nll_loss: a target class index is outside"):
            S.STen(o).to_numpy()
        S.STen(tw).release()
    # the flag is cleared by the raise: a valid call afterwards is clean
    o, tw = C.c_void_p(), C.c_void_p()
    lib.lamp_nll_loss_forward(C.byref(o), C.byref(tw), x, to_sten(torch.arange(8) % 5), None, 1, -100)
    assert np.isfinite(S.STen(o).item())
    lib.lamp_device_synchronize()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(16, 6, 8, 8), (4, 100, 8, 8), (64, 10), (1024, 256), (3, 5, 7), (2, 16, 32, 32)])
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm(gpu, dt, shape, training):
    x = closed_form(shape, 3, 4.0, dt) + 0.3
    Cc = shape[1]
    w, b = closed_form((Cc,), 1, 1.0, dt) + 1.0, closed_form((Cc,), 5, 1.0, dt)
    rm, rv = closed_form((Cc,), 7, 0.5, dt), closed_form((Cc,), 9, 0.5, dt) + 1.0
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = aten.native_batch_norm(x, w, b, rm_ref, rv_ref, training, 0.1, 1e-5)
    RM, RV = to_sten(rm), to_sten(rv)
    out = _out3()
    lib.lamp_native_batch_norm(out, to_sten(x), to_sten(w), to_sten(b), RM, RV, int(training), 0.1, 1e-5)
    y, sm, si = _wrap3(out)
    tol = FWD_TOL[dt] * 4
    assert_close(to_torch(y), ref[0].double(), tol, "bn y")
    if training:
        assert_close(to_torch(sm), ref[1].double(), tol, "save_mean")
        assert_close(to_torch(si), ref[2].double(), tol, "save_invstd")
        assert_close(to_torch(RM), rm_ref.double(), tol, "running_mean")
        assert_close(to_torch(RV), rv_ref.double(), tol, "running_var (unbiased)")
    gy = closed_form(shape, 11, 2.0, dt)
    save_mean, save_invstd = (ref[1], ref[2]) if training else (None, None)
    refb = aten.native_batch_norm_backward(gy, x, w, rm_ref, rv_ref, save_mean, save_invstd, training, 1e-5, [True, True, True])
    outb = _out3()
    lib.lamp_native_batch_norm_backward(outb, to_sten(gy), to_sten(x), to_sten(w), to_sten(rm_ref), to_sten(rv_ref),
                                        to_sten(ref[1]) if training else None, to_sten(ref[2]) if training else None,
                                        int(training), 1e-5, _mask3(1, 1, 1))
    dx, dw, db = _wrap3(outb)
    btol = {torch.float64: 1e-10, torch.float32: 2e-4, torch.bfloat16: 4e-2}[dt]
    assert_close(to_torch(dx), refb[0].double(), btol, "bn dx")
    assert_close(to_torch(dw), refb[1].double(), btol, "bn dweight")
    assert_close(to_torch(db), refb[2].double(), btol, "bn dbias")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(6, 5, 8, 8), (4, 16, 16, 16), (3, 7, 9, 9)])
def test_batch_norm_relu_fused_equals_unfused_pair(gpu, dt, shape, training):
    """lamp_native_batch_norm_relu(+_backward) must be BITWISE the pair native_batch_norm -> relu (and its backward)"""
    x = closed_form(shape, 3, 4.0, dt) + 0.3
    Cc = shape[1]
    w, b = closed_form((Cc,), 1, 1.0, dt) + 1.0, closed_form((Cc,), 5, 1.0, dt)
    rm, rv = closed_form((Cc,), 7, 0.5, dt), closed_form((Cc,), 9, 0.5, dt) + 1.0
    X, Wt, Bt = to_sten(x), to_sten(w), to_sten(b)
    out_u, out_f = _out3(), _out3()
    RMu, RVu, RMf, RVf = to_sten(rm), to_sten(rv), to_sten(rm), to_sten(rv)
    lib.lamp_native_batch_norm(out_u, X, Wt, Bt, RMu, RVu, int(training), 0.1, 1e-5)
    lib.lamp_native_batch_norm_relu(out_f, X, Wt, Bt, RMf, RVf, int(training), 0.1, 1e-5)
    yu, smu, siu = _wrap3(out_u)
    yf, smf, sif = _wrap3(out_f)
    assert np.array_equal(yu.relu().to_numpy(), yf.to_numpy())
    assert np.array_equal(RMu.to_numpy(), RMf.to_numpy()) and np.array_equal(RVu.to_numpy(), RVf.to_numpy())
    # against the oracle as well
    ref = torch.relu(aten.native_batch_norm(x, w, b, rm.clone(), rv.clone(), training, 0.1, 1e-5)[0])
    assert_close(to_torch(yf), ref.double(), FWD_TOL[dt] * 4, "fused bn+relu")
    gy = closed_form(shape, 11, 2.0, dt)
    GY = to_sten(gy)
    gm = C.c_void_p()
    lib.lamp_relu_backward(C.byref(gm), GY, yu, 0.0)
    outb_u, outb_f = _out3(), _out3()
    sm = smu if training else None
    si = siu if training else None
    lib.lamp_native_batch_norm_backward(outb_u, S.STen(gm), X, Wt, RMu, RVu, sm, si, int(training), 1e-5, _mask3(1, 1, 1))
    lib.lamp_native_batch_norm_relu_backward(outb_f, GY, X, Wt, Bt, RMu, RVu, sm, si, int(training), 1e-5, _mask3(1, 1, 1))
    for u, f, what in zip(_wrap3(outb_u), _wrap3(outb_f), ("dx", "dweight", "dbias")):
        assert np.array_equal(u.to_numpy(), f.to_numpy()), what


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(6, 5, 8, 8), (4, 16, 16, 16), (3, 7, 9, 9)])
def test_batch_norm_add_relu_fused_equals_unfused_chain(gpu, dt, shape):
    """lamp_native_batch_norm_add_relu(+_backward) must be BITWISE the chain native_batch_norm -> add -> relu and its backward"""
    x = closed_form(shape, 3, 4.0, dt) + 0.3
    addend = closed_form(shape, 29, 3.0, dt)
    Cc = shape[1]
    w, b = closed_form((Cc,), 1, 1.0, dt) + 1.0, closed_form((Cc,), 5, 1.0, dt)
    rm, rv = closed_form((Cc,), 7, 0.5, dt), closed_form((Cc,), 9, 0.5, dt) + 1.0
    X, AD, Wt, Bt = to_sten(x), to_sten(addend), to_sten(w), to_sten(b)
    out_u, out_f = _out3(), _out3()
    RMu, RVu, RMf, RVf = to_sten(rm), to_sten(rv), to_sten(rm), to_sten(rv)
    lib.lamp_native_batch_norm(out_u, X, Wt, Bt, RMu, RVu, 1, 0.1, 1e-5)
    lib.lamp_native_batch_norm_add_relu(out_f, X, AD, Wt, Bt, RMf, RVf, 1, 0.1, 1e-5)
    yu, smu, siu = _wrap3(out_u)
    yf, smf, sif = _wrap3(out_f)
    su = yu + AD                                            # the residual add, rounded to the element type
    assert np.array_equal(su.relu().to_numpy(), yf.to_numpy())
    assert np.array_equal(RMu.to_numpy(), RMf.to_numpy()) and np.array_equal(RVu.to_numpy(), RVf.to_numpy())
    ref = torch.relu((aten.native_batch_norm(x, w, b, rm.clone(), rv.clone(), True, 0.1, 1e-5)[0] + addend))
    assert_close(to_torch(yf), ref.double(), FWD_TOL[dt] * 4, "fused bn+add+relu")
    gy = closed_form(shape, 11, 2.0, dt)
    GY = to_sten(gy)
    gm = C.c_void_p()
    lib.lamp_relu_backward(C.byref(gm), GY, su, 0.0)
    G = S.STen(gm)
    outb_u = _out3()
    lib.lamp_native_batch_norm_backward(outb_u, G, X, Wt, RMu, RVu, smu, siu, 1, 1e-5, _mask3(1, 1, 1))
    out4 = (C.c_void_p * 4)()
    lib.lamp_native_batch_norm_add_relu_backward(out4, GY, X, AD, Wt, Bt, RMu, RVu, smu, siu, 1, 1e-5, (C.c_uint8 * 4)(1, 1, 1, 1))
    fused = [S.STen(out4[i]) for i in range(4)]
    for u, f, what in zip(list(_wrap3(outb_u)) + [G], fused, ("dx", "dweight", "dbias", "daddend")):
        assert np.array_equal(u.to_numpy(), f.to_numpy()), what


# (N, C, H, variant)  variant 0: batch norm, 1: + relu, 2: + addend + relu
BN_ONE_PASS_CASES = [(2048, 128, 8, 2), (2048, 128, 8, 1), (2048, 100, 8, 0), (2048, 6, 32, 1), (2048, 16, 16, 2), (2049, 64, 8, 1),
                     (2100, 5, 20, 2),                      # 50 packets per image row (a division, not a shift), ragged slices
                     (1024, 64, 8, 0), (37, 5, 12, 2), (3, 130, 8, 1), (512, 300, 8, 0)]   # small: the two kernels


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float64])
@pytest.mark.parametrize("case", BN_ONE_PASS_CASES)
def test_batch_norm_backward_in_one_pass(gpu, case, dt):
    """Training-mode batch-norm backward reads dy and x ONCE (bn_bwd_fused_kernel, and since round 5 bn_bwd_fused_fp_kernel for f32 / f64 - the
    precisions the reference's example runs in - whose channels are launched in co-resident chunks): the workgroups of a channel keep their
    slice in registers while they wait for each other's partial sums.  Checked against ATen (f32 arithmetic on the bf16 inputs; f64 for the
    wide types), at the ResNet step's shapes and at ragged ones; repeated launches (the wait counters reset themselves) and launches that
    alternate between two streams give bitwise the same tensors."""
    N, Cc, H, variant = case
    shape = (N, Cc, H, H)
    x = closed_form(shape, 3, 4.0, dt) + 0.3
    addend = closed_form(shape, 29, 3.0, dt)
    gy = closed_form(shape, 11, 2.0, dt)
    w, b = closed_form((Cc,), 1, 1.0, dt) + 1.0, closed_form((Cc,), 5, 1.0, dt)
    rm, rv = closed_form((Cc,), 7, 0.5, dt), closed_form((Cc,), 9, 0.5, dt) + 1.0
    X, AD, GY, Wt, Bt, RM, RV = (to_sten(t) for t in (x, addend, gy, w, b, rm, rv))
    fwd = _out3()
    lib.lamp_native_batch_norm(fwd, X, Wt, Bt, RM, RV, 1, 0.1, 1e-5)
    _, sm, si = _wrap3(fwd)

    def backward():
        if variant == 2:
            out4 = (C.c_void_p * 4)()
            lib.lamp_native_batch_norm_add_relu_backward(out4, GY, X, AD, Wt, Bt, RM, RV, sm, si, 1, 1e-5, (C.c_uint8 * 4)(1, 1, 1, 1))
            return [S.STen(out4[i]) for i in range(4)]
        out = _out3()
        if variant == 1:
            lib.lamp_native_batch_norm_relu_backward(out, GY, X, Wt, Bt, RM, RV, sm, si, 1, 1e-5, _mask3(1, 1, 1))
        else:
            lib.lamp_native_batch_norm_backward(out, GY, X, Wt, RM, RV, sm, si, 1, 1e-5, _mask3(1, 1, 1))
        return list(_wrap3(out))

    lib.lamp_kernel_timer_enable(1)
    first = backward()
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    if N >= 2048:                                           # the large activations (>= 8 packets per thread); small ones keep the two kernels
        assert b"bn_bwd_fused" in buf.value and b"bn_bwd_reduce" not in buf.value, buf.value.decode()
    first_np = [t.to_numpy() for t in first]
    # reference: the accumulation type's arithmetic on the same values, the mask from the ROUNDED pre-activation as the kernels take it
    at = torch.float64 if dt == torch.float64 else torch.float32
    xf, mean, invstd = x.to(at), to_torch(sm).to(at), to_torch(si).to(at)
    g = gy.to(at)
    if variant >= 1:
        pre = torch.addcmul(b.to(at).view(1, -1, 1, 1), xf - mean.view(1, -1, 1, 1), (invstd * w.to(at)).view(1, -1, 1, 1)).to(dt)
        if variant == 2:
            pre = (pre.to(at) + addend.to(at)).to(dt)
        g = torch.where(pre.to(at) < 0, torch.zeros_like(g), g)
    ref = aten.native_batch_norm_backward(g.double(), xf.double(), w.double(), None, None, mean.double(), invstd.double(), True, 1e-5, [True, True, True])
    tol = {torch.bfloat16: 4e-2, torch.float32: 2e-4, torch.float64: 1e-9}[dt]
    if dt == torch.bfloat16 or variant == 0:                # (wide types: an element whose pre-activation is within rounding of 0 may take the other branch)
        for got, want, what in zip(first, ref, ("dx", "dweight", "dbias")):
            assert_close(to_torch(got), want.double(), tol, what)
    if variant == 2 and dt == torch.bfloat16:
        assert torch.equal(to_torch(first[3]).float(), g), "the addend's gradient is the masked dy, exactly"
    if dt != torch.bfloat16:
        # the two-kernel form on the same inputs: same masks (same expression), sums in another order
        lib.lamp_bn_backward_mode(0)
        try:
            two = backward()
        finally:
            lib.lamp_bn_backward_mode(-1)
        for got, want, what in zip(first, two, ("dx", "dweight", "dbias", "daddend")):
            assert_close(to_torch(got), to_torch(want), tol, "one pass vs two kernels: " + what)
        if variant == 2:
            assert torch.equal(to_torch(first[3]), to_torch(two[3])), "the addend's gradient (the masked dy) is exact in both forms"
    for _ in range(5):
        for a, t in zip(first_np, backward()):
            assert np.array_equal(a, t.to_numpy()), "a repeated launch differs"
    # two streams in turn (the host orders the waiting kernels of different streams by an event)
    side = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(side))
    dflt = C.c_void_p(); lib.lamp_stream_get_default(0, C.byref(dflt))
    lib.lamp_device_synchronize()
    try:
        outs = []
        for i in range(6):
            lib.lamp_stream_set_current(side if i % 2 == 0 else dflt)
            outs.append(backward())
        lib.lamp_stream_synchronize(side)
    finally:
        lib.lamp_stream_set_current(dflt)
    lib.lamp_device_synchronize()
    for o in outs:
        for a, t in zip(first_np, o):
            assert np.array_equal(a, t.to_numpy()), "a launch on the other stream differs"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("affine", [(True, True), (True, False), (False, False)])
def test_layer_norm(gpu, dt, affine):
    x = closed_form((6, 7, 96), 3, 4.0, dt)
    w = closed_form((96,), 1, 1.0, dt) + 1.0 if affine[0] else None
    b = closed_form((96,), 5, 1.0, dt) if affine[1] else None
    ref = aten.native_layer_norm(x, [96], w, b, 1e-5)
    out = _out3()
    lib.lamp_native_layer_norm(out, to_sten(x), i64_array([96]), 1, to_sten(w) if w is not None else None,
                               to_sten(b) if b is not None else None, 1e-5)
    y, mean, rstd = _wrap3(out)
    tol = FWD_TOL[dt] * 4
    assert_close(to_torch(y), ref[0].double(), tol, "ln y")
    assert_close(to_torch(mean), ref[1].double(), tol, "ln mean")
    assert_close(to_torch(rstd), ref[2].double(), tol, "ln rstd")
    gy = closed_form((6, 7, 96), 13, 2.0, dt)
    refb = aten.native_layer_norm_backward(gy, x, [96], ref[1], ref[2], w, b, [True, w is not None, b is not None])
    outb = _out3()
    lib.lamp_native_layer_norm_backward(outb, to_sten(gy), to_sten(x), i64_array([96]), 1, to_sten(ref[1]), to_sten(ref[2]),
                                        to_sten(w) if w is not None else None, to_sten(b) if b is not None else None,
                                        _mask3(1, int(w is not None), int(b is not None)))
    dx, dw, db = _wrap3(outb)
    btol = {torch.float64: 1e-10, torch.float32: 2e-4, torch.bfloat16: 4e-2}[dt]
    assert_close(to_torch(dx), refb[0].double(), btol, "ln dx")
    if w is not None:
        assert_close(to_torch(dw), refb[1].double(), btol, "ln dweight")
    if b is not None:
        assert_close(to_torch(db), refb[2].double(), btol, "ln dbias")


CONV_CASES = [
    # (N, Cin, H, W, Cout, k, stride, pad, dil, groups)
    (2, 3, 32, 32, 6, 5, 1, 2, 1, 1),      # resnet stem
    (2, 6, 32, 32, 6, 3, 2, 1, 1, 1),      # res1.r1
    (2, 6, 32, 32, 6, 1, 2, 0, 1, 1),      # res1.l
    (3, 6, 16, 16, 6, 3, 1, 1, 1, 1),      # res1.r2
    (5, 6, 16, 16, 16, 3, 2, 1, 1, 1),     # res2.r1 (8x8 outputs: several images per wgrad round, ragged last round)
    (5, 6, 16, 16, 16, 1, 2, 0, 1, 1),     # res2.l
    (2100, 6, 16, 16, 6, 3, 2, 1, 1, 1),   # more images than workgroups on the narrow matrix-core path
    (2, 5, 16, 24, 7, 3, 1, 0, 1, 1),      # narrow path, no padding, non-square map
    (2, 16, 8, 8, 128, 3, 1, 1, 1, 1),     # res3.r1
    (2, 128, 8, 8, 128, 3, 1, 1, 1, 1),    # res3.r2
    (2, 128, 8, 8, 100, 3, 1, 1, 1, 1),    # res4.r1
    (2, 100, 8, 8, 100, 3, 1, 1, 1, 1),    # res4.r2
    (2, 128, 8, 8, 100, 1, 1, 0, 1, 1),    # res4.l
    (5, 100, 8, 8, 128, 3, 1, 1, 1, 1),    # odd batch on the implicit-GEMM path
    (7, 16, 8, 8, 128, 1, 1, 0, 1, 1),     # res3.l (1x1)
    (70, 128, 8, 8, 128, 3, 1, 1, 1, 1),   # several image splits in wgrad
    (9, 50, 8, 8, 70, 3, 1, 1, 1, 1),      # eight-wave wgrad: Cin not a multiple of 4 (16-byte partial-sum stores reach padding columns), Cout not of 16 (clamped dY rows), odd batch (the pair's missing image)
    (131, 34, 8, 8, 66, 3, 1, 1, 1, 1),    # ... two channels in the second slice of Cin, two in the fifth 16-row block of Cout, ragged image ranges
    (3, 4, 9, 7, 6, 3, 2, 1, 2, 2),        # odd sizes, dilation, groups
    (1, 2, 3, 3, 1, 3, 1, 0, 1, 1),        # reference KAT geometry (autograd.test.scala:2043)
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_convolution_forward_backward(gpu, dt, case):
    N, Cin, H, W, Cout, k, s, p, d, groups = case
    x = closed_form((N, Cin, H, W), 3, 2.0, dt)
    w = closed_form((Cout, Cin // groups, k, k), 17, 1.0, dt)
    b = closed_form((Cout,), 5, 1.0, dt)
    args = ([s, s], [p, p], [d, d], False, [0, 0], groups)
    # bf16: the reference is ATen's convolution in f64 ON THE bf16 OPERANDS (the exact result of these inputs); the kernels accumulate in
    # f32 and round once, so 2^-7 per element holds (was 1.6e-2 / 3e-2 against ATen's own bf16 kernels, which round
    # differently from one another)
    rdt = torch.float64 if dt == torch.bfloat16 else dt
    ref = aten.convolution(x.to(rdt), w.to(rdt), b.to(rdt), *args)
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(b), i64_array([s, s]), i64_array([p, p]), i64_array([d, d]), 2, 0,
                         i64_array([0, 0]), groups)
    tol = {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 2.0 ** -7}[dt]
    assert_close(to_torch(S.STen(o)), ref.double(), tol, "conv forward")
    gy = closed_form(tuple(ref.shape), 23, 1.0, dt)
    refb = aten.convolution_backward(gy.to(rdt), x.to(rdt), w.to(rdt), [Cout], *args, [True, True, True])
    out = _out3()
    lib.lamp_convolution_backward(out, to_sten(gy), to_sten(x), to_sten(w), i64_array([s, s]), i64_array([p, p]), i64_array([d, d]), 2, 0,
                                  i64_array([0, 0]), groups, _mask3(1, 1, 1))
    dx, dw, db = _wrap3(out)
    btol = {torch.float64: 1e-10, torch.float32: 1e-4, torch.bfloat16: 2.0 ** -7}[dt]
    assert_close(to_torch(dx), refb[0].double(), btol, "conv dgrad")
    assert_close(to_torch(dw), refb[1].double(), btol, "conv wgrad")
    assert_close(to_torch(db), refb[2].double(), btol, "conv bias grad")


@pytest.mark.parametrize("case", [(16, 128, 128, 3), (13, 128, 100, 3), (9, 100, 100, 3), (24, 16, 128, 3), (11, 128, 100, 1), (8, 16, 128, 1),
                                  (3, 100, 128, 3), (1100, 128, 128, 3),
                                  (16, 16, 16, 3), (13, 128, 16, 3), (9, 128, 64, 1), (21, 64, 48, 3),    # 1 / 4 channel tiles per wave
                                  (10, 40, 72, 3), (5, 33, 97, 3), (7, 36, 36, 3), (1032, 100, 100, 3)])  # K-tails of 8, 1 and 4 channels (fprop: Cin, dgrad: Cout)
def test_igemm_eight_image_kernel(gpu, case, monkeypatch):
    """ig_conv8d_kernel (one workgroup per CU, eight images, wave = image x all output channels; the default for > 64 output channels
    once the batch gives every CU a workgroup): forced on small and ragged batches (N not a multiple of 8, N < 8), fprop with bias and
    dgrad against ATen and against the two-image kernel (same products in f32, summed channel-chunk outer instead of tap outer: equal up
    to one bf16 rounding), and the batch-norm statistics it publishes equal to a statistics pass over its output."""
    import os
    N, Cin, Cout, k = case
    dt = torch.bfloat16
    x = closed_form((N, Cin, 8, 8), 3, 2.0, dt)
    w = closed_form((Cout, Cin, k, k), 17, 1.0, dt)
    b = closed_form((Cout,), 5, 1.0, dt)
    p = (k - 1) // 2
    args = ([1, 1], [p, p], [1, 1], False, [0, 0], 1)
    gy = closed_form((N, Cout, 8, 8), 23, 1.0, dt)

    def run(variant):
        monkeypatch.setenv("LAMP_IG_VARIANT", variant)
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(b), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0,
                             i64_array([0, 0]), 1)
        out = _out3()
        lib.lamp_convolution_backward(out, to_sten(gy), to_sten(x), to_sten(w), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0,
                                      i64_array([0, 0]), 1, _mask3(1, 0, 0))
        return S.STen(o), _wrap3(out)[0]

    yd, dxd = run("d")
    yb, dxb = run("b")
    monkeypatch.delenv("LAMP_IG_VARIANT")
    ref = aten.convolution(x, w, b, *args)
    refb = aten.convolution_backward(gy, x, w, [Cout], *args, [True, False, False])
    assert_close(to_torch(yd), ref.double(), 2.0 ** -7, "conv forward (8-image kernel)")
    assert_close(to_torch(dxd), refb[0].double(), 2.0 ** -7, "conv dgrad (8-image kernel)")
    assert_close(to_torch(yd), to_torch(yb).double(), 2.0 ** -7, "fprop: 8-image kernel vs 2-image kernel")     # one bf16 ulp
    assert_close(to_torch(dxd), to_torch(dxb).double(), 2.0 ** -7, "dgrad: 8-image kernel vs 2-image kernel")
    if N >= 2:
        # the statistics hand-off: a training-mode batch norm directly on the convolution's output takes the epilogue's partials
        monkeypatch.setenv("LAMP_IG_VARIANT", "d")
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(b), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0,
                             i64_array([0, 0]), 1)
        Y = S.STen(o)
        g, bb = closed_form((Cout,), 7, 1.0, dt) + 1.0, closed_form((Cout,), 9, 1.0, dt)
        out = _out3()
        lib.lamp_native_batch_norm(out, Y, to_sten(g), to_sten(bb), to_sten(torch.zeros(Cout, dtype=dt)), to_sten(torch.ones(Cout, dtype=dt)), 1, 0.1, 1e-5)
        yn, mean, invstd = _wrap3(out)
        # reference statistics in f32 of the bf16 values the convolution stored; save_mean / save_invstd come back in bf16
        rn = aten.native_batch_norm(to_torch(Y).float(), g.float(), bb.float(), torch.zeros(Cout), torch.ones(Cout), True, 0.1, 1e-5)
        assert_close(to_torch(mean), rn[1].double(), 2.0 ** -8, "save_mean from the epilogue's partials")
        assert_close(to_torch(invstd), rn[2].double(), 2.0 ** -8, "save_invstd from the epilogue's partials")
        assert_close(to_torch(yn), rn[0].double(), 2.0 ** -7, "batch norm on the hand-off statistics")
        monkeypatch.delenv("LAMP_IG_VARIANT")


@pytest.mark.parametrize("case", [(9, 100, 100), (1029, 100, 100), (12, 40, 72), (5, 33, 97), (6, 104, 41), (8, 68, 128)])
def test_igemm_k_tail_stages_multiply_what_the_padded_chunk_did(gpu, case, monkeypatch):
    """ig_conv8d_kernel's K-tail (round 6): a 3x3 whose K side ends 1 .. 8 channels past a whole 32-channel chunk multiplies those channels as three
    stages of four taps (tail image of ig_pack_body) instead of nine stages of 32 k that are mostly padding.  The products are the same f32
    values summed in another order: forward and input gradient with the tail stages (default) and without (LAMP_IG_KTAIL=0) agree within one
    bf16 rounding, both with the f64 convolution of the same operands, and the batch-norm partials the forward publishes describe its own output."""
    N, Cin, Cout = case
    dt = torch.bfloat16
    x = closed_form((N, Cin, 8, 8), 3, 2.0, dt)
    w = closed_form((Cout, Cin, 3, 3), 17, 1.0, dt)
    b = closed_form((Cout,), 5, 1.0, dt)
    gy = closed_form((N, Cout, 8, 8), 23, 1.0, dt)
    args = ([1, 1], [1, 1], [1, 1], False, [0, 0], 1)
    monkeypatch.setenv("LAMP_IG_VARIANT", "d")

    def run(ktail):
        monkeypatch.setenv("LAMP_IG_KTAIL", ktail)
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(b), i64_array([1, 1]), i64_array([1, 1]), i64_array([1, 1]), 2, 0,
                             i64_array([0, 0]), 1)
        out = _out3()
        lib.lamp_convolution_backward(out, to_sten(gy), to_sten(x), to_sten(w), i64_array([1, 1]), i64_array([1, 1]), i64_array([1, 1]), 2, 0,
                                      i64_array([0, 0]), 1, _mask3(1, 0, 0))
        return to_torch(S.STen(o)), to_torch(_wrap3(out)[0])

    y1, dx1 = run("1")
    y0, dx0 = run("0")
    ref = aten.convolution(x.double(), w.double(), b.double(), *args)
    refb = aten.convolution_backward(gy.double(), x.double(), w.double(), [Cout], *args, [True, False, False])
    assert_close(y1, ref, 2.0 ** -7, "forward with tail stages")
    assert_close(dx1, refb[0], 2.0 ** -7, "input gradient with tail stages")
    assert_close(y1, y0.double(), 2.0 ** -7, "forward: tail stages vs padded chunk")
    assert_close(dx1, dx0.double(), 2.0 ** -7, "input gradient: tail stages vs padded chunk")
    # at least as close to the exact result as the padded form, up to the noise of one rounding
    e1, e0 = (y1.double() - ref).norm(), (y0.double() - ref).norm()
    assert e1 <= 1.05 * e0 + 1e-12, (e1, e0)


def _timer_classes():
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    return {l.split()[0]: int(l.split()[1]) for l in buf.value.decode().splitlines() if l.strip()}


@pytest.mark.parametrize("case", [(16, 128, 128, 3), (13, 128, 100, 3), (9, 100, 100, 3), (24, 16, 128, 3), (11, 128, 100, 1), (8, 16, 128, 1),
                                  (3, 100, 128, 3), (1, 128, 128, 3), (5, 128, 16, 3), (6, 128, 16, 1), (7, 20, 36, 3), (10, 64, 48, 3),
                                  (4, 4, 128, 3), (9, 92, 84, 1), (130, 128, 128, 3), (1030, 32, 100, 3)])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_igemm_f32_kernels(gpu, case, dt):
    """conv_igemm_f32.hip - the f32 and f64 matrix-instruction convolutions of the 8x8 layers (v_mfma_f32_16x16x4_f32 /
    v_mfma_f64_16x16x4_f64: exact IEEE FMA chains): fprop with bias, dgrad (plain and with the fused addend), wgrad and the bias gradient
    against ATen in f64 on the same operands, at north_star's tolerances (f32: 1e-5 forward, 1e-4 backward; f64: 1e-12 / 1e-10;
    ops.scala:1547-1651); batches that are not multiples of the four (f64: two) images per workgroup, every count of 16-channel output
    tiles the kernels are instantiated for (1 = pixel split, 2, 4, 6, 7, 8), channel counts that are multiples of 4 but not of 16, one
    image range and many in the weight gradient.  The kernel classes must have run."""
    N, Cin, Cout, k = case
    sfx = "f32" if dt == torch.float32 else "f64"
    ftol, btol = (1e-5, 1e-4) if dt == torch.float32 else (1e-12, 1e-10)
    x = closed_form((N, Cin, 8, 8), 3, 2.0, dt)
    w = closed_form((Cout, Cin, k, k), 17, 1.0, dt)
    b = closed_form((Cout,), 5, 1.0, dt)
    p = (k - 1) // 2
    args = ([1, 1], [p, p], [1, 1], False, [0, 0], 1)
    gy = closed_form((N, Cout, 8, 8), 23, 1.0, dt)
    add = closed_form((N, Cin, 8, 8), 29, 1.0, dt)
    ref = aten.convolution(x.double(), w.double(), b.double(), *args)
    refb = aten.convolution_backward(gy.double(), x.double(), w.double(), [Cout], *args, [True, True, True])
    lib.lamp_kernel_timer_filter(None)
    _timer_classes()                          # drop what earlier tests left in the log
    lib.lamp_kernel_timer_enable(1)
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(b), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0,
                         i64_array([0, 0]), 1)
    out = _out3()
    lib.lamp_convolution_backward(out, to_sten(gy), to_sten(x), to_sten(w), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0,
                                  i64_array([0, 0]), 1, _mask3(1, 1, 1))
    dx, dw, db = _wrap3(out)
    oa = C.c_void_p()
    lib.lamp_convolution_backward_input_add(C.byref(oa), to_sten(gy), to_sten(x), to_sten(w), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2,
                                            i64_array([0, 0]), 1, to_sten(add))
    got_w = to_torch(dw)                      # resolves the deferred reduction of the partial sums
    lib.lamp_kernel_timer_enable(0)
    ran = _timer_classes()
    got_add = to_torch(S.STen(oa))
    assert ran.get("conv_igemm_fprop_dgrad_" + sfx, 0) == 3 and ran.get("conv_wgrad_igemm_" + sfx, 0) == 1, f"the {sfx} matrix-core kernels did not run: {ran}"
    assert_close(to_torch(S.STen(o)), ref, ftol, sfx + " igemm forward")
    assert_close(to_torch(dx), refb[0], btol, sfx + " igemm dgrad")
    assert_close(got_w, refb[1], btol, sfx + " igemm wgrad")
    assert_close(to_torch(db), refb[2], btol, "bias gradient")
    assert_close(got_add, refb[0] + add.double(), btol, sfx + " igemm dgrad + addend")
    # the fused addend is the unfused sum bit for bit: both are fl(fl(dgrad) + addend)
    assert torch.equal(got_add.to(dt), to_torch(dx).to(dt) + add)
    if dt == torch.float32 and N >= 2 and Cout > 16:
        # the statistics hand-off: a training-mode batch norm directly on the f32 convolution's output merges the epilogue's per-image
        # Welford triples instead of running its statistics pass - and gets ATen's statistics
        o2 = C.c_void_p()
        lib.lamp_convolution(C.byref(o2), to_sten(x), to_sten(w), to_sten(b), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0,
                             i64_array([0, 0]), 1)
        Y = S.STen(o2)
        gam, bet = closed_form((Cout,), 7, 1.0, dt) + 1.0, closed_form((Cout,), 9, 1.0, dt)
        _timer_classes()
        lib.lamp_kernel_timer_enable(1)
        out3 = _out3()
        lib.lamp_native_batch_norm(out3, Y, to_sten(gam), to_sten(bet), to_sten(torch.zeros(Cout, dtype=dt)), to_sten(torch.ones(Cout, dtype=dt)), 1, 0.1, 1e-5)
        yn, mean, invstd = _wrap3(out3)
        got_yn = to_torch(yn)
        lib.lamp_kernel_timer_enable(0)
        ran = _timer_classes()
        assert "bn_fwd_stats" not in ran, f"the batch norm ran its own statistics pass: {ran}"
        rn = aten.native_batch_norm(to_torch(Y).double(), gam.double(), bet.double(), torch.zeros(Cout, dtype=torch.float64), torch.ones(Cout, dtype=torch.float64), True, 0.1, 1e-5)
        assert_close(to_torch(mean), rn[1], 1e-5, "save_mean from the f32 epilogue's partials")
        assert_close(to_torch(invstd), rn[2], 1e-5, "save_invstd from the f32 epilogue's partials")
        assert_close(got_yn, rn[0], 1e-5, "batch norm on the hand-off statistics")


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_conv1d_and_transposed(gpu, dt):
    x = closed_form((2, 3, 11), 3, 2.0, dt)
    w = closed_form((4, 3, 3), 17, 1.0, dt)
    b = closed_form((4,), 5, 1.0, dt)
    args = ([2], [1], [1], False, [0], 1)
    ref = aten.convolution(x, w, b, *args)
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(b), i64_array([2]), i64_array([1]), i64_array([1]), 1, 0, i64_array([0]), 1)
    assert_close(to_torch(S.STen(o)), ref.double(), FWD_TOL[dt], "conv1d")
    # transposed 2-D
    xt = closed_form((2, 4, 5, 5), 3, 2.0, dt)
    wt = closed_form((4, 3, 3, 3), 7, 1.0, dt)
    bt = closed_form((3,), 2, 1.0, dt)
    targs = ([2, 2], [1, 1], [1, 1], True, [1, 1], 1)
    reft = aten.convolution(xt, wt, bt, *targs)
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), to_sten(xt), to_sten(wt), to_sten(bt), i64_array([2, 2]), i64_array([1, 1]), i64_array([1, 1]), 2, 1,
                         i64_array([1, 1]), 1)
    assert_close(to_torch(S.STen(o)), reft.double(), FWD_TOL[dt], "transposed conv")
    gy = closed_form(tuple(reft.shape), 23, 1.0, dt)
    refb = aten.convolution_backward(gy, xt, wt, [3], *targs, [True, True, True])
    out = _out3()
    lib.lamp_convolution_backward(out, to_sten(gy), to_sten(xt), to_sten(wt), i64_array([2, 2]), i64_array([1, 1]), i64_array([1, 1]), 2, 1,
                                  i64_array([1, 1]), 1, _mask3(1, 1, 1))
    dx, dw, db = _wrap3(out)
    assert_close(to_torch(dx), refb[0].double(), 1e-4, "transposed dgrad")
    assert_close(to_torch(dw), refb[1].double(), 1e-4, "transposed wgrad")
    assert_close(to_torch(db), refb[2].double(), 1e-4, "transposed bias grad")


@pytest.mark.parametrize("dt", DTYPES)
def test_pooling(gpu, dt):
    x = closed_form((3, 5, 8, 8), 3, 2.0, dt)
    X = to_sten(x)
    tol = FWD_TOL[dt]
    for (k, s, p) in [(8, 1, 0), (2, 2, 0), (3, 2, 1)]:
        ref = aten.avg_pool2d(x, [k], [s], [p], False, True, None)
        o = C.c_void_p(); lib.lamp_avg_pool2d(C.byref(o), X, k, s, p, 0, 1)
        assert_close(to_torch(S.STen(o)), ref.double(), tol * 2, "avg_pool2d")
        gy = closed_form(tuple(ref.shape), 9, 1.0, dt)
        refb = aten.avg_pool2d_backward(gy, x, [k], [s], [p], False, True, None)
        o = C.c_void_p(); lib.lamp_avg_pool2d_backward(C.byref(o), to_sten(gy), X, k, s, p, 0, 1)
        assert_close(to_torch(S.STen(o)), refb.double(), tol * 2, "avg_pool2d backward")
    for (k, s, p, d) in [(2, 2, 0, 1), (3, 2, 1, 1), (3, 1, 1, 2)]:
        ref, idx = aten.max_pool2d_with_indices(x, [k], [s], [p], [d], False)
        o, i = C.c_void_p(), C.c_void_p()
        lib.lamp_max_pool2d_with_indices(C.byref(o), C.byref(i), X, k, s, p, d, 0)
        O, I = S.STen(o), S.STen(i)
        assert_close(to_torch(O), ref.double(), 0.0, "max_pool values (exact)")
        assert np.array_equal(I.to_numpy(), idx.numpy()), "max_pool indices must be bit-exact"
        gy = closed_form(tuple(ref.shape), 9, 1.0, dt)
        refb = aten.max_pool2d_with_indices_backward(gy, x, [k], [s], [p], [d], False, idx)
        o = C.c_void_p(); lib.lamp_max_pool2d_with_indices_backward(C.byref(o), to_sten(gy), X, k, s, p, d, 0, I)
        assert_close(to_torch(S.STen(o)), refb.double(), tol * 2, "max_pool backward")


def test_index_ops_bit_exact(gpu):
    x = closed_form((50, 3), 3, 2.0, torch.float64)
    idx = (torch.arange(120) * 7) % 50
    X, I = to_sten(x), to_sten(idx)
    assert np.array_equal(X.indexSelect(0, I).to_numpy(), x.index_select(0, idx).numpy())
    assert np.array_equal(X.indexSelect(1, to_sten(torch.tensor([2, 0]))).to_numpy(), x.index_select(1, torch.tensor([2, 0])).numpy())
    src = closed_form((120, 3), 9, 1.0, torch.float64)
    got = X.indexAdd(0, I, to_sten(src)).to_numpy()
    np.testing.assert_allclose(got, x.index_add(0, idx, src).numpy(), rtol=1e-13, atol=1e-13)
    m = x > 0.1
    assert np.array_equal(X.maskedSelect(to_sten(m)).to_numpy(), x.masked_select(m).numpy())
    big = closed_form((100001,), 1, 2.0, torch.float32)
    mb = big > 0
    assert np.array_equal(to_sten(big).maskedSelect(to_sten(mb)).to_numpy(), big.masked_select(mb).numpy())
    assert np.array_equal(I.repeatInterleave(5, 0).to_numpy(), idx.repeat_interleave(5, 0).numpy())
    assert np.array_equal(I.oneHot(50).to_numpy(), torch.nn.functional.one_hot(idx, 50).numpy())
    d = closed_form((7, 1000), 5, 2.0, torch.float32)
    v, ix = to_sten(d).topk(10, 1, False, False)
    rv, rix = aten.topk(d, 10, 1, False, True)
    assert np.array_equal(np.sort(ix.to_numpy(), 1), np.sort(rix.numpy(), 1)), "top-k index sets"
    assert np.array_equal(np.sort(v.to_numpy(), 1), np.sort(rv.numpy(), 1))
    v, ix = to_sten(d).topk(3, 1, True, True)
    rv, rix = aten.topk(d, 3, 1, True, True)
    assert np.array_equal(v.to_numpy(), rv.numpy())


def test_rng_statistics(gpu):
    lib.lamp_manual_seed(42)
    u = S.STen.rand([200000], S.F32).to_numpy()
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 5e-3
    n = S.STen.normal(1.0, 2.0, [200000], S.F64).to_numpy()
    assert abs(n.mean() - 1.0) < 2e-2 and abs(n.std() - 2.0) < 2e-2
    r = S.STen.randint(0, 7, [100000]).to_numpy()
    assert r.min() == 0 and r.max() == 6 and r.dtype == np.int64
    lib.lamp_manual_seed(42)
    u2 = S.STen.rand([200000], S.F32).to_numpy()
    assert np.array_equal(u, u2), "same seed, same stream"
    ones = S.STen.ones([100000], S.F32)
    ones.dropout_(0.25, True)
    o = ones.to_numpy()
    assert set(np.unique(o)).issubset({0.0, np.float32(1 / 0.75)}) and abs((o == 0).mean() - 0.25) < 1e-2


def test_errors_are_loud(gpu):
    a = to_sten(closed_form((3, 4), 0, 1.0, torch.float32))
    b = to_sten(closed_form((5, 4), 0, 1.0, torch.float32))
    with pytest.raises(Exception, match="broadcast"):
        a + b
    with pytest.raises(Exception, match="multiplied"):
        a.mm(b)
    with pytest.raises(Exception, match="dtype"):
        a + to_sten(closed_form((3, 4), 0, 1.0, torch.float64))
    host = S.STen.zeros([3, 4], S.F32, device=S.CPU)
    assert host.relu().device == S.CPU                      # lamp's CPU device: element-wise where the tensor lives (tests/test_cpu_device.py)
    staged = host.logSoftMax(1)                                # a GPU-only operator on all-host arguments: staged through the GPU, host result
    assert staged.device == S.CPU and np.allclose(staged.to_numpy(), np.log(0.25))
    with pytest.raises(Exception, match="device"):
        a + host                                               # no silent transfer between devices
    with pytest.raises(Exception, match="host tensor"):
        lib.lamp_log_softmax_backward_data(C.byref(C.c_void_p()), a, host, 1)   # mixed devices on a GPU-only operator: its own message


def test_allocation_registry(gpu):
    import gc
    gc.collect()
    n0 = S.live_tensor_count()
    t = [S.STen.zeros([128, 128]) for _ in range(10)]
    assert S.live_tensor_count() == n0 + 10
    del t
    gc.collect()
    assert S.live_tensor_count() == n0


@pytest.mark.parametrize("cin,cout,hw,stride", [(128, 128, 8, 1), (6, 6, 16, 1), (6, 16, 16, 2), (3, 6, 32, 1)])
def test_igemm_pack_cache_sees_every_weight_write(gpu, cin, cout, hw, stride):
    """The implicit-GEMM and the narrow convolutions cache their packed weights (fprop and dgrad images) per storage version.
    Every way of changing the weights must invalidate them: in-place arithmetic, copy_, a write through a view, a host upload,
    the optimiser (which re-packs the cached images itself, in place)."""
    from lamp_amd import nn as NN
    dt = torch.bfloat16
    k = 5 if cin == 3 else 3
    pad = k // 2
    x = closed_form((4, cin, hw, hw), 3, 2.0, dt)
    w0 = closed_form((cout, cin, k, k), 17, 1.0, dt)
    b = torch.zeros(cout, dtype=dt)
    X, Bt = to_sten(x), to_sten(b)
    W = to_sten(w0)
    ho = (hw + 2 * pad - k) // stride + 1
    gy = closed_form((4, cout, ho, ho), 23, 1.0, dt)
    GY = to_sten(gy)

    def conv(Wt):
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), X, Wt, Bt, i64_array([stride, stride]), i64_array([pad, pad]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
        return to_torch(S.STen(o))

    def dgrad(Wt):
        out3 = (C.c_void_p * 3)()
        mask = (C.c_uint8 * 3)(1, 0, 0)
        lib.lamp_convolution_backward(out3, GY, X, Wt, i64_array([stride, stride]), i64_array([pad, pad]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1, mask)
        return to_torch(S.STen(out3[0]))

    def check(w_now, what):
        ref = aten.convolution(x, w_now, b, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1)
        assert_close(conv(W), ref.double(), 1.6e-2, what)
        gx = aten.convolution_backward(gy.float(), x.float(), w_now.float(), [0], [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        assert_close(dgrad(W), gx.double(), 1.6e-2, what + " (dgrad)")

    check(w0, "first call")
    check(w0, "cached call")
    lib.lamp_mul_scalar_(W, 0.5)
    check(to_torch(W).to(dt), "after mul_scalar_")
    w1 = closed_form((cout, cin, k, k), 5, 1.0, dt)
    lib.lamp_copy_(W, to_sten(w1), 0)
    check(w1, "after copy_")
    W.select(0, 3).fill_(0.25)                              # write through a view of the same storage
    w2 = w1.clone(); w2[3] = 0.25
    check(w2, "after a view write")
    w3 = closed_form((cout, cin, k, k), 9, 1.0, dt)
    W2 = to_sten(w3)                                        # a different tensor with different contents
    ref3 = aten.convolution(x, w3, b, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1)
    assert_close(conv(W2), ref3.double(), 1.6e-2, "another weight tensor")
    # the optimiser: one SGD step changes W, the next convolution must use the new values
    g = S.STen.ones([cout, cin, k, k], S.BF16, 0)
    opt = NN.SGDW([W], 0.5, 0.0)
    opt.step([g], 1.0)
    check(to_torch(W).to(dt), "after the optimiser step")
    opt.step([g], 1.0)
    check(to_torch(W).to(dt), "after a second optimiser step (images re-packed in place)")


@pytest.mark.parametrize("cin,cout,N,H,stride", [(6, 6, 16, 32, 2), (128, 100, 1024, 8, 1)])
def test_pair_launch_sees_every_write_to_either_filter(gpu, cin, cout, N, H, stride):
    """lamp_convolution_pair keeps ONE packed image per pair in the narrow kernel (both filters in it) and two cached images in the
    eight-image kernel: whichever way EITHER filter changes - in place, through the optimiser stepping only the 3x3, only the 1x1, or both
    (the optimiser re-packs the cached images itself, in place: a replayed HIP graph keeps their address) - the next pair launch equals the
    two separate convolutions on the current weights, bit for bit."""
    from lamp_amd import nn as NN
    dt = torch.bfloat16
    x = closed_form((N, cin, H, H), 3, 2.0, dt)
    WA, BA = to_sten(closed_form((cout, cin, 3, 3), 17, 0.2, dt)), to_sten(closed_form((cout,), 5, 1.0, dt))
    WB, BB = to_sten(closed_form((cout, cin, 1, 1), 23, 0.4, dt)), to_sten(closed_form((cout,), 13, 1.0, dt))
    X = to_sten(x)
    one, p1, p0, z, sd = i64_array([1, 1]), i64_array([1, 1]), i64_array([0, 0]), i64_array([0, 0]), i64_array([stride, stride])

    def check(what):
        o2 = (C.c_void_p * 2)()
        lib.lamp_convolution_pair(o2, X, WA, BA, sd, p1, one, WB, BB, sd, p0, one, 2, 1)
        pa, pb = S.STen(o2[0]), S.STen(o2[1])
        oa, ob = C.c_void_p(), C.c_void_p()
        lib.lamp_convolution(C.byref(oa), X, WA, BA, sd, p1, one, 2, 0, z, 1)
        lib.lamp_convolution(C.byref(ob), X, WB, BB, sd, p0, one, 2, 0, z, 1)
        assert torch.equal(to_torch(pa), to_torch(S.STen(oa))), what + ": 3x3"
        assert torch.equal(to_torch(pb), to_torch(S.STen(ob))), what + ": 1x1"
        # ... and the single convolutions themselves follow the weights (against the oracle on the CURRENT values)
        ra = aten.convolution(x.float(), to_torch(WA).float(), to_torch(BA).float(), [stride, stride], [1, 1], [1, 1], False, [0, 0], 1)
        assert_close(to_torch(pa), ra.double(), FWD_TOL[dt] * 4, what + ": 3x3 against the oracle")
        rb = aten.convolution(x.float(), to_torch(WB).float(), to_torch(BB).float(), [stride, stride], [0, 0], [1, 1], False, [0, 0], 1)
        assert_close(to_torch(pb), rb.double(), FWD_TOL[dt] * 4, what + ": 1x1 against the oracle")

    check("first call"); check("cached call")
    lib.lamp_mul_scalar_(WB, 0.5); check("after the 1x1 was scaled in place")
    lib.lamp_mul_scalar_(WA, -1.5); check("after the 3x3 was scaled in place")
    ga, gb = S.STen.ones([cout, cin, 3, 3], S.BF16, 0), S.STen.ones([cout, cin, 1, 1], S.BF16, 0)
    NN.SGDW([WA], 0.25, 0.0).step([ga], 1.0); check("after the optimiser stepped the 3x3 alone")
    NN.SGDW([WB], 0.25, 0.0).step([gb], 1.0); check("after the optimiser stepped the 1x1 alone")
    both = NN.SGDW([WA, WB], 0.125, 0.0)
    both.step([ga, gb], 1.0); check("after the optimiser stepped both")
    both.step([ga, gb], 1.0); check("after a second step of both (images re-packed in place)")


def test_pair_image_is_never_packed_again_from_a_filter_that_is_gone(gpu):
    """the narrow kernels' pair image used to remember RAW data pointers of both filters and the optimiser's re-pack hook
    read the absent one's - after a strided sibling (whose contiguous copy dies with the call) or a rebuilt shortcut that pointer dangled.  Now
    an image is re-packed only from tensors the optimiser is handed, a pair entry with one filter missing is dropped, and copies made inside
    the entry point are not cached at all.  The pair must keep equalling the two convolutions on the CURRENT weights through all of it."""
    from lamp_amd import nn as NN
    dt = torch.bfloat16
    cin = cout = 6
    N, H, stride = 16, 32, 2
    x = closed_form((N, cin, H, H), 3, 2.0, dt)
    X = to_sten(x)
    WA, BA = to_sten(closed_form((cout, cin, 3, 3), 17, 0.2, dt)), to_sten(closed_form((cout,), 5, 1.0, dt))
    wide = to_sten(closed_form((cout, 2 * cin, 1, 1), 23, 0.4, dt))
    BB = to_sten(closed_form((cout,), 13, 1.0, dt))
    one, p1, p0, z, sd = i64_array([1, 1]), i64_array([1, 1]), i64_array([0, 0]), i64_array([0, 0]), i64_array([stride, stride])

    def check(WB, what):
        o2 = (C.c_void_p * 2)()
        lib.lamp_convolution_pair(o2, X, WA, BA, sd, p1, one, WB, BB, sd, p0, one, 2, 1)
        pa, pb = S.STen(o2[0]), S.STen(o2[1])
        ra = aten.convolution(x.float(), to_torch(WA).float(), to_torch(BA).float(), [stride, stride], [1, 1], [1, 1], False, [0, 0], 1)
        rb = aten.convolution(x.float(), to_torch(WB).float(), to_torch(BB).float(), [stride, stride], [0, 0], [1, 1], False, [0, 0], 1)
        assert_close(to_torch(pa), ra.double(), FWD_TOL[dt] * 4, what + ": 3x3")
        assert_close(to_torch(pb), rb.double(), FWD_TOL[dt] * 4, what + ": 1x1")

    ga = S.STen.ones([cout, cin, 3, 3], S.BF16, 0)
    opt = NN.SGDW([WA], 0.125, 0.0)
    # (1) a strided sibling: every call makes (and frees) a contiguous copy of it
    WBs = wide.slice(1, 0, 2 * cin, 2)
    assert not WBs.is_contiguous()
    for i in range(4):
        check(WBs, f"strided sibling, round {i}")
        junk = [S.STen.full([cout * cin], 1e4, S.BF16, 0) for _ in range(8)]          # what the freed copy's pool block is handed out as next
        opt.step([ga], 1.0)                                                         # the hook visits the 3x3 alone
        del junk
    # (2) the shortcut is rebuilt while the 3x3 is kept
    for i in range(4):
        WB = to_sten(closed_form((cout, cin, 1, 1), 29 + i, 0.4, dt))
        check(WB, f"rebuilt shortcut {i}: first call"); check(WB, f"rebuilt shortcut {i}: cached call")
        del WB
        junk = [S.STen.full([cout * cin], -1e4, S.BF16, 0) for _ in range(8)]
        opt.step([ga], 1.0)
        del junk
    WB = to_sten(closed_form((cout, cin, 1, 1), 41, 0.4, dt))
    check(WB, "after the rebuilds")
    gb = S.STen.ones([cout, cin, 1, 1], S.BF16, 0)
    NN.SGDW([WA, WB], 0.125, 0.0).step([ga, gb], 1.0)
    check(WB, "both stepped")
    lib.lamp_device_synchronize()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("reduction", [1, 2])
def test_nll_loss_forward_accumulates_the_epoch_loss_in_its_launch(gpu, dt, reduction):
    """lamp_nll_loss_forward_accumulate_ = lamp_nll_loss_forward followed by acc.add_(loss, alpha = n) (IOLoops.scala:714), bitwise:
    loss, total weight and the accumulator; the accumulator's dtype and size are checked."""
    N, Cc = 257, 10
    x = torch.log_softmax(closed_form((N, Cc), 3, 4.0, dt).double(), 1).to(dt)
    t = (torch.arange(N) * 7) % Cc
    t[5] = -100
    w = closed_form((Cc,), 9, 1.0, dt) + 1.0
    X, Tt, Wt = to_sten(x), to_sten(t), to_sten(w)
    acc0 = closed_form((1,), 21, 8.0, dt)
    o, tw = C.c_void_p(), C.c_void_p()
    lib.lamp_nll_loss_forward(C.byref(o), C.byref(tw), X, Tt, Wt, reduction, -100)
    L, TW = S.STen(o), S.STen(tw)
    A1 = to_sten(acc0)
    lib.lamp_add_(A1, L.reshape(1), float(N))
    A2 = to_sten(acc0)
    o2, tw2 = C.c_void_p(), C.c_void_p()
    lib.lamp_nll_loss_forward_accumulate_(C.byref(o2), C.byref(tw2), X, Tt, Wt, reduction, -100, A2, float(N))
    L2, TW2 = S.STen(o2), S.STen(tw2)
    assert np.array_equal(L2.to_numpy(), L.to_numpy()) and np.array_equal(TW2.to_numpy(), TW.to_numpy())
    assert np.array_equal(A2.to_numpy(), A1.to_numpy()), (A2.to_numpy(), A1.to_numpy())
    with pytest.raises(Exception, match="one-element tensor of the input's dtype"):
        lib.lamp_nll_loss_forward_accumulate_(C.byref(o2), C.byref(tw2), X, Tt, Wt, reduction, -100, to_sten(closed_form((2,), 1, 1.0, dt)), 1.0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(64, 100, 8, 8), (2048, 100, 8, 8), (37, 10, 4, 4), (1, 7, 8, 8), (19, 130, 8, 8)])
def test_loss_launch_also_yields_the_pooled_log_softmax_input_gradient(gpu, dt, shape):
    """lamp_nll_loss_forward_pooled_gradient_ (round 6): the NllLoss forward of Cnn.resnet's tail and - from the same launch, for the derivative
    of one that backprop seeds a loss with (autograd.scala:264-282) - the input gradient of the pooled LogSoftMax in front of it, as ONE value per
    (sample, class) plane.  Loss, total weight and epoch accumulator BITWISE those of lamp_nll_loss_forward(_accumulate_); the plane values
    BITWISE every element of their plane in lamp_global_avg_pool_log_softmax_nll_backward(ones, ...) - with and without class weights, with an
    ignored class, mean and sum; and against the f64 mathematics."""
    if shape[0] == 2048 and dt != torch.bfloat16:
        pytest.skip("the large case only in the benchmark's dtype")
    N, Cc, H, _ = shape
    x = closed_form(shape, 3, 6.0, dt)
    X = to_sten(x)
    o = C.c_void_p(); lib.lamp_global_avg_pool_log_softmax(C.byref(o), X)
    Y = S.STen(o)
    target = (torch.arange(N) * 7) % Cc
    if N > 5:
        target[5] = 3
    T_ = to_sten(target)
    wts = closed_form((Cc,), 13, 1.0, dt).abs() + 0.5
    ones = S.STen.ones([], S.F64 if dt == torch.float64 else S.F32 if dt == torch.float32 else S.BF16, 0)
    for reduction in (1, 2):
        for W_ in (None, to_sten(wts)):
            for ignore in (-100, 3):
                lv, tw = C.c_void_p(), C.c_void_p()
                acc0 = closed_form((1,), 21, 8.0, dt)
                A1 = to_sten(acc0)
                lib.lamp_nll_loss_forward_accumulate_(C.byref(lv), C.byref(tw), Y, T_, W_, reduction, ignore, A1, float(N))
                LV, TW = S.STen(lv), S.STen(tw)
                one = C.c_void_p(); lib.lamp_global_avg_pool_log_softmax_nll_backward(C.byref(one), ones, T_, W_, reduction, ignore, TW, Y, X)
                DX = S.STen(one).to_numpy()
                A2 = to_sten(acc0)
                l2, t2, pg = C.c_void_p(), C.c_void_p(), C.c_void_p()
                lib.lamp_nll_loss_forward_pooled_gradient_(C.byref(l2), C.byref(t2), C.byref(pg), Y, T_, W_, reduction, ignore, A2, float(N), H * H)
                what = f"reduction {reduction}, weights {W_ is not None}, ignore {ignore}"
                assert pg.value, "no fused form for " + str(shape)
                assert np.array_equal(S.STen(l2).to_numpy(), LV.to_numpy()) and np.array_equal(S.STen(t2).to_numpy(), TW.to_numpy()), what
                assert np.array_equal(A2.to_numpy(), A1.to_numpy()), what
                PG = S.STen(pg)
                assert PG.shape == [N, Cc]
                view = PG.unsqueeze(2).unsqueeze(3).expand([N, Cc, H, H])
                assert view.strides[2:] == [0, 0] and (N == 1 or view.strides[:2] == [1, N])        # class-major values
                assert np.array_equal(view.to_numpy(), DX), what + ": plane values differ from the backward launch"
                # without an accumulator
                l3, t3, pg3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
                lib.lamp_nll_loss_forward_pooled_gradient_(C.byref(l3), C.byref(t3), C.byref(pg3), Y, T_, W_, reduction, ignore, None, 0.0, H * H)
                assert np.array_equal(S.STen(l3).to_numpy(), LV.to_numpy()) and np.array_equal(S.STen(pg3).to_numpy(), PG.to_numpy()), what
    # the mathematics (mean, weighted): d loss / d x[n, c, h, w] = w[t_n] (softmax(pool(x))[n, c] - [c == t_n]) / (sum w[t]) / (H W)
    xd = x.double().requires_grad_(True)
    lp = torch.log_softmax(xd.mean(dim=(2, 3)), dim=1)
    torch.nn.functional.nll_loss(lp, target, weight=wts.double(), reduction="mean").backward()
    l4, t4, pg4 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib.lamp_nll_loss_forward_pooled_gradient_(C.byref(l4), C.byref(t4), C.byref(pg4), Y, T_, to_sten(wts), 1, -100, None, 0.0, H * H)
    assert_close(to_torch(S.STen(pg4)), xd.grad[:, :, 0, 0], {torch.float64: 1e-10, torch.float32: 1e-3, torch.bfloat16: 6e-2}[dt], "plane gradient vs f64")
    with pytest.raises(Exception, match="reduction mean"):
        lib.lamp_nll_loss_forward_pooled_gradient_(C.byref(l4), C.byref(t4), C.byref(pg4), Y, T_, None, 0, -100, None, 0.0, H * H)


@pytest.mark.parametrize("shape", [(2048, 100, 8, 8), (256, 128, 8, 8), (70, 16, 16, 16)])
def test_batch_norm_pair_backward_reads_a_plane_broadcast_gradient_in_place(gpu, shape):
    """The one-pass backward of relu(bn(x) + bn2(x2)) handed the loss tail's gradient as the stride-0 view expand([N, C, 1, 1] -> [N, C, H, W])
    (lamp_nll_loss_forward_pooled_gradient_): all six results BITWISE those for the materialised tensor - the kernel reads the [N, C] values and
    the 26 MB gradient of the benchmark's last block is neither written nor read."""
    dt = torch.bfloat16
    N, Cc, H, _ = shape
    x, x2 = closed_form(shape, 3, 4.0, dt), closed_form(shape, 29, 3.0, dt)
    w, b, w2, b2 = (closed_form((Cc,), k, 1.0, dt) + (1.0 if k in (5, 11) else 0.0) for k in (5, 7, 11, 13))
    pg = closed_form((N, Cc), 17, 0.01, dt)
    X, X2, W_, B_, W2, B2, PG = (to_sten(t) for t in (x, x2, w, b, w2, b2, pg))
    z = lambda: to_sten(torch.zeros(Cc, dtype=dt))
    out5 = (C.c_void_p * 5)()
    lib.lamp_native_batch_norm2_add_relu(out5, X, W_, B_, z(), z(), X2, W2, B2, z(), z(), 0.1, 0.1, 1e-5, 1e-5)
    _, sm, si, sm2, si2 = (S.STen(h) for h in out5)
    view = PG.view(N, Cc, 1, 1).expand([N, Cc, H, H])
    dense = view.contiguous()
    assert view.strides == [Cc, 1, 0, 0] and dense.strides != view.strides
    # ... and class-major values (what the loss launch hands out): strides [1, N, 0, 0]
    PGT = to_sten(pg.t().contiguous()).transpose(0, 1)
    view_t = PGT.unsqueeze(2).unsqueeze(3).expand([N, Cc, H, H])
    assert view_t.strides == [1, N, 0, 0] and np.array_equal(view_t.to_numpy(), dense.to_numpy())
    mask = (C.c_uint8 * 6)(1, 1, 1, 1, 1, 1)
    res = {}
    for name, G in (("view", view), ("class-major view", view_t), ("dense", dense)):
        out6 = (C.c_void_p * 6)()
        lib.lamp_native_batch_norm2_add_relu_backward(out6, G, X, W_, B_, sm, si, X2, W2, B2, sm2, si2, 1e-5, 1e-5, mask)
        res[name] = [S.STen(h).to_numpy() for h in out6]
    for name in ("view", "class-major view"):
        for i, (a, b_) in enumerate(zip(res[name], res["dense"])):
            assert np.array_equal(a, b_), f"result {i} differs between the {name} and the materialised gradient"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(64, 100, 8, 8), (2048, 100, 8, 8), (33, 16, 16, 16), (5, 10, 8, 8), (4, 12, 12, 12), (3, 7, 32, 32)])
def test_last_block_and_tail_in_one_call_never_write_the_block_s_output(gpu, dt, shape):
    """lamp_native_batch_norm2_add_relu_pool_log_softmax (round 6): relu(bn(x) + bn2(x2)) -> AvgPool2D over the map -> Flatten -> LogSoftMax, the
    last block of Cnn.resnet with the network's tail (cnn.scala:129-136).  The block's output is read by the pool only, so the normalise kernel
    leaves its plane means instead (summed in the pool kernel's order) and the LogSoftMax runs on those: the log-probabilities, the four saved
    statistics and the running statistics are BITWISE those of lamp_native_batch_norm2_add_relu followed by lamp_global_avg_pool_log_softmax -
    also where the pooled form does not apply (12 x 12 maps: 18 packets per plane) and the two calls run inside the entry point."""
    if shape[0] == 2048 and dt != torch.bfloat16:
        pytest.skip("the large case only in the benchmark's dtype")
    N, Cc, H, _ = shape
    x, x2 = closed_form(shape, 3, 4.0, dt), closed_form(shape, 29, 3.0, dt)
    w, b, w2, b2 = (closed_form((Cc,), k, 1.0, dt) + (1.0 if k in (5, 11) else 0.0) for k in (5, 7, 11, 13))
    X, X2, W_, B_, W2, B2 = (to_sten(t) for t in (x, x2, w, b, w2, b2))
    rs = [to_sten(torch.zeros(Cc, dtype=dt)) for _ in range(4)] + [to_sten(torch.ones(Cc, dtype=dt)) for _ in range(4)]   # running means / variances of both paths
    lib.lamp_kernel_timer_enable(1)
    out5 = (C.c_void_p * 5)()
    lib.lamp_native_batch_norm2_add_relu_pool_log_softmax(out5, X, W_, B_, rs[0], rs[4], X2, W2, B2, rs[1], rs[5], 0.1, 0.2, 1e-5, 1e-4)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    got = [S.STen(h) for h in out5]
    ref5 = (C.c_void_p * 5)()
    lib.lamp_native_batch_norm2_add_relu(ref5, X, W_, B_, rs[2], rs[6], X2, W2, B2, rs[3], rs[7], 0.1, 0.2, 1e-5, 1e-4)
    ref = [S.STen(h) for h in ref5]
    o = C.c_void_p()
    lib.lamp_global_avg_pool_log_softmax(C.byref(o), ref[0])
    want = S.STen(o)
    assert got[0].shape == [N, Cc]
    assert np.array_equal(got[0].to_numpy(), want.to_numpy()), "log-probabilities differ from the two calls"
    for i in range(1, 5):
        assert np.array_equal(got[i].to_numpy(), ref[i].to_numpy()), f"saved statistic {i} differs"
    for a, b_ in ((0, 2), (1, 3), (4, 6), (5, 7)):
        assert np.array_equal(rs[a].to_numpy(), rs[b_].to_numpy()), "running statistics differ"
    # against the oracle: ATen's chain in f64 on the same operands
    f = torch.float64
    bn = lambda t, ww, bb: aten.native_batch_norm(t.to(f), ww.to(f), bb.to(f), None, None, True, 0.1, 1e-5)[0]
    yref = torch.relu(bn(x, w, b) + aten.native_batch_norm(x2.to(f), w2.to(f), b2.to(f), None, None, True, 0.2, 1e-4)[0])
    oref = torch.log_softmax(yref.mean(dim=(2, 3)), 1)
    assert_close(to_torch(got[0]), oref, {torch.float64: 1e-10, torch.float32: 2e-4, torch.bfloat16: 6e-2}[dt], "log-probabilities against the f64 chain")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(64, 100, 8, 8), (2048, 100, 8, 8), (5, 10, 4, 4), (7, 33, 16, 16), (3, 256, 8, 8), (4, 12, 7, 7), (2, 6, 2, 2)])
def test_global_avg_pool_log_softmax_is_bitwise_the_three_call_chain(gpu, dt, shape):
    """lamp_global_avg_pool_log_softmax(+_backward) = avg_pool2d(k = H) -> flatten -> log_softmax(1) and its backward, BITWISE: one
    kernel per direction where the planes are whole 16-byte packets (the tail of Cnn.resnet: cnn.scala:129-136), the three calls
    otherwise (7x7 maps, 256-wide rows that the packet row kernels would serve, 2x2 f32 maps)."""
    if shape[0] == 2048 and dt != torch.bfloat16:
        pytest.skip("the large case only in the benchmark's dtype")
    N, Cc, H, _ = shape
    x = closed_form(shape, 3, 6.0, dt)
    gy = closed_form((N, Cc), 11, 2.0, dt)
    X, GY = to_sten(x), to_sten(gy)
    p = C.c_void_p(); lib.lamp_avg_pool2d(C.byref(p), X, H, 1, 0, 0, 1)
    P = S.STen(p)
    f = C.c_void_p(); lib.lamp_reshape(C.byref(f), P, i64_array([N, Cc]), 2)
    F_ = S.STen(f)
    l = C.c_void_p(); lib.lamp_log_softmax(C.byref(l), F_, 1)
    L = S.STen(l)
    o = C.c_void_p(); lib.lamp_global_avg_pool_log_softmax(C.byref(o), X)
    Ofused = S.STen(o)
    assert Ofused.shape == [N, Cc]
    assert np.array_equal(Ofused.to_numpy(), L.to_numpy()), "forward differs from the chain"
    ref = torch.log_softmax(x.double().mean(dim=(2, 3)), dim=1)
    assert_close(to_torch(Ofused), ref, {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 3e-2}[dt], "forward vs f64")
    gi = C.c_void_p(); lib.lamp_log_softmax_backward_data(C.byref(gi), GY, L, 1)
    GI = S.STen(gi)
    g4 = C.c_void_p(); lib.lamp_reshape(C.byref(g4), GI, i64_array([N, Cc, 1, 1]), 4)
    G4 = S.STen(g4)
    dxc = C.c_void_p(); lib.lamp_avg_pool2d_backward(C.byref(dxc), G4, X, H, 1, 0, 0, 1)
    DXc = S.STen(dxc)
    dxf = C.c_void_p(); lib.lamp_global_avg_pool_log_softmax_backward(C.byref(dxf), GY, Ofused, X)
    DXf = S.STen(dxf)
    assert DXf.shape == list(shape)
    assert np.array_equal(DXf.to_numpy(), DXc.to_numpy()), "backward differs from the chain"
    # ... and with the NllLoss behind it (SupervisedModel.scala: the loss of the pooled log-probabilities): nll_loss_backward -> the call above
    # as ONE call, bitwise, for every reduction, with and without class weights, with an ignored class
    target = (torch.arange(N) * 7) % Cc
    wts = closed_form((Cc,), 13, 1.0, dt).abs() + 0.5
    T_ = to_sten(target)
    for reduction in (0, 1, 2):
        for W_ in (None, to_sten(wts)):
            for ignore in (-100, 7):
                lv, tw = C.c_void_p(), C.c_void_p()
                lib.lamp_nll_loss_forward(C.byref(lv), C.byref(tw), Ofused, T_, W_, reduction, ignore)
                LV, TW = S.STen(lv), S.STen(tw)
                gl_ = closed_form(tuple(LV.shape) if LV.shape else (1,), 17, 1.0, dt).reshape(LV.shape) + 1.0
                GL = to_sten(gl_)
                gyn = C.c_void_p(); lib.lamp_nll_loss_backward(C.byref(gyn), GL, Ofused, T_, W_, reduction, ignore, TW)
                two = C.c_void_p(); lib.lamp_global_avg_pool_log_softmax_backward(C.byref(two), S.STen(gyn), Ofused, X)
                one = C.c_void_p(); lib.lamp_global_avg_pool_log_softmax_nll_backward(C.byref(one), GL, T_, W_, reduction, ignore, TW, Ofused, X)
                assert np.array_equal(S.STen(one).to_numpy(), S.STen(two).to_numpy()), f"loss backward differs from the two calls (reduction {reduction}, ignore {ignore})"


# (N, Cin, H, Cout, k, stride, pad, dtype, fused?)  fused = the dgrad kernel has the accumulate epilogue for this geometry
DGRAD_ADD_CASES = [
    (1024, 128, 8, 128, 3, 1, 1, torch.bfloat16, True),     # eight-image implicit GEMM, 8 channel tiles
    (1024, 100, 8, 128, 3, 1, 1, torch.bfloat16, True),     # 7 tiles, 100 of 112 channels stored
    (1024, 64, 8, 128, 1, 1, 0, torch.bfloat16, True),      # 1x1, 4 tiles
    (1024, 16, 8, 128, 3, 1, 1, torch.bfloat16, True),      # 1 tile
    (1027, 128, 8, 64, 3, 1, 1, torch.bfloat16, True),      # ragged last workgroup
    (64, 128, 8, 128, 3, 1, 1, torch.bfloat16, True),       # small batch: the two-image kernel (eight waves), adds in its stores since round 5
    (600, 128, 8, 100, 3, 1, 1, torch.bfloat16, True),      # two-image kernel, four waves, 100 of 128 channels stored
    (65, 100, 8, 128, 1, 1, 0, torch.bfloat16, True),       # 1x1, ragged last workgroup
    (64, 64, 8, 128, 3, 1, 1, torch.bfloat16, True),        # at most 64 gradient channels out: the 64-row kernel
    (31, 16, 8, 64, 1, 1, 0, torch.bfloat16, True),
    (64, 6, 32, 6, 3, 1, 1, torch.bfloat16, True),          # narrow kernel, two output phases per MFMA
    (64, 16, 16, 16, 3, 1, 1, torch.bfloat16, True),        # narrow kernel, one phase per MFMA
    (64, 3, 32, 8, 5, 1, 2, torch.bfloat16, True),          # 5x5
    (64, 16, 32, 32, 3, 2, 1, torch.bfloat16, None),        # stride 2: whichever kernel serves it
    (8, 5, 12, 7, 3, 1, 1, torch.float32, False),           # f32: direct kernel, then an add
    (4, 6, 10, 4, 3, 1, 1, torch.float64, False),
    (64, 6, 32, 6, 3, 2, 1, torch.float32, True),           # round 5: the strided narrow f32 / f64 dgrad adds in its store (res1.r1 of the ResNet)
    (64, 6, 16, 16, 3, 2, 1, torch.float32, True),          # res2.r1
    (33, 6, 32, 6, 1, 2, 0, torch.float64, True),           # the 1x1 stride-2 shortcut
    (64, 16, 8, 16, 3, 1, 1, torch.float64, True),          # stride 1
]


@pytest.mark.parametrize("case", DGRAD_ADD_CASES)
def test_convolution_input_gradient_accumulates_in_the_kernel(gpu, case):
    """lamp_convolution_backward_input_add = lamp_convolution_backward(mask 1,0,0) followed by lamp_add, BITWISE (each step rounded to
    the dtype), with the add folded into the dgrad kernel's epilogue where that kernel has one (no elementwise launch then).
    autograd.scala:66-84: a Variable with two consumers accumulates its partial derivatives; this is the second `+=`."""
    N, Cin, H, Cout, k, stride, pad, dt, fused = case
    x = closed_form((N, Cin, H, H), 3, 2.0, dt)
    w = closed_form((Cout, Cin, k, k), 17, 0.5, dt)
    ho = (H + 2 * pad - k) // stride + 1
    gy = closed_form((N, Cout, ho, ho), 23, 1.0, dt)
    addend = closed_form((N, Cin, H, H), 29, 3.0, dt)
    X, W, GY, A = to_sten(x), to_sten(w), to_sten(gy), to_sten(addend)
    geom = (i64_array([stride, stride]), i64_array([pad, pad]), i64_array([1, 1]), 2)
    out3 = _out3()
    lib.lamp_convolution_backward(out3, GY, X, W, *geom, 0, i64_array([0, 0]), 1, _mask3(1, 0, 0))
    dx = S.STen(out3[0])
    chain = C.c_void_p()
    lib.lamp_add(C.byref(chain), A, dx, 1.0)
    lib.lamp_kernel_timer_enable(1)
    o = C.c_void_p()
    lib.lamp_convolution_backward_input_add(C.byref(o), GY, X, W, *geom, i64_array([0, 0]), 1, A)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    got, want = to_torch(S.STen(o)), to_torch(S.STen(chain))
    assert got.shape == want.shape and torch.equal(got, want), f"max diff {(got - want).abs().max().item()}"
    assert torch.equal(to_torch(A).double(), addend.double()), "the addend is not modified"
    if fused is not None:
        assert (b"elementwise" not in buf.value) == fused, buf.value.decode()
    gx = aten.convolution_backward(gy.float(), x.float(), w.float(), [0], [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    tol = {torch.float64: 1e-6, torch.float32: 1e-4, torch.bfloat16: 3e-2}[dt]
    assert_close(got, (gx.double() + addend.double()), tol, "dgrad + addend")


# (N, Cin, H, Cout_a, Cout_b, stride, dtype, addend?, one launch?)
DGRAD_PAIR_CASES = [
    (64, 6, 32, 6, 6, 2, torch.bfloat16, False, True),      # res1 of Cnn.resnet: 6 + 6 gradient channels on 16x16 -> 6 x 32x32
    (67, 6, 32, 6, 6, 2, torch.bfloat16, True, True),       # ragged batch, a third consumer's contribution already there
    (64, 6, 16, 16, 16, 2, torch.bfloat16, False, True),    # res2: 16 + 16 channels, sixteen k-steps
    (2050, 6, 16, 16, 16, 2, torch.bfloat16, True, True),   # more images than workgroups
    (33, 6, 16, 6, 6, 1, torch.bfloat16, False, True),      # stride 1
    (40, 6, 16, 12, 12, 2, torch.bfloat16, False, True),    # 48 K pairs: twelve k-steps in the plain form, 8 + 8 by row parity
    (40, 5, 16, 9, 7, 2, torch.bfloat16, True, True),       # ... 34 pairs, uneven halves
    (40, 6, 16, 8, 8, 2, torch.bfloat16, False, True),      # 32 pairs: both halves full (4 + 4)
    (16, 8, 32, 5, 3, 1, torch.bfloat16, False, True),
    (1024, 128, 8, 128, 128, 1, torch.bfloat16, False, "conv_igemm_fprop_dgrad"),   # res4: the eight-image kernel takes the second gradient as a second set of images
    (1027, 100, 8, 128, 128, 1, torch.bfloat16, True, "conv_igemm_fprop_dgrad"),    # ragged last workgroup, 7 channel tiles, a third contribution
    (1032, 128, 8, 100, 100, 1, torch.bfloat16, True, "conv_igemm_fprop_dgrad"),    # res4's own geometry: 100 gradient channels, the 3x3's last four as K-tail stages
    (1024, 16, 8, 128, 128, 1, torch.bfloat16, False, "conv_igemm_fprop_dgrad"),    # res3: 16 input channels (one tile)
    (1024, 64, 8, 64, 64, 1, torch.bfloat16, True, "conv_igemm_fprop_dgrad"),       # two 32-channel chunks, 4 tiles
    (64, 128, 8, 100, 100, 1, torch.bfloat16, True, "conv_igemm_fprop_dgrad"),     # small batch: the two-image kernel (eight waves) takes the second source too
    (600, 128, 8, 128, 128, 1, torch.bfloat16, False, "conv_igemm_fprop_dgrad"),   # ... with four waves
    (33, 16, 8, 128, 128, 1, torch.bfloat16, True, "conv_igemm_fprop_dgrad"),      # ... the 64-row kernel (16 gradient channels out), ragged
    (64, 64, 8, 64, 64, 1, torch.bfloat16, False, "conv_igemm_fprop_dgrad"),
    (8, 5, 12, 7, 4, 1, torch.float32, False, False),
    (4, 6, 10, 4, 4, 2, torch.float64, True, False),
]


@pytest.mark.parametrize("case", DGRAD_PAIR_CASES)
def test_input_gradient_of_a_block_s_two_first_convolutions(gpu, case):
    """lamp_convolution_backward_input_pair: the input gradient of the 3x3 and the 1x1 convolution lamp's residual block applies to its
    input (cnn.scala:16-20), summed - what autograd.scala:66-84 accumulates from the two consumers.  The narrow bf16 layers take both
    output gradients in one launch (the second tensor as extra channels of the staged image, its filter as their centre tap), summed in f32
    and rounded once: within one bf16 rounding of the chain convolution_backward -> convolution_backward_input_add, and at least as close
    to the f32 oracle.  The eight-image implicit-GEMM kernel (8x8 maps, N >= 4 x CUs) does the same with two sets of images in turn and one
    set of accumulators.  Every other geometry runs that chain inside the entry point, bitwise."""
    N, Cin, H, Ca, Cb, stride, dt, with_add, one_launch = case
    x = closed_form((N, Cin, H, H), 3, 2.0, dt)
    wa, wb = closed_form((Ca, Cin, 3, 3), 17, 0.5, dt), closed_form((Cb, Cin, 1, 1), 19, 0.7, dt)
    ho = (H + 2 - 3) // stride + 1
    assert ho == (H - 1) // stride + 1
    ga, gb = closed_form((N, Ca, ho, ho), 23, 1.0, dt), closed_form((N, Cb, ho, ho), 31, 1.0, dt)
    addend = closed_form((N, Cin, H, H), 29, 3.0, dt)
    X, WA, WB, GA, GB, A = to_sten(x), to_sten(wa), to_sten(wb), to_sten(ga), to_sten(gb), to_sten(addend)
    sd, p1, p0, one, z = i64_array([stride, stride]), i64_array([1, 1]), i64_array([0, 0]), i64_array([1, 1]), i64_array([0, 0])
    # the chain, in the order backprop reaches the block: the shortcut's gradient (+ addend), then the 3x3's added to it
    if with_add:
        first = C.c_void_p()
        lib.lamp_convolution_backward_input_add(C.byref(first), GB, X, WB, sd, p0, one, 2, z, 1, A)
        first = S.STen(first)
    else:
        out3 = _out3()
        lib.lamp_convolution_backward(out3, GB, X, WB, sd, p0, one, 2, 0, z, 1, _mask3(1, 0, 0))
        first = S.STen(out3[0])
    chain = C.c_void_p()
    lib.lamp_convolution_backward_input_add(C.byref(chain), GA, X, WA, sd, p1, one, 2, z, 1, first)
    chain = to_torch(S.STen(chain))
    lib.lamp_kernel_timer_enable(1)
    o = C.c_void_p()
    lib.lamp_convolution_backward_input_pair(C.byref(o), X, GA, WA, sd, p1, one, GB, WB, sd, p0, one, 2, 1, A if with_add else None)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    got = to_torch(S.STen(o))
    rep = buf.value.decode()
    f = torch.float64 if dt == torch.float64 else torch.float32
    ref = (aten.convolution_backward(ga.to(f), x.to(f), wa.to(f), [0], [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0] +
           aten.convolution_backward(gb.to(f), x.to(f), wb.to(f), [0], [stride, stride], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]).double()
    if with_add:
        ref = ref + addend.double()
    tol = {torch.float64: 1e-6, torch.float32: 1e-4, torch.bfloat16: 3e-2}[dt]
    assert_close(got, ref, tol, "pair against the oracle")
    if one_launch:
        cls = "conv_dgrad_narrow" if one_launch is True else one_launch
        lines = [l for l in rep.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith(cls) and int(lines[0].split()[1]) == 1, rep
        assert_close(got, chain.double(), 2.0 ** -6, "pair against the chain")   # (each of the chain's two roundings is relative to its own term)
        err_pair, err_chain = (got.double() - ref).abs().mean().item(), (chain.double() - ref).abs().mean().item()
        assert err_pair <= err_chain * 1.02, f"one rounding should not be further from the oracle than two: {err_pair} vs {err_chain}"
    else:
        assert torch.equal(got, chain), f"max diff {(got - chain).abs().max().item()}"
    assert torch.equal(to_torch(A).double(), addend.double()), "the addend is not modified"


# (N, Cin, H, Cout_a, Cout_b, stride, dtype, one launch?)
WGRAD_PAIR_CASES = [
    (64, 6, 32, 6, 6, 2, torch.bfloat16, True),         # res1 of Cnn.resnet
    (67, 6, 32, 6, 6, 2, torch.bfloat16, True),         # ragged batch
    (2050, 6, 16, 6, 6, 1, torch.bfloat16, True),       # stride 1, more image groups than workgroups
    (16, 8, 32, 5, 3, 1, torch.bfloat16, True),
    (33, 3, 32, 8, 8, 2, torch.bfloat16, True),         # all 16 rows of the tile
    (64, 6, 16, 16, 16, 2, torch.bfloat16, False),      # res2: 32 rows do not fit one tile - two launches
    (64, 128, 8, 128, 128, 1, torch.bfloat16, "conv_wgrad_igemm"),     # implicit-GEMM layers with more than 32 input channels: the eight-wave kernel stages the second gradient too
    (1027, 128, 8, 100, 100, 1, torch.bfloat16, "conv_wgrad_igemm"),   # res4 of Cnn.resnet, ragged
    (2048, 64, 8, 128, 100, 1, torch.bfloat16, "conv_wgrad_igemm"),    # different output-channel counts
    (64, 16, 8, 128, 128, 1, torch.bfloat16, "conv_wgrad_igemm"),      # res3: 16 input channels, the four-wave kernel
    (1024, 16, 8, 128, 100, 1, torch.bfloat16, "conv_wgrad_igemm"),
    (100, 64, 8, 64, 48, 1, torch.bfloat16, "conv_wgrad_igemm"),       # two slices of Cin, at most 64 output channels
    (8, 5, 12, 7, 4, 1, torch.float32, False),
    (4, 6, 10, 4, 4, 2, torch.float64, False),
]


@pytest.mark.parametrize("case", WGRAD_PAIR_CASES)
def test_weight_gradients_of_a_block_s_two_first_convolutions(gpu, case):
    """lamp_convolution_backward_weight_pair: dW of the 3x3 and of the 1x1 convolution lamp's residual block applies to its input
    (cnn.scala:16-20).  On the narrow bf16 layers with at most 16 output channels together the two output gradients share the rows of the
    matrix-core tile and x is staged once: the 3x3's gradient BITWISE that of its own launch (same products in the same order), the 1x1's
    - the centre tap of the extra rows - equal up to the order the image groups are summed in (f32 sums rounded to bf16).  The eight-wave
    implicit-GEMM kernel (8x8 maps, more than 32 input and 64 output channels) stages the second gradient beside the first and keeps its
    centre-tap product in eight more registers.  Everything else runs the two lamp_convolution_backward calls inside the entry point, bitwise."""
    N, Cin, H, Ca, Cb, stride, dt, one_launch = case
    x = closed_form((N, Cin, H, H), 3, 2.0, dt)
    wa, wb = closed_form((Ca, Cin, 3, 3), 17, 0.5, dt), closed_form((Cb, Cin, 1, 1), 19, 0.7, dt)
    ho = (H - 1) // stride + 1
    ga, gb = closed_form((N, Ca, ho, ho), 23, 1.0, dt), closed_form((N, Cb, ho, ho), 31, 1.0, dt)
    X, WA, WB, GA, GB = to_sten(x), to_sten(wa), to_sten(wb), to_sten(ga), to_sten(gb)
    sd, p1, p0, one, z = i64_array([stride, stride]), i64_array([1, 1]), i64_array([0, 0]), i64_array([1, 1]), i64_array([0, 0])

    def single(G, W, pad):
        out3 = _out3()
        lib.lamp_convolution_backward(out3, G, X, W, sd, pad, one, 2, 0, z, 1, _mask3(0, 1, 0))
        return to_torch(S.STen(out3[1]))
    da, db = single(GA, WA, p1), single(GB, WB, p0)
    lib.lamp_kernel_timer_enable(1)
    o2 = (C.c_void_p * 2)()
    lib.lamp_convolution_backward_weight_pair(o2, X, GA, WA, sd, p1, one, GB, WB, sd, p0, one, 2, 1)
    pa, pb = to_torch(S.STen(o2[0])), to_torch(S.STen(o2[1]))       # (reading them runs the deferred reductions)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    rep = buf.value.decode()
    assert pa.shape == wa.shape and pb.shape == wb.shape
    assert torch.equal(pa, da), f"3x3: max diff {(pa - da).abs().max().item()}"
    if one_launch:
        cls = "conv_wgrad_narrow" if one_launch is True else one_launch
        lines = [l for l in rep.splitlines() if l.startswith("conv_wgrad") and not l.startswith("conv_wgrad_reduce")]
        assert len(lines) == 1 and lines[0].startswith(cls) and int(lines[0].split()[1]) == 1, rep
        assert_close(pb, db.double(), 2.0 ** -7, "1x1 against its own launch")
    else:
        assert torch.equal(pb, db), f"1x1: max diff {(pb - db).abs().max().item()}"
    f = torch.float64 if dt == torch.float64 else torch.float32
    ra = aten.convolution_backward(ga.to(f), x.to(f), wa.to(f), [0], [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    rb = aten.convolution_backward(gb.to(f), x.to(f), wb.to(f), [0], [stride, stride], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    tol = {torch.float64: 1e-6, torch.float32: 2e-4, torch.bfloat16: 2e-2}[dt]
    assert_close(pa, ra.double(), tol, "3x3 against the oracle")
    assert_close(pb, rb.double(), tol, "1x1 against the oracle")


@pytest.mark.parametrize("min_ips", ["2", "3"])
def test_eight_wave_weight_gradient_dma_and_register_forms_agree_bit_for_bit(gpu, min_ips):
    """`ig_wgrad8h_kernel` brings its dY tiles in by LDS-DMA through inline asm with hand-counted `s_waitcnt vmcnt` (correct only while
    hipcc keeps its own waits where they are).  The register form of the same kernel (`LAMP_WG8H_DMA=0`, a run-time switch and the fallback) does
    the same arithmetic in the same order, so over odd batches, image ranges of one to three images, Cout < 128 and Cin off the 32-channel grid
    - alone and with the shortcut's gradient riding along - the two must agree BIT FOR BIT.  One child process per form (the switch is read once)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for dma in ("0", "1"):
        env = dict(os.environ, LAMP_WG8H_DMA=dma, LAMP_WGRAD_MIN_IPS=min_ips)
        r = subprocess.run([sys.executable, os.path.join(here, "wg8h_digest.py")], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[dma] = r.stdout.strip().splitlines()
    assert len(outs["0"]) == len(outs["1"]) >= 10
    assert outs["0"][-1] == outs["1"][-1] and "conv_wgrad_igemm" in outs["1"][-1], outs["1"][-1]
    for a, b in zip(outs["0"], outs["1"]):
        assert a == b, f"register form {a} != DMA form {b}"


def test_stride_two_input_gradient_pairs_by_row_parity(gpu, tmp_path):
    """Round 6 (`ncv_fwd2_kernel<.., PAR>`): the input gradient of a stride-2 pair is a stride-1 correlation over the zero-DILATED output gradient,
    so an output row meets non-zero rows under one parity of the filter row only; the kernel packs its K pairs by that parity and a super-tile takes
    rows of one parity - half the k-steps.  The skipped products were exact zeros: against the plain form (LAMP_NCV_DGRAD_PARITY=0) the results may
    differ only by the order of the f32 sums - at most one bf16 rounding, on few elements."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    got = {}
    for flag in ("0", "1"):
        f = str(tmp_path / f"par{flag}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "dgrad_parity_dump.py"), f], env=dict(os.environ, LAMP_NCV_DGRAD_PARITY=flag), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[flag] = np.load(f)
    for k in got["0"].files:
        a, b = got["0"][k].astype(np.float64), got["1"][k].astype(np.float64)
        assert a.shape == b.shape and np.isfinite(b).all()
        d = np.abs(a - b)
        scale = np.abs(a) + np.abs(a).mean()
        assert (d <= 2.0 ** -7 * scale).all(), f"{k}: max relative difference {(d / scale).max():.3e}"
        assert (d > 0).mean() < 0.25, f"{k}: {100 * (d > 0).mean():.1f} % of the elements differ"


def test_input_gradient_pair_checks_its_arguments(gpu):
    dt = torch.bfloat16
    x = closed_form((4, 6, 8, 8), 3, 2.0, dt)
    wa, wb = closed_form((5, 6, 3, 3), 17, 0.5, dt), closed_form((5, 6, 1, 1), 19, 0.5, dt)
    gy = closed_form((4, 5, 8, 8), 23, 1.0, dt)
    one, p1, p0 = i64_array([1, 1]), i64_array([1, 1]), i64_array([0, 0])
    o = C.c_void_p()
    with pytest.raises(Exception, match="does not match its forward output shape"):
        lib.lamp_convolution_backward_input_pair(C.byref(o), to_sten(x), to_sten(gy), to_sten(wa), one, p1, one, to_sten(x), to_sten(wb), one, p0, one, 2, 1, None)
    with pytest.raises(Exception, match="dtype mismatch"):
        lib.lamp_convolution_backward_input_pair(C.byref(o), to_sten(x), to_sten(gy), to_sten(wa), one, p1, one, to_sten(gy.float()), to_sten(wb), one, p0, one, 2, 1, None)
    with pytest.raises(Exception, match="does not have the input's shape"):
        lib.lamp_convolution_backward_input_pair(C.byref(o), to_sten(x), to_sten(gy), to_sten(wa), one, p1, one, to_sten(gy), to_sten(wb), one, p0, one, 2, 1, to_sten(gy))


def test_convolution_input_gradient_accumulate_checks_its_arguments(gpu):
    x = closed_form((4, 6, 8, 8), 3, 2.0, torch.bfloat16)
    w = closed_form((5, 6, 3, 3), 17, 0.5, torch.bfloat16)
    gy = closed_form((4, 5, 8, 8), 23, 1.0, torch.bfloat16)
    o = C.c_void_p()
    geom = (i64_array([1, 1]), i64_array([1, 1]), i64_array([1, 1]), 2, i64_array([0, 0]), 1)
    with pytest.raises(Exception, match="does not have the input's shape"):
        lib.lamp_convolution_backward_input_add(C.byref(o), to_sten(gy), to_sten(x), to_sten(w), *geom, to_sten(gy))
    with pytest.raises(Exception, match="dtype mismatch"):
        lib.lamp_convolution_backward_input_add(C.byref(o), to_sten(gy), to_sten(x), to_sten(w), *geom, to_sten(x.float()))


def test_weight_gradient_reductions_are_deferred_and_batched(gpu):
    """The reductions of the bf16 convolutions' weight-gradient partial sums are registered and run in ONE launch at
    lamp_flush_deferred (end of backprop) or when anything asks for the tensor (here: the copy to the host).  Results are those of
    ATen within bf16 tolerance and identical whether a flush came first or the access triggered it."""
    dt = torch.bfloat16
    cases = [(128, 128, 8, 3), (6, 6, 16, 3), (16, 128, 8, 1)]
    refs, got_access, got_flush = [], [], []
    for rnd in range(2):
        handles = []
        for (cin, cout, hw, k) in cases:
            x = closed_form((8, cin, hw, hw), 3, 2.0, dt)
            w = closed_form((cout, cin, k, k), 17, 1.0, dt)
            gy = closed_form((8, cout, hw, hw), 23, 1.0, dt)
            out3 = (C.c_void_p * 3)()
            mask = (C.c_uint8 * 3)(0, 1, 0)
            lib.lamp_convolution_backward(out3, to_sten(gy), to_sten(x), to_sten(w), i64_array([1, 1]), i64_array([k // 2, k // 2]), i64_array([1, 1]), 2, 0,
                                          i64_array([0, 0]), 1, mask)
            handles.append(S.STen(out3[1]))
            if rnd == 0:
                refs.append(aten.convolution_backward(gy.float(), x.float(), w.float(), [0], [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1,
                                                      [False, True, False])[1])
        if rnd == 1:
            lib.lamp_flush_deferred()
        (got_flush if rnd == 1 else got_access).extend(to_torch(h) for h in handles)
    for r, a, f in zip(refs, got_access, got_flush):
        assert_close(a, r.double(), 1.6e-2, "deferred weight gradient")
        assert torch.equal(a, f)


def test_two_layers_weight_gradients_share_one_launch(gpu):
    """Round 6: at large batches the eight-wave weight-gradient kernel takes TWO layers in one launch - a layer that qualifies is parked (its gradient
    marked pending) until the next one arrives, each then gets half the workgroups, twice the images per workgroup and leaves half the partial
    sums (igemm_wgrad_group; LAMP_WGRAD_GROUP=0: every layer at once).  The gradients are those of the f64 convolution within bf16 tolerance and
    within the summation-order noise of the separate launches; a layer that stays alone is launched by the flush or by the first read of its
    gradient; the launch count says which happened."""
    import os, subprocess, sys
    dt = torch.bfloat16
    N = 2048
    layers = [(100, 100, 7), (128, 128, 11), (128, 100, 13)]

    def wgrad(cin, cout, seed):
        x = closed_form((N, cin, 8, 8), seed, 2.0, dt)
        w = closed_form((cout, cin, 3, 3), 17, 1.0, dt)
        gy = closed_form((N, cout, 8, 8), seed + 16, 1.0, dt)
        out3 = (C.c_void_p * 3)()
        lib.lamp_convolution_backward(out3, to_sten(gy), to_sten(x), to_sten(w), i64_array([1, 1]), i64_array([1, 1]), i64_array([1, 1]), 2, 0,
                                      i64_array([0, 0]), 1, (C.c_uint8 * 3)(0, 1, 0))
        ref = aten.convolution_backward(gy.double(), x.double(), w.double(), [0], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return S.STen(out3[1]), ref

    def launches():
        buf = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_report(buf, len(buf))
        return {l.split()[0]: int(l.split()[1]) for l in buf.value.decode().splitlines() if l.strip()}

    lib.lamp_device_synchronize()
    lib.lamp_kernel_timer_enable(1)
    hs = [wgrad(*l) for l in layers]                       # layers 0 and 1 in one launch; layer 2 parked
    mid = launches().get("conv_wgrad_igemm", 0)
    got2 = to_torch(hs[2][0])                              # the read launches it alone
    after = launches()
    lib.lamp_kernel_timer_enable(0)
    assert mid == 1 and after.get("conv_wgrad_igemm", 0) == 1, (mid, after)        # (the report starts a new count)
    got = [to_torch(hs[0][0]), to_torch(hs[1][0]), got2]
    for g, (_, ref) in zip(got, hs):
        assert_close(g, ref, 2.0 ** -7, "grouped weight gradient against the f64 convolution")
    # ... and against the separate launches (another process: the switch is read once): only the order of the image ranges' f32 sums differs
    code = ("import sys, torch, ctypes as C\n"
            "from tests.test_ops_gpu import *\n"
            "from tests.test_ops_gpu import _out3\n"
            "import numpy as np\n"
            "outs = []\n"
            "for (cin, cout, seed) in %r:\n"
            "    x = closed_form((%d, cin, 8, 8), seed, 2.0, torch.bfloat16); w = closed_form((cout, cin, 3, 3), 17, 1.0, torch.bfloat16); gy = closed_form((%d, cout, 8, 8), seed + 16, 1.0, torch.bfloat16)\n"
            "    out3 = (C.c_void_p * 3)()\n"
            "    lib.lamp_convolution_backward(out3, to_sten(gy), to_sten(x), to_sten(w), i64_array([1, 1]), i64_array([1, 1]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1, (C.c_uint8 * 3)(0, 1, 0))\n"
            "    outs.append(to_torch(S.STen(out3[1])).float().numpy())\n"
            "np.savez(sys.argv[1], *outs)\n") % (layers, N, N)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "alone.npz")
        out = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=dict(os.environ, LAMP_WGRAD_GROUP="0", PYTHONPATH=root), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        alone = np.load(f)
        for i, g in enumerate(got):
            a = torch.from_numpy(alone[f"arr_{i}"]).double()
            assert_close(g, a, 2.0 ** -7, "grouped against separate launches")
            assert (g.double() - a).abs().gt(0).float().mean().item() < 0.25, "more than a quarter of the elements changed their rounding"


@pytest.mark.parametrize("cin,cout,k,N", [(128, 128, 3, 64), (64, 64, 3, 64), (32, 64, 1, 64), (128, 100, 3, 64),
                                         (128, 128, 3, 1024), (16, 16, 3, 1024), (128, 100, 1, 1024)])   # N = 1024: the eight-image kernel, one triple per workgroup
def test_batch_norm_takes_its_statistics_from_the_convolution(gpu, cin, cout, k, N):
    """A bf16 8x8 convolution on the implicit-GEMM path leaves per-image Welford triples of its output for the batch norm that
    consumes it (conv_stats_publish / conv_stats_lookup): that batch norm launches no statistics kernel, and its outputs are those
    of a batch norm run on a COPY of the tensor (different storage: no hand-off) up to the merge order of the partial statistics.
    Writing into the tensor invalidates the hand-off."""
    dt = torch.bfloat16
    x = closed_form((N, cin, 8, 8), 3, 2.0, dt)
    w = closed_form((cout, cin, k, k), 17, 0.2, dt)
    bias = closed_form((cout,), 5, 1.0, dt)
    g, b = closed_form((cout,), 1, 1.0, dt) + 1.0, closed_form((cout,), 9, 1.0, dt)
    rm, rv = closed_form((cout,), 7, 0.5, dt), closed_form((cout,), 11, 0.5, dt) + 1.0
    o = C.c_void_p()
    p_ = (k - 1) // 2
    lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(bias), i64_array([1, 1]), i64_array([p_, p_]), i64_array([1, 1]), 2, 0,
                         i64_array([0, 0]), 1)
    Y = S.STen(o)
    Ycopy = Y.clone()

    def bn(t):
        out = _out3()
        RM, RV = to_sten(rm), to_sten(rv)
        lib.lamp_kernel_timer_enable(1)
        lib.lamp_native_batch_norm_relu(out, t, to_sten(g), to_sten(b), RM, RV, 1, 0.1, 1e-5)
        buf = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_report(buf, len(buf))
        lib.lamp_kernel_timer_enable(0)
        return _wrap3(out), RM, RV, buf.value

    (y1, m1, i1), RM1, RV1, rep1 = bn(Y)
    (y2, m2, i2), RM2, RV2, rep2 = bn(Ycopy)
    assert b"bn_fwd_stats" not in rep1, "the statistics pass ran although the convolution had published them"
    assert b"bn_fwd_stats" in rep2
    assert_close(to_torch(m1), to_torch(m2), 1e-2, "save_mean")          # bf16-rounded values: equal up to one rounding step
    assert_close(to_torch(i1), to_torch(i2), 1e-2, "save_invstd")
    assert_close(to_torch(y1), to_torch(y2), 2e-2, "normalised output")
    assert_close(to_torch(RV1), to_torch(RV2), 1e-2, "running_var")
    ref = torch.relu(aten.native_batch_norm(to_torch(Ycopy).to(dt), g, b, rm.clone(), rv.clone(), True, 0.1, 1e-5)[0])
    assert_close(to_torch(y1), ref.double(), FWD_TOL[dt] * 4, "against the oracle")
    half = Y.slice(0, 0, N // 2)                                             # a batch slice shares storage, offset and version: no hand-off
    _, _, _, rep_half = bn(half)
    assert b"bn_fwd_stats" in rep_half, "the statistics of the whole batch were used for a slice of it"
    lib.lamp_mul_(Y, Y.onesLike())                                           # any write through the handle bumps the storage version
    _, _, _, rep3 = bn(Y)
    assert b"bn_fwd_stats" in rep3, "stale statistics were used after the tensor had been written"


@pytest.mark.parametrize("cin,cout,k,H,stride,N,offset", [(3, 6, 5, 32, 1, 64, 0), (6, 6, 3, 32, 2, 67, 0), (6, 6, 1, 32, 2, 64, 0),
                                                          (6, 6, 3, 16, 1, 2050, 0), (6, 6, 3, 16, 1, 2048, 0), (6, 16, 3, 16, 2, 33, 0),
                                                          (6, 16, 1, 16, 2, 1100, 0), (8, 5, 3, 32, 1, 9, 0), (3, 6, 5, 32, 1, 2, 0),
                                                          (6, 6, 3, 32, 2, 64, 40.0), (6, 16, 3, 16, 2, 2048, -25.0)])
def test_batch_norm_takes_its_statistics_from_the_narrow_convolution(gpu, cin, cout, k, H, stride, N, offset):
    """The narrow bf16 layers of Cnn.resnet (stem 3 -> 6 5x5, res1 / res2: 6 and 16 channels on 32x32 / 16x16 maps, cnn.scala:89-131) run on
    ncv_fwd2_kernel, whose epilogue leaves one Welford triple per image and channel over the values it stores: the batch norm behind such a
    convolution launches no statistics pass and returns what a batch norm of a COPY of the tensor returns, up to the merge order (ragged
    batches, more images than workgroups - one triple per workgroup when they divide evenly -, the smallest batch that has statistics,
    and outputs whose mean is tens of standard deviations away from zero: the kernel's sums are not shifted)."""
    dt = torch.bfloat16
    x = closed_form((N, cin, H, H), 3, 2.0, dt)
    w = closed_form((cout, cin, k, k), 17, 0.3 if not offset else 0.05, dt)
    bias = closed_form((cout,), 5, 1.0, dt) + offset
    g, b = closed_form((cout,), 1, 1.0, dt) + 1.0, closed_form((cout,), 9, 1.0, dt)
    rm, rv = closed_form((cout,), 7, 0.5, dt), closed_form((cout,), 11, 0.5, dt) + 1.0
    o = C.c_void_p()
    p_ = (k - 1) // 2
    lib.lamp_kernel_timer_enable(1)
    lib.lamp_convolution(C.byref(o), to_sten(x), to_sten(w), to_sten(bias), i64_array([stride, stride]), i64_array([p_, p_]), i64_array([1, 1]), 2, 0,
                         i64_array([0, 0]), 1)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    assert b"conv_fwd_narrow" in buf.value, "this geometry is meant to run on the narrow kernel"
    Y = S.STen(o)
    ref_y = aten.convolution(x, w, bias, [stride, stride], [p_, p_], [1, 1], False, [0, 0], 1)
    assert_close(to_torch(Y), ref_y.double(), FWD_TOL[dt] * 4, "the convolution itself")
    Ycopy = Y.clone()

    def bn(t):
        out = _out3()
        RM, RV = to_sten(rm), to_sten(rv)
        lib.lamp_kernel_timer_enable(1)
        lib.lamp_native_batch_norm_relu(out, t, to_sten(g), to_sten(b), RM, RV, 1, 0.1, 1e-5)
        rep = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_report(rep, len(rep))
        lib.lamp_kernel_timer_enable(0)
        return _wrap3(out), RM, RV, rep.value

    (y1, m1, i1), RM1, RV1, rep1 = bn(Y)
    (y2, m2, i2), RM2, RV2, rep2 = bn(Ycopy)
    assert b"bn_fwd_stats" not in rep1, "the statistics pass ran although the convolution had published them"
    assert b"bn_fwd_stats" in rep2
    # the triples against the tensor itself: mean and variance per channel in f64
    yt = to_torch(Ycopy).double()
    assert_close(to_torch(m1), yt.mean(dim=(0, 2, 3)), 1e-2, "save_mean against the tensor")
    assert_close(to_torch(i1), 1.0 / torch.sqrt(yt.var(dim=(0, 2, 3), unbiased=False) + 1e-5), 1e-2, "save_invstd against the tensor")
    assert_close(to_torch(m1), to_torch(m2), 1e-2, "save_mean")          # bf16-rounded values: equal up to one rounding step
    assert_close(to_torch(i1), to_torch(i2), 1e-2, "save_invstd")
    assert_close(to_torch(y1), to_torch(y2), 2e-2, "normalised output")
    assert_close(to_torch(RM1), to_torch(RM2), 1e-2, "running_mean")
    assert_close(to_torch(RV1), to_torch(RV2), 1e-2, "running_var")
    lib.lamp_mul_(Y, Y.onesLike())                                           # any write through the handle bumps the storage version
    _, _, _, rep3 = bn(Y)
    assert b"bn_fwd_stats" in rep3, "stale statistics were used after the tensor had been written"


@pytest.mark.parametrize("cin,cout,N,H,stride", [(128, 100, 1024, 8, 1), (16, 128, 1024, 8, 1), (128, 128, 1032, 8, 1), (100, 100, 1024, 8, 1),
                                                 (64, 64, 1024, 8, 1), (16, 16, 1027, 8, 1), (128, 100, 64, 8, 1), (6, 6, 64, 8, 1),
                                                 (16, 128, 33, 8, 1), (128, 128, 600, 8, 1), (64, 64, 64, 8, 1),
                                                 (6, 6, 67, 32, 2), (3, 8, 16, 32, 2), (6, 5, 9, 32, 1), (6, 16, 33, 16, 2)])
def test_convolution_pair_is_the_two_convolutions_in_one_launch(gpu, cin, cout, N, H, stride):
    """lamp_convolution_pair(x, 3x3, 1x1) - the two branches of lamp's residual block start with a Conv2D on the same input (cnn.scala:16-20,
    38-78) - returns BITWISE what two lamp_convolution calls return, hands each output's batch-norm statistics to its consumer as the single
    convolutions do, and is ONE launch of the eight-image kernel where that kernel takes the geometry (N >= 4 x CUs, bf16, 8x8 maps) or of the
    narrow kernel where both filters' output channels fit the MFMA's 16 columns (res1 of Cnn.resnet: 6 + 6 channels, 32 x 32, stride 2); every
    other geometry runs the two calls inside the entry point."""
    dt = torch.bfloat16
    x = closed_form((N, cin, H, H), 3, 2.0, dt)
    wa, ba = closed_form((cout, cin, 3, 3), 17, 0.2, dt), closed_form((cout,), 5, 1.0, dt)
    wb, bb = closed_form((cout, cin, 1, 1), 23, 0.4, dt), closed_form((cout,), 13, 1.0, dt)
    X, WA, BA, WB, BB = to_sten(x), to_sten(wa), to_sten(ba), to_sten(wb), to_sten(bb)
    one, p1, p0, z, sd = i64_array([1, 1]), i64_array([1, 1]), i64_array([0, 0]), i64_array([0, 0]), i64_array([stride, stride])

    def single(W, B, pad):
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), X, W, B, sd, pad, one, 2, 0, z, 1)
        return S.STen(o)
    ya, yb = single(WA, BA, p1), single(WB, BB, p0)
    o2 = (C.c_void_p * 2)()
    lib.lamp_kernel_timer_enable(1)
    lib.lamp_convolution_pair(o2, X, WA, BA, sd, p1, one, WB, BB, sd, p0, one, 2, 1)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    pa, pb = S.STen(o2[0]), S.STen(o2[1])
    assert torch.equal(to_torch(pa), to_torch(ya)), "3x3 output of the pair differs from the single convolution"
    assert torch.equal(to_torch(pb), to_torch(yb)), "1x1 output of the pair differs from the single convolution"
    launches = {ln.split()[0]: int(ln.split()[1]) for ln in buf.value.decode().splitlines() if ln.strip()}       # "tag count total_ms flops bytes"
    # the eight-image kernel / (round 5) the two-image kernel with 128-row weight stages at small batches / the narrow kernel's spare columns
    fused = (H == 8 and cin >= 8 and (N >= 1024 or cout > 64)) or (H == 32 and 2 * cout <= 16)
    assert sum(n for t, n in launches.items() if t.startswith("conv_")) == (1 if fused else 2), (fused, launches)
    # against the oracle (ATen f32 on the same bf16 values)
    ra = aten.convolution(x.float(), wa.float(), ba.float(), [stride, stride], [1, 1], [1, 1], False, [0, 0], 1)
    rb = aten.convolution(x.float(), wb.float(), bb.float(), [stride, stride], [0, 0], [1, 1], False, [0, 0], 1)
    assert_close(to_torch(pa), ra.double(), FWD_TOL[dt] * 4, "3x3 against the oracle")
    assert_close(to_torch(pb), rb.double(), FWD_TOL[dt] * 4, "1x1 against the oracle")
    # the statistics hand-off of BOTH outputs: the batch norm that reads them launches no statistics pass and returns the bits it returns
    # for the single convolutions' outputs
    g, b = closed_form((cout,), 1, 1.0, dt) + 1.0, closed_form((cout,), 9, 1.0, dt)

    def bn(t):
        out = _out3()
        RM, RV = to_sten(torch.zeros(cout, dtype=dt)), to_sten(torch.ones(cout, dtype=dt))
        lib.lamp_kernel_timer_enable(1)
        lib.lamp_native_batch_norm_relu(out, t, to_sten(g), to_sten(b), RM, RV, 1, 0.1, 1e-5)
        rb_ = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_report(rb_, len(rb_))
        lib.lamp_kernel_timer_enable(0)
        return [to_torch(h) for h in _wrap3(out)] + [to_torch(RM), to_torch(RV)], rb_.value
    for got, want, name in ((pa, ya, "3x3"), (pb, yb, "1x1")):
        (r1, rep1), (r2, rep2) = bn(got), bn(want)
        assert (b"bn_fwd_stats" in rep1) == (b"bn_fwd_stats" in rep2), f"{name}: the pair's output carries a different hand-off"
        if (cin >= 8 and H == 8) or (H > 8 and N >= 2):
            assert b"bn_fwd_stats" not in rep1, f"{name}: no statistics were handed over"
        for u, v in zip(r1, r2):
            if H == 8:
                assert torch.equal(u, v), f"{name}: batch norm of the pair's output differs from batch norm of the single convolution's"
            else:
                # the narrow kernel deals a pair's 12 - 16 output columns to its lanes differently from a single filter's 6 - 8 (one window
                # phase per MFMA instead of two): the same values are summed in another order
                assert_close(u, v.double(), 2e-2, f"{name}: batch norm of the pair's output against that of the single convolution's")


def _conv_wgrad_bf16(x, w, gy, k):
    out3 = (C.c_void_p * 3)()
    mask = (C.c_uint8 * 3)(0, 1, 0)
    lib.lamp_convolution_backward(out3, gy, x, w, i64_array([1, 1]), i64_array([k // 2, k // 2]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1, mask)
    return S.STen(out3[1])


def test_deferred_reductions_and_graph_capture(gpu):
    """(a) a bf16 convolution backward issued through the C ABI INSIDE a capture whose dW nobody touches before the capture
    ends must still be part of the graph (lamp_graph_end_capture queues the registered reductions on the capturing stream): every
    replay produces dW for the batch then in the input buffer; (b) reductions registered BEFORE lamp_graph_begin_capture run eagerly
    at its start instead of being recorded by the first pointer access inside the capture."""
    dt = torch.bfloat16
    cin, cout, hw, k = 128, 128, 8, 3
    xs = [closed_form((16, cin, hw, hw), 3 + 7 * i, 2.0, dt) for i in range(3)]
    w = closed_form((cout, cin, k, k), 17, 1.0, dt)
    gy = closed_form((16, cout, hw, hw), 23, 1.0, dt)
    W, GY = to_sten(w), to_sten(gy)
    eager = [to_torch(_conv_wgrad_bf16(to_sten(x), W, GY, k)) for x in xs]
    st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(st))
    dflt = C.c_void_p(); lib.lamp_stream_get_default(0, C.byref(dflt))
    lib.lamp_device_synchronize()
    lib.lamp_stream_set_current(st)
    try:
        x_buf = to_sten(xs[0])
        _conv_wgrad_bf16(x_buf, W, GY, k).to_numpy()            # warm-up on this stream (attributes, caches)
        before = _conv_wgrad_bf16(to_sten(xs[1]), W, GY, k)     # (b) registered, not yet reduced, when the capture begins
        lib.lamp_graph_begin_capture()
        dw = _conv_wgrad_bf16(x_buf, W, GY, k)                  # (a) never touched inside the capture
        g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
        lib.lamp_stream_synchronize(st)
        assert torch.equal(to_torch(before), eager[1]), "a reduction registered before the capture must have run eagerly"
        for i in (2, 0, 1):
            x_buf.copyFrom(to_sten(xs[i]))
            lib.lamp_graph_launch(g)
            lib.lamp_stream_synchronize(st)
            assert torch.equal(to_torch(dw), eager[i]), f"replay {i}: dW is not the eager result - the reduction is not in the graph"
        lib.lamp_graph_release(g)
    finally:
        lib.lamp_stream_set_current(dflt)
    lib.lamp_device_synchronize()


def test_deferred_reductions_belong_to_the_thread_that_registered_them(gpu):
    """the replica threads of the single-process data-parallel step finish backprop at about the same time; one
    thread's flush must not take (and clear the flags of) another thread's reductions.  Four threads, each on its own stream, run
    bf16 weight gradients + flush + an in-place scaling many times; every result must equal the single-threaded one bit for bit."""
    import threading
    dt = torch.bfloat16
    cin, cout, hw, k = 128, 128, 8, 3
    w = closed_form((cout, cin, k, k), 17, 1.0, dt)
    gy = closed_form((16, cout, hw, hw), 23, 1.0, dt)
    xs = [closed_form((16, cin, hw, hw), 3 + 11 * i, 2.0, dt) for i in range(4)]
    W, GY = to_sten(w), to_sten(gy)
    XS = [to_sten(x) for x in xs]
    want = []
    for X in XS:
        d = _conv_wgrad_bf16(X, W, GY, k)
        lib.lamp_flush_deferred()
        lib.lamp_mul_scalar_(d, 0.5)
        want.append(to_torch(d))
    lib.lamp_device_synchronize()
    errors = []
    barrier = threading.Barrier(4)

    def work(i):
        try:
            st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(st)); lib.lamp_stream_set_current(st)
            for rep in range(40):
                barrier.wait()
                d = _conv_wgrad_bf16(XS[i], W, GY, k)
                if rep % 2 == 0:
                    lib.lamp_flush_deferred()               # the natural batching point ...
                lib.lamp_mul_scalar_(d, 0.5)                # ... or the first pointer access resolves it
                got = to_torch(d)
                if not torch.equal(got, want[i]):
                    errors.append(f"thread {i} repetition {rep}: weight gradient differs (max |d| {(got - want[i]).abs().max().item()})")
                    break
        except Exception as e:                             # noqa: BLE001
            errors.append(f"thread {i}: {e}")
            barrier.abort()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    lib.lamp_device_synchronize()
    assert not errors, errors


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float64])
@pytest.mark.parametrize("held", [32, 64, 128, 224])
def test_one_pass_batch_norm_backward_under_cu_pressure(gpu, held, dt):
    """the one-pass batch-norm backward (workgroups of a channel wait for each other's partial sums) on a device
    it does NOT have to itself.  `held` single-wave workgroups spin for 3 ms on a high-priority second stream - with 512-thread,
    register-file-filling batch-norm workgroups that is `held` CUs taken away - while the six large maps of the ResNet step run their
    backward on the compute stream.  The kernel has to finish (workgroups are handed out in launch order, so the channels complete
    one after the other as CUs free up) with bitwise the results it gives on an idle device, which in turn agree with the two-kernel
    form; marked shared through lamp_device_shared_hint the host takes the two kernels by itself.  f32 / f64 (round 5): the chunked launches
    of bn_bwd_fused_fp_kernel size their chunks for an idle chip - under pressure a chunk is NOT fully co-resident and completes channel
    by channel all the same."""
    maps = [(2048, 6, 32, 1), (2048, 16, 16, 2), (2048, 128, 8, 2), (2048, 128, 8, 1), (2048, 100, 8, 0), (2048, 16, 16, 1)]
    hi = C.c_void_p(); lib.lamp_stream_get_from_pool(1, 0, C.byref(hi))
    tensors = []
    for (N, Cc, H, variant) in maps:
        shape = (N, Cc, H, H)
        x = closed_form(shape, 3, 4.0, dt) + 0.3
        ad = closed_form(shape, 29, 3.0, dt)
        gy = closed_form(shape, 11, 2.0, dt)
        w, b = closed_form((Cc,), 1, 1.0, dt) + 1.0, closed_form((Cc,), 5, 1.0, dt)
        rm, rv = closed_form((Cc,), 7, 0.5, dt), closed_form((Cc,), 9, 0.5, dt) + 1.0
        X, AD, GY, Wt, Bt, RM, RV = (to_sten(t) for t in (x, ad, gy, w, b, rm, rv))
        fwd = _out3()
        lib.lamp_native_batch_norm(fwd, X, Wt, Bt, RM, RV, 1, 0.1, 1e-5)
        _, sm, si = _wrap3(fwd)
        tensors.append((variant, X, AD, GY, Wt, Bt, RM, RV, sm, si))

    def backward_all():
        outs = []
        for (variant, X, AD, GY, Wt, Bt, RM, RV, sm, si) in tensors:
            if variant == 2:
                out4 = (C.c_void_p * 4)()
                lib.lamp_native_batch_norm_add_relu_backward(out4, GY, X, AD, Wt, Bt, RM, RV, sm, si, 1, 1e-5, (C.c_uint8 * 4)(1, 1, 1, 1))
                outs.extend(S.STen(out4[i]) for i in range(4))
            else:
                out = _out3()
                if variant == 1:
                    lib.lamp_native_batch_norm_relu_backward(out, GY, X, Wt, Bt, RM, RV, sm, si, 1, 1e-5, _mask3(1, 1, 1))
                else:
                    lib.lamp_native_batch_norm_backward(out, GY, X, Wt, RM, RV, sm, si, 1, 1e-5, _mask3(1, 1, 1))
                outs.extend(_wrap3(out))
        return outs

    def classes(fn):
        lib.lamp_kernel_timer_enable(1)
        r = fn()
        buf = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_report(buf, len(buf))
        lib.lamp_kernel_timer_enable(0)
        return r, buf.value

    try:
        lib.lamp_bn_backward_mode(0)
        two, rep = classes(backward_all)
        assert b"bn_bwd_fused" not in rep
        two = [t.to_numpy() for t in two]
        lib.lamp_bn_backward_mode(1)
        alone, rep = classes(backward_all)
        assert rep.count(b"bn_bwd_fused") == 1 and b"bn_bwd_reduce" not in rep, rep.decode()
        alone = [t.to_numpy() for t in alone]
        lib.lamp_device_synchronize()
        for rnd in range(3):
            lib.lamp_debug_occupy_cus(held, 3000.0, hi)            # 3 ms on the high-priority stream, then the backward passes beside it
            pressed = backward_all()
            lib.lamp_device_synchronize()                          # raises if a workgroup gave up waiting (device-side assertion)
            for a, p in zip(alone, pressed):
                assert np.array_equal(a, p.to_numpy()), f"round {rnd}: results under CU pressure differ from the idle device's"
        # the two forms agree (different summation order of the channel sums: bf16 tolerance on dweight / dbias, dx follows them)
        for a, t in zip(alone, two):
            assert_close(torch.from_numpy(a), torch.from_numpy(t).double(), {torch.bfloat16: 4e-2, torch.float32: 2e-4, torch.float64: 1e-9}[dt],
                         "one-pass vs two-kernel batch-norm backward")
        # marked shared: the default rule takes the two kernels (what the eager data-parallel step does while its all-reduce is in flight)
        lib.lamp_bn_backward_mode(-1)
        lib.lamp_device_shared_hint(0, 1)
        try:
            shared, rep = classes(backward_all)
            assert b"bn_bwd_fused" not in rep and b"bn_bwd_reduce" in rep, rep.decode()
            for a, t in zip(two, shared):
                assert np.array_equal(a, t.to_numpy())
            lib.lamp_bn_backward_mode(2)                           # explicit override: one pass although shared
            _, rep = classes(backward_all)
            assert b"bn_bwd_fused" in rep
        finally:
            lib.lamp_device_shared_hint(0, -1)
    finally:
        lib.lamp_bn_backward_mode(-1)
        lib.lamp_device_synchronize()


# (shape, which backward form serves it): small maps / f32 / f64 take the chain inside the entry point (bitwise), large bf16 maps the
# one-pass dual kernel (different summation order of the channel sums: bf16 tolerance)
BN_PAIR_SHAPES = [(6, 5, 8, 8), (4, 16, 16, 16), (3, 7, 9, 9), (2048, 128, 8, 8), (2048, 6, 16, 16), (2049, 16, 8, 8), (300, 100, 8, 8), (2100, 5, 20, 20)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", BN_PAIR_SHAPES)
def test_batch_norm_pair_add_relu_equals_the_chain(gpu, dt, shape):
    """lamp_native_batch_norm2_add_relu(+_backward) - relu(bn(x) + bn2(x2)), the tail of every residual block of Cnn.resnet - against the
    chain it replaces: native_batch_norm(x2) -> native_batch_norm_add_relu(x, that) and their two backward calls.  Forward: BITWISE (y,
    both pairs of saved statistics, both pairs of running statistics).  Backward: bitwise where the entry point runs the chain itself
    (f32 / f64), within bf16 rounding of the channel sums where the one-pass kernel serves it (bf16; asserted through the kernel
    timers), and against ATen in f32 on the same inputs."""
    if shape[0] >= 300 and dt != torch.bfloat16:
        pytest.skip("the large shapes are the bf16 one-pass cases")
    x = closed_form(shape, 3, 4.0, dt) + 0.3
    x2 = closed_form(shape, 31, 3.0, dt) - 0.2
    Cc = shape[1]
    w, b = closed_form((Cc,), 1, 1.0, dt) + 1.0, closed_form((Cc,), 5, 1.0, dt)
    w2, b2 = closed_form((Cc,), 13, 1.0, dt) + 0.8, closed_form((Cc,), 17, 1.0, dt)
    rm, rv = closed_form((Cc,), 7, 0.5, dt), closed_form((Cc,), 9, 0.5, dt) + 1.0
    rm2, rv2 = closed_form((Cc,), 19, 0.5, dt), closed_form((Cc,), 23, 0.5, dt) + 1.0
    X, X2, Wt, Bt, W2, B2 = (to_sten(t) for t in (x, x2, w, b, w2, b2))
    RMc, RVc, RM2c, RV2c = (to_sten(t) for t in (rm, rv, rm2, rv2))      # the chain's running statistics
    RMf, RVf, RM2f, RV2f = (to_sten(t) for t in (rm, rv, rm2, rv2))      # the fused op's
    # the chain
    o_l = _out3()
    lib.lamp_native_batch_norm(o_l, X2, W2, B2, RM2c, RV2c, 1, 0.1, 1e-5)
    l, sm2c, si2c = _wrap3(o_l)
    o_c = _out3()
    lib.lamp_native_batch_norm_add_relu(o_c, X, l, Wt, Bt, RMc, RVc, 1, 0.1, 1e-5)
    yc, smc, sic = _wrap3(o_c)
    # one op
    o5 = (C.c_void_p * 5)()
    lib.lamp_native_batch_norm2_add_relu(o5, X, Wt, Bt, RMf, RVf, X2, W2, B2, RM2f, RV2f, 0.1, 0.1, 1e-5, 1e-5)
    yf, smf, sif, sm2f, si2f = (S.STen(o5[i]) for i in range(5))
    for a, c_, what in ((yf, yc, "y"), (smf, smc, "save_mean"), (sif, sic, "save_invstd"), (sm2f, sm2c, "save_mean2"), (si2f, si2c, "save_invstd2"),
                        (RMf, RMc, "running_mean"), (RVf, RVc, "running_var"), (RM2f, RM2c, "running_mean2"), (RV2f, RV2c, "running_var2")):
        assert np.array_equal(a.to_numpy(), c_.to_numpy()), what
    ref_l = aten.native_batch_norm(x2.float(), w2.float(), b2.float(), None, None, True, 0.1, 1e-5)[0]
    ref_y = torch.relu(aten.native_batch_norm(x.float(), w.float(), b.float(), None, None, True, 0.1, 1e-5)[0] + ref_l)
    assert_close(to_torch(yf), ref_y.double(), FWD_TOL[dt] * 6 if dt != torch.float64 else 1e-5, "relu(bn + bn2)", scale="max")
    # backward
    gy = closed_form(shape, 11, 2.0, dt)
    GY = to_sten(gy)
    o4 = (C.c_void_p * 4)()
    lib.lamp_native_batch_norm_add_relu_backward(o4, GY, X, l, Wt, Bt, RMc, RVc, smc, sic, 1, 1e-5, (C.c_uint8 * 4)(1, 1, 1, 1))
    dxc, dwc, dbc, dlc = (S.STen(o4[i]) for i in range(4))
    o3 = _out3()
    lib.lamp_native_batch_norm_backward(o3, dlc, X2, W2, RM2c, RV2c, sm2c, si2c, 1, 1e-5, _mask3(1, 1, 1))
    dx2c, dw2c, db2c = _wrap3(o3)
    chain = [dxc, dwc, dbc, dx2c, dw2c, db2c]
    lib.lamp_kernel_timer_enable(1)
    o6 = (C.c_void_p * 6)()
    lib.lamp_native_batch_norm2_add_relu_backward(o6, GY, X, Wt, Bt, smf, sif, X2, W2, B2, sm2f, si2f, 1e-5, 1e-5, (C.c_uint8 * 6)(1, 1, 1, 1, 1, 1))
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    fused = [S.STen(o6[i]) for i in range(6)]
    names = ("dx", "dweight", "dbias", "dx2", "dweight2", "dbias2")
    one_pass = b"bn_bwd_fused" in buf.value
    if dt == torch.bfloat16 and shape[2] % 8 == 0 and shape[0] >= 300:
        assert one_pass and b"bn_bwd_reduce" not in buf.value, buf.value.decode()     # ONE launch for both batch norms
    if one_pass:
        for f, c_, what in zip(fused, chain, names):
            assert_close(to_torch(f), to_torch(c_).double(), 4e-2, f"one-pass {what} vs the chain", scale="max")
        # repeated launches: bitwise the same (the slots return to rest; nothing is accumulated atomically)
        first = [f.to_numpy() for f in fused]
        for _ in range(3):
            o6b = (C.c_void_p * 6)()
            lib.lamp_native_batch_norm2_add_relu_backward(o6b, GY, X, Wt, Bt, smf, sif, X2, W2, B2, sm2f, si2f, 1e-5, 1e-5, (C.c_uint8 * 6)(1, 1, 1, 1, 1, 1))
            for a, h in zip(first, o6b):
                assert np.array_equal(a, S.STen(h).to_numpy())
    else:
        for f, c_, what in zip(fused, chain, names):
            assert np.array_equal(f.to_numpy(), c_.to_numpy()), what
    # ATen in f32 on the same (rounded) inputs: mask from the rounded pre-activation
    xf, x2f, g = x.float(), x2.float(), gy.float()
    mean, invstd = to_torch(smf).float(), to_torch(sif).float()
    mean2, invstd2 = to_torch(sm2f).float(), to_torch(si2f).float()
    view = (1, -1, 1, 1)
    lr = ((x2f - mean2.view(view)) * (invstd2 * w2.float()).view(view) + b2.float().view(view)).to(dt)
    pre = ((xf - mean.view(view)) * (invstd * w.float()).view(view) + b.float().view(view)).to(dt)
    pre = (pre.float() + lr.float()).to(dt)
    g = torch.where(pre.float() < 0, torch.zeros_like(g), g)
    r1 = aten.native_batch_norm_backward(g, xf, w.float(), None, None, mean, invstd, True, 1e-5, [True, True, True])
    r2 = aten.native_batch_norm_backward(g, x2f, w2.float(), None, None, mean2, invstd2, True, 1e-5, [True, True, True])
    tol = {torch.float64: 1e-4, torch.float32: 1e-3, torch.bfloat16: 4e-2}[dt]     # (the f64 reference above is computed in f32)
    for f, r, what in zip(fused, list(r1) + list(r2), names):
        assert_close(to_torch(f), r.double(), tol, f"{what} vs ATen", scale="max")
    # a subset of the gradients only
    o6c = (C.c_void_p * 6)()
    lib.lamp_native_batch_norm2_add_relu_backward(o6c, GY, X, Wt, Bt, smf, sif, X2, W2, B2, sm2f, si2f, 1e-5, 1e-5, (C.c_uint8 * 6)(0, 1, 0, 1, 0, 1))
    assert [bool(h) for h in o6c] == [False, True, False, True, False, True]
    for i in (1, 3, 5):
        assert np.array_equal(S.STen(o6c[i]).to_numpy(), fused[i].to_numpy()), names[i]


# ---- the batch norm + relu between two convolutions, folded into the second convolution (round 3) ---------------------------------------
# (N, Cin, H, W, Cout, k): the first two are the geometries whose kernels fold the table (eight-image fprop, eight-wave wgrad); the others
# take the materialising path of the same entry points
BN_CONV_SHAPES = [(1100, 128, 8, 8, 128, 3), (1030, 100, 8, 8, 100, 3), (1100, 40, 8, 8, 72, 3), (6, 5, 8, 8, 7, 3), (3, 16, 16, 16, 8, 1), (40, 128, 8, 8, 128, 3),
                  (1100, 64, 8, 8, 64, 3), (5, 6, 4, 4, 3, 3)]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", BN_CONV_SHAPES)
def test_convolution_of_batch_norm_relu_equals_the_chain(gpu, dt, shape):
    """lamp_batch_norm_affine + lamp_convolution_bn_relu_input(+_backward) - the middle of lamp's residual block (cnn.scala:38-60) with the
    batch norm applied while the convolution stages its input - against the chain it replaces: native_batch_norm_relu -> convolution and
    convolution_backward -> native_batch_norm_relu_backward.  Statistics, running statistics and the convolution's output: BITWISE.  The
    input gradient and the bias gradient: bitwise (same kernels); the weight gradient: bitwise too (the same kernel multiplies the same
    rebuilt operand)."""
    N, Ci, H, W, Co, k = shape
    if N >= 1000 and dt != torch.bfloat16:
        pytest.skip("the large shapes are the bf16 folding cases")
    p = k // 2
    x = closed_form((N, Ci, H, W), 3, 4.0, dt) + 0.3
    g, b = closed_form((Ci,), 1, 1.0, dt) + 1.0, closed_form((Ci,), 5, 1.0, dt)
    rm, rv = closed_form((Ci,), 7, 0.5, dt), closed_form((Ci,), 9, 0.5, dt) + 1.0
    w = closed_form((Co, Ci, k, k), 17, 0.5, dt)
    cb = closed_form((Co,), 29, 1.0, dt)
    X, G, B, Wt, CB = (to_sten(t) for t in (x, g, b, w, cb))
    RMc, RVc, RMf, RVf = (to_sten(t) for t in (rm, rv, rm, rv))
    one, pad = i64_array([1, 1]), i64_array([p, p])
    # the chain
    o3 = _out3()
    lib.lamp_native_batch_norm_relu(o3, X, G, B, RMc, RVc, 1, 0.1, 1e-5)
    act, smc, sic = _wrap3(o3)
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), act, Wt, CB, one, pad, one, 2, 0, i64_array([0, 0]), 1)
    yc = S.STen(o)
    # folded
    folds = C.c_int(-1)
    lib.lamp_convolution_bn_relu_input_folds(C.byref(folds), X, Wt, one, pad, one, 2, 1)
    expect_fold = dt == torch.bfloat16 and N >= 1024 and Ci > 32 and Co > 64 and k == 3
    assert folds.value == int(expect_fold)
    a3 = _out3()
    lib.lamp_batch_norm_affine(a3, X, G, B, RMf, RVf, 0.1, 1e-5)
    aff, smf, sif = _wrap3(a3)
    assert aff.shape == [Ci, 4] or tuple(aff.shape) == (Ci, 4)
    lib.lamp_kernel_timer_enable(1)
    o = C.c_void_p()
    lib.lamp_convolution_bn_relu_input(C.byref(o), X, aff, Wt, CB, one, pad, one, 2, 1)
    yf = S.STen(o)
    for a, c_, what in ((yf, yc, "y"), (smf, smc, "save_mean"), (sif, sic, "save_invstd"), (RMf, RMc, "running_mean"), (RVf, RVc, "running_var")):
        assert np.array_equal(a.to_numpy(), c_.to_numpy()), what
    # the table is (mean, invstd * weight, bias) of the saved (rounded) statistics
    t = to_torch(aff).float()
    assert torch.equal(t[:, 0], to_torch(smf).float()) and torch.equal(t[:, 1], to_torch(sif).float() * g.float()) and torch.equal(t[:, 2], b.float())
    # backward
    gy = closed_form((N, Co, H, W), 11, 2.0, dt)
    GY = to_sten(gy)
    c3 = _out3()
    lib.lamp_convolution_backward(c3, GY, act, Wt, one, pad, one, 2, 0, i64_array([0, 0]), 1, _mask3(1, 1, 1))
    dactc, dwc, dcbc = _wrap3(c3)
    f3 = _out3()
    lib.lamp_convolution_bn_relu_input_backward(f3, GY, X, aff, Wt, one, pad, one, 2, 1, _mask3(1, 1, 1))
    dactf, dwf, dcbf = _wrap3(f3)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    for a, c_, what in ((dactf, dactc, "d activation"), (dwf, dwc, "dweight"), (dcbf, dcbc, "dbias")):
        assert np.array_equal(a.to_numpy(), c_.to_numpy()), what
    # a subset of the gradients
    f3b = _out3()
    lib.lamp_convolution_bn_relu_input_backward(f3b, GY, X, aff, Wt, one, pad, one, 2, 1, _mask3(0, 1, 0))
    assert [bool(h) for h in f3b] == [False, True, False]
    assert np.array_equal(S.STen(f3b[1]).to_numpy(), dwf.to_numpy())
    # and against ATen in f64 on the same rounded operands
    view = (1, -1, 1, 1)
    pre = torch.relu(torch.addcmul(b.float().view(view), x.float() - to_torch(smf).float().view(view), (to_torch(sif).float() * g.float()).view(view)).to(dt))
    ref = aten.convolution(pre.double(), w.double(), cb.double(), [1, 1], [p, p], [1, 1], False, [0, 0], 1)
    assert_close(to_torch(yf), ref, {torch.float32: 1e-4, torch.bfloat16: 2.0 ** -6}[dt], "conv(relu(bn(x))) vs ATen", scale="max")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float64])
def test_cat_and_chunk_in_one_launch(gpu, dt):
    """cat of contiguous blocks (16-byte multiples: one launch, `col_blocks_kernel`; anything else: a copy per input) and its inverse
    lamp_chunk_contiguous, along every dimension, against torch."""
    for shape, dim, n in (((768, 768), 1, 3), ((5, 8, 16), 2, 2), ((5, 8, 16), 1, 4), ((6, 24), 0, 3), ((7, 5), 1, 2), ((3, 8), 1, 8)):
        parts = [closed_form(shape, 3 + 5 * i, 2.0, dt) for i in range(n)]
        hs = [to_sten(p) for p in parts]
        arr = (C.c_void_p * n)(*[h.h for h in hs])
        o = C.c_void_p()
        lib.lamp_cat(C.byref(o), arr, n, dim)
        whole = S.STen(o)
        ref = torch.cat(parts, dim)
        assert torch.equal(to_torch(whole), ref.double()), (shape, dim, n)
        outs = (C.c_void_p * n)()
        lib.lamp_chunk_contiguous(outs, whole, n, dim)
        for i in range(n):
            assert torch.equal(to_torch(S.STen(outs[i])), parts[i].double()), (shape, dim, n, i)
    with pytest.raises(Exception, match="equal chunks"):
        lib.lamp_chunk_contiguous((C.c_void_p * 2)(), to_sten(closed_form((3, 5), 1, 1.0, dt)), 2, 1)


# ---- the sorting / overwriting-scatter / triangle names of the STen surface --------------------------------------------
def _out1():
    return C.c_void_p()


@pytest.mark.parametrize("shape,dim", [((7,), 0), ((5, 33), 1), ((5, 33), 0), ((3, 4, 50), -1), ((2, 3000), 1), ((70001,), 0), ((4, 1), 1), ((6, 2048), 1)])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64, torch.int64, torch.bfloat16])
def test_sort_argsort_are_the_stable_sort(gpu, shape, dim, dt):
    """STen.sort / argsort (STen.scala:1761, 1592): values and int64 indices bit-exact against ATen's STABLE sort in both directions - ties
    (the closed form repeats every 1009 elements; the integer case has few distinct values), NaN above everything, rows shorter and
    longer than one LDS chunk, lengths that are no power of two."""
    x = closed_form(shape, 3, 50.0, torch.float64)
    if dt == torch.int64:
        x = (x.round() % 7).to(torch.int64)
    else:
        x = x.to(dt)
        if x.numel() > 5:
            x.view(-1)[3] = float("nan"); x.view(-1)[x.numel() // 2] = float("nan")
    X = to_sten(x)
    for desc in (0, 1):
        ref_v, ref_i = torch.sort(x, dim=dim, descending=bool(desc), stable=True)
        v, i = _out1(), _out1()
        lib.lamp_sort(C.byref(v), C.byref(i), X, dim, desc)
        V, I = S.STen(v), S.STen(i)
        assert np.array_equal(I.to_numpy(), ref_i.numpy()), f"sort indices (descending={desc})"
        got = to_torch(V).to(torch.float64)
        assert torch.equal(torch.nan_to_num(got, nan=1e300), torch.nan_to_num(ref_v.to(torch.float64), nan=1e300)), "sort values"
        a = _out1()
        lib.lamp_argsort(C.byref(a), X, 1, dim, desc)
        assert np.array_equal(S.STen(a).to_numpy(), ref_i.numpy()), "argsort"
    mv, mi = _out1(), _out1()
    lib.lamp_median_dim(C.byref(mv), C.byref(mi), X, dim, 0)
    if dt != torch.bfloat16 and not (dt != torch.int64 and x.numel() > 5):          # ATen's median propagates NaN: compare on NaN-free inputs only
        rv, ri = torch.median(x, dim=dim)
        assert torch.equal(to_torch(S.STen(mv)).to(torch.float64), rv.to(torch.float64)), "median values"
        # the position of the median may be any index holding that value: the value at it must be the median
        assert torch.equal(torch.gather(x, dim if x.ndim else 0, torch.from_numpy(S.STen(mi).to_numpy()).unsqueeze(dim)).squeeze(dim).to(torch.float64), rv.to(torch.float64))


def test_unique_and_bincount(gpu):
    """STen.unique (sorted, with inverse and counts: STen.scala:1037-1055) and bincount (:1034) against ATen"""
    for n, mod in [(1, 3), (10, 3), (5000, 37), (70000, 1009)]:
        x = ((torch.arange(n) * 7919) % mod).to(torch.int64).reshape(-1)
        for t in (x, x.to(torch.float32) * 0.5, x.reshape(-1, 1) if n > 1 else x):
            v, inv, cnt = _out1(), _out1(), _out1()
            lib.lamp_unique(C.byref(v), C.byref(inv), C.byref(cnt), to_sten(t))
            rv, rinv, rcnt = torch.unique(t, sorted=True, return_inverse=True, return_counts=True)
            assert np.array_equal(S.STen(v).to_numpy(), rv.numpy()) and np.array_equal(S.STen(inv).to_numpy(), rinv.numpy()) and np.array_equal(S.STen(cnt).to_numpy(), rcnt.numpy())
        o = _out1()
        lib.lamp_bincount(C.byref(o), to_sten(x), None, mod + 5)
        assert np.array_equal(S.STen(o).to_numpy(), torch.bincount(x, minlength=mod + 5).numpy())
        w = closed_form((n,), 5, 2.0, torch.float64)
        o = _out1()
        lib.lamp_bincount(C.byref(o), to_sten(x), to_sten(w), 0)
        np.testing.assert_allclose(S.STen(o).to_numpy(), torch.bincount(x, weights=w).numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64, torch.int64, torch.bfloat16])
def test_overwriting_scatters(gpu, dt):
    """scatter (tensor and scalar source), index_put (with and without accumulation, broadcast indices and values, negative indices), put and
    index_copy (STen.scala:1412-1423, 1715-1726) against ATen; index sets without duplicates where the result would be unspecified."""
    x = closed_form((6, 9, 4), 3, 20.0, torch.float64).round().to(dt)
    src = closed_form((6, 5, 4), 9, 20.0, torch.float64).round().to(dt)
    idx = torch.stack([torch.randperm(9, generator=torch.Generator().manual_seed(s))[:5] for s in range(24)]).reshape(6, 4, 5).permute(0, 2, 1).contiguous()
    X, SRC, IDX = to_sten(x), to_sten(src), to_sten(idx)
    o = _out1(); lib.lamp_scatter(C.byref(o), X, 1, IDX, SRC)
    assert torch.equal(to_torch(S.STen(o)).double(), x.scatter(1, idx, src).double())
    o = _out1(); lib.lamp_scatter_value(C.byref(o), X, 1, IDX, 3.0)
    assert torch.equal(to_torch(S.STen(o)).double(), x.scatter(1, idx, 3.0).double())
    # index_put: two index tensors (the second broadcast from one row), values broadcast over the trailing dimension
    i0 = torch.tensor([[0], [4], [-1]]); i1 = torch.tensor([[1, 3, 8]])
    vals = closed_form((3, 3, 1), 4, 10.0, torch.float64).round().to(dt)
    I0, I1 = to_sten(i0), to_sten(i1)
    hs = (C.c_void_p * 2)(I0.h, I1.h)
    for acc in (0, 1):
        o = _out1(); lib.lamp_index_put(C.byref(o), X, hs, 2, to_sten(vals), acc)
        assert torch.equal(to_torch(S.STen(o)).double(), x.index_put((i0, i1), vals, accumulate=bool(acc)).double()), f"index_put accumulate={acc}"
    # accumulation with duplicates is defined: sums
    d0 = torch.tensor([2, 2, 2, 4])
    dv = closed_form((4, 9, 4), 6, 4.0, torch.float64).round().to(dt)
    D0 = to_sten(d0)
    hs1 = (C.c_void_p * 1)(D0.h)
    o = _out1(); lib.lamp_index_put(C.byref(o), X, hs1, 1, to_sten(dv), 1)
    assert torch.equal(to_torch(S.STen(o)).double(), x.index_put((d0,), dv, accumulate=True).double())
    pi = torch.tensor([[0, 17], [215, 100]]); pv = closed_form((2, 2), 8, 9.0, torch.float64).round().to(dt)
    for acc in (0, 1):
        o = _out1(); lib.lamp_put(C.byref(o), X, to_sten(pi), to_sten(pv), acc)
        assert torch.equal(to_torch(S.STen(o)).double(), x.put(pi, pv, accumulate=bool(acc)).double())
    ci = torch.tensor([7, 0, 3]); cs = closed_form((6, 3, 4), 2, 9.0, torch.float64).round().to(dt)
    o = _out1(); lib.lamp_index_copy(C.byref(o), X, 1, to_sten(ci), to_sten(cs))
    assert torch.equal(to_torch(S.STen(o)).double(), x.index_copy(1, ci, cs).double())
    with pytest.raises(Exception):                       # out of range: the device assertion surfaces at the next host wait
        far = to_sten(torch.full((6, 1, 4), 9))
        bad = _out1(); lib.lamp_scatter_value(C.byref(bad), X, 1, far, 1.0)
        S.STen(bad).to_numpy()


@pytest.mark.parametrize("dt", [torch.float32, torch.float64, torch.int64, torch.bfloat16])
def test_triangles_and_diagonals(gpu, dt):
    """tril / tril_ / triu, diagonal (a view) and trace (STen.scala:1883-1886, 1322)"""
    x = closed_form((3, 5, 7), 3, 20.0, torch.float64).round().to(dt)
    X = to_sten(x)
    for k in (-2, 0, 1, 9):
        o = _out1(); lib.lamp_tril(C.byref(o), X, k)
        assert torch.equal(to_torch(S.STen(o)).double(), x.tril(k).double())
        o = _out1(); lib.lamp_triu(C.byref(o), X, k)
        assert torch.equal(to_torch(S.STen(o)).double(), x.triu(k).double())
    Y = to_sten(x.clone())
    lib.lamp_tril_out(Y, Y, -1)                        # STen.tril_: out is self
    assert torch.equal(to_torch(Y).double(), x.tril(-1).double())
    for off, d1, d2 in [(0, 0, 1), (1, 1, 2), (-2, 2, 0), (0, -1, -2)]:
        o = _out1(); lib.lamp_diagonal(C.byref(o), X, off, d1, d2)
        D = S.STen(o)
        assert torch.equal(to_torch(D).double(), x.diagonal(off, d1, d2).double())
    m = x[0]
    o = _out1(); lib.lamp_trace(C.byref(o), to_sten(m))
    assert float(to_torch(S.STen(o)).double()) == float(m.double().trace())
    # diagonal is a view: writing through it changes the matrix
    M = to_sten(torch.zeros(4, 4, dtype=torch.float64))
    o = _out1(); lib.lamp_diagonal(C.byref(o), M, 0, 0, 1)
    S.STen(o).fill_(2.0)
    assert np.array_equal(M.to_numpy(), 2.0 * np.eye(4))


def test_sorting_family_special_values(gpu):
    """-0.0 and +0.0 are ONE value for ATen's comparisons (a stable sort keeps their order, unique counts them once), every NaN is a
    value of its own for unique, median propagates NaN with the position of the slice's first NaN, and multinomial rejects negative / NaN /
    all-zero rows (raised at the next host wait, as the other device-side checks)."""
    nan = float("nan")
    z = torch.tensor([0.0, -0.0, -0.0, 0.0, -1.0, 0.0], dtype=torch.float32)
    sv, si = _out1(), _out1()
    lib.lamp_sort(C.byref(sv), C.byref(si), to_sten(z), 0, 0)
    rv, ri = torch.sort(z, stable=True)
    assert np.array_equal(S.STen(si).to_numpy(), ri.numpy()) and np.array_equal(np.signbit(S.STen(sv).to_numpy()), np.signbit(rv.numpy()))
    for dt in (torch.float32, torch.float64):
        x = torch.tensor([0.0, -0.0, nan, 2.0, nan, 1.0, 2.0], dtype=dt)
        v, inv, cnt = _out1(), _out1(), _out1()
        lib.lamp_unique(C.byref(v), C.byref(inv), C.byref(cnt), to_sten(x))
        rv, rinv, rcnt = torch.ops.aten._unique2(x, True, True, True)
        gv = S.STen(v).to_numpy()
        assert np.array_equal(gv, rv.numpy(), equal_nan=True) and np.array_equal(S.STen(cnt).to_numpy(), rcnt.numpy()), (gv, rv)
        # each NaN element maps to one of the NaN slots, each slot used once (which NaN lands where is unspecified)
        gi = S.STen(inv).to_numpy()
        assert np.array_equal(gi[[0, 1, 3, 5, 6]], rinv.numpy()[[0, 1, 3, 5, 6]]) and sorted(gi[[2, 4]].tolist()) == sorted(rinv.numpy()[[2, 4]].tolist())
        m = torch.tensor([[1.0, nan, 0.5, nan], [3.0, 1.0, 2.0, 0.0], [nan, nan, nan, nan]], dtype=dt)
        for dim in (0, 1):
            mv, mi = _out1(), _out1()
            lib.lamp_median_dim(C.byref(mv), C.byref(mi), to_sten(m), dim, 0)
            rmv, rmi = torch.ops.aten.median.dim(m, dim, False)
            assert np.array_equal(S.STen(mv).to_numpy(), rmv.numpy(), equal_nan=True), (dim, S.STen(mv).to_numpy(), rmv)
            nanrows = torch.isnan(rmv).numpy()
            assert np.array_equal(S.STen(mi).to_numpy()[nanrows], rmi.numpy()[nanrows]), "position of the first NaN"
    for bad in ([0.2, -0.1, 0.9], [0.0, 0.0, 0.0], [0.5, nan, 0.5], [0.5, float("inf"), 0.5]):
        for repl in (0, 1):
            with pytest.raises(Exception, match="multinomial"):
                o = _out1()
                lib.lamp_multinomial(C.byref(o), to_sten(torch.tensor([[0.3, 0.3, 0.4], bad], dtype=torch.float32)), 2, repl)
                S.STen(o).to_numpy()                     # the host wait that reports the device-side check
    o = _out1(); lib.lamp_multinomial(C.byref(o), to_sten(torch.tensor([0.0, 1.0, 0.0])), 3, 1)
    assert S.STen(o).to_numpy().tolist() == [1, 1, 1]


def test_mode_unique_along_a_dimension_and_cartesian_product(gpu):
    """STen.mode (STen.scala:1561), STen.unique(dim, ...) (:1059), STen.uniqueConsecutive (:1068), STen.cartesianProduct (:674) against ATen:
    values and integer outputs bit-exact (mode: the smallest most frequent value and the position of its LAST occurrence; unique_dim: slices
    sorted lexicographically over their flattened elements - the third dimension of a 3-D tensor included, where the flattening order matters)."""
    g = torch.Generator().manual_seed(7)
    for shape, dim in [((5,), 0), ((6, 9), 1), ((6, 9), 0), ((4, 5, 6), 1), ((4, 5, 6), 2), ((300, 7), 0), ((3, 2000), 1)]:
        for dt in (torch.int64, torch.float32):
            x = torch.randint(0, 4, shape, generator=g).to(dt)
            for keep in (0, 1):
                mv, mi = _out1(), _out1()
                lib.lamp_mode(C.byref(mv), C.byref(mi), to_sten(x), dim, keep)
                rv, ri = torch.mode(x, dim, bool(keep))
                gi = torch.from_numpy(S.STen(mi).to_numpy())
                assert np.array_equal(S.STen(mv).to_numpy(), rv.numpy()), (shape, dim, dt, keep)
                # the position: ATen's CPU kernel sorts (value, index) pairs with an unstable sort, so WHICH occurrence it reports is
                # unspecified for long slices; this library always reports the last one (what ATen reports for short slices)
                gk = gi if keep else gi.unsqueeze(dim)
                assert torch.equal(torch.gather(x, dim, gk), rv if keep else rv.unsqueeze(dim)), "the index does not hold the mode"
                last = (x == (rv if keep else rv.unsqueeze(dim))).to(torch.int64) * torch.arange(x.shape[dim]).view([-1 if d == dim else 1 for d in range(x.ndim)])
                assert torch.equal(gk, last.max(dim, keepdim=True).values), "not the LAST occurrence"
                if x.shape[dim] <= 16:
                    assert torch.equal(gi, ri), (shape, dim, dt, keep)
            for name, ref in (("lamp_unique_dim", lambda t: torch.unique(t, sorted=True, return_inverse=True, return_counts=True, dim=dim)),
                              ("lamp_unique_consecutive", lambda t: torch.unique_consecutive(t, return_inverse=True, return_counts=True, dim=dim))):
                v, inv, cnt = _out1(), _out1(), _out1()
                getattr(lib, name)(C.byref(v), C.byref(inv), C.byref(cnt), to_sten(x), dim)
                rv, rinv, rcnt = ref(x)
                assert np.array_equal(S.STen(v).to_numpy(), rv.numpy()), (name, shape, dim, dt)
                assert np.array_equal(S.STen(inv).to_numpy(), rinv.numpy()) and np.array_equal(S.STen(cnt).to_numpy(), rcnt.numpy()), (name, shape, dim, dt)
                v2 = _out1()
                getattr(lib, name)(C.byref(v2), None, None, to_sten(x), dim)
                assert np.array_equal(S.STen(v2).to_numpy(), rv.numpy())
    ts = [torch.tensor([1.0, 2.0, 3.0]), torch.tensor([4.0, 5.0]), torch.tensor([7.0, 8.0, 9.0, 10.0])]
    for k in (1, 2, 3):
        o = _out1()
        hs = [to_sten(t) for t in ts[:k]]
        lib.lamp_cartesian_prod(C.byref(o), handle_array([h.h for h in hs]), k)
        assert np.array_equal(S.STen(o).to_numpy(), torch.cartesian_prod(*ts[:k]).numpy())


def test_randperm_and_multinomial(gpu):
    """STen.randperm (a permutation; positions are uniform) and STen.multinomial (with replacement: frequencies follow the weights; without:
    distinct indices, zero-weight categories never drawn, the first draw follows the weights) - the sampler of the language model
    (languagemodel/package.scala:100)."""
    for n in (1, 2, 1000, 70000):
        o = _out1(); lib.lamp_randperm(C.byref(o), n, S.I64, 0)
        p = S.STen(o).to_numpy()
        assert p.dtype == np.int64 and np.array_equal(np.sort(p), np.arange(n))
    # uniformity: the position of element 0 over many small permutations
    pos = []
    for _ in range(300):
        o = _out1(); lib.lamp_randperm(C.byref(o), 8, S.I64, 0)
        pos.append(int(np.where(S.STen(o).to_numpy() == 0)[0][0]))
    counts = np.bincount(pos, minlength=8)
    assert counts.min() >= 15 and counts.max() <= 70, counts          # expectation 37.5, sd 5.7
    o = _out1(); lib.lamp_randperm(C.byref(o), 5, S.I64, -1)
    assert S.STen(o).device == S.CPU
    w = torch.tensor([[0.1, 0.0, 0.3, 0.6], [5.0, 5.0, 0.0, 0.0]], dtype=torch.float32)
    o = _out1(); lib.lamp_multinomial(C.byref(o), to_sten(w), 20000, 1)
    s = S.STen(o).to_numpy()
    assert s.shape == (2, 20000) and s.dtype == np.int64
    f0 = np.bincount(s[0], minlength=4) / 20000.0
    f1 = np.bincount(s[1], minlength=4) / 20000.0
    assert np.abs(f0 - np.array([0.1, 0.0, 0.3, 0.6])).max() < 0.015 and f0[1] == 0.0, f0
    assert np.abs(f1 - np.array([0.5, 0.5, 0.0, 0.0])).max() < 0.015 and f1[2] == 0.0 and f1[3] == 0.0, f1
    firsts = []
    wv = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.0], dtype=torch.float64)
    for _ in range(400):
        o = _out1(); lib.lamp_multinomial(C.byref(o), to_sten(wv), 4, 0)
        d = S.STen(o).to_numpy()
        assert d.shape == (4,) and sorted(d.tolist()) == [0, 1, 2, 3], "without replacement: the four categories with weight, each once"
        firsts.append(int(d[0]))
    ff = np.bincount(firsts, minlength=5) / 400.0
    assert np.abs(ff[:4] - np.array([0.1, 0.2, 0.3, 0.4])).max() < 0.08, ff
    with pytest.raises(Exception):
        bad = _out1(); lib.lamp_multinomial(C.byref(bad), to_sten(wv), 9, 0)
