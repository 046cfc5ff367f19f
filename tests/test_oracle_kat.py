"""Pins the oracle (oracle/lamp_oracle.py) to the reference's own known-answer tests.  CPU only.

Every constant comes from tests/golden/reference_kats.json (transcribed from the reference's
scalatest sources, file:line recorded there).  The rule is the reference's: forward value equal
to 4 decimals, autograd gradient equal to the central finite difference to 4 decimals.
"""
import numpy as np
import pytest
import torch

from oracle import lamp_oracle as O
from tests import kats
from tests.backends import OracleBackend

B = OracleBackend()


@pytest.mark.parametrize("name", sorted(kats.CASES))
def test_autograd_kat_value_and_gradient(name):
    value, grad = kats.run_case(B, name)
    assert round(value, 4) == round(kats.EXPECTED[name], 4), (name, value, kats.EXPECTED[name])
    fd = kats.finite_difference(B, name)
    # flattened and NaN == NaN, as the reference compares (`.toVec.roundTo(4) ==` on saddle vectors: autograd.test.scala:135, 176)
    assert np.array_equal(np.round(grad.reshape(-1), 4) + 0.0, np.round(fd.reshape(-1), 4) + 0.0, equal_nan=True), (name, grad, fd)


@pytest.mark.parametrize("name", sorted(kats.SDPA))
def test_fused_attention_kat_value_and_gradient(name):
    """autograd.test.scala:219-285 (CUDA only in the reference): the fused operator on q / k / v of shape (1, 8, 1, 8), f32.
    Value to 4 decimals; the autograd gradient is exact (0 for q and k, 1 for v) and equals the reference's central difference
    wherever f32 can resolve it (the sum of 64 f32 values near 704 moves in steps of 6e-5: the v case compares to 1 decimal)."""
    value, grad = kats.run_sdpa_case(B, name)
    assert round(value, 4) == round(kats.SDPA[name]["expected"], 4)
    assert np.array_equal(grad, np.full(64, 1.0 if name.endswith("v") else 0.0))
    fd = kats.sdpa_finite_difference(B, name)
    digits = 1 if name.endswith("v") else 4
    assert np.array_equal(np.round(grad, digits) + 0.0, np.round(fd, digits) + 0.0), (grad, fd)


def test_exact_constants_at_full_precision():
    # the KATs the reference states with all digits
    for name in ("softmax", "exp", "l2 logistic regression loss - nll_loss", "nn Logistic 2 - wrt weight", "nn Mlp1 - wrt first weight"):
        value, _ = kats.run_case(B, name, backprop=False)
        assert abs(value - kats.EXPECTED[name]) <= 1e-9 * max(1.0, abs(kats.EXPECTED[name])), (name, value)


def test_adamw_kats():
    g = kats.GOLDEN["adamw"]
    for key in ("no_weight_decay", "weight_decay"):
        p = torch.tensor([g["init"]], dtype=torch.float64)
        grad = torch.tensor([g["gradients"]], dtype=torch.float64)
        opt = O.AdamW([p], weightDecay=g[key]["weightDecay"], learningRate=g["learningRate"], beta1=g["beta1"], beta2=g["beta2"])
        opt.step([grad], 1.0)
        digits = g[key].get("roundTo")
        if digits:
            assert np.array_equal(np.round(p.numpy()[0], digits), np.round(g[key]["step1"], digits))
        else:
            assert p.numpy()[0].tolist() == g[key]["step1"]          # exact, as in the reference test
        opt.step([grad], 1.0)
        if digits:
            assert np.array_equal(np.round(p.numpy()[0], digits), np.round(g[key]["step2"], digits))
        else:
            assert p.numpy()[0].tolist() == g[key]["step2"]
    # fp16 parameters with a fp32 working copy
    p = torch.tensor([g["init"]], dtype=torch.float64).half()
    grad = torch.tensor([g["gradients"]], dtype=torch.float64).half()
    opt = O.AdamW([p], weightDecay=g["half_mixed"]["weightDecay"], learningRate=g["learningRate"], beta1=g["beta1"], beta2=g["beta2"],
                  mixedPrecision=True)
    opt.step([grad], 1.0)
    assert p.double().numpy()[0].tolist() == g["half_mixed"]["step1"]
    opt.step([grad], 1.0)
    assert p.double().numpy()[0].tolist() == g["half_mixed"]["step2"]


def test_sgd_kats():
    g = kats.GOLDEN["sgd"]
    for key in ("noop", "no_momentum_no_wd", "no_momentum"):
        p = torch.ones(1, 2, dtype=torch.float64)
        O.SGDW([p], g[key]["lr"], g[key]["wd"]).step([torch.tensor([g[key]["grad"]], dtype=torch.float64)], 1.0)
        assert np.allclose(p.numpy()[0], g[key]["expect"], rtol=0, atol=1e-15)
    p = torch.ones(1, 2, dtype=torch.float64)
    opt = O.SGDW([p], g["two_steps"]["lr"], g["two_steps"]["wd"])
    grad = torch.tensor([g["two_steps"]["grad"]], dtype=torch.float64)
    opt.step([grad], 1.0)
    assert np.array_equal(np.round(p.numpy()[0], 4), np.round(g["two_steps"]["step1"], 4))
    opt.step([grad], 1.0)
    assert np.array_equal(np.round(p.numpy()[0], 4), np.round(g["two_steps"]["step2"], 4))


def test_gradient_clipping_kat():
    g = kats.GOLDEN["gradient_clipping"]
    ts = [torch.ones(s, dtype=torch.float64) for s in g["shapes"]]
    O.gradient_clipping_in_place(ts, g["theta"])
    assert np.array_equal(np.round(ts[0].numpy().reshape(-1), 4), np.round(np.full(6, g["expect"]), 4))


def test_umap_edge_weights_kat():
    # lamp-umap/src/test/scala/lamp/umap/umap.test.scala:10-53
    data = torch.tensor([[1.0, 4.0], [2.0, 5.0], [3.0, 6.0]], dtype=torch.float64)
    knn = O.knn_minibatched(data, data, 3, 100)
    # sorted=false in the reference; order neighbours by distance like the reference's result
    d2 = O.squared_euclidean_distance(data, data)
    knn = torch.stack([row[torch.argsort(d2[i][row], stable=True)] for i, row in enumerate(knn)])
    dist = [[float(torch.linalg.vector_norm(data[i] - data[j])) for j in knn[i]] for i in range(3)]
    rows = O.edge_weights(dist, knn.tolist())
    exp = [(0., 1., 1.), (0., 2., 0.), (1., 0., 1.), (1., 2., 1.), (2., 1., 1.), (2., 0., 0.)]
    assert rows == exp, rows


def test_knn_is_exact_on_integer_points():
    # lamp-knn test style: exact squared distances and neighbour sets
    data = torch.tensor([[0., 0.], [1., 0.], [0., 2.], [5., 5.]], dtype=torch.float64)
    d = O.squared_euclidean_distance(data, data)
    assert d.tolist() == [[0., 1., 4., 50.], [1., 0., 5., 41.], [4., 5., 0., 34.], [50., 41., 34., 0.]]
    idx = O.knn_minibatched(data, data, 2, 3)
    assert [set(r) for r in idx.tolist()] == [{0, 1}, {1, 0}, {2, 0}, {3, 2}]


def test_reference_semantic_traps():
    # relu gradient at exactly 0 is 1 (ops.scala:918-935), unlike ATen's threshold_backward
    x = O.param(torch.tensor([[0.0, -1.0, 2.0]], dtype=torch.float64))
    x.relu().sum().backprop()
    assert x.grad.tolist() == [[1.0, 0.0, 1.0]]
    # gradients accumulate into the pre-allocated buffer
    x = O.param(torch.ones(2, 2, dtype=torch.float64))
    (x + x).sum().backprop()
    assert x.grad.tolist() == [[2.0, 2.0], [2.0, 2.0]]
    # batch norm: running_var updated with the unbiased estimate, starting from 0 (BatchNorm2D.scala:62-66)
    bn = O.make_bn(3, torch.float64, 0, True)
    xin = O.const(torch.arange(3 * 3 * 2 * 2, dtype=torch.float64).reshape(3, 3, 2, 2))
    bn.forward(xin)
    per_channel = xin.value.transpose(0, 1).reshape(3, -1)
    assert torch.allclose(bn.runningVar.value, 0.1 * per_channel.var(1, unbiased=True))
    assert torch.allclose(bn.runningMean.value, 0.1 * per_channel.mean(1))
