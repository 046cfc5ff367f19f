"""HIP path vs the reference's known-answer tests and vs the oracle, through the host C ABI.

1. Every KAT of tests/kats.py (the reference's own `testGradientAndValue` cases) runs on the GPU in
   float64 with the reference's acceptance rule: value to 4 decimals, autograd gradient == central
   finite difference to 4 decimals.
2. Optimiser KATs (exact AdamW/SGDW values, gradient clipping).
3. Composite parity vs the oracle on deterministic closed-form inputs: MLP step (BASELINE config 1),
   CIFAR ResNet step, several optimiser steps; f32 forward <= 1e-5, gradients <= 1e-3 (BASELINE.json).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from lamp_amd import autograd as A
from lamp_amd import nn
from lamp_amd import sten as S
from lamp_amd._capi import lib
from oracle import lamp_oracle as O
from tests import kats
from tests.backends import HipBackend, OracleBackend
from tests.util import assert_close, to_sten, to_torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(kats.CASES))
def test_reference_kat_on_gpu(gpu, name):
    B = HipBackend()
    value, grad = kats.run_case(B, name)
    assert round(value, 4) == round(kats.EXPECTED[name], 4), (name, value, kats.EXPECTED[name])
    fd = kats.finite_difference(B, name)
    # flattened and NaN == NaN, as the reference compares (`.toVec.roundTo(4) ==` on saddle vectors: autograd.test.scala:135, 176)
    assert np.array_equal(np.round(grad.reshape(-1), 4) + 0.0, np.round(fd.reshape(-1), 4) + 0.0, equal_nan=True), (name, grad, fd)
    # and bitwise-close to the oracle in float64
    ovalue, ograd = kats.run_case(OracleBackend(), name)
    assert abs(value - ovalue) <= 1e-12 * max(1.0, abs(ovalue))
    np.testing.assert_allclose(grad, ograd, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("device", [0, S.CPU])
@pytest.mark.parametrize("name", sorted(kats.SDPA))
def test_fused_attention_kat_on_gpu(gpu, name, device):
    """autograd.test.scala:219-285 through lamp_scaled_dot_product_attention (+ _backward) in f32: value to 4 decimals, the exact
    gradient, the reference's finite-difference rule where f32 resolves it, and agreement with the oracle (ATen's fused CPU operator).
    Also with host tensors (the CPU device: staged through the GPU)."""
    B = HipBackend(device=device)
    value, grad = kats.run_sdpa_case(B, name)
    assert round(value, 4) == round(kats.SDPA[name]["expected"], 4)
    assert np.array_equal(grad, np.full(64, 1.0 if name.endswith("v") else 0.0))
    fd = kats.sdpa_finite_difference(B, name)
    digits = 1 if name.endswith("v") else 4
    assert np.array_equal(np.round(grad, digits) + 0.0, np.round(fd, digits) + 0.0), (grad, fd)
    ovalue, ograd = kats.run_sdpa_case(OracleBackend(), name)
    assert value == ovalue and np.array_equal(grad, ograd)


def test_adamw_kats_on_gpu(gpu):
    g = kats.GOLDEN["adamw"]
    for key in ("no_weight_decay", "weight_decay"):
        p = S.STen.from_numpy(np.array([g["init"]]), dtype=S.F64)
        grad = S.STen.from_numpy(np.array([g["gradients"]]), dtype=S.F64)
        opt = nn.AdamW([p], weightDecay=g[key]["weightDecay"], learningRate=g["learningRate"], beta1=g["beta1"], beta2=g["beta2"])
        for step in ("step1", "step2"):
            opt.step([grad], 1.0)
            np.testing.assert_allclose(p.to_numpy()[0], g[key][step], rtol=1e-14, atol=0)
            assert np.array_equal(np.round(p.to_numpy()[0], 10), np.round(g[key][step], 10))
        assert opt.state[0].to_numpy() == 2.0          # stepCount scalar is state()[0] (AdamW.scala:97)
    # mixed precision (bf16 parameters, f32 working copy): same trajectory as the oracle
    p0 = torch.tensor([g["init"]], dtype=torch.float64).bfloat16()
    gr = torch.tensor([g["gradients"]], dtype=torch.float64).bfloat16()
    po = p0.clone()
    oo = O.AdamW([po], weightDecay=1e-5, learningRate=0.1, beta1=0.999, beta2=0.9, mixedPrecision=True)
    ph, gh = to_sten(p0), to_sten(gr)
    oh = nn.AdamW([ph], weightDecay=1e-5, learningRate=0.1, beta1=0.999, beta2=0.9, mixedPrecision=True)
    for _ in range(3):
        oo.step([gr], 1.0); oh.step([gh], 1.0)
        assert np.array_equal(ph.to_numpy(), po.float().numpy()), "bf16 parameters must round identically"


def test_fp16_mixed_precision_adamw_kat_on_gpu(gpu):
    """adamw.test.scala:96-127: half-precision parameters and gradients, f32 working copy and moments: the exact values of the
    reference after one and two steps (0.89990234375 -> 0.79931640625), computed by the fused kernel on f16 tensors."""
    g = kats.GOLDEN["adamw"]
    h = g["half_mixed"]
    p = S.STen.from_numpy(np.array([g["init"]], dtype=np.float16), dtype=S.F16)
    grad = S.STen.from_numpy(np.array([g["gradients"]], dtype=np.float16), dtype=S.F16)
    opt = nn.AdamW([p], weightDecay=h["weightDecay"], learningRate=g["learningRate"], beta1=g["beta1"], beta2=g["beta2"], mixedPrecision=True)
    opt.step([grad], 1.0)
    assert p.to_numpy().astype(np.float64)[0].tolist() == h["step1"]
    opt.step([grad], 1.0)
    assert p.to_numpy().astype(np.float64)[0].tolist() == h["step2"]
    assert [t.dtype for t in opt.state[1:]] == [S.F32] * (len(opt.state) - 1), "moments and working copy are f32"


@pytest.mark.parametrize("dt", [torch.float16])
def test_half_precision_compute(gpu, dt):
    """f16 (lamp's HalfPrecision, scalar type byte 5) is a compute type of the library: casts, element-wise arithmetic, reductions,
    mm / bmm on the f16 matrix cores, softmax - against ATen on CPU at f16 resolution (2^-10 per element)."""
    from tests.util import closed_form
    a, b = closed_form((37, 64), 3, 2.0, dt), closed_form((37, 64), 5, 2.0, dt)
    A_, B_ = to_sten(a), to_sten(b)
    assert A_.dtype == S.F16
    tol = 2.0 ** -10
    assert_close(to_torch(A_ + B_), (a + b).double(), tol, "add")
    assert_close(to_torch(A_ * B_), (a * b).double(), tol, "mul")
    assert_close(to_torch(A_.castToFloat()), a.float().double(), 0.0, "f16 -> f32 is exact")
    assert_close(to_torch(to_sten(a.float()).castToType(S.F16)), a.double(), 0.0, "f32 -> f16 round trip")
    assert_close(to_torch(A_.exp()), a.float().exp().half().double(), tol * 2, "exp")
    assert_close(to_torch(A_.sum([1], False)), a.float().sum(1).half().double(), tol * 2, "row sums (f32 accumulation)")
    w = closed_form((64, 48), 9, 1.0, dt)
    assert_close(to_torch(A_.mm(to_sten(w))), (a.double() @ w.double()), tol * 2, "mm")
    assert_close(to_torch(A_.t.mm(to_sten(b))), (a.double().t() @ b.double()), tol * 2, "mm, transposed operand")
    x3, y3 = closed_form((5, 20, 16), 1, 1.0, dt), closed_form((5, 16, 24), 2, 1.0, dt)
    assert_close(to_torch(to_sten(x3).bmm(to_sten(y3))), x3.double() @ y3.double(), tol * 2, "bmm")
    assert_close(to_torch(A_.logSoftMax(1)), torch.log_softmax(a.float(), 1).double(), tol * 4, "log_softmax")
    # gradients flow in f16 through the host operators
    pa, pw = A.param(A_), A.param(to_sten(w))
    pa.mm(pw).relu().sum().backprop()
    ref_a, ref_w = a.double().requires_grad_(True), w.double().requires_grad_(True)
    (ref_a @ ref_w).relu().sum().backward()
    assert_close(to_torch(pa.partialDerivative), ref_a.grad, tol * 4, "dA", scale="max")
    assert_close(to_torch(pw.partialDerivative), ref_w.grad, tol * 4, "dW", scale="max")


def test_sgd_and_clipping_kats_on_gpu(gpu):
    g = kats.GOLDEN["sgd"]
    for key in ("noop", "no_momentum_no_wd", "no_momentum"):
        p = S.STen.ones([1, 2], S.F64)
        nn.SGDW([p], g[key]["lr"], g[key]["wd"]).step([S.STen.from_numpy(np.array([g[key]["grad"]]))], 1.0)
        np.testing.assert_allclose(p.to_numpy()[0], g[key]["expect"], rtol=0, atol=1e-15)
    p = S.STen.ones([1, 2], S.F64)
    opt = nn.SGDW([p], 1.0, 0.1)
    grad = S.STen.from_numpy(np.array([g["two_steps"]["grad"]]))
    opt.step([grad], 1.0)
    assert np.array_equal(np.round(p.to_numpy()[0], 4), np.round(g["two_steps"]["step1"], 4))
    opt.step([grad], 1.0)
    assert np.array_equal(np.round(p.to_numpy()[0], 4), np.round(g["two_steps"]["step2"], 4))
    # momentum variant vs oracle
    po = torch.ones(1, 2, dtype=torch.float64); ph = S.STen.ones([1, 2], S.F64)
    oo, oh = O.SGDW([po], 0.5, 0.01, momentum=0.9), nn.SGDW([ph], 0.5, 0.01, momentum=0.9)
    gt = torch.tensor([[0.5, 0.75]], dtype=torch.float64)
    for _ in range(3):
        oo.step([gt], 1.0); oh.step([to_sten(gt)], 1.0)
    np.testing.assert_allclose(ph.to_numpy(), po.numpy(), rtol=1e-14)
    c = kats.GOLDEN["gradient_clipping"]
    ts = [S.STen.ones(s, S.F64) for s in c["shapes"]]
    nn.gradientClippingInPlace(ts, c["theta"])
    assert np.array_equal(np.round(ts[0].to_numpy().reshape(-1), 4), np.round(np.full(6, c["expect"]), 4))
    np.testing.assert_allclose(ts[1].to_numpy().reshape(-1), np.full(2, c["expect"]), rtol=1e-14)


def test_semantic_traps_on_gpu(gpu):
    x = A.param(S.STen.from_numpy(np.array([[0.0, -1.0, 2.0]])))
    x.relu().sum().backprop()
    assert x.partialDerivative.to_numpy().tolist() == [[1.0, 0.0, 1.0]]          # gradient at 0 is 1
    x = A.param(S.STen.ones([2, 2], S.F64))
    (x + x).sum().backprop()
    assert x.partialDerivative.to_numpy().tolist() == [[2.0, 2.0], [2.0, 2.0]]   # accumulate contract
    c = A.const(S.STen.ones([2, 2], S.F64))
    assert c.partialDerivative is None and not c.needsGrad
    # Linear = mm then broadcast add of a [1, out] bias; BN running_var starts at 0
    m = nn.BatchNorm2D(3, S.F64)
    st = m.state
    assert [v.needsGrad for v in st] == [True, True, False, False]
    assert st[3].value.to_numpy().tolist() == [0.0, 0.0, 0.0]
    lin = nn.Linear(4, 3, S.F64)
    assert lin.state[1].shape == [1, 3]


def _load_from_oracle(hip_module, oracle_module, dtype):
    hip_module.load([S.STen.from_numpy(v.value.detach().double().numpy(), dtype=dtype) for v in oracle_module.state()])


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_mlp_step_matches_oracle(gpu, dt):
    """BASELINE config 1: MLP(784 -> 256 -> 10, BatchNorm, relu, dropout 0) -> logSoftMax -> NLL(ones), B = 1024."""
    ldt = S.F64 if dt == torch.float64 else S.F32
    om = O.Sequential(O.mlp(784, 10, [256], dt), O.Fun(lambda v: v.logSoftMax(1)))
    hm = nn.Sequential(nn.MLP(784, 10, [256], ldt), nn.Fun("logsoftmax", 1))
    _load_from_oracle(hm, om, ldt)
    x = O.closed_form(1024 * 784, 0, 1.0, dt).reshape(1024, 784)
    target = torch.arange(1024) % 10
    cw = torch.ones(10, dtype=dt)
    oloss, ograds = O.training_step(om, O.nll_loss(10, cw), x, target, None)
    model = nn.SupervisedModel(hm, nn.SupervisedModel.NLL, to_sten(cw))
    acc = S.STen.zeros([1], ldt)
    n, hgrads = model.addTotalLossAndReturnGradientsAndNumExamples(to_sten(x), to_sten(target), acc)
    assert n == 1024
    ftol, btol = (1e-11, 1e-9) if dt == torch.float64 else (1e-5, 1e-3)
    assert_close(to_torch(acc) / 1024.0, oloss.double().reshape(1), ftol, "loss")
    assert len(hgrads) == len(ograds) == 6
    for i, (hg, og) in enumerate(zip(hgrads, ograds)):
        assert_close(to_torch(hg), og.double(), btol, f"gradient {i}")
    # running statistics were updated identically
    for hv, ov in zip(hm.state, om.state()):
        assert_close(to_torch(hv.value), ov.value.double(), ftol * 10, "state after step")


@pytest.mark.parametrize("dt,B", [(torch.float64, 4), (torch.float32, 16)])
def test_resnet_training_steps_match_oracle(gpu, dt, B):
    """Cnn.resnet(100) forward + backprop + AdamW, 2 steps, closed-form weights and batch."""
    ldt = S.F64 if dt == torch.float64 else S.F32
    om = O.resnet(100, dt)
    hm = nn.resnet(100, 0.0, ldt)
    assert len(hm.state) == len(om.state()) == 74 and len(hm.parameters) == 37
    _load_from_oracle(hm, om, ldt)
    x = O.closed_form(B * 3 * 32 * 32, 5, 1.0, dt).reshape(B, 3, 32, 32)
    target = (torch.arange(B) * 7) % 100
    cw = torch.ones(100, dtype=dt)
    oopt = O.AdamW([p.value for p in om.parameters()], weightDecay=0.0, learningRate=1e-3, beta1=0.9, beta2=0.95)
    hopt = nn.AdamW([p.value for p in hm.parameters], weightDecay=0.0, learningRate=1e-3, beta1=0.9, beta2=0.95)
    model = nn.SupervisedModel(hm, nn.SupervisedModel.NLL, to_sten(cw))
    X, T = to_sten(x), to_sten(target)
    ftol, btol = (1e-10, 1e-8) if dt == torch.float64 else (1e-5, 1e-3)
    for step in range(2):
        oloss, ograds = O.training_step(om, O.nll_loss(100, cw), x, target, None)
        acc = S.STen.zeros([1], ldt)
        buf = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_filter(None)
        lib.lamp_kernel_timer_report(buf, len(buf))          # drops what earlier tests left in the log
        lib.lamp_kernel_timer_enable(1)
        n, hgrads = model.addTotalLossAndReturnGradientsAndNumExamples(X, T, acc)
        grads_t = [to_torch(hg) for hg in hgrads]
        lib.lamp_kernel_timer_enable(0)
        lib.lamp_kernel_timer_report(buf, len(buf))
        # the reference's own precisions (cifar100.scala:127-129: double unless --single) must run on the f32 / f64 matrix-core convolutions,
        # not on the direct kernels
        sfx = "f32" if dt == torch.float32 else "f64"
        ran = {l.split()[0]: int(l.split()[1]) for l in buf.value.decode().splitlines() if l.strip()}
        assert ran.get("conv_igemm_fprop_dgrad_" + sfx, 0) >= 12 and ran.get("conv_wgrad_igemm_" + sfx, 0) == 6, f"{sfx} matrix-core convolutions did not run: {ran}"
        assert not any(k.startswith("conv_") and k.endswith("_direct") for k in ran), f"a wide layer fell to the direct kernels: {ran}"
        assert_close(to_torch(acc) / B, oloss.double().reshape(1), ftol, f"loss step {step}")
        for i, (hg, og) in enumerate(zip(grads_t, ograds)):
            assert_close(hg, og.double(), btol, f"step {step} gradient {i} shape {list(og.shape)}")
        oopt.step(ograds, 1.0)
        hopt.step(hgrads, 1.0)
        for i, (hv, ov) in enumerate(zip(hm.state, om.state())):
            assert_close(to_torch(hv.value), ov.value.double(), btol, f"step {step} state {i}")
    # eval mode uses the running statistics
    hm.asEval()
    for mod in om.mods[1].mods:
        pass
    out_h = hm.forward(A.const(X)).value
    assert out_h.shape == [B, 100]


def test_train_step_entry_point(gpu):
    """lamp_model_train_step == gradients + optimizer.step (single process, no communicator)."""
    hm1, hm2 = nn.resnet(100, 0.0, S.F32), nn.resnet(100, 0.0, S.F32)
    hm2.load([v.value for v in hm1.state])
    x = to_sten(O.closed_form(8 * 3 * 32 * 32, 5, 1.0, torch.float32).reshape(8, 3, 32, 32))
    t = to_sten((torch.arange(8) * 3) % 100)
    cw = S.STen.ones([100], S.F32)
    m1, m2 = nn.SupervisedModel(hm1, 0, cw), nn.SupervisedModel(hm2, 0, cw)
    o1 = nn.AdamW_factory(0.0)([p.value for p in hm1.parameters])
    o2 = nn.AdamW_factory(0.0)([p.value for p in hm2.parameters])
    for _ in range(2):
        n, g = m1.addTotalLossAndReturnGradientsAndNumExamples(x, t, None)
        o1.step(g, 1.0)
        assert m2.train_step(o2, x, t) == 8
    for a, b in zip(hm1.state, hm2.state):
        assert np.array_equal(a.value.to_numpy(), b.value.to_numpy())


@pytest.mark.parametrize("dt,D,tol", [(torch.float32, 16, 1e-4), (torch.bfloat16, 64, 4e-2)])
@pytest.mark.parametrize("causal", [False, True])
def test_scaled_dot_product_attention_op(gpu, dt, D, tol, causal):
    """The ScaledDotProductAttention op (ops.scala:2342-2390): one backward call feeds the three parents; a parent used twice
    (q is also the value here) accumulates both contributions.  f32 takes the composed kernels, bf16 with head dim 64 the fused ones."""
    g = torch.Generator().manual_seed(9)
    Bz, H, Sq = 2, 2, 70
    q0, k0 = (torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(dt) for _ in range(2))
    w0 = torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(dt)
    qd, kd = (t.double().clone().requires_grad_(True) for t in (q0, k0))
    sc = qd @ kd.transpose(-1, -2) / np.sqrt(D)
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(Sq, Sq, dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(sc, -1) @ qd
    (ref * w0.double()).sum().backward()
    q, k = A.param(to_sten(q0)), A.param(to_sten(k0))
    out = q.scaledDotProductAttention(k, q, causal)
    assert_close(to_torch(out.value), ref.detach(), tol, "attention output")
    (out * A.const(to_sten(w0))).sum().backprop()
    assert_close(to_torch(q.partialDerivative), qd.grad, tol, "dq (+ dv: q is also the value)")
    assert_close(to_torch(k.partialDerivative), kd.grad, tol, "dk")


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("causal", [False, True])
def test_scaled_dot_product_attention_with_attention_bias(gpu, dt, tol, causal):
    """attentionBias: Option[STen] of ScaledDotProductAttention (ops.scala:2342-2390): softmax(q k^T / sqrt(d) + bias (+ causal mask)) v;
    the bias is a plain tensor (no gradient), here a (1, heads, Sq, Sk) table broadcast over the batch - e.g. relative-position biases"""
    g = torch.Generator().manual_seed(11)
    Bz, H, Sq, D = 2, 3, 40, 64
    q0, k0, v0 = (torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(dt) for _ in range(3))
    bias0 = (torch.randn(1, H, Sq, Sq, generator=g, dtype=torch.float64) * 2).to(dt)
    w0 = torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(dt)
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q0, k0, v0))
    sc = qd @ kd.transpose(-1, -2) / np.sqrt(D) + bias0.double()
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(Sq, Sq, dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(sc, -1) @ vd
    (ref * w0.double()).sum().backward()
    q, k, v = A.param(to_sten(q0)), A.param(to_sten(k0)), A.param(to_sten(v0))
    out = q.scaledDotProductAttention(k, v, causal, to_sten(bias0))
    assert_close(to_torch(out.value), ref.detach(), tol, "attention output", scale="max")
    (out * A.const(to_sten(w0))).sum().backprop()
    for name, a, b in (("dq", q, qd), ("dk", k, kd), ("dv", v, vd)):
        assert_close(to_torch(a.partialDerivative), b.grad, tol, name, scale="max")


def test_hip_graph_replays_the_gradient_computation(gpu):
    """launch-bound steps: forward + backprop captured once into a HIP graph (lamp_graph_*), replayed on new batches written into the
    captured input buffer; gradients and the loss accumulator equal the eager ones, the optimiser runs eagerly between replays"""
    import ctypes as C
    from lamp_amd._capi import lib
    dt, ldt = torch.float32, S.F32
    om = O.Sequential(O.mlp(48, 5, [32], dt), O.Fun(lambda v: v.logSoftMax(1)))
    def fresh():
        m = nn.Sequential(nn.MLP(48, 5, [32], ldt), nn.Fun("logsoftmax", 1)); _load_from_oracle(m, om, ldt); return m
    cw = to_sten(torch.ones(5, dtype=dt))
    batches = [(O.closed_form(64 * 48, 11 * i, 1.0, dt).reshape(64, 48), (torch.arange(64) + i) % 5) for i in range(3)]
    # eager reference: three AdamW steps
    em = fresh(); emodel = nn.SupervisedModel(em, nn.SupervisedModel.NLL, cw)
    eopt = nn.AdamW([p.value for p in em.parameters], weightDecay=0.0, learningRate=1e-2)
    eacc = S.STen.zeros([1], ldt)
    for x, t in batches:
        emodel.train_step(eopt, to_sten(x), to_sten(t), eacc)
    # captured: same three steps, gradients from graph replays
    st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(st)); lib.lamp_stream_set_current(st)
    try:
        gm = fresh(); gmodel = nn.SupervisedModel(gm, nn.SupervisedModel.NLL, cw)
        gopt = nn.AdamW([p.value for p in gm.parameters], weightDecay=0.0, learningRate=1e-2)
        gacc = S.STen.zeros([1], ldt)
        x_buf, t_buf = to_sten(batches[0][0]), to_sten(batches[0][1])
        scratch = S.STen.zeros([1], ldt)
        gmodel.addTotalLossAndReturnGradientsAndNumExamples(x_buf, t_buf, scratch)       # eager warm-up (attributes, caches)
        for v, ov in zip(gm.state, om.state()):                                            # undo the running-statistics update
            v.value.copyFrom(to_sten(ov.value))
        lib.lamp_graph_begin_capture()
        n, grads = gmodel.addTotalLossAndReturnGradientsAndNumExamples(x_buf, t_buf, gacc)
        g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
        assert n == 64
        for x, t in batches:
            x_buf.copyFrom(to_sten(x)); t_buf.copyFrom(to_sten(t))
            lib.lamp_graph_launch(g)
            gopt.step(grads, 1.0)
        lib.lamp_device_synchronize()
        for a, b in zip(gm.state, em.state):
            assert_close(to_torch(a.value), to_torch(b.value).double(), 1e-6, "state after three captured steps")
        assert_close(to_torch(gacc), to_torch(eacc).double(), 1e-6, "loss accumulator")
        lib.lamp_graph_release(g)
    finally:
        d = C.c_void_p(); lib.lamp_stream_get_default(0, C.byref(d)); lib.lamp_stream_set_current(d)
        lib.lamp_stream_release(st); lib.lamp_stream_release(d)


@pytest.mark.gpu
@pytest.mark.parametrize("ldt,tol", [(S.F32, 1e-6), (S.F64, 1e-12), (S.BF16, 2.0 ** -7)], ids=["f32", "f64", "bf16"])
def test_replayed_resnet_step_tracks_the_weights(gpu, ldt, tol):
    """Cnn.resnet forward + backprop captured once and replayed while AdamW updates the weights eagerly between replays: the convolutions'
    packed weight images (implicit-GEMM, narrow, small - found in their caches during the capture, so the graph holds no pack launch) must
    follow every update (the optimisers' re-pack hook).  Three replayed steps == three eager steps."""
    import ctypes as C
    from lamp_amd._capi import lib
    B = 8
    lib.lamp_manual_seed(77)
    em = nn.resnet(100, 0.0, ldt)
    gm = nn.resnet(100, 0.0, ldt)
    gm.load([v.value for v in em.state])
    cw = S.STen.ones([100], ldt)
    tdt = {S.F32: torch.float32, S.F64: torch.float64, S.BF16: torch.bfloat16}[ldt]
    x = to_sten(O.closed_form(B * 3 * 32 * 32, 5, 1.0, torch.float32).reshape(B, 3, 32, 32).to(tdt))
    t = to_sten((torch.arange(B) * 7) % 100)
    emodel, gmodel = nn.SupervisedModel(em, nn.SupervisedModel.NLL, cw), nn.SupervisedModel(gm, nn.SupervisedModel.NLL, cw)
    eopt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-2, mixedPrecision=(ldt == S.BF16))([p.value for p in em.parameters])
    gopt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-2, mixedPrecision=(ldt == S.BF16))([p.value for p in gm.parameters])
    st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(st)); lib.lamp_stream_set_current(st)
    try:
        # step 1 eagerly on both (fills the caches), steps 2 - 4: eager vs replayed
        for model, opt in ((emodel, eopt), (gmodel, gopt)):
            n, g1 = model.addTotalLossAndReturnGradientsAndNumExamples(x, t, None)
            opt.step(g1, 1.0)
        for _ in range(3):
            n, ge = emodel.addTotalLossAndReturnGradientsAndNumExamples(x, t, None)
            eopt.step(ge, 1.0)
        lib.lamp_device_synchronize()
        lib.lamp_graph_begin_capture()
        n, grads = gmodel.addTotalLossAndReturnGradientsAndNumExamples(x, t, None)
        g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
        for _ in range(3):
            lib.lamp_graph_launch(g)
            gopt.step(grads, 1.0)
        lib.lamp_device_synchronize()
        for i, (a, b) in enumerate(zip(gm.state, em.state)):
            assert_close(to_torch(a.value), to_torch(b.value).double(), tol, f"state {i} after three replayed steps")
        lib.lamp_graph_release(g)
    finally:
        d = C.c_void_p(); lib.lamp_stream_get_default(0, C.byref(d)); lib.lamp_stream_set_current(d)
        lib.lamp_stream_release(st); lib.lamp_stream_release(d)


@pytest.mark.gpu
def test_backprop_recorded_into_a_graph_does_not_poison_later_eager_passes(gpu):
    """The gradient of a one-element loss starts from a cached constant 1 (per thread, device, dtype, stream).  A capture must not
    create that constant: its fill would only be RECORDED, and an eager backprop on the same stream before the first replay would
    start from uninitialised memory.  Here the very first backprop on a fresh stream in f64 happens inside a capture, the graph is
    never launched, and the eager pass that follows must still give the reference gradient."""
    import ctypes as C
    from lamp_amd import autograd as AG
    from lamp_amd._capi import lib
    x = O.closed_form(24, 3, 2.0, torch.float64).reshape(4, 6)
    st = C.c_void_p(); lib.lamp_stream_get_from_pool(1, 0, C.byref(st)); lib.lamp_stream_set_current(st)   # a stream no other test computes f64 losses on
    try:
        X = to_sten(x)                                         # uploads synchronise: not inside a capture

        def loss_and_grad():
            v = AG.param(X)
            (v * v).sum().backprop()
            return v.partialDerivative
        lib.lamp_device_synchronize()
        lib.lamp_graph_begin_capture()
        recorded = loss_and_grad()
        g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
        eager = loss_and_grad()                                 # the graph has not been launched
        lib.lamp_device_synchronize()
        assert_close(to_torch(eager), (2 * x).double(), 1e-12, "eager gradient after a capture")
        lib.lamp_graph_launch(g)
        lib.lamp_device_synchronize()
        assert_close(to_torch(recorded), (2 * x).double(), 1e-12, "gradient of the replayed graph")
        lib.lamp_graph_release(g)
    finally:
        d = C.c_void_p(); lib.lamp_stream_get_default(0, C.byref(d)); lib.lamp_stream_set_current(d)
        lib.lamp_stream_release(st); lib.lamp_stream_release(d)
