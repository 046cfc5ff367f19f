"""kNN, UMAP and attention entry points vs the oracle (lamp-knn, lamp-umap, Transformer composed attention)."""
import ctypes as C

import math

import numpy as np
import pytest
import torch

from lamp_amd import autograd as A
from lamp_amd import nn
from lamp_amd import sten as S
from lamp_amd._capi import lib, f64_array
from oracle import lamp_oracle as O
from tests.util import assert_close, to_sten, to_torch

pytestmark = pytest.mark.gpu


def _points(n, d, dtype):
    # 16 separated clusters + uniform jitter from a seeded generator: neighbour distances differ by far more
    # than the rounding noise of |q|^2 + |x|^2 - 2 q.x, so the k-NN SETS are well defined (no ties)
    g = torch.Generator().manual_seed(12345)
    jitter = torch.rand(n, d, generator=g, dtype=torch.float64)
    centers = (torch.arange(n) % 16).double().reshape(n, 1) * 3.0
    return (jitter + centers).to(dtype)


def _well_separated_queries(data64, candidates, k, gap):
    """keep the query rows whose k-th and (k+1)-th neighbour distances differ by more than `gap` (in f64), so that the
    k-NN SET is decided by the data and not by the rounding noise of the |q|^2 + |x|^2 - 2 q.x form."""
    d = O.squared_euclidean_distance(data64[candidates], data64)
    v, _ = torch.topk(d, k + 1, 1, largest=False, sorted=True)
    return candidates[(v[:, k] - v[:, k - 1]) > gap]


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_knn_matches_reference_algorithm(gpu, dt):
    data64 = _points(3000, 16, torch.float64)
    rows = _well_separated_queries(data64, torch.arange(400), 10, 2e-2)
    assert len(rows) > 100
    data, query = data64.to(dt), data64[rows].to(dt)
    ref = O.knn_minibatched(data, query, 10, 100)
    i, d = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(d), to_sten(data), to_sten(query), 10)
    I, D = S.STen(i), S.STen(d)
    got = I.to_numpy()
    assert got.dtype == np.int64 and got.shape == (len(rows), 10)
    assert np.array_equal(np.sort(got, 1), np.sort(ref.numpy(), 1)), "neighbour index sets must be exact"
    assert (np.sort(got, 1) == rows.numpy()[:, None]).any(1).all(), "self is among the neighbours (umap.scala:327-329)"
    # |q|^2 + |x|^2 - 2 q.x cancels catastrophically in f32 (|q|^2 ~ 3e4 here, in the reference as well), so the values are
    # checked against the exact f64 distances with the error bound of that formula: a few ulp of |q|^2
    exact = torch.gather(O.squared_euclidean_distance(data64[rows], data64), 1, torch.from_numpy(got))
    bound = (2e-2 if dt == torch.float32 else 1e-9)
    assert (to_torch(D).double() - exact).abs().max().item() <= bound, "distances"


def test_knn_chunked_merge(gpu):
    # k = 17 is beyond the fused kernel (k <= 16): column-chunked GEMM + top-k + merge of the per-chunk winners; 40000 points do
    # not fit the per-call distance block for 512 queries, so several chunks are merged
    data64 = _points(40000, 8, torch.float64)
    rows = _well_separated_queries(data64, torch.arange(0, 40000, 13)[:3000], 17, 1e-2)[:512]
    assert len(rows) > 100
    data, query = data64.float(), data64[rows].float()
    ref = O.knn_minibatched(data, query, 17, 128)
    lib.lamp_kernel_timer_enable(1)
    i = C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), None, to_sten(data), to_sten(query), 17)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    assert b"knn_fused" not in buf.value
    assert np.array_equal(np.sort(S.STen(i).to_numpy(), 1), np.sort(ref.numpy(), 1))


def test_knn_narrow_widths_are_padded_onto_the_fused_kernel(gpu):
    """feature widths other than 64 / 128 (here 8, 50, 100) are zero-padded: same neighbours as the reference algorithm"""
    for d in (8, 50, 100):
        g = torch.Generator().manual_seed(d)
        data64 = torch.rand(2000, d, generator=g, dtype=torch.float64) + (torch.arange(2000) % 16).double().reshape(2000, 1) * 0.25
        rows = _well_separated_queries(data64, torch.arange(0, 2000, 5), 10, 1e-2)
        assert len(rows) > 50
        ref = O.knn_minibatched(data64.float(), data64[rows].float(), 10, 100)
        lib.lamp_kernel_timer_enable(1)
        i = C.c_void_p()
        lib.lamp_knn_squared_euclidean(C.byref(i), None, to_sten(data64.float()), to_sten(data64[rows].float()), 10)
        buf = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_report(buf, len(buf))
        lib.lamp_kernel_timer_enable(0)
        assert b"knn_fused_f32" in buf.value
        assert np.array_equal(np.sort(S.STen(i).to_numpy(), 1), np.sort(ref.numpy(), 1))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("n,nq,d,k", [(3000, 400, 128, 10), (777, 257, 64, 16), (64, 130, 128, 1), (20000, 33, 128, 10), (100, 100, 64, 7)])
def test_knn_fused_topk(gpu, n, nq, d, k, dt):
    """f32 with 64 / 128 features and k <= 16 takes knn_fused.hip (top-k inside the distance GEMM): ragged tile counts, fewer
    points than one tile, k = 1 and k = 16, against the reference algorithm of the oracle on well separated rows."""
    # clusters 0.25 apart: |x|^2 stays below 3e3, so the f32 form |q|^2 + |x|^2 - 2 q.x is good to ~1e-3 and rows whose k-th and
    # (k+1)-th neighbours differ by 2e-2 have a k-NN SET decided by the data
    g = torch.Generator().manual_seed(777)
    data64 = torch.rand(n, d, generator=g, dtype=torch.float64) + (torch.arange(n) % 16).double().reshape(n, 1) * 0.25
    rows = _well_separated_queries(data64, torch.arange(0, n, max(n // nq, 1))[:nq], k, 2e-2 if dt == torch.float32 else 1e-9)
    assert len(rows) >= min(nq, n) // 4
    data, query = data64.to(dt), data64[rows].to(dt)
    ref = O.knn_minibatched(data, query, k, 100)
    lib.lamp_kernel_timer_enable(1)
    i, dd = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(dd), to_sten(data), to_sten(query), k)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    assert (b"knn_fused_f32" if dt == torch.float32 else b"knn_fused_f64") in buf.value, "the fused kernel did not run"
    got, dist = S.STen(i).to_numpy(), S.STen(dd).to_numpy()
    assert got.shape == (len(rows), k) and got.dtype == np.int64
    assert np.array_equal(np.sort(got, 1), np.sort(ref.numpy(), 1)), "neighbour index sets must be exact"
    assert (np.diff(dist, axis=1) >= 0).all(), "neighbours come sorted by distance"
    exact = torch.gather(O.squared_euclidean_distance(data64[rows], data64), 1, torch.from_numpy(got))
    assert (torch.from_numpy(dist).double() - exact).abs().max().item() <= (5e-3 if dt == torch.float32 else 1e-9)


@pytest.mark.parametrize("npdt", [np.float32, np.float64])
def test_knn_fused_ties_take_the_lower_index(gpu, npdt):
    """Exact duplicates have bit-identical distances: the lower index wins, inside one tile, across tiles and at the k-th place."""
    rng = np.random.default_rng(2)
    base = rng.random((50, 128)).astype(npdt)
    data = np.concatenate([base, base, base[:25], rng.random((200, 128)).astype(npdt) + 3.0])   # copies at +50 and +100
    query = base[:40]
    i, dd = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(dd), S.STen.from_numpy(data, 0), S.STen.from_numpy(query, 0), 2)
    got = S.STen(i).to_numpy()
    assert np.array_equal(got[:, 0], np.arange(40)) and np.array_equal(got[:, 1], np.arange(40) + 50)
    i3 = C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i3), None, S.STen.from_numpy(data, 0), S.STen.from_numpy(query, 0), 3)
    got3 = S.STen(i3).to_numpy()
    assert np.array_equal(got3[:25, 2], np.arange(25) + 100)


@pytest.mark.parametrize("d,k", [(128, 5), (40, 3), (200, 4)])
def test_knn_jaccard_and_host_api(gpu, d, k):
    """lamp.knn.JaccardDistance on binary feature vectors (fused kernel for d <= 128, composed blocks above) against the reference
    algorithm, and the host-side API (knnSearch / classification / regression) of lamp_amd.knn."""
    from lamp_amd import knn as K
    g = torch.Generator().manual_seed(3)
    n = 1500
    data = (torch.rand(n, d, generator=g) < 0.3).double()
    data[:, 0] = 1.0                                                  # no empty rows (0 / 0)
    dist = O.jaccard_distance(data, data)
    v, _ = torch.topk(dist, k + 1, 1, largest=False, sorted=True)
    rows = torch.arange(n)[(v[:, k] - v[:, k - 1]) > 1e-6][:400]      # the k-NN set is decided by the data
    assert len(rows) > 50
    ref = O.knn_minibatched(data, data[rows], k, 100, O.jaccard_distance).numpy()
    for precision, dt in (("f64", np.float64), ("f32", np.float32)):
        got = K.knnSearch(data.numpy(), data[rows].numpy(), k, K.JaccardDistance, 0, precision, 100)
        assert got.dtype == np.int32 and np.array_equal(np.sort(got, 1), np.sort(ref, 1)), precision
    # Euclidean through the same API + the host post-processing
    labels = (np.arange(n) % 3).astype(np.int64)
    idx = K.knnSearch(data.numpy(), data[:10].numpy(), 4, K.SquaredEuclideanDistance, 0, "f64")
    cls = K.classification(labels, idx, 3, False)
    assert cls.shape == (10, 3) and np.allclose(cls.sum(1), 1.0)
    assert np.allclose(K.regression(labels.astype(np.float64), idx), labels[idx].mean(1))
    assert np.allclose(K.classification(labels, idx, 3, True), np.log(cls + 1e-6))


@pytest.mark.parametrize("min_dist", [0.0, 0.3])
def test_umap_fused_loss_and_gradient(gpu, min_dist):
    n, e1, e2 = 200, 700, 3000
    loc = _points(n, 2, torch.float64) * 0.1
    g = torch.Generator().manual_seed(0)
    i1 = torch.randint(0, n, (e1,), generator=g); i2 = (i1 + 1 + torch.randint(0, n - 1, (e1,), generator=g)) % n
    i3 = torch.randint(0, n, (e2,), generator=g); i4 = (i3 + 1 + torch.randint(0, n - 1, (e2,), generator=g)) % n
    b = torch.rand(e1, generator=g, dtype=torch.float64)
    lv = O.param(loc.clone())
    L = O.umap_loss(lv, i1, i2, i3, i4, b, minDist=min_dist)
    L.backprop()
    # the reference accumulates the four IndexSelect backwards as out = 2*out + scatter: weights 1, 2, 4, 8
    grad = S.STen.zeros([n, 2], S.F64)
    lo = C.c_void_p()
    lib.lamp_umap_loss_grad(C.byref(lo), grad, to_sten(loc), to_sten(i1), to_sten(i2), to_sten(b), to_sten(i3), to_sten(i4), min_dist, 1, 1.0,
                            f64_array([1.0, 2.0, 4.0, 8.0]))
    assert_close(to_torch(S.STen(lo)).reshape(()), L.value.double().reshape(()), 1e-12, "loss")
    assert_close(to_torch(grad), lv.grad.double(), 1e-10, "gradient (reference accumulation order)")
    # and the op-by-op HIP path agrees with both
    hv = A.param(to_sten(loc))
    from tests.test_apps_helpers import hip_umap_loss
    HL = hip_umap_loss(hv, to_sten(i1), to_sten(i2), to_sten(i3), to_sten(i4), to_sten(b), min_dist)
    HL.backprop()
    assert_close(to_torch(HL.value).reshape(()), L.value.double().reshape(()), 1e-12, "op-by-op loss")
    assert_close(to_torch(hv.partialDerivative), lv.grad.double(), 1e-10, "op-by-op gradient")


def test_umap_layout_iterations_follow_the_oracle(gpu):
    """umap.scala:238-283: zeroGrad, backprop, AdamW(lr 0.1, wd 0, clip 1, beta2 0.95).step - 5 iterations."""
    n, e1, e2 = 120, 400, 2000
    loc0 = _points(n, 2, torch.float64) * 0.05
    g = torch.Generator().manual_seed(1)
    i1 = torch.randint(0, n, (e1,), generator=g); i2 = (i1 + 1 + torch.randint(0, n - 1, (e1,), generator=g)) % n
    i3 = torch.randint(0, n, (e2,), generator=g); i4 = (i3 + 1 + torch.randint(0, n - 1, (e2,), generator=g)) % n
    b = torch.rand(e1, generator=g, dtype=torch.float64)
    lo = O.param(loc0.clone())
    oopt = O.AdamW([lo.value], 0.0, 0.1, 0.9, 0.95, clip=1.0)
    H = to_sten(loc0)
    hopt = nn.AdamW([H], 0.0, 0.1, 0.9, 0.95, clip=1.0)
    grad = S.STen.zeros([n, 2], S.F64)
    I1, I2, I3, I4, Bt = (to_sten(t) for t in (i1, i2, i3, i4, b))
    for _ in range(5):
        lo.zeroGrad()
        L = O.umap_loss(lo, i1, i2, i3, i4, b)
        L.backprop()
        oopt.step([lo.grad], 1.0)
        grad.zero_()
        out = C.c_void_p()
        lib.lamp_umap_loss_grad(C.byref(out), grad, H, I1, I2, Bt, I3, I4, 0.0, 1, 1.0, f64_array([1.0, 2.0, 4.0, 8.0]))
        hopt.step([grad], 1.0)
        assert_close(to_torch(S.STen(out)).reshape(()), L.value.double().reshape(()), 1e-9, "loss along the trajectory")
    assert_close(to_torch(H), lo.value.double(), 1e-8, "layout after 5 iterations")


@pytest.mark.parametrize("dt", [torch.float64, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
def test_attention_matches_composed_softmax(gpu, dt, causal):
    Bz, H, Sq, D = 2, 3, 40, 16
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(dt) for _ in range(3))
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    scores = qd @ kd.transpose(-1, -2) / np.sqrt(D)
    if causal:
        scores = scores.masked_fill(torch.triu(torch.ones(Sq, Sq, dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(scores, -1) @ vd
    lse_ref = torch.logsumexp(scores, -1)
    o, l = C.c_void_p(), C.c_void_p()
    lib.lamp_scaled_dot_product_attention(C.byref(o), C.byref(l), to_sten(q), to_sten(k), to_sten(v), int(causal), 0.0)
    Ot, Lt = S.STen(o), S.STen(l)
    tol = {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 3e-2}[dt]
    assert_close(to_torch(Ot), ref.detach(), tol, "attention output")
    assert_close(to_torch(Lt), lse_ref.detach(), tol, "logsumexp")
    go = torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(dt)
    ref.backward(go.double())
    out3 = (C.c_void_p * 3)()
    lib.lamp_scaled_dot_product_attention_backward(out3, to_sten(go), to_sten(q), to_sten(k), to_sten(v), Ot, Lt, int(causal), 0.0)
    btol = {torch.float64: 1e-10, torch.float32: 1e-3, torch.bfloat16: 6e-2}[dt]
    for name, h, r in zip(("dq", "dk", "dv"), out3, (qd.grad, kd.grad, vd.grad)):
        assert_close(to_torch(S.STen(h)), r, btol, name)


@pytest.mark.parametrize("shape", [(2, 3, 200, 200, 64), (1, 2, 130, 333, 128), (2, 2, 64, 64, 128), (1, 1, 1, 70, 64), (1, 2, 257, 96, 64)])
@pytest.mark.parametrize("causal", [False, True])
def test_flash_attention_bf16(gpu, shape, causal):
    """bf16 with head dim 64 / 128 takes the fused flash kernel (attention.hip): ragged Sq / Sk, Sq != Sk, causal masks that
    leave whole key tiles unused.  Checked against an f64 softmax(QK^T)V of the same bf16-rounded inputs."""
    Bz, H, Sq, Sk, D = shape
    g = torch.Generator().manual_seed(11)
    q = torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(torch.bfloat16)
    k, v = (torch.randn(Bz, H, Sk, D, generator=g, dtype=torch.float64).to(torch.bfloat16) for _ in range(2))
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    scores = qd @ kd.transpose(-1, -2) / np.sqrt(D)
    if causal:
        scores = scores.masked_fill(torch.triu(torch.ones(Sq, Sk, dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(scores, -1) @ vd
    lse_ref = torch.logsumexp(scores, -1)
    lib.lamp_kernel_timer_enable(1)
    o, l = C.c_void_p(), C.c_void_p()
    lib.lamp_scaled_dot_product_attention(C.byref(o), C.byref(l), to_sten(q), to_sten(k), to_sten(v), int(causal), 0.0)
    Ot, Lt = S.STen(o), S.STen(l)
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    assert b"sdpa_flash_fwd" in buf.value, "the fused kernel did not run"
    assert_close(to_torch(Ot), ref.detach(), 2e-2, "attention output")
    assert_close(to_torch(Lt), lse_ref.detach(), 2e-2, "logsumexp")
    go = torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(torch.bfloat16)
    ref.backward(go.double())
    out3 = (C.c_void_p * 3)()
    lib.lamp_kernel_timer_enable(1)
    lib.lamp_scaled_dot_product_attention_backward(out3, to_sten(go), to_sten(q), to_sten(k), to_sten(v), Ot, Lt, int(causal), 0.0)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    assert b"sdpa_flash_bwd_dq" in buf.value and b"sdpa_flash_bwd_dkv" in buf.value, "the fused backward did not run"
    assert Lt.dtype == S.F32
    for name, h, r in zip(("dq", "dk", "dv"), out3, (qd.grad, kd.grad, vd.grad)):
        assert_close(to_torch(S.STen(h)), r, 2e-2, name, scale="max")     # sums over S of bf16-rounded P / dS: the error scales with sqrt(S) * 2^-9 * |terms|, not with the element


@pytest.mark.parametrize("shape", [(3, 37, 12), (2, 5, 16), (1, 9, 1), (2, 3, 5)])
@pytest.mark.parametrize("causal", [False, True])
def test_short_sequence_attention_bf16(gpu, shape, causal):
    """S <= 16 with head width 64 (the problems lamp's MultiheadAttention actually hands the fused operator: one per token, over the
    head axis) take the wave-per-problem kernels of attention_small.hip; output, logsumexp and the three gradients against f64."""
    Bz, H, Sq = shape
    D = 64
    g = torch.Generator().manual_seed(5)
    q, k, v, go = (torch.randn(Bz, H, Sq, D, generator=g, dtype=torch.float64).to(torch.bfloat16) for _ in range(4))
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    scores = qd @ kd.transpose(-1, -2) / np.sqrt(D)
    if causal:
        scores = scores.masked_fill(torch.triu(torch.ones(Sq, Sq, dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(scores, -1) @ vd
    lse_ref = torch.logsumexp(scores, -1)
    ref.backward(go.double())
    buf = C.create_string_buffer(1 << 16)
    lib.lamp_kernel_timer_enable(1)
    o, l = C.c_void_p(), C.c_void_p()
    lib.lamp_scaled_dot_product_attention(C.byref(o), C.byref(l), to_sten(q), to_sten(k), to_sten(v), int(causal), 0.0)
    Ot, Lt = S.STen(o), S.STen(l)
    out3 = (C.c_void_p * 3)()
    lib.lamp_scaled_dot_product_attention_backward(out3, to_sten(go), to_sten(q), to_sten(k), to_sten(v), Ot, Lt, int(causal), 0.0)
    lib.lamp_kernel_timer_report(buf, len(buf))
    lib.lamp_kernel_timer_enable(0)
    assert b"sdpa_small_fwd" in buf.value and b"sdpa_small_bwd" in buf.value, "the short-sequence kernels did not run"
    assert_close(to_torch(Ot), ref.detach(), 1e-2, "attention output")
    assert_close(to_torch(Lt), lse_ref.detach(), 1e-5, "logsumexp")
    for name, h, r in zip(("dq", "dk", "dv"), out3, (qd.grad, kd.grad, vd.grad)):
        assert_close(to_torch(S.STen(h)), r, 1e-2, name, scale="max")     # dS is rounded to bf16 before the dq / dk products


def test_umap_skip_self_equals_masked_pairs(gpu):
    """lamp_umap_loss_grad_skip_self folds `mask = i.ne(j); i.maskedSelect(mask); j.maskedSelect(mask)` (umap.scala:221-227) into
    the kernel: same loss and gradient as the explicit compaction, for 2-D (lane-pair kernel) and 3-D (generic kernel) layouts."""
    rng = np.random.default_rng(4)
    n, e1, e2 = 300, 1500, 6000
    for dim in (2, 3):
        loc = S.STen.from_numpy(rng.random((n, dim)), 0, S.F64)
        a1 = np.sort(rng.integers(0, n, e1)); a2 = (a1 + 1 + rng.integers(0, n - 1, e1)) % n
        a3 = np.repeat(a1, 4)[:e2]; a4 = rng.integers(0, n, e2); a4[::7] = a3[::7]          # every 7th negative hits itself
        b = S.STen.from_numpy(rng.random(e1), 0, S.F64)
        w = f64_array([1.0, 2.0, 4.0, 8.0])
        keep = a3 != a4
        res = []
        for fn, i3, i4 in ((lib.lamp_umap_loss_grad, a3[keep], a4[keep]), (lib.lamp_umap_loss_grad_skip_self, a3, a4)):
            g = S.STen.zeros([n, dim], S.F64)
            lo = C.c_void_p()
            fn(C.byref(lo), g, loc, S.STen.from_numpy(a1, 0), S.STen.from_numpy(a2, 0), b, S.STen.from_numpy(i3, 0), S.STen.from_numpy(i4, 0), 0.0, 1, 1.5, w)
            res.append((float(S.STen(lo).to_numpy()), g.to_numpy()))
        assert abs(res[0][0] - res[1][0]) <= 1e-12 * abs(res[0][0])
        assert np.abs(res[0][1] - res[1][1]).max() <= 1e-12 * np.abs(res[0][1]).max()


@pytest.mark.parametrize("min_dist", [0.0, 0.1])
def test_umap_negatives_drawn_inside_the_kernel(gpu, min_dist):
    """Round 6: lamp_umap_loss_grad_sampled draws Umap.optimize's negatives (ii = index1.repeatInterleave(n), jj = randint(0, total - 1),
    umap.scala:211-213) inside the layout kernel from a counter-based generator; lamp_umap_negatives writes the SAME draws out when it takes the
    generator at the same point.  So: loss and gradient of the sampled call == lamp_umap_loss_grad_skip_self on the materialised negatives (the
    self-hits dropped and the repulsion normalised by the pairs kept, both), for 2-D (drawn in the kernel) and 3-D (written out inside the entry
    point) layouts; the draws are in [0, high), cover it uniformly, and differ between edges, between the negatives of an edge and between calls."""
    rng = np.random.default_rng(9)
    n, e1, neg = 257, 4000, 5
    a1 = np.sort(rng.integers(0, n, e1)); a2 = (a1 + 1 + rng.integers(0, n - 1, e1)) % n
    I1, I2 = S.STen.from_numpy(a1, 0), S.STen.from_numpy(a2, 0)
    b = S.STen.from_numpy(rng.random(e1), 0, S.F64)
    w = f64_array([1.0, 2.0, 4.0, 8.0])
    for dim in (2, 3):
        loc = S.STen.from_numpy(rng.random((n, dim)), 0, S.F64)
        lib.lamp_manual_seed(77)
        ii, jj = C.c_void_p(), C.c_void_p()
        lib.lamp_umap_negatives(C.byref(ii), C.byref(jj), I1, neg, n - 1)
        II, JJ = S.STen(ii), S.STen(jj)
        ii_np, jj_np = II.to_numpy(), JJ.to_numpy()
        assert np.array_equal(ii_np, np.repeat(a1, neg))
        assert jj_np.min() >= 0 and jj_np.max() == n - 2                      # randint's high is exclusive (the reference never draws the last point)
        counts = np.bincount(jj_np, minlength=n - 1)
        assert counts.min() > 0 and abs(counts.mean() - e1 * neg / (n - 1)) < 1e-9 and counts.std() < 3.0 * np.sqrt(e1 * neg / (n - 1))
        assert (ii_np == jj_np).sum() > 0                                    # some negatives do hit their own point: the case the kernel must drop
        per_edge = jj_np.reshape(e1, neg)
        assert (per_edge[:, 0] != per_edge[:, 1]).mean() > 0.98 and (per_edge[0] != per_edge[1]).any()
        g_ref = S.STen.zeros([n, dim], S.F64)
        lo = C.c_void_p()
        lib.lamp_umap_loss_grad_skip_self(C.byref(lo), g_ref, loc, I1, I2, b, II, JJ, min_dist, 1, 1.5, w)
        l_ref = float(S.STen(lo).to_numpy())
        lib.lamp_manual_seed(77)
        g = S.STen.zeros([n, dim], S.F64)
        lo2 = C.c_void_p()
        lib.lamp_umap_loss_grad_sampled(C.byref(lo2), g, loc, I1, I2, b, neg, n - 1, min_dist, 1, 1.5, w)
        l = float(S.STen(lo2).to_numpy())
        assert abs(l - l_ref) <= 1e-12 * abs(l_ref), (dim, l, l_ref)
        assert np.abs(g.to_numpy() - g_ref.to_numpy()).max() <= 1e-11 * np.abs(g_ref.to_numpy()).max()
        # the next call takes the generator further: other negatives
        ii2, jj2 = C.c_void_p(), C.c_void_p()
        lib.lamp_umap_negatives(C.byref(ii2), C.byref(jj2), I1, neg, n - 1)
        assert (S.STen(jj2).to_numpy() != jj_np).mean() > 0.9
    with pytest.raises(Exception, match="negatives per edge"):
        lib.lamp_umap_negatives(C.byref(ii), C.byref(jj), I1, 0, n - 1)


def test_umap_sharded_layout_building_blocks(gpu):
    """Edge list split over ranks: the sum over the shards of lamp_umap_loss_grad_sharded (global normalisers handed in) equals the
    unsharded loss / gradient; with one rank optimize_sharded reproduces optimize."""
    from lamp_amd import umap as U
    rng = np.random.default_rng(8)
    n, e1 = 400, 2400
    loc = S.STen.from_numpy(rng.random((n, 2)), 0, S.F64)
    a1 = np.sort(rng.integers(0, n, e1)); a2 = (a1 + 1 + rng.integers(0, n - 1, e1)) % n
    bb = rng.random(e1)
    a3 = np.repeat(a1, 3); a4 = rng.integers(0, n, a3.shape[0]); a4[::11] = a3[::11]
    w = f64_array([1.0, 2.0, 4.0, 8.0])
    T = lambda a, dt=None: S.STen.from_numpy(a, 0, dt)
    g_all = S.STen.zeros([n, 2], S.F64)
    lo = C.c_void_p()
    lib.lamp_umap_loss_grad_skip_self(C.byref(lo), g_all, loc, T(a1), T(a2), T(bb, S.F64), T(a3), T(a4), 0.0, 1, 1.0, w)
    loss_all = float(S.STen(lo).to_numpy())
    kept = C.c_void_p()
    lib.lamp_count_ne(C.byref(kept), T(a3), T(a4))
    kept = S.STen(kept)
    assert int(kept.to_numpy()[0]) == int((a3 != a4).sum())
    bsum = T(np.array([bb.sum()]), S.F64)
    world = 3
    g_sum, loss_sum = np.zeros((n, 2)), 0.0
    for r in range(world):
        e = np.arange(r, e1, world)
        neg = np.concatenate([np.arange(3 * i, 3 * i + 3) for i in e])
        g = S.STen.zeros([n, 2], S.F64)
        lo = C.c_void_p()
        lib.lamp_umap_loss_grad_sharded(C.byref(lo), g, loc, T(a1[e]), T(a2[e]), T(bb[e], S.F64), T(a3[neg]), T(a4[neg]), 0.0, 1, 1.0, w, bsum, kept)
        g_sum += g.to_numpy(); loss_sum += float(S.STen(lo).to_numpy())
    assert abs(loss_sum - loss_all) <= 1e-11 * abs(loss_all)
    assert np.abs(g_sum - g_all.to_numpy()).max() <= 1e-11 * np.abs(g_all.to_numpy()).max()
    # one rank: the sharded driver is the plain one
    knn_idx = (np.arange(n)[:, None] + 1 + rng.integers(0, n - 1, (n, 6))) % n
    knn_idx[:, 0] = np.arange(n)
    dist = np.sort(rng.random((n, 6)), 1); dist[:, 0] = 0.0
    ew = U.edge_weights(T(dist, S.F64), T(knn_idx.astype(np.int64)))
    l1, v1 = U.optimize(ew, n, 0.1, 5, 0.0, 5, 42, True, 1.0, 0, 2)
    l2, v2 = U.optimize_sharded(ew, n, 0.1, 5, 0.0, 5, 42, True, 1.0, 0, 2, None, 1, 0)
    assert abs(v1 - v2) <= 1e-9 * abs(v1) and np.abs(l1.to_numpy() - l2.to_numpy()).max() <= 1e-9


def test_knn_and_umap_at_full_size_properties(gpu):
    """BASELINE config 5 at its full size (1M x 128 f32 points, k = 10): size-independent properties instead of an oracle.
    kNN (262,144 query rows against all 1M points): the query itself is a neighbour at distance ~0, indices are in range and
    distinct per row, and no sampled non-neighbour is closer than the k-th neighbour.  UMAP layout (1M points, 10M attractive
    + 50M repulsive pairs): with unit term weights every pair pushes its two end points with opposite forces, so the
    gradient sums to zero over the points."""
    n, d, k, nq = 1_000_000, 128, 10, 262_144
    rng = np.random.default_rng(5)
    pts = rng.random((n, d), dtype=np.float32) + (np.arange(n) % 16)[:, None].astype(np.float32)
    data = S.STen.from_numpy(pts, 0, S.F32)
    rows = np.sort(rng.choice(n, nq, replace=False))
    query = S.STen.from_numpy(pts[rows], 0, S.F32)
    i, dd = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(dd), data, query, k)
    idx, dist = S.STen(i).to_numpy(), S.STen(dd).to_numpy()
    assert idx.shape == (nq, k) and idx.min() >= 0 and idx.max() < n
    assert (idx == rows[:, None]).any(1).all(), "self is among the neighbours"
    assert (np.sort(idx, 1)[:, 1:] != np.sort(idx, 1)[:, :-1]).all(), "neighbours are distinct"
    # |q|^2 + |x|^2 - 2qx in f32 with |x|^2 ~ 1e4: absolute error of a few 1e-3
    self_pos = (idx == rows[:, None]).argmax(1)
    assert np.abs(dist[np.arange(nq), self_pos]).max() < 5e-2
    probe = rng.integers(0, n, nq)
    sample = rng.choice(nq, 4096, replace=False)
    exact = ((pts[rows[sample]].astype(np.float64) - pts[probe[sample]].astype(np.float64)) ** 2).sum(1)
    kth = dist[sample].max(1).astype(np.float64)
    in_set = (idx[sample] == probe[sample, None]).any(1)
    assert (in_set | (exact >= kth - 5e-2)).all(), "a non-neighbour is never closer than the k-th neighbour"

    # UMAP layout step on the full edge list shape
    ne = n * k
    loc = S.STen.from_numpy(rng.random((n, 2)), 0, S.F64)
    a1 = np.repeat(np.arange(n, dtype=np.int64), k)
    a3 = rng.integers(0, n, ne * 5).astype(np.int64)
    i1 = S.STen.from_numpy(a1, 0)
    i2 = S.STen.from_numpy((a1 + 1 + rng.integers(0, n - 1, ne)) % n, 0)             # never a self pair (log 0)
    b = S.STen.from_numpy(rng.random(ne), 0, S.F64)
    i3 = S.STen.from_numpy(a3, 0)
    i4 = S.STen.from_numpy((a3 + 1 + rng.integers(0, n - 1, ne * 5)) % n, 0)
    grad = S.STen.zeros([n, 2], S.F64, 0)
    w = (C.c_double * 4)(1.0, 1.0, 1.0, 1.0)
    o = C.c_void_p()
    lib.lamp_umap_loss_grad(C.byref(o), grad, loc, i1, i2, b, i3, i4, 0.1, 1, 1.0, w)
    g = grad.to_numpy()
    assert np.isfinite(g).all() and np.isfinite(S.STen(o).to_numpy()).all()
    assert np.abs(g.sum(0)).max() <= 1e-9 * np.abs(g).sum(), "pairwise forces cancel"

    # UMAP edge weights on a 1M x 10 neighbour table (self in column 0): every non-self neighbour yields one row, rows are in
    # emission order, and sampled rows agree with the oracle's rho / bisection / fuzzy union computed for those rows only
    kd = np.sort(rng.random((n, k)), axis=1); kd[:, 0] = 0.0
    ki = rng.integers(0, n, (n, k)).astype(np.int64); ki[:, 0] = np.arange(n)
    o = C.c_void_p()
    lib.lamp_umap_edge_weights(C.byref(o), S.STen.from_numpy(kd, 0, S.F64), S.STen.from_numpy(ki, 0))
    ew = S.STen(o).to_numpy()
    keep = ki != np.arange(n)[:, None]
    assert ew.shape == (int(keep.sum()), 3)
    assert np.array_equal(ew[:, 0], np.repeat(np.arange(n), k)[keep.reshape(-1)].astype(np.float64))
    assert np.array_equal(ew[:, 1], ki[keep].astype(np.float64))
    assert (ew[:, 2] >= 0).all() and (ew[:, 2] <= 1 + 1e-12).all()
    log2k = math.log(k) / math.log(2.0)

    def rho_sigma(row):
        r = min(d for d in kd[row] if d > 0)
        return r, O._binary_search(log2k, lambda s_: sum(math.exp((-1 * max(0.0, d - r)) / s_) for d in kd[row]))
    flat_pos = np.flatnonzero(keep.reshape(-1))
    for e in rng.choice(len(flat_pos), 200, replace=False):
        i, jidx = divmod(int(flat_pos[e]), k)
        j = int(ki[i, jidx])
        ri, si = rho_sigma(i)
        wij = math.exp((-1 * max(0.0, kd[i, jidx] - ri)) / si)
        wji = 0.0
        hits = np.flatnonzero(ki[j] == i)
        if len(hits):
            rj, sj = rho_sigma(j)
            wji = math.exp((-1 * max(0.0, kd[j, hits[0]] - rj)) / sj)
        assert abs(ew[e, 2] - (wij + wji - wij * wji)) < 1e-9


def test_umap_edge_weights_reference_kat_and_oracle(gpu):
    """lamp_umap_edge_weights vs the reference's KAT (umap.test.scala:10-53, the oracle is pinned to it in test_oracle_kat.py)
    and vs the oracle on a 600-point kNN graph (rho / bisection sigma / fuzzy union, rows in emission order)."""
    data = torch.tensor([[1.0, 4.0], [2.0, 5.0], [3.0, 6.0]], dtype=torch.float64)
    knn = O.knn_minibatched(data, data, 3, 100)
    d2 = O.squared_euclidean_distance(data, data)
    knn = torch.stack([row[torch.argsort(d2[i][row], stable=True)] for i, row in enumerate(knn)])
    dist = torch.tensor([[float(torch.linalg.vector_norm(data[i] - data[j])) for j in knn[i]] for i in range(3)], dtype=torch.float64)
    o = C.c_void_p()
    lib.lamp_umap_edge_weights(C.byref(o), to_sten(dist), to_sten(knn))
    got = S.STen(o).to_numpy()
    exp = np.array([(0., 1., 1.), (0., 2., 0.), (1., 0., 1.), (1., 2., 1.), (2., 1., 1.), (2., 0., 0.)])
    assert got.shape == exp.shape and np.array_equal(got[:, :2], exp[:, :2]) and np.allclose(got[:, 2], exp[:, 2], atol=1e-6), got

    pts = _points(600, 5, torch.float64)
    idx = O.knn_minibatched(pts, pts, 8, 100)
    dd = torch.stack([torch.linalg.vector_norm(pts[i] - pts[idx[i]], dim=1) for i in range(600)])
    rows = np.array(O.edge_weights(dd.tolist(), idx.tolist()))
    o = C.c_void_p()
    lib.lamp_umap_edge_weights(C.byref(o), to_sten(dd), to_sten(idx))
    got = S.STen(o).to_numpy()
    assert got.shape == rows.shape
    assert np.array_equal(got[:, :2], rows[:, :2]), "edge list (i, j) and its order"
    assert np.abs(got[:, 2] - rows[:, 2]).max() < 1e-9


def test_umap_pipeline_end_to_end(gpu):
    """lamp_amd.umap.umap mirrors Umap.umap: with 0 iterations the graph b equals the oracle's edge weights on the oracle's kNN
    (exact f64 distances); with iterations the loss goes down, runs are reproducible for a seed, and graph neighbours end up much
    closer in the layout than random pairs."""
    from lamp_amd import umap as U
    rng = np.random.default_rng(3)
    centers = np.array([[0.0] * 6, [8.0] * 6, [-8.0, 8.0, -8.0, 8.0, -8.0, 8.0]])
    data = np.concatenate([c + rng.standard_normal((60, 6)) * 0.5 for c in centers])
    t = torch.from_numpy(data)
    idx = O.knn_minibatched(t, t, 6, 50)
    d2 = O.squared_euclidean_distance(t, t)
    idx = torch.stack([row[torch.argsort(d2[i][row], stable=True)] for i, row in enumerate(idx)])
    dist = torch.stack([torch.linalg.vector_norm(t[i] - t[idx[i]], dim=1) for i in range(len(t))])
    ref_rows = {(int(a), int(b)): w for a, b, w in O.edge_weights(dist.tolist(), idx.tolist())}
    layout0, b, _ = U.umap(data, k=6, iterations=0, randomSeed=7)
    got = b.to_numpy()
    assert len(got) == len(ref_rows)
    for i, j, w in got:
        assert abs(ref_rows[(int(i), int(j))] - w) < 1e-9
    assert layout0.shape == [180, 2]

    losses = []
    layout, b2, last = U.umap(data, k=6, iterations=150, lr=0.1, randomSeed=7, log=lambda s: losses.append(float(s.split(",")[-1].strip(" )"))))
    layout_again, _, last_again = U.umap(data, k=6, iterations=150, lr=0.1, randomSeed=7)
    # same seed, same samples; the f64 atomic scatter-adds of the gradient are summed in a run-dependent order (as index_add is
    # on any GPU), so agreement is to rounding, not bitwise
    assert np.abs(layout.to_numpy() - layout_again.to_numpy()).max() < 1e-3 and abs(last - last_again) < 1e-4 * max(1.0, abs(last)), \
        (np.abs(layout.to_numpy() - layout_again.to_numpy()).max(), last, last_again)
    assert len(losses) == 150 and np.mean(losses[-10:]) < 0.8 * np.mean(losses[:10]), "the loss (negated objective, umap.scala:163-175) goes down"
    L = layout.to_numpy()
    e = b2.to_numpy()
    edge_len = np.linalg.norm(L[e[:, 0].astype(int)] - L[e[:, 1].astype(int)], axis=1).mean()
    ri, rj = rng.integers(0, 180, 4000), rng.integers(0, 180, 4000)
    rand_len = np.linalg.norm(L[ri] - L[rj], axis=1).mean()
    assert edge_len < 0.5 * rand_len, (edge_len, rand_len)      # graph neighbours end up close, random pairs do not


def _mnist_fixture():
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "mnist_first1000.npz"))
    return d["pixels"].astype(np.float64), d["labels"].astype(np.int64)


def test_umap_mnist_meets_the_reference_acceptance(gpu):
    """The reference's own end-to-end UMAP test (umap.test.scala:57-79, "mnist"): the first 1000 MNIST records, numDim = 2,
    positiveSamples = 5000, negativeSampleSize = 5, 1000 iterations -> `assert(loss < 0.7)`.  The reference reads /mnist_train.csv.gz,
    which is not in its tree; the fixture holds the first 1000 records of the MNIST resource it does carry
    (lamp-core/src/test/resources/mnist_test.csv.gz, scripts/make_mnist_fixture.py).

    What that assertion looks at is the loss of the LAST iteration - one random subsample of 5000 edges and 25000 negatives - and
    neither random stream of the reference (Cmwc5 for the initial layout, libtorch for the samples) can be reproduced.  So (VERDICT r2
    item 5): the ORACLE runs the same recipe with its own generator over five seeds, the HIP path over the same five seeds, and
      * the mean of the last 50 iterations meets the reference's 0.7 for the median HIP seed (the bar, freed of the single-sample noise;
        on these records - the first 1000 of the TEST set, the train set is not in the reference's tree - the oracle itself ends above
        0.7 for one seed in five: 0.66 ... 0.73, measured HIP 0.67 ... 0.72),
      * the HIP distribution of that mean lies inside the oracle's (same mean within 0.02, i.e. ~1.5 of the oracle's own seed-to-seed
        standard deviations),
      * the last-iteration loss stays within the oracle's own last-iteration range widened by its noise (3 sigma of one run)."""
    import torch
    from lamp_amd import umap as U
    data, labels = _mnist_fixture()
    t = torch.from_numpy(data)
    idx = O.knn_minibatched(t, t, 10, 1000)
    d2 = O.squared_euclidean_distance(t, t)
    idx = torch.stack([row[torch.argsort(d2[i][row], stable=True)] for i, row in enumerate(idx)])
    dist = torch.stack([torch.linalg.vector_norm(t[i] - t[idx[i]], dim=1) for i in range(len(t))])
    rows = torch.tensor(O.edge_weights(dist.tolist(), idx.tolist()), dtype=torch.float64)
    seeds = (42, 1, 7, 11, 123)
    o_mean, o_last, o_std = [], [], []
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 8))       # 30k-element ops: ATen's OpenMP loops crawl on the 128+ threads of a GPU host
    try:
        for seed in seeds:
            losses = []
            O.umap_optimize(rows, 1000, 0.1, 1000, 0.0, 5, seed, positiveSamples=5000, losses=losses)
            o_mean.append(float(np.mean(losses[-50:]))); o_last.append(losses[-1]); o_std.append(float(np.std(losses[-50:])))
    finally:
        torch.set_num_threads(threads)
    h_mean, h_last, lay = [], [], None
    for seed in seeds:
        losses = []
        layout, b, loss = U.umap(data, numDim=2, positiveSamples=5000, negativeSampleSize=5, iterations=1000, randomSeed=seed, lossHistory=losses)
        assert len(losses) == 50 and abs(losses[-1] - float(loss)) < 1e-12
        h_mean.append(float(np.mean(losses[-50:]))); h_last.append(losses[-1])
        if seed == 42:
            lay = layout.to_numpy()
            assert np.array_equal(b.to_numpy()[:, :2], rows.numpy()[:, :2]) and np.abs(b.to_numpy()[:, 2] - rows.numpy()[:, 2]).max() < 1e-9, "same UMAP graph as the oracle"
    print(f"UMAP MNIST, mean loss of the last 50 iterations: HIP {np.round(h_mean, 4).tolist()} oracle {np.round(o_mean, 4).tolist()}; "
          f"last iteration: HIP {np.round(h_last, 4).tolist()} oracle {np.round(o_last, 4).tolist()}; oracle noise of one run {np.mean(o_std):.4f}")
    assert float(np.median(h_mean)) < 0.7, f"the reference's bar on the mean of the last 50 iterations: {h_mean}"
    assert abs(np.mean(h_mean) - np.mean(o_mean)) < 0.02, (h_mean, o_mean)
    assert min(o_mean) - 0.02 <= min(h_mean) and max(h_mean) <= max(o_mean) + 0.02, (h_mean, o_mean)
    # (the same 0.02 of slack as for the means: a single iteration's loss of five seeds strays three of ITS OWN sigmas below the oracle's five now
    # and then - round 6 saw 0.616 against a bound of 0.619 in one run of two, the float atomics' order being the only difference between them)
    noise = 3.0 * max(o_std) + 0.02
    assert min(o_last) - noise <= min(h_last) and max(h_last) <= max(o_last) + noise, (h_last, o_last, noise)
    assert lay.shape == (1000, 2) and np.isfinite(lay).all()
    # the embedding means something: most of a point's 10 nearest neighbours in the layout carry its digit
    d2l = ((lay[:, None, :] - lay[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2l, np.inf)
    nn = np.argsort(d2l, 1)[:, :10]
    purity = (labels[nn] == labels[:, None]).mean()
    assert purity > 0.6, f"neighbour purity {purity}"                  # measured 0.74


# ---- large f32 / f64 searches through the f16 matrix pipe (kernels/knn_split.hip, round 3) ---------------------------------------------------
def _knn(data, query, k):
    i, d = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(d), to_sten(data), to_sten(query), k)
    return S.STen(i).to_numpy(), S.STen(d).to_numpy()


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("n,nq,d,k", [(20000, 1500, 128, 10), (5000, 700, 64, 12), (16, 40, 128, 3), (3001, 257, 128, 1), (70, 300, 64, 10)])
def test_knn_split_filter_returns_the_exact_search(gpu, n, nq, d, k, dt):
    """knn_split.hip forced on small problems (mode 2): standard-normal data, where the filter proves (nearly) every query, against the exact
    fused kernel (mode 0) and the oracle.  Index sets equal on every row whose k-th / (k+1)-th neighbours are separated by more than the
    f32 formula's own noise; distances within that noise; rows sorted by distance."""
    g = torch.Generator().manual_seed(4242 + n)
    data64 = torch.randn(n, d, generator=g, dtype=torch.float64)
    rows = _well_separated_queries(data64, torch.arange(0, n, max(n // nq, 1))[:nq], min(k, n - 1), 1e-3) if n > k else torch.arange(min(nq, n))
    data, query = data64.to(dt), data64[rows].to(dt)
    try:
        lib.lamp_knn_split_mode(0)
        ei, ed = _knn(data, query, k)
        lib.lamp_knn_split_mode(2)
        lib.lamp_kernel_timer_enable(1)
        si, sd = _knn(data, query, k)
        buf = C.create_string_buffer(1 << 16)
        lib.lamp_kernel_timer_report(buf, len(buf))
        lib.lamp_kernel_timer_enable(0)
        failed, planes = C.c_int64(-1), C.c_int(-1)
        lib.lamp_knn_split_last_failed(C.byref(failed))
        lib.lamp_knn_split_last_planes(C.byref(planes))
    finally:
        lib.lamp_knn_split_mode(1)
    assert b"knn_split_f16" in buf.value, "the split kernel did not run"
    assert planes.value == 2
    assert si.dtype == np.int64 and si.shape == (len(rows), k)
    assert np.array_equal(np.sort(si, 1), np.sort(ei, 1)), "same neighbour sets as the exact kernel"
    ref = O.knn_minibatched(data, query, k, 100)
    assert np.array_equal(np.sort(si, 1), np.sort(ref.numpy(), 1)), "and as the reference algorithm"
    assert (np.diff(sd, axis=1) >= 0).all()
    exact = torch.gather(O.squared_euclidean_distance(data64[rows], data64), 1, torch.from_numpy(si))
    assert (torch.from_numpy(sd).double() - exact).abs().max().item() <= (2e-4 if dt == torch.float32 else 1e-10)     # f32: |q|^2 ~ d, a few ulp of 2 d
    assert 0 <= failed.value <= max(2, len(rows) // 50), f"{failed.value} of {len(rows)} queries needed the exact kernel on centred data"


def test_knn_split_filter_far_from_the_origin_and_falls_back_on_duplicates(gpu):
    """bench.py's kNN points in small (uniform jitter + 16 clusters one unit apart in every coordinate: |x|^2 up to 3e4, where the f32
    formula itself is only good to ~5e-3): the filter works on the centred rows, split into two f16 planes (22 bits).  Rows with a well
    separated neighbourhood get the exact kernel's set; queries whose k-th neighbour has exact duplicates cannot be proven and go
    through the exact kernel (ties -> lower index on either path)."""
    g = torch.Generator().manual_seed(99)
    n, d, k = 20000, 128, 10
    data64 = torch.rand(n, d, generator=g, dtype=torch.float64) + (torch.arange(n) % 16).double().reshape(n, 1)
    data64[160:176] = data64[144:160]                                 # exact duplicates (same cluster: i % 16 is kept)
    data64[300:320] = data64[300]                                     # twenty copies of one point: its 10th and 16th neighbours tie
    data, query = data64.float(), data64[:1200].float()
    try:
        lib.lamp_knn_split_mode(0)
        ei, ed = _knn(data, query, k)
        lib.lamp_knn_split_mode(2)
        si, sd = _knn(data, query, k)
        failed, planes = C.c_int64(-1), C.c_int(-1)
        lib.lamp_knn_split_last_failed(C.byref(failed))
        lib.lamp_knn_split_last_planes(C.byref(planes))
    finally:
        lib.lamp_knn_split_mode(1)
    assert planes.value == 2
    assert 20 <= failed.value <= 600, f"{failed.value} of 1200 queries needed the exact kernel"     # the twenty copies at least
    sep = _well_separated_queries(data64, torch.arange(1200), k, 5e-2).numpy()
    assert len(sep) > 300
    assert np.array_equal(np.sort(si[sep], 1), np.sort(ei[sep], 1)), "same neighbour sets as the exact kernel where the data decides them"
    assert np.abs(sd.astype(np.float64) - ed).max() <= 5e-2          # the f32 formula's own noise at |q|^2 ~ 3e4
    assert np.array_equal(si[144:160, 0], np.arange(144, 160)) and np.array_equal(si[160:176, 0], np.arange(144, 160)), "a duplicate's first neighbour is the lower index"
    assert np.array_equal(si[144:176, 1], np.r_[np.arange(160, 176), np.arange(160, 176)]), "then its copy"
    assert np.array_equal(si[300:320], ei[300:320]) and np.array_equal(sd[300:320], ed[300:320]), "unprovable queries come from the exact kernel, bitwise"
    assert np.array_equal(si[300:320], np.tile(np.arange(300, 310), (20, 1))), "ten of twenty equal points: the ten lowest indices"


def test_knn_split_proof_holds_with_a_far_outlier_and_near_ties(gpu):
    """the per-query proof must also cover the f32 arithmetic of what the filter compares (centred norms accumulated in f32, centred
    rows rounded to f32).  f64 data - where nothing else covers it - in a tight cluster away from the origin, ONE far outlier opposite the mean
    (max |x_c|^2 is then ~1e6 times a neighbour's) and queries whose 1st .. 30th neighbours sit on a shell with radii 1e-7 apart (squared
    distances 2e-10 apart: above the f64 formula's own noise at |q|^2 = 1600, far below anything an f16 filter resolves): the filter cannot order them, so those queries must fail the proof and come from the exact kernel - the neighbour sets are the exact search's
    on EVERY row."""
    g = torch.Generator().manual_seed(31)
    n, d, k, nshell = 6000, 64, 10, 40
    data = 5.0 + 1e-2 * torch.randn(n, d, generator=g, dtype=torch.float64)
    data[n - 1] = -1000.0                                                     # the outlier
    base = 200
    for s_ in range(nshell):                                                  # query s_: thirty points on a thin shell around it
        c = data[s_].clone()
        dirs = torch.randn(30, d, generator=g, dtype=torch.float64)
        dirs /= dirs.norm(dim=1, keepdim=True)
        radii = 1e-3 * (1.0 + 1e-4 * torch.arange(30, dtype=torch.float64))
        data[base + 30 * s_: base + 30 * (s_ + 1)] = c + dirs * radii.reshape(30, 1)
    query = data[:400].clone()
    try:
        lib.lamp_knn_split_mode(0)
        ei, ed = _knn(data, query, k)
        lib.lamp_knn_split_mode(2)
        si, sd = _knn(data, query, k)
        failed = C.c_int64(-1)
        lib.lamp_knn_split_last_failed(C.byref(failed))
    finally:
        lib.lamp_knn_split_mode(1)
    exact = O.squared_euclidean_distance(query, data)
    ref = torch.topk(exact, k, 1, largest=False, sorted=True)[1].numpy()
    v = torch.topk(exact, k + 1, 1, largest=False, sorted=True)[0]
    sep = ((v[:, k] - v[:, k - 1]) > 1e-11).numpy()                          # rows the f64 formula decides (its noise at |q|^2 = 1600 is ~4e-13)
    assert sep[:nshell].all() and sep.sum() > 300
    assert np.array_equal(np.sort(ei[sep], 1), np.sort(ref[sep], 1)), "the exact kernel resolves squared distances 2e-10 apart in f64"
    assert np.array_equal(np.sort(si, 1), np.sort(ei, 1)), "filter + proof + fallback = the exact search on every row"
    assert failed.value >= nshell, f"only {failed.value} queries failed the proof: the shell queries cannot be proven by an f16 filter"


def test_knn_few_queries_slice_the_data_set(gpu):
    """A handful of queries against many points: the exact kernel cuts the data set into slices over blockIdx.y (a workgroup per query block
    would stream all of it alone) and merges the per-slice lists - same indices and values as the unsliced search of the same rows."""
    g = torch.Generator().manual_seed(7)
    n, d, k = 70000, 128, 16
    data = torch.randn(n, d, generator=g)
    data[500] = data[40]; data[69999] = data[40]                      # ties across slices
    lib.lamp_knn_split_mode(0)
    try:
        qs = torch.cat([data[:300], data[69990:]])
        few_i, few_d = _knn(data, qs, k)                              # 310 queries: 3 workgroups -> sliced
        pad = torch.cat([qs, data[1000:1000 + 40000]])                # the same rows inside a search large enough not to be sliced
        all_i, all_d = _knn(data, pad, k)
    finally:
        lib.lamp_knn_split_mode(1)
    assert np.array_equal(few_i, all_i[:310]) and np.array_equal(few_d, all_d[:310])
    assert few_i[40, 0] == 40 and few_i[40, 1] == 500 and few_i[40, 2] == 69999, "equal distances: ascending index, across slices too"
@pytest.mark.gpu
def test_knn_default_mode_takes_the_filter_on_the_bench_points(gpu):
    """The default mode decides by a sample whether the f16 filter pays.  On bench.py's kNN points (uniform [0, 1)^128 + 16 clusters along the
    diagonal, |x - mean|^2 up to ~7200) it must: a too generous error bound makes the sample decline and the search falls back to the exact
    kernel at a third of the speed (round 4: the bound's third term at 2^-20 with c = 2^-18 did exactly that, unnoticed by the parity
    tests).  Same neighbours as the exact search."""
    n, nq, d, k = 131072, 32768, 128, 10                      # 4.3e9 distance evaluations: above the default mode's threshold
    rng = np.random.default_rng(0)
    pts = rng.random((n, d), dtype=np.float32) + (np.arange(n) % 16)[:, None].astype(np.float32)
    X = S.STen.from_numpy(pts, 0, S.F32)
    Q = X.slice(0, 1000, 1000 + nq)
    planes, failed = C.c_int(-1), C.c_int64(-1)
    try:
        lib.lamp_knn_split_mode(0)
        e, ed = C.c_void_p(), C.c_void_p(); lib.lamp_knn_squared_euclidean(C.byref(e), C.byref(ed), X, Q, k + 1)
        exact, exact_d = S.STen(e).to_numpy(), S.STen(ed).to_numpy()
        lib.lamp_knn_split_mode(1)
        f, fd = C.c_void_p(), C.c_void_p(); lib.lamp_knn_squared_euclidean(C.byref(f), C.byref(fd), X, Q, k)
        got, got_d = S.STen(f).to_numpy(), S.STen(fd).to_numpy()
        lib.lamp_knn_split_last_planes(C.byref(planes)); lib.lamp_knn_split_last_failed(C.byref(failed))
    finally:
        lib.lamp_knn_split_mode(1)
    assert planes.value == 2, "the default mode declined the filter on the bench's points"
    assert failed.value < nq // 10, f"{failed.value} of {nq} queries unproven"
    # |x|^2 reaches 3e4 here: one f32 ulp of the formula's intermediates is 2 - 4e-3 and the exact kernel's dot product is a chain of 128 f32
    # fmas at that magnitude (noise ~ 1e-2 in d, bounded by the 2^-21 (|q|^2 + |x|^2) = 0.03 of the proof); the re-rank rounds an f64 dot
    # product once.  Rows whose k-th and (k+1)-th neighbours are closer than that may legitimately swap them
    clear = (exact_d[:, k] - exact_d[:, k - 1]) > 0.1
    assert clear.mean() > 0.15, f"only {clear.mean():.3f} of the rows are clearly separated"
    assert np.array_equal(np.sort(got[clear], 1), np.sort(exact[clear, :k], 1)), "neighbour sets differ where the f32 formula can tell them apart"
    assert np.abs(got_d - exact_d[:, :k]).max() < 0.06


