"""Transformer family + language model: oracle pinned to the reference's tests (CPU), HIP path vs oracle (GPU).

Reference tests mirrored here: lamp-core/src/test/scala/lamp/nn/maskedsoftmax.test.scala:13-88 (mask known answers) and
lamp-core/src/test/scala/lamp/nn/nn.test.scala:700-860 ("transformer encoder" / "linearized transformer encoder": the sum of the
output is 0.0 to 4 decimals and the autograd gradients equal central finite differences to 6 decimals, all weights 2, b2 and the
scales 1, input arange(12).view(2, 3, 2), maxLength ones(2, 3)).
"""
import math

import numpy as np
import pytest
import torch

from oracle import lamp_oracle as O
from oracle import lamp_transformer_oracle as T
from tests.util import assert_close, closed_form, to_sten, to_torch


# ---------------------------------------------------------------------------------------------------------------------------
# oracle vs the reference's own tests (CPU)
# ---------------------------------------------------------------------------------------------------------------------------
def test_oracle_sequence_mask_known_answers():
    ones = lambda *s: O.const(torch.ones(*s, dtype=torch.float64))
    m = T.sequence_mask(torch.tensor([2, 3]), ones(2, 4, 3), 0.0).value              # "1D"
    assert m[0].tolist() == [[1, 1, 0]] * 4 and m[1].tolist() == [[1, 1, 1]] * 4
    m = T.sequence_mask(torch.tensor([2, 3]), ones(2, 4, 4), 0.0).value              # "1D symm"
    assert m[0].tolist() == [[1, 1, 0, 0]] * 4 and m[1].tolist() == [[1, 1, 1, 0]] * 4
    m = T.sequence_mask(torch.tensor([[2, 3], [2, 1]]), ones(2, 2, 3), 0.0).value    # "2D"
    assert m[0].tolist() == [[1, 1, 0], [1, 1, 1]] and m[1].tolist() == [[1, 1, 0], [1, 0, 0]]


def _reference_test_encoder(linearized, dtype=torch.float64, is_cuda=False):
    """nn.test.scala:722-775: in 2, 5 heads x 4 hidden, mlp 7, all weights 2 except b2 / scales = 1, gptOrder false."""
    two = lambda *s: O.param(torch.ones(*s, dtype=dtype) * 2)
    one = lambda *s: O.param(torch.ones(*s, dtype=dtype))
    att = T.MultiheadAttention(two(2, 20), two(2, 20), two(2, 20), two(20, 2), 5, linearized, False, is_cuda=is_cuda)
    return T.TransformerEncoder([T.TransformerEncoderBlock(att, two(2, 7), two(1, 7), two(7, 2), one(1, 2), one(2), one(2), False)])


@pytest.mark.parametrize("linearized", [False, True])
def test_oracle_transformer_encoder_reference_test(linearized):
    x = O.const(torch.arange(12, dtype=torch.float64).view(2, 3, 2))
    mx = torch.ones(2, 3, dtype=torch.int64)
    enc = _reference_test_encoder(linearized)
    out = enc.forward(x, mx).sum()
    assert round(out.value.item(), 4) == 0.0
    grads = enc.gradients(out)
    params = enc.parameters()
    assert len(params) == 10
    eps = 1e-3
    for p, g in zip(params, grads):
        flat = p.value.view(-1)
        num = torch.zeros_like(flat)
        for i in range(flat.numel()):
            old = flat[i].item()
            flat[i] = old + eps; a = enc.forward(x, mx).sum().value.item()
            flat[i] = old - eps; b = enc.forward(x, mx).sum().value.item()
            flat[i] = old
            num[i] = (a - b) / (2 * eps)
        assert np.array_equal(np.round(g.view(-1).numpy(), 6) + 0.0, np.round(num.numpy(), 6) + 0.0)


def test_oracle_fused_branch_reads_dim1_as_heads():
    """The fused call of Transformer.scala:930-945 attends over the axis the views put in dimension 2 (the heads)."""
    torch.manual_seed(1)
    B, S, H, D = 2, 8, 3, 4
    q, k, v = (torch.randn(B, S, H, D, dtype=torch.float64) for _ in range(3))
    got = T.ScaledDotProductAttention(O.const(q), O.const(k), O.const(v), True).value.value
    w = torch.einsum("bshd,bsgd->bshg", q, k) / math.sqrt(D)
    w = w.masked_fill(torch.ones(H, H).triu(1).bool(), float("-inf")).softmax(-1)
    assert torch.allclose(got, torch.einsum("bshg,bsgd->bshd", w, v), atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------------
# HIP path vs oracle, through the C ABI
# ---------------------------------------------------------------------------------------------------------------------------
def _hip():
    from lamp_amd import autograd as A, nn, sten as S, transformer as TR
    return A, nn, S, TR


def _lamp_dtype(S, dt): return {torch.float64: S.F64, torch.float32: S.F32, torch.bfloat16: S.BF16}[dt]


def _load(hm, om, S, dt):
    hm.load([S.STen.from_numpy(v.value.detach().double().numpy(), dtype=_lamp_dtype(S, dt)) for v in om.state()])


def _rand_params(shapes, dt, salt=0, scale=1.0):
    return [O.param(closed_form(s, salt + 17 * i, scale, dt)) for i, s in enumerate(shapes)]


def _oracle_mha(dq, hidden, heads, out, dt, linearized, causal, salt=0, is_cuda=False):
    """is_cuda=False (default): the composed branch = the reference's ATen CPU path, the parity target of the HIP library's default mode"""
    wq, wk, wv, wo = _rand_params([(dq, hidden * heads), (dq, hidden * heads), (dq, hidden * heads), (hidden * heads, out)], dt, salt, 1.0)
    return T.MultiheadAttention(wq, wk, wv, wo, heads, linearized, causal, is_cuda=is_cuda)


def _oracle_encoder_block(in_, hidden, heads, mlp, dt, linearized, gpt, causal, salt=0):
    att = _oracle_mha(in_, hidden, heads, in_, dt, linearized, causal, salt)
    w1, b1, w2, b2, s1, s2 = _rand_params([(in_, mlp), (1, mlp), (mlp, in_), (1, in_), (in_,), (in_,)], dt, salt + 100, 1.0)
    return T.TransformerEncoderBlock(att, w1, b1, w2, b2, s1, s2, gpt)


def _compare(hm, om, hout, oout, dt, S):
    from lamp_amd.autograd import const as A_const
    from tests.util import FWD_TOL, BWD_TOL
    assert_close(to_torch(hout.value), oout.value.double(), FWD_TOL[dt] * 4, "forward")
    # a weighted sum: the plain sum of a layer-normalised output is identically 0 and so are its gradients
    c = closed_form(tuple(oout.shape), 31, 1.0, dt)
    hg = hm.gradients((hout * A_const(to_sten(c))).sum())
    og = om.gradients((oout * O.const(c)).sum())
    assert len(hg) == len(og) and len(og) > 0
    for i, (a, b) in enumerate(zip(hg, og)):
        assert_close(to_torch(a), b.double(), BWD_TOL[dt], f"gradient {i}")


@pytest.mark.gpu
def test_sequence_mask_known_answers_on_gpu(gpu):
    A, nn, S, TR = _hip()
    ones = lambda *s: A.const(S.STen.ones(list(s), S.F64))
    mask = TR.MultiheadAttention.sequenceMask
    m = mask(to_sten(torch.tensor([2, 3])), ones(2, 4, 3), 0.0).value.to_numpy()
    assert m[0].tolist() == [[1, 1, 0]] * 4 and m[1].tolist() == [[1, 1, 1]] * 4
    m = mask(to_sten(torch.tensor([2, 3])), ones(2, 4, 4), 0.0).value.to_numpy()
    assert m[0].tolist() == [[1, 1, 0, 0]] * 4 and m[1].tolist() == [[1, 1, 1, 0]] * 4
    m = mask(to_sten(torch.tensor([[2, 3], [2, 1]])), ones(2, 2, 3), 0.0).value.to_numpy()
    assert m[0].tolist() == [[1, 1, 0], [1, 1, 1]] and m[1].tolist() == [[1, 1, 0], [1, 0, 0]]
    x = closed_form((2, 3, 5), 3, 4.0, torch.float64)
    mx = torch.tensor([[1, 5, 2], [3, 4, 1]])
    got = TR.MultiheadAttention.maskedSoftmax(A.const(to_sten(x)), to_sten(mx)).value
    assert_close(to_torch(got), T.masked_softmax(O.const(x), mx).value, 1e-12, "maskedSoftmax")


@pytest.mark.gpu
@pytest.mark.parametrize("linearized", [False, True])
def test_reference_transformer_encoder_test_on_gpu(gpu, linearized):
    """nn.test.scala:700-860 on the GPU in f64: value 0.0 to 4 decimals, gradients = the oracle's (which equal finite differences)."""
    A, nn, S, TR = _hip()
    om = _reference_test_encoder(linearized)
    hm = TR.TransformerEncoder(1, 2, 4, 5, 7, 0.0, S.F64, 0, linearized, False, False)
    assert len(hm.state) == 10 and [v.shape for v in hm.state] == [v.shape for v in om.state()]
    _load(hm, om, S, torch.float64)
    x = torch.arange(12, dtype=torch.float64).view(2, 3, 2)
    mx = torch.ones(2, 3, dtype=torch.int64)
    hout = hm.forward(A.const(to_sten(x)), to_sten(mx))
    hsum = hout.sum()
    assert round(float(hsum.value.to_numpy()), 4) == 0.0
    oout = om.forward(O.const(x), mx)
    assert_close(to_torch(hout.value), oout.value, 1e-11, "forward")
    # the reference's acceptance rule: gradients equal to 6 decimals (the oracle's equal central finite differences, see above)
    for a, b in zip(hm.gradients(hsum), om.gradients(oout.sum())):
        assert np.array_equal(np.round(to_torch(a).numpy(), 6) + 0.0, np.round(b.numpy(), 6) + 0.0)


@pytest.fixture
def as_written_for_cuda():
    """opt-in mode of the HIP library: the fused call exactly as the reference writes it for CUDA (dimension 1 read as heads)"""
    from lamp_amd import transformer as TR
    prev = TR.MultiheadAttention.fusedCallAsWritten(True)
    yield
    TR.MultiheadAttention.fusedCallAsWritten(prev)


def _mha_case(dt, case, is_cuda):
    A, nn, S, TR = _hip()
    B, Sq, heads, hidden, dq, out = 3, 6, 4, 8, 10, 5
    linearized = case.startswith("linearized")
    causal = case in ("causal", "fused_causal")
    if case.startswith("fused"):
        Sq = 16                                     # aligned sequence: the gate of the fused branch (Transformer.scala:946-951)
    om = _oracle_mha(dq, hidden, heads, out, dt, linearized, causal, is_cuda=is_cuda)
    hm = TR.MultiheadAttention(dq, dq, dq, hidden, out, 0.0, heads, _lamp_dtype(S, dt), 0, linearized, causal)
    _load(hm, om, S, dt)
    q = closed_form((B, Sq, dq), 5, 2.0, dt)
    k = closed_form((B, Sq + (3 if case == "cross" else 0), dq), 6, 2.0, dt)
    v = closed_form(tuple(k.shape), 7, 2.0, dt)
    mx = None
    if case in ("maxlen2d", "linearized_maxlen"):
        mx = (torch.arange(B * heads * Sq).view(B * heads, Sq) % Sq + 1)[:B]     # batch x queries
    qv, kv, vv = (O.param(t) for t in (q, k, v))
    oout = om.forward(qv, kv, vv, mx)
    hq, hk, hv = (A.param(to_sten(t)) for t in (q, k, v))
    hout = hm.forward(hq, hk, hv, to_sten(mx) if mx is not None else None)
    from tests.util import FWD_TOL, BWD_TOL
    assert_close(to_torch(hout.value), oout.value.double(), FWD_TOL[dt] * 4, "forward")
    hout.sum().backprop(); oout.sum().backprop()
    for name, a, b in (("dq", hq, qv), ("dk", hk, kv), ("dv", hv, vv)):
        assert_close(to_torch(a.partialDerivative), b.grad.double(), BWD_TOL[dt], name)
    for i, (a, b) in enumerate(zip(hm.parameters, om.parameters())):
        assert_close(to_torch(a.partialDerivative), b.grad.double(), BWD_TOL[dt], f"weight {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
@pytest.mark.parametrize("case", ["plain", "maxlen2d", "causal", "linearized", "linearized_maxlen", "fused", "fused_causal", "cross"])
def test_multihead_attention_matches_oracle(gpu, dt, case):
    """default mode = the composed branch of the reference's CPU path, also for aligned sequences ("fused*": per-head attention over
    the 16 positions; without a mask the raw scores, :797-801)"""
    _mha_case(dt, case, is_cuda=False)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
@pytest.mark.parametrize("case", ["fused", "fused_causal", "causal", "plain"])
def test_multihead_attention_as_written_for_cuda(gpu, as_written_for_cuda, dt, case):
    """opt-in: the fused call with (batch, sequence, heads, d) views read as (batch, heads, sequence, d) - what the reference's CUDA
    branch hands to ATen (test_oracle_fused_branch_reads_dim1_as_heads states what that computes)"""
    _mha_case(dt, case, is_cuda=True)


@pytest.mark.gpu
@pytest.mark.parametrize("Sq,heads,hidden", [(384, 12, 64), (200, 3, 64), (128, 2, 128), (16, 4, 64)])
def test_multihead_attention_bf16_flash_path_has_cpu_semantics(gpu, Sq, heads, hidden):
    """bf16, head width 64 / 128, causal: the default mode runs the flash kernels on strided (batch, heads, sequence, d) views of the
    projections (no transposed copies) and must agree with the composed f32 oracle - per-head attention over the SEQUENCE - at bf16
    resolution; the kernel classes that ran are checked, and the as-written-for-CUDA reading of the same call differs measurably."""
    import ctypes as C
    from lamp_amd._capi import lib
    A, nn, S, TR = _hip()
    B, dq, out = 2, 64, 48
    om = _oracle_mha(dq, hidden, heads, out, torch.float32, False, True)
    for v in om.state():
        v.value.copy_((v.value * (2.0 / v.value.shape[0] ** 0.5)).bfloat16().float())
    hm = TR.MultiheadAttention(dq, dq, dq, hidden, out, 0.0, heads, S.BF16, 0, False, True)
    _load(hm, om, S, torch.bfloat16)
    x = closed_form((B, Sq, dq), 5, 2.0, torch.bfloat16)
    xv = O.param(x.float())
    oout = om.forward(xv)
    hx = A.param(to_sten(x))
    lib.lamp_kernel_timer_filter(None); lib.lamp_kernel_timer_enable(1)
    hout = hm.forward(hx)
    c = closed_form(tuple(oout.shape), 31, 1.0, torch.bfloat16)
    (hout * A.const(to_sten(c))).sum().backprop()
    lib.lamp_kernel_timer_enable(0)
    buf = C.create_string_buffer(1 << 16); lib.lamp_kernel_timer_report(buf, len(buf))
    ran = {l.split()[0] for l in buf.value.decode().splitlines()}
    assert {"sdpa_flash_fwd", "sdpa_flash_bwd_dq", "sdpa_flash_bwd_dkv"} <= ran, ran
    (oout * O.const(c.float())).sum().backprop()
    # four bf16 stages (projections, attention, output projection): chained roundings -> scale="max"
    assert_close(to_torch(hout.value), oout.value.double(), 2.0 ** -6, "forward", scale="max")
    assert_close(to_torch(hx.partialDerivative), xv.grad.double(), 2.0 ** -5, "dx", scale="max")
    for i, (a, b) in enumerate(zip(hm.parameters, om.parameters())):
        assert_close(to_torch(a.partialDerivative), b.grad.double(), 2.0 ** -5, f"weight {i}", scale="max")
    if Sq % 8 == 0:
        prev = TR.MultiheadAttention.fusedCallAsWritten(True)
        try:
            other = to_torch(hm.forward(A.const(to_sten(x))).value)
        finally:
            TR.MultiheadAttention.fusedCallAsWritten(prev)
        assert (other.double() - oout.value.double()).abs().max() > 0.05 * oout.value.abs().max(), "the two readings must differ"


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
@pytest.mark.parametrize("gpt,causal,linearized,seq", [(True, True, False, 16), (True, True, False, 6), (False, False, False, 6), (True, False, True, 5),
                                                        (False, True, False, 8)])
def test_transformer_encoder_block_matches_oracle(gpu, dt, gpt, causal, linearized, seq):
    A, nn, S, TR = _hip()
    in_, hidden, heads, mlp = 12, 4, 3, 20
    om = _oracle_encoder_block(in_, hidden, heads, mlp, dt, linearized, gpt, causal)
    hm = TR.TransformerEncoderBlock(in_, hidden, heads, mlp, in_, 0.0, _lamp_dtype(S, dt), 0, linearized, gpt, causal)
    assert [v.shape for v in hm.state] == [v.shape for v in om.state()]
    _load(hm, om, S, dt)
    x = closed_form((2, seq, in_), 9, 2.0, dt)
    mx = None if (causal or seq % 8 == 0) else torch.full((2, seq), seq - 1, dtype=torch.int64)
    hout = hm.forward(A.const(to_sten(x)), to_sten(mx) if mx is not None else None)
    _compare(hm, om, hout, om.forward(O.const(x), mx), dt, S)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_transformer_decoder_block_and_transformer_match_oracle(gpu, dt):
    A, nn, S, TR = _hip()
    in_, hidden, heads, mlp = 8, 4, 2, 12
    dd = _oracle_mha(in_, hidden, heads, in_, dt, False, True, 1)
    ed = _oracle_mha(in_, hidden, heads, in_, dt, False, False, 2)
    w1, b1, w2, b2 = _rand_params([(in_, mlp), (1, mlp), (mlp, in_), (1, in_)], dt, 300, 1.0)
    od = T.TransformerDecoderBlock(dd, ed, w1, b1, w2, b2)
    hd = TR.TransformerDecoderBlock(in_, hidden, heads, mlp, in_, 0.0, _lamp_dtype(S, dt), 0, False, True, False)
    assert [v.shape for v in hd.state] == [v.shape for v in od.state()]
    _load(hd, od, S, dt)
    dec = closed_form((2, 5, in_), 11, 2.0, dt)
    enc = closed_form((2, 7, in_), 12, 2.0, dt)
    hout = hd.forward(A.const(to_sten(dec)), A.const(to_sten(enc)))
    _compare(hd, od, hout, od.forward(O.const(dec), O.const(enc)), dt, S)

    # Transformer = gpt-order encoder + decoder (Transformer.scala:330-365), 2 blocks each
    oenc = T.TransformerEncoder([_oracle_encoder_block(in_, hidden, heads, mlp, dt, False, True, False, 1000 * i) for i in range(2)])
    odecs = []
    for i in range(2):
        a, b = _oracle_mha(in_, hidden, heads, in_, dt, False, True, 2000 + i), _oracle_mha(in_, hidden, heads, in_, dt, False, False, 3000 + i)
        odecs.append(T.TransformerDecoderBlock(a, b, *_rand_params([(in_, mlp), (1, mlp), (mlp, in_), (1, in_)], dt, 4000 + i, 1.0)))

    class OT(O.Module):
        def state(self): return oenc.state() + [s for d in odecs for s in d.state()]
    ot = OT()
    ht = TR.Transformer(2, in_, hidden, heads, mlp, 0.0, _lamp_dtype(S, dt), 0)
    assert [v.shape for v in ht.state] == [v.shape for v in ot.state()]
    _load(ht, ot, S, dt)
    encmx = torch.tensor([[7] * 7, [4] * 7])
    eo = oenc.forward(O.const(enc), encmx)
    oo = O.const(dec)
    for d in odecs:
        oo = d.forward(oo, eo, None)
    hout = ht.forward(A.const(to_sten(dec)), A.const(to_sten(enc)), None, to_sten(encmx))
    _compare(ht, ot, hout, oo, dt, S)


@pytest.mark.gpu
def test_positional_embeddings(gpu):
    A, nn, S, TR = _hip()
    for (L, D) in ((7, 6), (5, 5), (1, 2)):
        got = TR.PositionalEmbedding.vaswani(L, D, S.F64)
        assert_close(to_torch(got), T.vaswani(L, D), 1e-15, "vaswani")
    emb = TR.Embedding(11, 6, S.F32)
    w = closed_form((11, 6), 1, 1.0, torch.float32)
    emb.load([to_sten(w)])
    pos = TR.PositionalEmbedding.vaswani(4, 6, S.F32)
    tokens = torch.tensor([[1, 5, 10, 0], [3, 3, 2, 9]])
    for add in (True, False):
        te = TR.TransformerEmbedding(emb, add, pos)
        assert [v.shape for v in te.state] == [[4, 6], [11, 6]]
        got = to_torch(te.forward(A.const(to_sten(tokens))).value)
        e = w[tokens]
        p = T.vaswani(4, 6).float().unsqueeze(0)
        want = e + p if add else torch.cat([e, p.repeat(2, 1, 1)], 2)
        assert_close(got, want, 1e-6, "TransformerEmbedding")


def _oracle_lm(vocab, ctx, dim, heads, blocks, dt, pad):
    enc = T.TransformerEncoder([_oracle_encoder_block(dim, dim // heads, heads, dim * 4, dt, False, True, True, 1000 * i) for i in range(blocks)])
    te, pe = _rand_params([(vocab, dim), (ctx, dim)], dt, 77, 1.0)
    return T.LanguageModel(te, pe, enc, pad)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
@pytest.mark.parametrize("seq,positions", [(16, False), (6, False), (16, True)])
def test_language_model_loss_matches_oracle(gpu, dt, seq, positions):
    """LanguageModelLoss (lm.scala:44-59) = the module of example-autoregressivelm: loss value and all gradients."""
    A, nn, S, TR = _hip()
    vocab, ctx, dim, heads, blocks, pad = 19, 16, 12, 3, 2, -1000
    om = _oracle_lm(vocab, ctx, dim, heads, blocks, dt, pad)
    hm = TR.LanguageModelLoss(ctx, vocab, blocks, dim, dim // heads, heads, dim * 4, 0.0, pad, _lamp_dtype(S, dt), 0)
    assert [v.shape for v in hm.state] == [v.shape for v in om.state()]
    _load(hm, om, S, dt)
    tokens = (torch.arange(3 * seq).view(3, seq) * 7 + 3) % vocab
    pos = torch.tensor([[0, 5], [seq + 1, seq + 2], [2 * seq + 3, 3 * seq - 1]]) if positions else None
    target = ((torch.arange(3 * 2).view(3, 2) if positions else torch.arange(3 * seq).view(3, seq)) * 5 + 1) % vocab
    target[0, 1] = pad                                                    # ignored position
    oloss = om.loss(tokens, target, None, pos)
    hloss = hm.forward(A.const(to_sten(tokens)), to_sten(target), None, to_sten(pos) if pos is not None else None)
    ftol, btol = (1e-11, 1e-9) if dt == torch.float64 else (1e-5, 1e-3)
    assert_close(to_torch(hloss.value).reshape(()), oloss.value.double(), ftol, "loss")
    hg = hm.gradients(hloss); og = om.gradients(oloss)
    assert len(hg) == len(og) == 2 + 10 * blocks
    for i, (a, b) in enumerate(zip(hg, og)):
        assert_close(to_torch(a), b.double(), btol, f"gradient {i}")
    enc, logits = hm.languageModel(A.const(to_sten(tokens)), None, to_sten(pos) if pos is not None else None)
    oenc, ologits = om.forward(tokens, None, pos)
    assert_close(to_torch(enc.value), oenc.value.double(), ftol * 10, "encoded")
    assert_close(to_torch(logits.value), ologits.value.double(), ftol * 10, "logits")


@pytest.mark.gpu
def test_language_model_training_steps_with_identity_loss(gpu):
    """SupervisedModel(LanguageModelLoss, LossFunctions.Identity) + AdamW (example-autoregressivelm train.scala:40-66): two steps = oracle."""
    A, nn, S, TR = _hip()
    dt = torch.float32
    vocab, ctx, dim, heads, blocks, pad = 17, 8, 8, 2, 1, -1000
    om = _oracle_lm(vocab, ctx, dim, heads, blocks, dt, pad)
    hm = TR.LanguageModelLoss(ctx, vocab, blocks, dim, dim // heads, heads, dim * 4, 0.0, pad, S.F32, 0)
    _load(hm, om, S, dt)
    tokens = (torch.arange(4 * 8).view(4, 8) * 3 + 1) % vocab
    target = (tokens + 1) % vocab
    model = nn.SupervisedModel(hm, nn.SupervisedModel.IDENTITY)
    hopt = nn.AdamW([p.value for p in hm.parameters], weightDecay=0.01, learningRate=1e-2, clip=1.0)
    oopt = O.AdamW([p.value for p in om.parameters()], weightDecay=0.01, learningRate=1e-2, clip=1.0)
    acc = S.STen.zeros([1], S.F32)
    losses = []
    for step in range(2):
        n = model.train_step(hopt, to_sten(tokens), to_sten(target), acc)
        assert n == 4
        ol = om.loss(tokens, target)
        losses.append(ol.value.item())
        oopt.step(om.gradients(ol), 1.0)
        for i, (hv, ov) in enumerate(zip(hm.state, om.state())):
            assert_close(to_torch(hv.value), ov.value.double(), 2e-4, f"state {i} after step {step}")
    assert abs(float(acc.to_numpy()[0]) - 4 * sum(losses)) <= 1e-4 * 4 * sum(losses)
    assert losses[1] < losses[0]


@pytest.mark.gpu
def test_language_model_bf16_flash_path_matches_f32_oracle(gpu):
    """bf16 with head width 64: causal attention runs the flash kernels (attention.hip) on (batch, heads, sequence, d) views of the
    projections - per-head attention over the sequence, the arithmetic of the reference's CPU path; the loss and every gradient agree
    with the composed f32 oracle on the same bf16-rounded weights at bf16 resolution"""
    A, nn, S, TR = _hip()
    vocab, ctx, dim, heads, blocks, pad = 32, 16, 128, 2, 2, -1000
    om = _oracle_lm(vocab, ctx, dim, heads, blocks, torch.float32, pad)
    for v in om.state():                                   # weights of a trained-model magnitude, representable in bf16
        v.value.copy_((v.value * (4.0 / v.value.shape[-1] ** 0.5)).bfloat16().float())
    hm = TR.LanguageModelLoss(ctx, vocab, blocks, dim, dim // heads, heads, dim * 4, 0.0, pad, S.BF16, 0)
    _load(hm, om, S, torch.bfloat16)
    tokens = (torch.arange(4 * ctx).view(4, ctx) * 7 + 3) % vocab
    target = (tokens * 5 + 1) % vocab
    oloss = om.loss(tokens, target)
    hloss = hm.forward(A.const(to_sten(tokens)), to_sten(target))
    assert_close(to_torch(hloss.value).reshape(()).double(), oloss.value.double(), 2e-2, "loss")
    hg = hm.gradients(hloss); og = om.gradients(oloss)
    for i, (a, b) in enumerate(zip(hg, og)):
        assert_close(to_torch(a).double(), b.double(), 3e-2, f"gradient {i}", scale="max")     # a whole network in bf16: chained roundings


@pytest.mark.gpu
@pytest.mark.parametrize("B,Sq,heads,d", [(2, 384, 12, 64), (3, 128, 2, 128), (1, 64, 3, 64)])
def test_packed_self_attention_equals_three_projections(gpu, B, Sq, heads, d):
    """F::packed_self_attention (one product x . [Wq | Wk | Wv], the flash kernels on the three column blocks, dq | dk | dv written as the
    column blocks of one buffer) against the three-projection composition it replaces: the forward is BITWISE the same (each output
    element is the same dot product in the same order), dX is rounded once instead of three times and the weight gradients come from one
    product - compared at bf16 resolution, and the packed gradient buffer is asserted through the kernel classes that ran."""
    import ctypes as C
    from lamp_amd._capi import lib
    A, nn, S, TR = _hip()
    dt = torch.bfloat16
    inn, HD = 96, heads * d
    x = closed_form((B, Sq, inn), 5, 2.0, dt)
    ws = [closed_form((inn, HD), 11 + 7 * i, 2.0 / inn ** 0.5, dt) for i in range(3)]
    c = closed_form((B, Sq, HD), 31, 1.0, dt)

    def run(packed):
        hx = A.param(to_sten(x))
        hw = [A.param(to_sten(w)) for w in ws]
        if packed:
            out = A.PackedSelfAttention(hx, hw[0], hw[1], hw[2], heads, True)
        else:
            x2 = hx.view([B * Sq, inn])
            q, k, v = (x2.mm(w).view([B, Sq, heads, d]).transpose(1, 2) for w in hw)       # (B, heads, S, d) strided views
            out = q.scaledDotProductAttention(k, v, True).transpose(1, 2).reshape([B, Sq, HD])
        (out * A.const(to_sten(c))).sum().backprop()
        return to_torch(out.value), to_torch(hx.partialDerivative), [to_torch(w.partialDerivative) for w in hw]

    lib.lamp_kernel_timer_filter(None); lib.lamp_kernel_timer_enable(1)
    po, pdx, pdw = run(True)
    lib.lamp_kernel_timer_enable(0)
    buf = C.create_string_buffer(1 << 16); lib.lamp_kernel_timer_report(buf, len(buf))
    ran = {l.split()[0]: int(l.split()[1]) for l in buf.value.decode().splitlines()}
    assert {"sdpa_flash_fwd", "sdpa_flash_bwd_dq", "sdpa_flash_bwd_dkv"} <= set(ran), ran
    uo, udx, udw = run(False)
    assert torch.equal(po, uo), "the packed projection must give bitwise the three projections' values"
    assert_close(pdx, udx.double(), 2.0 ** -6, "dx", scale="max")
    for i, (a, b) in enumerate(zip(pdw, udw)):
        assert_close(a, b.double(), 2.0 ** -6, f"dW{i}", scale="max")
    # and against f64 on the same bf16 inputs
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    q, k, v = ((xr.reshape(B * Sq, inn) @ w).reshape(B, Sq, heads, d).transpose(1, 2) for w in wr)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(B, Sq, HD)
    (ref * c.double()).sum().backward()
    assert_close(po, ref.detach(), 2.0 ** -6, "forward vs f64", scale="max")
    assert_close(pdx, xr.grad, 2.0 ** -5, "dx vs f64", scale="max")
    for i, (a, b) in enumerate(zip(pdw, wr)):
        assert_close(a, b.grad, 2.0 ** -5, f"dW{i} vs f64", scale="max")
