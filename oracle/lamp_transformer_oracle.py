"""CPU oracle for lamp's transformer family and language model.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product never does.

Restates, operator by operator, lamp-core/src/main/scala/lamp/nn/Transformer.scala (MultiheadAttention :572-1008,
TransformerEncoderBlock :212-260, TransformerDecoderBlock :263-307, Transformer :310-326, PositionalEmbedding.vaswani
:1022-1045, TransformerEmbedding :1105-1125), nn/Embedding.scala:17-27 and nn/languagemodel/lm.scala:44-190 on top of the
Variable / Op restatement in lamp_oracle.py, issuing the same ATen calls through torch on CPU tensors.

Pinned: tests/test_transformer.py checks the mask helpers against the known answers of the reference's
lamp-core/src/test/scala/lamp/nn/maskedsoftmax.test.scala:13-88 and the encoder against the expected value (0.0) and the
numeric-gradient check of lamp-core/src/test/scala/lamp/nn/nn.test.scala:700-860 ("transformer encoder", "linearized
transformer encoder").  `is_cuda` selects which branch of MultiheadAttention.multiheadAttention runs (:921-945).  The DEFAULT is
False: the composed branch, the only one the reference's ATen CPU path (the parity target) and its CPU tests ever take.  is_cuda=True
restates the fused branch with the same (batch, sequence, heads, d) views the reference hands to ATen's scaled-dot-product operator,
which reads dimension 1 as heads - kept to pin the HIP library's opt-in "as written for CUDA" mode.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from oracle.lamp_oracle import (Variable, Op, const, param, Module, Mult, LayerNormOp, Gelu, Sigmoid, IndexSelect, NllLoss, LogSoftMax,
                         unbroadcast)

aten = torch.ops.aten


class Reshape(Op):            # ops.scala:33-41
    def __init__(self, a, shape):
        self.params = [(a, lambda p, out: out.add_(p.reshape(out.shape)))]
        self.value = Variable(a.value.reshape(*shape), self)


class MaskFill(Op):           # ops.scala:148-159
    def __init__(self, inp, mask: torch.Tensor, fill: float):
        self.params = [(inp, lambda p, out: out.add_(p.masked_fill(mask, 0.0)))]
        self.value = Variable(inp.value.masked_fill(mask, fill), self)


class Embedding(Op):          # ops.scala:2141-2190
    def __init__(self, inp: Variable, weight: Variable):
        self.params = [(weight, lambda p, out: out.add_(aten.embedding_backward(p, inp.value, weight.value.shape[0], 0, False, False)))]
        self.value = Variable(aten.embedding(weight.value, inp.value, 0, False, False), self)


class ScaledDotProductAttention(Op):   # ops.scala:2342-2390: ATen's fused operator, inputs read as (batch, heads, sequence, d)
    def __init__(self, q, k, v, is_causal):
        leaves = [t.value.detach().clone().requires_grad_(True) for t in (q, k, v)]
        with torch.enable_grad():
            out = torch.nn.functional.scaled_dot_product_attention(*leaves, attn_mask=None, dropout_p=0.0, is_causal=is_causal)
        grads = {}

        def back(i):
            def f(p, o):
                if "g" not in grads:
                    grads["g"] = torch.autograd.grad(out, leaves, p)
                o.add_(grads["g"][i])
            return f
        self.params = [(q, back(0)), (k, back(1)), (v, back(2))]
        self.value = Variable(out.detach(), self)


def mm1(a: Variable, b: Variable, reshape=False) -> Variable:
    shape = a.shape
    a2 = Reshape(a, [-1, shape[-1]]).value if reshape else a.view([-1, shape[-1]])
    return a2.mm(b).view(shape[:-1] + [-1])


def swish1(x: Variable) -> Variable:     # ops.scala Swish1: x * sigmoid(x)
    return x * x.sigmoid()


# ---- MultiheadAttention companion (Transformer.scala:667-1008) ---------------------------------------
def sequence_mask(maxLength: torch.Tensor, maskable: Variable, fill: float) -> Variable:
    mv = maskable.value
    if maxLength.dim() == 2:
        assert maxLength.shape[1] == mv.shape[1] and maxLength.shape[0] == mv.shape[0]
        mask = torch.arange(0, mv.shape[2], 1, dtype=mv.dtype).view(1, 1, -1) >= maxLength.unsqueeze(2)
    else:
        assert maxLength.dim() == 1 and maxLength.shape[0] == mv.shape[0]
        mask = (torch.arange(0, mv.shape[2], 1, dtype=mv.dtype).unsqueeze(0) >= maxLength.unsqueeze(1)).unsqueeze(1)
    return MaskFill(maskable, mask, fill).value


def masked_softmax(inp: Variable, maxLength: torch.Tensor) -> Variable:
    return sequence_mask(maxLength, inp, float("-inf")).logSoftMax(2).exp()


def scaled_dot_product_attention(q, k, v, maxLength):     # dropout = 0
    d = q.shape[2]
    scores = q.bmm(k.transpose(1, 2)) * (1.0 / math.sqrt(float(d)))
    weights = scores if maxLength is None else masked_softmax(scores, maxLength)
    return weights.bmm(v)


def linearized_attention(q, k, v, maxLength):             # dropout = 0
    qF = swish1(q) + 1.0
    maskable = swish1(k) + 1.0
    kF = maskable if maxLength is None else sequence_mask(maxLength, maskable, 0.0)
    tmp1 = kF.transpose(1, 2).bmm(v)
    tmp2 = kF.sum([1], True).transpose(1, 2)
    return qF.bmm(tmp1) / (qF.bmm(tmp2) + 1e-5)


def multihead_attention(query, keys, values, maxLength, wQ, wK, wV, wO, numHeads, linearized, causalMask, is_cuda):
    def transpose_in(x, h):
        s = x.shape
        t = x.view([s[0], s[1], h, -1]).transpose(1, 2)
        s2 = t.shape
        return Reshape(t, [-1, s2[2], s2[3]]).value

    def transpose_out(x, h):
        s = x.shape
        t = x.view([-1, h, s[1], s[2]]).transpose(1, 2)
        s2 = t.shape
        return Reshape(t, [s2[0], s2[1], -1]).value
    q1, k1, v1 = mm1(query, wQ, True), mm1(keys, wK, True), mm1(values, wV, True)
    nQ, nK, nV, nB = q1.shape[1], k1.shape[1], v1.shape[1], q1.shape[0]
    aligned = nQ % 8 == 0 and nK % 8 == 0 and nV % 8 == 0
    efficient = is_cuda and aligned and nQ == nK and not linearized and (causalMask or maxLength is None)
    if efficient:
        att = ScaledDotProductAttention(q1.view([nB, nQ, numHeads, -1]), k1.view([nB, nQ, numHeads, -1]), v1.view([nB, nQ, numHeads, -1]),
                                        causalMask).value.flatten(2, 3)
    else:
        q1t, k1t, v1t = transpose_in(q1, numHeads), transpose_in(k1, numHeads), transpose_in(v1, numHeads)
        if causalMask and maxLength is None:
            single = torch.arange(1, nQ + 1, 1, dtype=q1t.value.dtype).unsqueeze(0)
            mx = single.repeat(nB * numHeads, 1)
        elif maxLength is not None:
            mx = maxLength.repeat(numHeads, 1)
        else:
            mx = None
        out = linearized_attention(q1t, k1t, v1t, mx) if linearized else scaled_dot_product_attention(q1t, k1t, v1t, mx)
        att = transpose_out(out, numHeads)
    return mm1(att, wO, True)


class MultiheadAttention(Module):
    def __init__(self, wQ, wK, wV, wO, numHeads, linearized, causalMask, is_cuda=False):
        self.wQ, self.wK, self.wV, self.wO = wQ, wK, wV, wO
        self.numHeads, self.linearized, self.causalMask, self.is_cuda = numHeads, linearized, causalMask, is_cuda

    def state(self): return [self.wQ, self.wK, self.wV, self.wO]

    def forward(self, q, k=None, v=None, maxLength=None):
        k = q if k is None else k
        v = q if v is None else v
        return multihead_attention(q, k, v, maxLength, self.wQ, self.wK, self.wV, self.wO, self.numHeads, self.linearized, self.causalMask,
                                   self.is_cuda)


def layer_norm(x: Variable, shape) -> Variable:    # LayerNorm(List(in), tOpt): no scale, no bias, eps 1e-5
    return LayerNormOp(x, None, None, shape, 1e-5).value


class TransformerEncoderBlock(Module):
    def __init__(self, attention, w1, b1, w2, b2, scale1, scale2, gptOrder):
        self.attention, self.w1, self.b1, self.w2, self.b2, self.scale1, self.scale2, self.gptOrder = attention, w1, b1, w2, b2, scale1, scale2, gptOrder

    def state(self): return self.attention.state() + [self.w1, self.w2, self.b1, self.b2, self.scale1, self.scale2]

    def forward(self, x, maxLength=None):
        n = [x.shape[-1]]
        if self.gptOrder:
            a1 = layer_norm(x, n)
            a2 = self.attention.forward(a1, a1, a1, maxLength) * self.scale1 + x
            a3 = layer_norm(a2, n)
            return (mm1((mm1(a3, self.w1) + self.b1).gelu(), self.w2) + self.b2) * self.scale2 + a2
        a1 = self.attention.forward(x, x, x, maxLength)
        a2 = layer_norm(a1 + x, n)
        a3 = mm1((mm1(a2, self.w1) + self.b1).gelu(), self.w2) + self.b2
        return layer_norm(a3 + a3, n)


class TransformerEncoder(Module):
    def __init__(self, blocks): self.blocks = blocks
    def state(self): return [s for b in self.blocks for s in b.state()]

    def forward(self, x, maxLength=None):
        for b in self.blocks:
            x = b.forward(x, maxLength)
        return x


class TransformerDecoderBlock(Module):
    def __init__(self, attDD, attED, w1, b1, w2, b2):
        self.attDD, self.attED, self.w1, self.b1, self.w2, self.b2 = attDD, attED, w1, b1, w2, b2

    def state(self): return self.attDD.state() + self.attED.state() + [self.w1, self.w2, self.b1, self.b2]

    def forward(self, decoderInput, encoderOutput, maxLength=None):
        n = [decoderInput.shape[-1]]
        a1 = layer_norm(decoderInput, n)
        a2 = self.attDD.forward(a1, a1, a1, maxLength) + decoderInput
        a3 = layer_norm(a2, n)
        a4 = layer_norm(encoderOutput, n)
        a5 = a2 + self.attED.forward(a3, a4, a4, None)
        a6 = layer_norm(a5, n)
        return mm1((mm1(a6, self.w1) + self.b1).gelu(), self.w2) + self.b2 + a5


def vaswani(sequenceLength, dimension) -> torch.Tensor:
    m = torch.zeros(sequenceLength, dimension, dtype=torch.float64)
    for i in range(sequenceLength):
        for j in range(dimension // 2):
            a = i / math.pow(10000.0, (2.0 * j) / dimension)
            m[i, 2 * j] = math.sin(a)
            if 2 * j + 1 < dimension:
                m[i, 2 * j + 1] = math.cos(a)
    return m


class LanguageModel(Module):      # lm.scala:137-190 + LanguageModelLoss :44-59
    def __init__(self, tokenEmbedding: Variable, positionEmbedding: Variable, encoder: TransformerEncoder, padToken=-100):
        self.tokenEmbedding, self.positionEmbedding, self.encoder, self.padToken = tokenEmbedding, positionEmbedding, encoder, padToken

    def state(self): return [self.tokenEmbedding, self.positionEmbedding] + self.encoder.state()

    def forward(self, tokens: torch.Tensor, maxLength=None, positions=None):
        pos = const(torch.arange(0, tokens.shape[1], 1, dtype=tokens.dtype).unsqueeze(0))
        embedded = Embedding(const(tokens), self.tokenEmbedding).value + Embedding(pos, self.positionEmbedding).value
        enc = self.encoder.forward(embedded, maxLength)
        encoded = layer_norm(enc, [enc.shape[-1]])
        at = encoded
        if positions is not None:
            e = encoded.shape[2]
            at = encoded.view([-1, e]).indexSelect(0, const(positions.view(-1))).view([encoded.shape[0], positions.shape[1], e])
        logits = mm1(at, self.tokenEmbedding.transpose(0, 1))
        return encoded, logits

    def loss(self, tokens, target, maxLength=None, positions=None):
        _, logits = self.forward(tokens, maxLength, positions)
        weights = torch.ones(self.tokenEmbedding.shape[0], dtype=logits.value.dtype)
        return logits.logSoftMax(2).flatten(0, 1).nllLoss(target.view(-1), weights, 1, self.padToken)
