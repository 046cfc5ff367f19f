"""lamp.nn's transformer family and lamp.nn.languagemodel over the host C ABI.

Reference: lamp-core/src/main/scala/lamp/nn/Transformer.scala, nn/Embedding.scala, nn/languagemodel/lm.scala.  Modules whose
input is a tuple / case class in the reference take the same members here, `None` for an absent Option.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from ._capi import lib, handle_array
from .autograd import Variable
from .nn import Module, _mk
from .sten import STen, F32


def _forward_multi(m: Module, vars_: Sequence[Variable], tensors: Sequence[Optional[STen]]) -> Variable:
    hs = handle_array([v.h for v in vars_])
    ts = (C.c_void_p * max(1, len(tensors)))(*[(t.h if t is not None else None) for t in tensors])
    o = C.c_void_p()
    lib.lamp_module_forward_multi(m.h, hs, len(vars_), ts, len(tensors), C.byref(o))
    return Variable(o)


class _Multi(Module):
    def __init__(self, m: Module):
        super().__init__(m.h); m.h = None


class Embedding(_Multi):
    def __init__(self, classes, dimensions, dtype=F32, device=0):
        super().__init__(_mk("lamp_module_embedding", classes, dimensions, dtype, device))


class MultiheadAttention(_Multi):
    """MultiheadAttention.apply (Transformer.scala:619-641); forward((q, k, v, maxLength))."""
    def __init__(self, dQ, dK, dV, hiddenPerHead, out, dropout, numHeads, dtype=F32, device=0, linearized=False, causalMask=False):
        super().__init__(_mk("lamp_module_multihead_attention", dQ, dK, dV, hiddenPerHead, out, float(dropout), numHeads, dtype, device,
                             int(linearized), int(causalMask)))

    def forward(self, q, k=None, v=None, maxLength: Optional[STen] = None):
        vs = [q] if k is None else [q, k, v]
        return _forward_multi(self, vs, [maxLength])

    @staticmethod
    def fusedCallAsWritten(on: bool) -> bool:
        """process-wide switch (see lamp_attention_fused_call_as_written in include/lamp_host.h); returns the previous setting.
        Default False: the fused branch computes what the reference's CPU path computes (per-head attention over the sequence)."""
        prev = C.c_int(0); lib.lamp_attention_fused_call_as_written(int(bool(on)), C.byref(prev)); return bool(prev.value)

    @staticmethod
    def sequenceMask(maxLength: STen, maskable: Variable, fill: float) -> Variable:
        o = C.c_void_p(); lib.lamp_sequence_mask(C.byref(o), maxLength.h, maskable.h, float(fill)); return Variable(o)
    sequenceMaskValidLength1D = sequenceMask
    sequenceMaskValidLength2D = sequenceMask

    @staticmethod
    def maskedSoftmax(input: Variable, maxLength: STen) -> Variable:
        o = C.c_void_p(); lib.lamp_masked_softmax(C.byref(o), input.h, maxLength.h); return Variable(o)


class TransformerEncoderBlock(_Multi):
    def __init__(self, in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, out, dropout, dtype=F32, device=0, linearized=False,
                 gptOrder=False, causalMask=False):
        super().__init__(_mk("lamp_module_transformer_encoder_block", in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, out,
                             float(dropout), dtype, device, int(linearized), int(gptOrder), int(causalMask)))

    def forward(self, x, maxLength: Optional[STen] = None): return _forward_multi(self, [x], [maxLength])


class TransformerEncoder(_Multi):
    def __init__(self, numBlocks, in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, dropout, dtype=F32, device=0, linearized=False,
                 gptOrder=False, causalMask=False):
        super().__init__(_mk("lamp_module_transformer_encoder", numBlocks, in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim,
                             float(dropout), dtype, device, int(linearized), int(gptOrder), int(causalMask)))

    def forward(self, x, maxLength: Optional[STen] = None): return _forward_multi(self, [x], [maxLength])


class TransformerDecoderBlock(_Multi):
    def __init__(self, in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, out, dropout, dtype=F32, device=0, linearized=False,
                 decoderDecoderCausalMask=True, encoderDecoderCausalMask=False):
        super().__init__(_mk("lamp_module_transformer_decoder_block", in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, out,
                             float(dropout), dtype, device, int(linearized), int(decoderDecoderCausalMask), int(encoderDecoderCausalMask)))

    def forward(self, decoderInput, encoderOutput, maxLength: Optional[STen] = None):
        return _forward_multi(self, [decoderInput, encoderOutput], [maxLength])


class Transformer(_Multi):
    def __init__(self, numBlocks, in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, dropout, dtype=F32, device=0, linearized=False,
                 encoderCausalMask=False, decoderDecoderCausalMask=True, encoderDecoderCausalMask=False):
        super().__init__(_mk("lamp_module_transformer", numBlocks, in_, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, float(dropout),
                             dtype, device, int(linearized), int(encoderCausalMask), int(decoderDecoderCausalMask), int(encoderDecoderCausalMask)))

    def forward(self, decoderInput, encoderInput, decoderMaxLength: Optional[STen] = None, encoderMaxLength: Optional[STen] = None):
        return _forward_multi(self, [decoderInput, encoderInput], [decoderMaxLength, encoderMaxLength])


class PositionalEmbedding:
    @staticmethod
    def vaswani(sequenceLength, dimension, dtype=F32, device=0) -> STen:
        o = C.c_void_p(); lib.lamp_positional_embedding_vaswani(C.byref(o), sequenceLength, dimension, dtype, device); return STen(o)


class TransformerEmbedding(_Multi):
    def __init__(self, embedding: Embedding, addPositionalEmbedding: bool, positionalEmbedding: STen):
        super().__init__(_mk("lamp_module_transformer_embedding", embedding.h, int(addPositionalEmbedding), positionalEmbedding.h))
        self._keep = (embedding, positionalEmbedding)


class LanguageModelModule(_Multi):
    """LanguageModelModule.apply (lm.scala:194-232); forward returns (encoded, languageModelLogits)."""
    def __init__(self, maxLength, vocabularySize, numBlocks, embeddingDim, attentionHiddenPerHeadDim, attentionNumHeads, encoderMlpHiddenDim,
                 dropout, dtype=F32, device=0, linearized=False):
        super().__init__(_mk("lamp_module_language_model", maxLength, vocabularySize, numBlocks, embeddingDim, attentionHiddenPerHeadDim,
                             attentionNumHeads, encoderMlpHiddenDim, float(dropout), dtype, device, int(linearized)))

    def forward(self, tokens: Variable, maxLength: Optional[STen] = None, positions: Optional[STen] = None):
        e, l = C.c_void_p(), C.c_void_p()
        lib.lamp_language_model_forward(self.h, tokens.h, maxLength.h if maxLength is not None else None,
                                        positions.h if positions is not None else None, C.byref(e), C.byref(l))
        return Variable(e), Variable(l)


class LanguageModelLoss(_Multi):
    """LanguageModelLoss.apply (lm.scala:63-91); forward(LossInput) -> mean NLL ignoring padToken."""
    def __init__(self, maxLength, vocabularySize, numBlocks, embeddingDim, attentionHiddenPerHeadDim, attentionNumHeads, encoderMlpHiddenDim,
                 dropout, padToken, dtype=F32, device=0, linearized=False):
        super().__init__(_mk("lamp_module_language_model_loss", maxLength, vocabularySize, numBlocks, embeddingDim, attentionHiddenPerHeadDim,
                             attentionNumHeads, encoderMlpHiddenDim, float(dropout), padToken, dtype, device, int(linearized)))

    def forward(self, tokens: Variable, languageModelTarget: STen, maxLength: Optional[STen] = None, positions: Optional[STen] = None):
        return _forward_multi(self, [tokens], [languageModelTarget, maxLength, positions])

    def languageModel(self, tokens: Variable, maxLength: Optional[STen] = None, positions: Optional[STen] = None):
        return LanguageModelModule.forward(self, tokens, maxLength, positions)
