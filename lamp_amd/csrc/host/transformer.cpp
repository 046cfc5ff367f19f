// lamp.nn's transformer family - see transformer.h for the reference map.
#include "transformer.h"

namespace lamp {
namespace host {

namespace {
Var init_linear(int64_t in, int64_t out, int dtype, int device) {     // nn/package.scala:102-109
  return make_param(ops::normal(0.0, std::sqrt(2.0 / (double)(out + in)), {in, out}, dtype, device));
}
// a.view(-1, last).mm(b).view(shape.dropRight(1) :+ -1)
Var mm1(const Var& a, const Var& b, bool reshape_first = false) {
  auto shape = a->shape();
  Var a2 = reshape_first ? F::reshape(a, {-1, shape.back()}) : F::view(a, {-1, shape.back()});
  shape.back() = -1;
  return F::view(F::mm(a2, b), shape);
}
// mm1(a, w) + b with the bias folded into the GEMM epilogue (same values as the chain, see F::linear_bias)
Var mm1_bias(const Var& a, const Var& w, const Var& b) {
  static const bool fused = !(getenv("LAMP_LINEAR_BIAS_FUSED") && atoi(getenv("LAMP_LINEAR_BIAS_FUSED")) == 0);
  const auto bsh = b->shape();
  if (!fused || bsh.size() != 2 || bsh[0] != 1 || bsh[1] != w->value.size(1)) return F::add(mm1(a, w), b);
  auto shape = a->shape();
  Var a2 = F::view(a, {-1, shape.back()});
  shape.back() = -1;
  return F::view(F::linear_bias(a2, w, b), shape);
}
// x * scale + residual (Transformer.scala:244-247) in one pass with the values of the chain
Var mult_add(const Var& x, const Var& scale, const Var& residual) {
  static const bool fused = !(getenv("LAMP_MULT_ADD_FUSED") && atoi(getenv("LAMP_MULT_ADD_FUSED")) == 0);
  if (!fused || x->value.dtype() != scale->value.dtype() || x->value.dtype() != residual->value.dtype()) return F::add(F::mult(x, scale), residual);
  return F::mult_add(x, scale, residual);
}
Ten arange_like(int64_t start, int64_t end, const Ten& options) {
  lamp_tensor* o = nullptr;
  HCALL(lamp_arange(&o, (double)start, (double)end, 1.0, options.dtype(), options.device()));
  return Ten(o);
}
Ten unsqueeze(const Ten& a, int64_t dim) { lamp_tensor* o = nullptr; HCALL(lamp_unsqueeze(&o, a.h(), dim)); return Ten(o); }
// ATen's ge promotes a floating tensor against a long tensor to the floating type
Ten ge(const Ten& a, const Ten& b) {
  Ten b2 = b.dtype() == a.dtype() ? b : ops::cast(b, a.dtype());
  lamp_tensor* o = nullptr; HCALL(lamp_ge(&o, a.h(), b2.h())); return Ten(o);
}
// ATen repeat: leading dimensions are added when `reps` is longer than the shape, then every dimension is tiled
Ten repeat(const Ten& a, const std::vector<int64_t>& reps) {
  std::vector<int64_t> shape = a.shape();
  LAMP_CHECK(reps.size() >= shape.size(), "repeat: number of repeat dims can not be smaller than number of dims of tensor");
  while (shape.size() < reps.size()) shape.insert(shape.begin(), 1);
  // [s0, s1, ...] -> view [1, s0, 1, s1, ...] -> expand [r0, s0, r1, s1, ...] -> reshape [r0 * s0, ...]
  std::vector<int64_t> v, e, r;
  for (size_t i = 0; i < shape.size(); i++) {
    v.push_back(1); v.push_back(shape[i]);
    e.push_back(reps[i]); e.push_back(shape[i]);
    r.push_back(reps[i] * shape[i]);
  }
  Ten viewed = ops::reshape(a, v);
  lamp_tensor* o = nullptr;
  HCALL(lamp_expand(&o, viewed.h(), e.data(), (int)e.size()));
  return ops::reshape(Ten(o), r);
}
Var swish1(const Var& x) { return F::mult(x, F::sigmoid(x)); }
}  // namespace

// ---- Embedding ---------------------------------------------------------------------------------------
std::shared_ptr<Embedding> Embedding::make(int64_t classes, int64_t dimensions, int dtype, int device) {
  auto m = std::make_shared<Embedding>();
  m->weights = make_param(ops::normal(0.0, std::sqrt(2.0 / (double)(classes + dimensions)), {classes, dimensions}, dtype, device));
  return m;
}

// ---- MultiheadAttention --------------------------------------------------------------------------------
std::shared_ptr<MultiheadAttention> MultiheadAttention::make(int64_t dQ, int64_t dK, int64_t dV, int64_t hiddenPerHead, int64_t out, double dropout,
                                                             int64_t numHeads, int dtype, int device, bool linearized, bool causalMask) {
  auto m = std::make_shared<MultiheadAttention>();
  m->wQ = init_linear(dQ, hiddenPerHead * numHeads, dtype, device);
  m->wK = init_linear(dK, hiddenPerHead * numHeads, dtype, device);
  m->wV = init_linear(dV, hiddenPerHead * numHeads, dtype, device);
  m->wO = init_linear(hiddenPerHead * numHeads, out, dtype, device);
  m->dropout = dropout; m->train = true; m->numHeads = numHeads; m->linearized = linearized; m->causalMask = causalMask;
  return m;
}
Var MultiheadAttention::attend(const Var& q, const Var& k, const Var& v, const Ten& maxLength) {
  return multiheadAttention(q, k, v, maxLength, dropout, train, wQ, wK, wV, wO, numHeads, linearized, causalMask);
}
Var MultiheadAttention::forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) {
  LAMP_CHECK(xs.size() == 1 || xs.size() == 3, "MultiheadAttention takes (query, keys, values, maxLength?)");
  const Ten mx = aux.empty() ? Ten() : aux[0];
  return xs.size() == 3 ? attend(xs[0], xs[1], xs[2], mx) : attend(xs[0], xs[0], xs[0], mx);
}

// Transformer.scala:667-749.  2-D maxLength: cell (i, j, k) is masked iff k >= maxLength(i, j); 1-D: iff k >= maxLength(i).
// The arange is made with the maskable's options (its floating type), the comparison promotes against the long tensor.
Var MultiheadAttention::sequenceMask(const Ten& maxLength, const Var& maskable, double fill) {
  const Ten& mv = maskable->value;
  LAMP_CHECK(mv.ndim() == 3, "sequenceMask: maskable is batch x seq x ???");
  Ten mask;
  if (maxLength.ndim() == 2) {
    LAMP_CHECK(maxLength.size(1) == mv.size(1) && maxLength.size(0) == mv.size(0), "assertion failed (sequenceMaskValidLength2D)");
    mask = ge(ops::view(arange_like(0, mv.size(2), mv), {1, 1, -1}), unsqueeze(maxLength, 2));
  } else {
    LAMP_CHECK(maxLength.ndim() == 1 && maxLength.size(0) == mv.size(0), "assertion failed (sequenceMaskValidLength1D)");
    mask = unsqueeze(ge(unsqueeze(arange_like(0, mv.size(2), mv), 0), unsqueeze(maxLength, 1)), 1);
  }
  return F::mask_fill(maskable, mask, fill);
}
Var MultiheadAttention::maskedSoftmax(const Var& input, const Ten& maxLength) {   // :751-762
  return F::exp(F::log_softmax(sequenceMask(maxLength, input, -INFINITY), 2));
}
Var MultiheadAttention::scaledDotProductAttention(const Var& q, const Var& k, const Var& v, const Ten& maxLength, double dropout, bool trainDropout) {   // :784-804
  const double d = (double)q->value.size(2);
  Var scores = F::const_mult(F::bmm(q, F::transpose(k, 1, 2)), 1.0 / std::sqrt(d));
  // maxLength.fold(scores)(mx => maskedSoftmax(scores, mx)): without a mask the scores are used as they are
  Var weights = F::dropout(maxLength.defined() ? maskedSoftmax(scores, maxLength) : scores, dropout, trainDropout);
  return F::bmm(weights, v);
}
Var MultiheadAttention::linearizedAttention(const Var& q, const Var& k, const Var& v, const Ten& maxLength, double dropout, bool trainDropout) {   // :826-854
  Var qF = F::const_add(swish1(q), 1.0);
  Var maskable = F::dropout(F::const_add(swish1(k), 1.0), dropout, trainDropout);
  Var kF = maxLength.defined() ? sequenceMask(maxLength, maskable, 0.0) : maskable;
  Var tmp1 = F::bmm(F::transpose(kF, 1, 2), v);
  Var tmp2 = F::transpose(F::sum(kF, {1}, true), 1, 2);
  Var enumerator = F::bmm(qF, tmp1);
  Var denom = F::bmm(qF, tmp2);
  return F::div(enumerator, F::const_add(denom, 1e-5));
}
bool& MultiheadAttention::fused_call_as_written() {
  static bool v = [] { const char* e = getenv("LAMP_ATTENTION_AS_WRITTEN_FOR_CUDA"); return e && e[0] == '1'; }();
  return v;
}
Var MultiheadAttention::multiheadAttention(const Var& query, const Var& keys, const Var& values, const Ten& maxLength, double dropout,
                                           bool trainDropout, const Var& wQuery, const Var& wKeys, const Var& wValues, const Var& wOutput,
                                           int64_t numHeads, bool linearized, bool causalMask) {   // :889-1006
  auto transposeIn = [](const Var& x, int64_t h) {     // a x b x c -> (a * h) x b x (c / h)
    auto s = x->shape();
    Var t = F::transpose(F::view(x, {s[0], s[1], h, -1}), 1, 2);
    auto s2 = t->shape();
    return F::reshape(t, {-1, s2[2], s2[3]});
  };
  auto transposeOut = [](const Var& x, int64_t h) {    // (a * h) x b x c -> a x b x (c * h)
    auto s = x->shape();
    Var t = F::transpose(F::view(x, {-1, h, s[1], s[2]}), 1, 2);
    auto s2 = t->shape();
    return F::reshape(t, {s2[0], s2[1], -1});
  };
  // self-attention on bf16 with the fused kernels' head widths: one projection product, packed operands (LAMP_FUSE_QKV=0: three products)
  static const bool fuse_qkv = [] { const char* e = getenv("LAMP_FUSE_QKV"); return !(e && e[0] == '0'); }();
  if (fuse_qkv && !fused_call_as_written() && query.get() == keys.get() && keys.get() == values.get() && causalMask && !maxLength.defined() && !linearized &&
      (dropout == 0.0 || !trainDropout) && query->value.dtype() == kBF16 && query->value.ndim() == 3 && query->value.h()->is_device() &&
      wQuery->shape() == wKeys->shape() && wKeys->shape() == wValues->shape() && wQuery->value.size(1) % numHeads == 0 &&
      (wQuery->value.size(1) / numHeads == 64 || wQuery->value.size(1) / numHeads == 128) && query->value.size(1) % 8 == 0)
    return mm1(F::packed_self_attention(query, wQuery, wKeys, wValues, numHeads, true), wOutput, true);
  Var q1 = mm1(query, wQuery, true), k1 = mm1(keys, wKeys, true), v1 = mm1(values, wValues, true);
  const int64_t nQ = q1->value.size(1), nK = k1->value.size(1), nV = v1->value.size(1), nB = q1->value.size(0);
  const bool aligned = nQ % 8 == 0 && nK % 8 == 0 && nV % 8 == 0;
  Var attention;
  if (fused_call_as_written()) {
    // opt-in: the reference's CUDA branch literally.  isCuda is always true behind this library
    const bool useEfficientAttentionKernel = aligned && nQ == nK && !linearized && (causalMask || !maxLength.defined()) && (dropout == 0.0 || !trainDropout);
    if (useEfficientAttentionKernel) {
      // (batch, sequence, heads, d) views (:930-945); the operator reads dimension 1 as the heads and dimension 2 as the sequence
      attention = F::flatten(F::scaled_dot_product_attention(F::view(q1, {nB, nQ, numHeads, -1}), F::view(k1, {nB, nQ, numHeads, -1}),
                                                             F::view(v1, {nB, nQ, numHeads, -1}), causalMask), 2, 3);
      return mm1(attention, wOutput, true);
    }
  }
  // Default: the arithmetic of the composed branch, which is what the reference's CPU path always runs.  With a causal mask and no
  // explicit lengths that branch is softmax(Q K^T / sqrt(d) + causal mask) V per (batch, head) - exactly the fused operator on
  // (batch, heads, sequence, d) operands, so the flash kernels take strided views of the projections (no transposeIn copies) and
  // hand back a (batch, heads, sequence, d) view of (batch, sequence, heads, d) storage (transposeOut is free).  Every other case
  // (no mask: raw scores without softmax, :797-801; explicit maxLength; linearized; active dropout) runs composed as written.
  const bool fusedEqualsComposed = causalMask && !maxLength.defined() && nQ == nK && !linearized && (dropout == 0.0 || !trainDropout) &&
                                   q1->value.dtype() == kBF16;
  if (fusedEqualsComposed) {
    auto heads_first = [&](const Var& x, int64_t n) { return F::transpose(F::view(x, {nB, n, numHeads, -1}), 1, 2); };   // (B, H, S, d) view
    Var o = F::scaled_dot_product_attention(heads_first(q1, nQ), heads_first(k1, nK), heads_first(v1, nV), true);
    attention = F::flatten(F::transpose(o, 1, 2), 2, 3);
  } else {
    Var q1t = transposeIn(q1, numHeads), k1t = transposeIn(k1, numHeads), v1t = transposeIn(v1, numHeads);
    Ten maxLengthRepeated;
    if (causalMask && !maxLength.defined()) {
      Ten single = unsqueeze(arange_like(1, nQ + 1, q1t->value), 0);
      maxLengthRepeated = repeat(single, {nB * numHeads, 1});
    } else if (maxLength.defined()) {
      maxLengthRepeated = repeat(maxLength, {numHeads, 1});
    }
    Var output = linearized ? linearizedAttention(q1t, k1t, v1t, maxLengthRepeated, dropout, trainDropout)
                            : scaledDotProductAttention(q1t, k1t, v1t, maxLengthRepeated, dropout, trainDropout);
    attention = transposeOut(output, numHeads);
  }
  return mm1(attention, wOutput, true);
}

// LayerNorm(List(in), tOpt): neither scale nor bias (nn/LayerNorm.scala:44-50 defaults) - the norms of the blocks carry no state.
// ---- encoder -------------------------------------------------------------------------------------------
std::shared_ptr<TransformerEncoderBlock> TransformerEncoderBlock::make(int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                                                       int64_t mlpHiddenDim, int64_t out, double dropout, int dtype, int device,
                                                                       bool linearized, bool gptOrder, bool causalMask) {   // :492-530
  auto m = std::make_shared<TransformerEncoderBlock>();
  m->attention = MultiheadAttention::make(in, in, in, attentionHiddenPerHeadDim, in, dropout, attentionNumHeads, dtype, device, linearized, causalMask);
  m->gptOrder = gptOrder;
  m->layerNorm1 = std::static_pointer_cast<LayerNorm>(LayerNorm::make({in}, dtype, device, false, false));
  m->layerNorm2 = std::static_pointer_cast<LayerNorm>(LayerNorm::make({in}, dtype, device, false, false));
  m->w1 = init_linear(in, mlpHiddenDim, dtype, device);
  m->b1 = make_param(ops::zeros({1, mlpHiddenDim}, dtype, device));
  m->w2 = init_linear(mlpHiddenDim, out, dtype, device);
  m->b2 = make_param(ops::zeros({1, out}, dtype, device));
  m->scale1 = make_param(ops::normal(0.0, 0.0001, {in}, dtype, device));
  m->scale2 = make_param(ops::normal(0.0, 0.0001, {in}, dtype, device));
  m->dropout = dropout; m->train = true;
  return m;
}
void TransformerEncoderBlock::collect_state(std::vector<Var>& o) {   // :227-235
  attention->collect_state(o); layerNorm1->collect_state(o); layerNorm2->collect_state(o);
  o.push_back(w1); o.push_back(w2); o.push_back(b1); o.push_back(b2); o.push_back(scale1); o.push_back(scale2);
}
Var TransformerEncoderBlock::block(const Var& input, const Ten& maxLength) {   // :237-258
  if (gptOrder) {
    Var a1 = layerNorm1->forward(F::dropout(input, dropout, train));
    Var a2 = mult_add(attention->attend(a1, a1, a1, maxLength), scale1, input);
    Var a3 = layerNorm2->forward(F::dropout(a2, dropout, train));
    Var a4 = mult_add(mm1_bias(F::gelu(mm1_bias(a3, w1, b1)), w2, b2), scale2, a2);
    return a4;
  }
  Var a1 = attention->attend(input, input, input, maxLength);
  Var a2 = layerNorm1->forward(F::add(F::dropout(a1, dropout, train), input));
  Var a3 = mm1_bias(F::gelu(mm1_bias(a2, w1, b1)), w2, b2);
  return layerNorm2->forward(F::add(F::dropout(a3, dropout, train), a3));   // a3.dropout + a3 (sic, :255)
}
std::shared_ptr<TransformerEncoder> TransformerEncoder::make(int64_t numBlocks, int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                                             int64_t mlpHiddenDim, double dropout, int dtype, int device, bool linearized,
                                                             bool gptOrder, bool causalMask) {   // :76-102
  auto m = std::make_shared<TransformerEncoder>();
  for (int64_t i = 0; i < numBlocks; i++)
    m->blocks.push_back(TransformerEncoderBlock::make(in, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, in, dropout, dtype, device,
                                                      linearized, gptOrder, causalMask));
  return m;
}

// ---- decoder -------------------------------------------------------------------------------------------
std::shared_ptr<TransformerDecoderBlock> TransformerDecoderBlock::make(int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                                                       int64_t mlpHiddenDim, int64_t out, double dropout, int dtype, int device,
                                                                       bool linearized, bool ddCausal, bool edCausal) {   // :384-432
  auto m = std::make_shared<TransformerDecoderBlock>();
  m->attentionDecoderDecoder = MultiheadAttention::make(in, in, in, attentionHiddenPerHeadDim, in, dropout, attentionNumHeads, dtype, device, linearized, ddCausal);
  m->attentionEncoderDecoder = MultiheadAttention::make(in, in, in, attentionHiddenPerHeadDim, in, dropout, attentionNumHeads, dtype, device, linearized, edCausal);
  auto ln = [&]() { return std::static_pointer_cast<LayerNorm>(LayerNorm::make({in}, dtype, device, false, false)); };
  m->layerNorm1 = ln(); m->layerNorm2 = ln(); m->layerNorm3 = ln(); m->layerNorm4 = ln();
  m->w1 = init_linear(in, mlpHiddenDim, dtype, device);
  m->b1 = make_param(ops::zeros({1, mlpHiddenDim}, dtype, device));
  m->w2 = init_linear(mlpHiddenDim, out, dtype, device);
  m->b2 = make_param(ops::zeros({1, out}, dtype, device));
  m->dropout = dropout; m->train = true;
  return m;
}
void TransformerDecoderBlock::collect_state(std::vector<Var>& o) {   // :278-284
  attentionDecoderDecoder->collect_state(o); attentionEncoderDecoder->collect_state(o);
  layerNorm1->collect_state(o); layerNorm2->collect_state(o); layerNorm3->collect_state(o); layerNorm4->collect_state(o);
  o.push_back(w1); o.push_back(w2); o.push_back(b1); o.push_back(b2);
}
Var TransformerDecoderBlock::block(const Var& decoderInput, const Var& encoderOutput, const Ten& maxLength) {   // :286-305
  Var a1 = layerNorm1->forward(F::dropout(decoderInput, dropout, train));
  Var a2 = F::add(attentionDecoderDecoder->attend(a1, a1, a1, maxLength), decoderInput);
  Var a3 = layerNorm2->forward(F::dropout(a2, dropout, train));
  Var a4 = layerNorm3->forward(F::dropout(encoderOutput, dropout, train));
  Var a5 = F::add(a2, attentionEncoderDecoder->attend(a3, a4, a4, Ten()));
  Var a6 = layerNorm4->forward(F::dropout(a5, dropout, train));
  return F::add(mm1_bias(F::gelu(mm1_bias(a6, w1, b1)), w2, b2), a5);
}
std::shared_ptr<Transformer> Transformer::make(int64_t numBlocks, int64_t in, int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads,
                                               int64_t mlpHiddenDim, double dropout, int dtype, int device, bool linearized, bool encoderCausalMask,
                                               bool ddCausal, bool edCausal) {   // :330-365
  auto m = std::make_shared<Transformer>();
  m->encoder = TransformerEncoder::make(numBlocks, in, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, dropout, dtype, device, linearized,
                                        true, encoderCausalMask);
  m->decoder = std::make_shared<TransformerDecoder>();
  for (int64_t i = 0; i < numBlocks; i++)
    m->decoder->blocks.push_back(TransformerDecoderBlock::make(in, attentionHiddenPerHeadDim, attentionNumHeads, mlpHiddenDim, in, dropout, dtype, device,
                                                               linearized, ddCausal, edCausal));
  return m;
}

// ---- embeddings -----------------------------------------------------------------------------------------
Ten positional_embedding_vaswani(int64_t sequenceLength, int64_t dimension, int dtype, int device) {   // :1022-1045
  std::vector<double> m((size_t)(sequenceLength * dimension), 0.0);
  const int64_t N = dimension / 2;
  for (int64_t i = 0; i < sequenceLength; i++)
    for (int64_t j = 0; j < N; j++) {
      const double a = (double)i / std::pow(10000.0, (2.0 * (double)j) / (double)dimension);
      m[i * dimension + 2 * j] = std::sin(a);
      if (2 * j + 1 < dimension) m[i * dimension + 2 * j + 1] = std::cos(a);
    }
  Ten t = ops::zeros({sequenceLength, dimension}, /*f64*/ 7, device);
  if (!m.empty()) HCALL(lamp_copy_from_host(t.h(), m.data(), m.size() * sizeof(double)));
  return dtype == 7 ? t : ops::cast(t, dtype);
}
Var TransformerEmbedding::forward(const Var& x) {   // :1114-1124
  Var embedded = embedding->forward(x);
  auto ps = positionalEmbedding->shape();
  ps.insert(ps.begin(), 1);
  Var viewed = F::view(positionalEmbedding, ps);
  if (addPositionalEmbedding) return F::add(embedded, viewed);
  return F::concatenate({embedded, make_const(repeat(viewed->value, {embedded->value.size(0), 1, 1}))}, 2);
}

// ---- language model ---------------------------------------------------------------------------------------
std::shared_ptr<LanguageModelModule> LanguageModelModule::make(int64_t maxLength, int64_t vocabularySize, int64_t numBlocks, int64_t embeddingDim,
                                                               int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads, int64_t encoderMlpHiddenDim,
                                                               double dropout, int dtype, int device, bool linearized) {   // lm.scala:194-232
  auto m = std::make_shared<LanguageModelModule>();
  m->tokenEmbedding = Embedding::make(vocabularySize, embeddingDim, dtype, device);
  m->positionEmbedding = Embedding::make(maxLength, embeddingDim, dtype, device);
  m->encoder = TransformerEncoder::make(numBlocks, embeddingDim, attentionHiddenPerHeadDim, attentionNumHeads, encoderMlpHiddenDim, dropout, dtype, device,
                                        linearized, /*gptOrder*/ true, /*causalMask*/ true);
  m->finalNorm = std::static_pointer_cast<LayerNorm>(LayerNorm::make({embeddingDim}, dtype, device, false, false));
  return m;
}
std::pair<Var, Var> LanguageModelModule::run(const Var& tokens, const Ten& maxLength, const Ten& positions) {   // lm.scala:146-188
  const Ten& tv = tokens->value;
  LAMP_CHECK(tv.ndim() == 2, "LanguageModelInput.tokens is batch x sequence (long)");
  Var pos = make_const(unsqueeze(arange_like(0, tv.size(1), tv), 0));
  Var embedded = F::add(tokenEmbedding->forward(tokens), positionEmbedding->forward(pos));
  Var encoded = finalNorm->forward(encoder->encode(embedded, maxLength));
  Var at = encoded;
  if (positions.defined()) {
    const int64_t e = encoded->value.size(2);
    at = F::view(F::index_select(F::view(encoded, {-1, e}), 0, make_const(ops::view(positions, {-1}))), {encoded->value.size(0), positions.size(1), e});
  }
  Var logits = mm1(at, F::transpose(tokenEmbedding->weights, 0, 1));
  return {encoded, logits};
}
std::shared_ptr<LanguageModelLoss> LanguageModelLoss::make(int64_t maxLength, int64_t vocabularySize, int64_t numBlocks, int64_t embeddingDim,
                                                           int64_t attentionHiddenPerHeadDim, int64_t attentionNumHeads, int64_t encoderMlpHiddenDim,
                                                           double dropout, int64_t padToken, int dtype, int device, bool linearized) {   // lm.scala:63-91
  auto m = std::make_shared<LanguageModelLoss>();
  m->languageModel = LanguageModelModule::make(maxLength, vocabularySize, numBlocks, embeddingDim, attentionHiddenPerHeadDim, attentionNumHeads,
                                               encoderMlpHiddenDim, dropout, dtype, device, linearized);
  m->classWeights = ops::ones({vocabularySize}, dtype, device);
  m->padToken = padToken;
  return m;
}
Var LanguageModelLoss::loss(const Var& tokens, const Ten& target, const Ten& maxLength, const Ten& positions) {   // lm.scala:49-58
  const auto expect = positions.defined() ? positions.shape() : tokens->shape();
  LAMP_CHECK(target.shape() == expect, "assertion failed: languageModelTarget.shape == positions.getOrElse(tokens).shape (lm.scala:29-33)");
  Var logits = languageModel->run(tokens, maxLength, positions).second;
  return F::nll_loss(F::flatten(F::log_softmax(logits, 2), 0, 1), ops::view(target, {-1}), classWeights, /*Mean*/ 1, padToken);
}

}  // namespace host
}  // namespace lamp
