// extern "C" surface of the host layer (include/lamp_host.h).
#include "nn.h"
#include "transformer.h"
#include "../../../include/lamp_host.h"

#include <cstring>
#include <map>

using namespace lamp;
using namespace lamp::host;

struct lamp_optimizer { std::shared_ptr<Optimizer> o; };
struct lamp_model {
  SupervisedModel model;
  DataParallel dp;
};

namespace {
lamp_var* wrap(const Var& v) { auto* r = new lamp_var(); r->v = v; return r; }
lamp_module* wrapm(const Mod& m) { auto* r = new lamp_module(); r->m = m; return r; }
lamp_tensor* give(const Ten& t) {
  if (!t.defined()) return nullptr;
  lamp_tensor* r = nullptr;
  HCALL(lamp_tensor_retain(t.h(), &r));
  return r;
}
}  // namespace

extern "C" {

int lamp_var_const(lamp_var** out, const lamp_tensor* value) { LAMP_API_BEGIN *out = wrap(make_const(borrow(value))); LAMP_API_END }
int lamp_var_param(lamp_var** out, const lamp_tensor* value) { LAMP_API_BEGIN *out = wrap(make_param(borrow(value))); LAMP_API_END }
int lamp_var_value(const lamp_var* v, lamp_tensor** out) { LAMP_API_BEGIN *out = give(v->v->value); LAMP_API_END }
int lamp_var_grad(const lamp_var* v, lamp_tensor** out) { LAMP_API_BEGIN *out = give(v->v->grad_tensor()); LAMP_API_END }
int lamp_var_needs_grad(const lamp_var* v, int* out) { LAMP_API_BEGIN *out = v->v->needsGrad(); LAMP_API_END }
int lamp_var_zero_grad(lamp_var* v) { LAMP_API_BEGIN v->v->zeroGrad(); LAMP_API_END }
int lamp_var_backprop(lamp_var* v) { LAMP_API_BEGIN backprop(v->v); LAMP_API_END }
int lamp_var_wengert_size(const lamp_var* v, int64_t* out) { LAMP_API_BEGIN *out = (int64_t)topological_sort(v->v.get()).size(); LAMP_API_END }
int lamp_var_release(lamp_var* v) { LAMP_API_BEGIN delete v; LAMP_API_END }

int lamp_op_apply(lamp_var** out, const char* name, lamp_var* const* vars, int nvars, lamp_tensor* const* tensors, int ntensors, const double* d,
                  int nd, const int64_t* iv, int ni) {
  LAMP_API_BEGIN
  const std::string n(name);
  auto V = [&](int k) -> Var { LAMP_CHECK(k < nvars, n << ": missing Variable argument " << k); return vars[k] ? vars[k]->v : nullptr; };
  auto T = [&](int k) -> Ten { LAMP_CHECK(k < ntensors && tensors[k], n << ": missing tensor argument " << k); return borrow(tensors[k]); };
  auto D = [&](int k) -> double { LAMP_CHECK(k < nd, n << ": missing double argument " << k); return d[k]; };
  auto I = [&](int k) -> int64_t { LAMP_CHECK(k < ni, n << ": missing long argument " << k); return iv[k]; };
  auto IV = [&](int from, int count) { LAMP_CHECK(from + count <= ni, n << ": missing long arguments"); return std::vector<int64_t>(iv + from, iv + from + count); };
  Var r;
  if (n == "Transpose") r = F::transpose(V(0), ni > 0 ? I(0) : 0, ni > 1 ? I(1) : 1);
  else if (n == "View") r = F::view(V(0), IV(0, ni));
  else if (n == "Reshape") r = F::reshape(V(0), IV(0, ni));
  else if (n == "Flatten") r = F::flatten(V(0), I(0), I(1));
  else if (n == "Concatenate") { std::vector<Var> as; for (int k = 0; k < nvars; k++) as.push_back(V(k)); r = F::concatenate(as, I(0)); }
  else if (n == "Add") r = F::add(V(0), V(1));
  else if (n == "ConstAdd") r = F::const_add(V(0), D(0));
  else if (n == "Minus") r = F::minus(V(0), V(1));
  else if (n == "ConstMult") r = F::const_mult(V(0), D(0));
  else if (n == "Mult") r = F::mult(V(0), V(1));
  else if (n == "Div") r = F::div(V(0), V(1));
  else if (n == "Sum") r = F::sum(V(0), IV(1, ni - 1), I(0) != 0);                 // i = [keepDim, dims...]
  else if (n == "Mean") r = F::mean(V(0), IV(1, ni - 1), I(0) != 0);
  else if (n == "Norm2") r = F::norm2(V(0), IV(1, ni - 1), I(0) != 0);
  else if (n == "MatMul") r = F::mm(V(0), V(1));
  else if (n == "BatchedMatMul") r = F::bmm(V(0), V(1));
  else if (n == "Exp") r = F::exp(V(0));
  else if (n == "Log") r = F::log(V(0));
  else if (n == "Log1p") r = F::log1p(V(0));
  else if (n == "Sin") r = F::sin(V(0));
  else if (n == "Cos") r = F::cos(V(0));
  else if (n == "Tanh") r = F::tanh(V(0));
  else if (n == "PowConst") r = F::pow_const(V(0), D(0));
  else if (n == "Relu") r = F::relu(V(0));
  else if (n == "LeakyRelu") r = F::leaky_relu(V(0), D(0));
  else if (n == "Gelu") r = F::gelu(V(0));
  else if (n == "Sigmoid") r = F::sigmoid(V(0));
  else if (n == "HardSwish") r = F::hardswish(V(0));
  else if (n == "Softplus") r = F::softplus(V(0), D(0), D(1));
  else if (n == "LogSoftMax") r = F::log_softmax(V(0), I(0));
  else if (n == "Dropout") r = F::dropout(V(0), D(0), I(0) != 0);
  else if (n == "NllLoss") r = F::nll_loss(V(0), T(0), T(1), I(0), I(1));            // tensors = [target, weights], i = [reduction, ignore]
  else if (n == "MseLoss") r = F::mse_loss(V(0), T(0), I(0));
  else if (n == "IndexSelect") r = F::index_select(V(0), I(0), V(1));
  else if (n == "EuclideanDistance") r = F::euclidean_distance(V(0), V(1), I(0));
  else if (n == "CappedShiftedNegativeExponential") r = F::capped_shifted_negative_exponential(V(0), D(0));
  else if (n == "ScaledDotProductAttention") r = F::scaled_dot_product_attention(V(0), V(1), V(2), I(0) != 0, ntensors > 0 && tensors[0] ? T(0) : Ten());   // i = [isCausal], tensors = [attentionBias?]
  else if (n == "PackedSelfAttention") r = F::packed_self_attention(V(0), V(1), V(2), V(3), I(0), I(1) != 0);   // (x, wQ, wK, wV), i = [numHeads, isCausal]
  else if (n == "Convolution") {
    // i = [nspatial, stride.., padding.., dilation.., transposed, outputPadding.., groups]
    const int ns = (int)I(0);
    r = F::convolution(V(0), V(1), V(2), IV(1, ns), IV(1 + ns, ns), IV(1 + 2 * ns, ns), I(1 + 3 * ns) != 0, IV(2 + 3 * ns, ns), I(2 + 4 * ns));
  }
  else if (n == "AvgPool2D") r = F::avg_pool2d(V(0), I(0), I(1), I(2));
  else if (n == "MaxPool2D") r = F::max_pool2d(V(0), I(0), I(1), I(2), I(3));
  else if (n == "BatchNorm") r = F::batch_norm(V(0), V(1), V(2), T(0), T(1), I(0) != 0, D(0), D(1));   // tensors = [runningMean, runningVar]
  else if (n == "BatchNorm2D") r = F::batch_norm_2d(V(0), V(1), V(2), T(0), T(1), I(0) != 0, D(0), D(1));
  else if (n == "LayerNormOp") r = F::layer_norm(V(0), V(1), V(2), IV(0, ni), D(0));
  else if (n == "Embedding") r = F::embedding(V(0), V(1));
  else if (n == "MaskFill") r = F::mask_fill(V(0), T(0), D(0));                       // tensors = [mask], d = [fill]
  else if (n == "Stack") { std::vector<Var> as; for (int k = 0; k < nvars; k++) as.push_back(V(k)); r = F::stack(as, I(0)); }
  else if (n == "Select") r = F::select(V(0), I(0), I(1));
  else if (n == "Slice") r = F::slice(V(0), I(0), I(1), I(2), I(3));
  else if (n == "MaskSelect") r = F::mask_select(V(0), V(1));
  else if (n == "IndexFill") r = F::index_fill(V(0), I(0), V(1), D(0));
  else if (n == "Where") r = F::where(T(0), V(0), V(1));                               // tensors = [condition]
  else if (n == "Assign") r = F::assign(V(0), V(1));                                   // (abandon, keep)
  else if (n == "CastToPrecision") r = F::cast_to_precision(V(0), (int)I(0));          // i = [scalar type byte]
  else if (n == "ScatterAdd") r = F::scatter_add(V(0), V(1), I(0), I(1));              // i = [dim, maxIndex]
  else if (n == "IndexAdd") r = F::index_add(V(0), V(1), I(0), I(1));
  else if (n == "IndexAddToTarget") r = F::index_add_to_target(V(0), V(1), V(2), I(0)); // (target, src, index)
  else if (n == "RepeatInterleave") r = F::repeat_interleave(V(0), V(1), I(0));
  else if (n == "ExpandAs") r = F::expand_as(V(0), T(0));
  else if (n == "Expand") r = F::expand(V(0), IV(0, ni));
  else if (n == "Tan") r = F::tan(V(0));
  else if (n == "ArcTan") r = F::atan(V(0));
  else if (n == "Pow") r = F::pow(V(0), V(1));
  else if (n == "ElementWiseMinimum") r = F::minimum(V(0), V(1));
  else if (n == "ElementWiseMaximum") r = F::maximum(V(0), V(1));
  else if (n == "Variance") r = F::variance(V(0), IV(0, ni));
  else if (n == "SquaredFrobeniusMatrixNorm") r = F::squared_frobenius(V(0));
  else if (n == "WeightNorm") r = F::weight_norm(V(0), V(1), I(0));
  else if (n == "SmoothL1Loss") r = F::smooth_l1_loss(V(0), T(0), I(0), D(0));         // tensors = [target], i = [reduction], d = [beta]
  else if (n == "BinaryCrossEntropyWithLogitsLoss") r = F::binary_cross_entropy_with_logits(V(0), T(0), ntensors > 1 && tensors[1] ? T(1) : Ten(), I(0));
  else if (n == "MaxPool1D") r = F::max_pool1d(V(0), I(0), I(1), I(2), I(3));
  else if (n == "Diag") r = F::diag(V(0), I(0));
  else if (n == "Cross") r = F::cross(V(0), V(1), I(0));
  else if (n == "ArgMax") r = F::argmax(V(0), I(0), I(1) != 0);
  else if (n == "OneHot") r = F::one_hot(V(0), I(0));
  else if (n == "EqWhere") r = F::eq_where(V(0), I(0));
  else LAMP_CHECK(false, "unknown Op '" << n << "'");
  *out = wrap(r);
  LAMP_API_END
}

// ---- modules ---------------------------------------------------------------------------------------
int lamp_module_linear(lamp_module** out, int64_t in, int64_t outf, int dtype, int device, int bias) {
  LAMP_API_BEGIN *out = wrapm(Linear::make(in, outf, dtype, device, bias)); LAMP_API_END
}
int lamp_module_conv2d(lamp_module** out, int64_t inC, int64_t outC, int64_t k, int dtype, int device, int bias, int64_t stride, int64_t padding,
                       int64_t dilation, int64_t groups) {
  LAMP_API_BEGIN *out = wrapm(Conv2D::make(inC, outC, k, dtype, device, bias, stride, padding, dilation, groups)); LAMP_API_END
}
int lamp_module_batch_norm(lamp_module** out, int64_t features, int dtype, int device, int two_d) {
  LAMP_API_BEGIN *out = wrapm(BatchNorm::make(features, dtype, device, two_d)); LAMP_API_END
}
int lamp_module_layer_norm(lamp_module** out, const int64_t* shape, int nshape, int dtype, int device, int scale, int bias) {
  LAMP_API_BEGIN *out = wrapm(LayerNorm::make(std::vector<int64_t>(shape, shape + nshape), dtype, device, scale, bias)); LAMP_API_END
}
int lamp_module_dropout(lamp_module** out, double p) { LAMP_API_BEGIN *out = wrapm(std::make_shared<Dropout>(p, true)); LAMP_API_END }
int lamp_module_fun(lamp_module** out, const char* name, double a, double b) { LAMP_API_BEGIN *out = wrapm(make_fun(name, a, b)); LAMP_API_END }
int lamp_module_sequential(lamp_module** out, lamp_module* const* mods, int n) {
  LAMP_API_BEGIN
  std::vector<Mod> ms;
  for (int i = 0; i < n; i++) ms.push_back(mods[i]->m);
  *out = wrapm(std::make_shared<Sequential>(ms));
  LAMP_API_END
}
int lamp_module_residual(lamp_module** out, lamp_module* right, lamp_module* left) {
  LAMP_API_BEGIN *out = wrapm(std::make_shared<Residual>(right->m, left ? left->m : nullptr)); LAMP_API_END
}
int lamp_module_mlp(lamp_module** out, int64_t in, int64_t outf, const int64_t* hidden, int nhidden, int dtype, int device, double dropout,
                    int lastNonLinearity, const char* activation, int norm, int bias) {
  LAMP_API_BEGIN
  *out = wrapm(mlp(in, outf, std::vector<int64_t>(hidden, hidden + nhidden), dtype, device, dropout, lastNonLinearity, activation, norm, bias));
  LAMP_API_END
}
int lamp_module_resnet(lamp_module** out, int64_t num_classes, double dropout, int dtype, int device) {
  LAMP_API_BEGIN *out = wrapm(cnn_resnet(num_classes, dropout, dtype, device)); LAMP_API_END
}
// ---- transformer family (Transformer.scala, lm.scala) -----------------------------------------------------
int lamp_module_embedding(lamp_module** out, int64_t classes, int64_t dimensions, int dtype, int device) {
  LAMP_API_BEGIN *out = wrapm(Embedding::make(classes, dimensions, dtype, device)); LAMP_API_END
}
int lamp_module_multihead_attention(lamp_module** out, int64_t dQ, int64_t dK, int64_t dV, int64_t hidden_per_head, int64_t outf, double dropout,
                                    int64_t num_heads, int dtype, int device, int linearized, int causal_mask) {
  LAMP_API_BEGIN *out = wrapm(MultiheadAttention::make(dQ, dK, dV, hidden_per_head, outf, dropout, num_heads, dtype, device, linearized, causal_mask)); LAMP_API_END
}
int lamp_module_transformer_encoder_block(lamp_module** out, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                                          int64_t mlp_hidden, int64_t outf, double dropout, int dtype, int device, int linearized, int gpt_order,
                                          int causal_mask) {
  LAMP_API_BEGIN
  *out = wrapm(TransformerEncoderBlock::make(in, attention_hidden_per_head, attention_num_heads, mlp_hidden, outf, dropout, dtype, device, linearized,
                                             gpt_order, causal_mask));
  LAMP_API_END
}
int lamp_module_transformer_encoder(lamp_module** out, int64_t num_blocks, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                                    int64_t mlp_hidden, double dropout, int dtype, int device, int linearized, int gpt_order, int causal_mask) {
  LAMP_API_BEGIN
  *out = wrapm(TransformerEncoder::make(num_blocks, in, attention_hidden_per_head, attention_num_heads, mlp_hidden, dropout, dtype, device, linearized,
                                        gpt_order, causal_mask));
  LAMP_API_END
}
int lamp_module_transformer_decoder_block(lamp_module** out, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                                          int64_t mlp_hidden, int64_t outf, double dropout, int dtype, int device, int linearized,
                                          int decoder_decoder_causal_mask, int encoder_decoder_causal_mask) {
  LAMP_API_BEGIN
  *out = wrapm(TransformerDecoderBlock::make(in, attention_hidden_per_head, attention_num_heads, mlp_hidden, outf, dropout, dtype, device, linearized,
                                             decoder_decoder_causal_mask, encoder_decoder_causal_mask));
  LAMP_API_END
}
int lamp_module_transformer(lamp_module** out, int64_t num_blocks, int64_t in, int64_t attention_hidden_per_head, int64_t attention_num_heads,
                            int64_t mlp_hidden, double dropout, int dtype, int device, int linearized, int encoder_causal_mask,
                            int decoder_decoder_causal_mask, int encoder_decoder_causal_mask) {
  LAMP_API_BEGIN
  *out = wrapm(Transformer::make(num_blocks, in, attention_hidden_per_head, attention_num_heads, mlp_hidden, dropout, dtype, device, linearized,
                                 encoder_causal_mask, decoder_decoder_causal_mask, encoder_decoder_causal_mask));
  LAMP_API_END
}
int lamp_positional_embedding_vaswani(lamp_tensor** out, int64_t sequence_length, int64_t dimension, int dtype, int device) {
  LAMP_API_BEGIN *out = give(positional_embedding_vaswani(sequence_length, dimension, dtype, device)); LAMP_API_END
}
int lamp_module_transformer_embedding(lamp_module** out, lamp_module* embedding, int add_positional_embedding, const lamp_tensor* positional_embedding) {
  LAMP_API_BEGIN
  auto e = std::dynamic_pointer_cast<Embedding>(embedding->m);
  LAMP_CHECK(e, "TransformerEmbedding: `embedding` must be an Embedding module");
  auto m = std::make_shared<TransformerEmbedding>();
  m->embedding = e;
  m->addPositionalEmbedding = add_positional_embedding;
  m->positionalEmbedding = make_const(borrow(positional_embedding));
  *out = wrapm(m);
  LAMP_API_END
}
int lamp_module_language_model(lamp_module** out, int64_t max_length, int64_t vocabulary_size, int64_t num_blocks, int64_t embedding_dim,
                               int64_t attention_hidden_per_head, int64_t attention_num_heads, int64_t encoder_mlp_hidden, double dropout, int dtype,
                               int device, int linearized) {
  LAMP_API_BEGIN
  *out = wrapm(LanguageModelModule::make(max_length, vocabulary_size, num_blocks, embedding_dim, attention_hidden_per_head, attention_num_heads,
                                         encoder_mlp_hidden, dropout, dtype, device, linearized));
  LAMP_API_END
}
int lamp_module_language_model_loss(lamp_module** out, int64_t max_length, int64_t vocabulary_size, int64_t num_blocks, int64_t embedding_dim,
                                    int64_t attention_hidden_per_head, int64_t attention_num_heads, int64_t encoder_mlp_hidden, double dropout,
                                    int64_t pad_token, int dtype, int device, int linearized) {
  LAMP_API_BEGIN
  *out = wrapm(LanguageModelLoss::make(max_length, vocabulary_size, num_blocks, embedding_dim, attention_hidden_per_head, attention_num_heads,
                                       encoder_mlp_hidden, dropout, pad_token, dtype, device, linearized));
  LAMP_API_END
}
int lamp_language_model_forward(lamp_module* m, lamp_var* tokens, const lamp_tensor* max_length, const lamp_tensor* positions, lamp_var** encoded,
                                lamp_var** logits) {
  LAMP_API_BEGIN
  auto lm = std::dynamic_pointer_cast<LanguageModelModule>(m->m);
  if (!lm) if (auto l = std::dynamic_pointer_cast<LanguageModelLoss>(m->m)) lm = l->languageModel;
  LAMP_CHECK(lm, "lamp_language_model_forward: not a LanguageModelModule / LanguageModelLoss");
  auto r = lm->run(tokens->v, max_length ? borrow(max_length) : Ten(), positions ? borrow(positions) : Ten());
  if (encoded) *encoded = wrap(r.first);
  if (logits) *logits = wrap(r.second);
  LAMP_API_END
}
int lamp_sequence_mask(lamp_var** out, const lamp_tensor* max_length, lamp_var* maskable, double fill) {
  LAMP_API_BEGIN *out = wrap(MultiheadAttention::sequenceMask(borrow(max_length), maskable->v, fill)); LAMP_API_END
}
int lamp_masked_softmax(lamp_var** out, lamp_var* input, const lamp_tensor* max_length) {
  LAMP_API_BEGIN *out = wrap(MultiheadAttention::maskedSoftmax(input->v, borrow(max_length))); LAMP_API_END
}
int lamp_module_forward_multi(lamp_module* m, lamp_var* const* vars, int nvars, const lamp_tensor* const* tensors, int ntensors, lamp_var** out) {
  LAMP_API_BEGIN
  std::vector<Var> xs;
  for (int i = 0; i < nvars; i++) { LAMP_CHECK(vars[i], "forward_multi: NULL variable"); xs.push_back(vars[i]->v); }
  std::vector<Ten> aux;
  for (int i = 0; i < ntensors; i++) aux.push_back(tensors[i] ? borrow(tensors[i]) : Ten());
  *out = wrap(m->m->forward_multi(xs, aux));
  LAMP_API_END
}
int lamp_module_forward(lamp_module* m, lamp_var* x, lamp_var** out) { LAMP_API_BEGIN *out = wrap(m->m->forward(x->v)); LAMP_API_END }
int lamp_module_num_state(lamp_module* m, int64_t* out) { LAMP_API_BEGIN *out = (int64_t)m->m->state().size(); LAMP_API_END }
int lamp_module_state(lamp_module* m, int64_t index, lamp_var** out) {
  LAMP_API_BEGIN
  auto s = m->m->state();
  LAMP_CHECK(index >= 0 && index < (int64_t)s.size(), "state index out of range");
  *out = wrap(s[index]);
  LAMP_API_END
}
int lamp_module_set_training(lamp_module* m, int training) { LAMP_API_BEGIN m->m->set_training(training); LAMP_API_END }
int lamp_module_zero_grad(lamp_module* m) { LAMP_API_BEGIN m->m->zeroGrad(); LAMP_API_END }
int lamp_module_release(lamp_module* m) { LAMP_API_BEGIN delete m; LAMP_API_END }

// ---- optimisers ---------------------------------------------------------------------------------------
int lamp_optimizer_adamw(lamp_optimizer** out, lamp_tensor* const* params, int n, double wd, double lr, double b1, double b2, double eps, double clip,
                         int debias, int mixed) {
  LAMP_API_BEGIN
  std::vector<Ten> ps;
  for (int i = 0; i < n; i++) ps.push_back(borrow(params[i]));
  auto* o = new lamp_optimizer();
  o->o = std::make_shared<AdamW>(ps, wd, lr, b1, b2, eps, clip >= 0, clip, debias, mixed);
  *out = o;
  LAMP_API_END
}
int lamp_optimizer_adamw_tagged(lamp_optimizer** out, lamp_tensor* const* params, int n, const double* wd, const double* lr, const double* b1,
                                const double* b2, double eps, double clip, int debias, int mixed) {
  LAMP_API_BEGIN
  LAMP_CHECK(wd && lr && b1 && b2, "lamp_optimizer_adamw_tagged: one value per parameter for every hyperparameter");
  std::vector<Ten> ps;
  for (int i = 0; i < n; i++) ps.push_back(borrow(params[i]));
  auto a = std::make_shared<AdamW>(ps, 0.0, 0.0, 0.0, 0.0, eps, clip >= 0, clip, debias, mixed);
  a->weightDecayPer.assign(wd, wd + n); a->learningRatePer.assign(lr, lr + n); a->beta1Per.assign(b1, b1 + n); a->beta2Per.assign(b2, b2 + n);
  auto* o = new lamp_optimizer();
  o->o = a;
  *out = o;
  LAMP_API_END
}
int lamp_optimizer_sgdw(lamp_optimizer** out, lamp_tensor* const* params, int n, double lr, double wd, double momentum, double clip) {
  LAMP_API_BEGIN
  std::vector<Ten> ps;
  for (int i = 0; i < n; i++) ps.push_back(borrow(params[i]));
  auto* o = new lamp_optimizer();
  o->o = std::make_shared<SGDW>(ps, lr, wd, momentum >= 0, momentum, clip >= 0, clip);
  *out = o;
  LAMP_API_END
}
int lamp_optimizer_step(lamp_optimizer* o, lamp_tensor* const* gradients, int n, double schedule_factor) {
  LAMP_API_BEGIN
  std::vector<Ten> gs;
  for (int i = 0; i < n; i++) gs.push_back(gradients[i] ? borrow(gradients[i]) : Ten());
  o->o->step(gs, schedule_factor);
  LAMP_API_END
}
int lamp_optimizer_num_state(lamp_optimizer* o, int64_t* out) { LAMP_API_BEGIN *out = (int64_t)o->o->state().size(); LAMP_API_END }
int lamp_optimizer_state(lamp_optimizer* o, int64_t index, lamp_tensor** out) {
  LAMP_API_BEGIN
  auto s = o->o->state();
  LAMP_CHECK(index >= 0 && index < (int64_t)s.size(), "optimizer state index out of range");
  *out = give(s[index]);
  LAMP_API_END
}
int lamp_optimizer_load(lamp_optimizer* o, lamp_tensor* const* tensors, int n) {
  LAMP_API_BEGIN
  std::vector<Ten> ts;
  for (int i = 0; i < n; i++) { LAMP_CHECK(tensors[i], "Optimizer.load: NULL tensor"); ts.push_back(borrow(tensors[i])); }
  o->o->load(ts);
  LAMP_API_END
}
int lamp_optimizer_release(lamp_optimizer* o) { LAMP_API_BEGIN delete o; LAMP_API_END }
int lamp_gradient_clipping_in_place(lamp_tensor* const* gradients, int n, double theta) {
  LAMP_API_BEGIN
  std::vector<lamp_tensor*> g;
  for (int i = 0; i < n; i++) if (gradients[i]) g.push_back(gradients[i]);
  if (!g.empty()) HCALL(lamp_gradient_clipping_(g.data(), (int)g.size(), theta));
  LAMP_API_END
}

// ---- supervised model ------------------------------------------------------------------------------------
int lamp_model_create(lamp_model** out, lamp_module* module, int loss_kind, const lamp_tensor* class_weights, int64_t reduction, int64_t ignore) {
  LAMP_API_BEGIN
  auto* m = new lamp_model();
  m->model.module = module->m;
  m->model.loss_kind = loss_kind;
  if (class_weights) m->model.classWeights = borrow(class_weights);
  if (loss_kind == 0) LAMP_CHECK(class_weights, "LossFunctions.NLL always passes class weights (LossFunctions.scala:39-55)");
  m->model.reduction = reduction;
  m->model.ignore = ignore;
  *out = m;
  LAMP_API_END
}
int lamp_model_gradients(lamp_model* m, const lamp_tensor* samples, const lamp_tensor* target, lamp_tensor* acc, int zero_grad, int64_t* num_examples) {
  LAMP_API_BEGIN
  *num_examples = m->model.addTotalLossAndReturnGradientsAndNumExamples(borrow(samples), borrow(target), acc ? borrow(acc) : Ten(), zero_grad, nullptr);
  LAMP_API_END
}
int lamp_model_forward_loss(lamp_model* m, const lamp_tensor* samples, const lamp_tensor* target, lamp_tensor* acc, int64_t* num_examples) {
  LAMP_API_BEGIN
  *num_examples = m->model.addTotalLossAndReturnNumExamples(borrow(samples), borrow(target), acc ? borrow(acc) : Ten());
  LAMP_API_END
}
int lamp_model_train_step(lamp_model* m, lamp_optimizer* o, lamp_comm* comm, const lamp_tensor* samples, const lamp_tensor* target, lamp_tensor* acc,
                          int64_t* num_examples) {
  LAMP_API_BEGIN
  m->dp.comm = comm;
  *num_examples = m->dp.step(m->model, *o->o, borrow(samples), borrow(target), acc ? borrow(acc) : Ten());
  LAMP_API_END
}
int lamp_model_train_step_scheduled(lamp_model* m, lamp_optimizer* o, lamp_comm* comm, const lamp_tensor* samples, const lamp_tensor* target,
                                    lamp_tensor* acc, double schedule_factor, int64_t* num_examples) {
  LAMP_API_BEGIN
  m->dp.comm = comm;
  *num_examples = m->dp.step(m->model, *o->o, borrow(samples), borrow(target), acc ? borrow(acc) : Ten(), schedule_factor);
  LAMP_API_END
}
int lamp_model_exchange_and_step(lamp_model* m, lamp_optimizer* o, lamp_comm* comm, lamp_tensor* const* grads, int ngrads, int64_t num_examples,
                                 double schedule_factor) {
  LAMP_API_BEGIN
  LAMP_CHECK(comm, "NULL communicator");
  m->dp.comm = comm;
  std::vector<Ten> gs;
  for (int i = 0; i < ngrads; i++) { LAMP_CHECK(grads[i], "NULL gradient"); gs.push_back(borrow(grads[i])); }
  m->dp.exchange_and_step(m->model, *o->o, gs, num_examples, schedule_factor);
  LAMP_API_END
}
int lamp_model_sync_state(lamp_model* m, lamp_optimizer* o, lamp_comm* comm, int root) {
  LAMP_API_BEGIN
  LAMP_CHECK(comm, "NULL communicator");
  m->dp.comm = comm;
  m->dp.sync_state(m->model, *o->o, root);
  LAMP_API_END
}
int lamp_data_parallel_step(lamp_model* main_model, lamp_optimizer* o, lamp_model* const* replicas, int nreplicas, const lamp_tensor* const* samples,
                            const lamp_tensor* const* targets, lamp_tensor* const* accs, int zero_grad, int step, double schedule_factor,
                            int64_t* num_examples) {
  LAMP_API_BEGIN
  std::vector<SupervisedModel*> reps;
  for (int i = 0; i < nreplicas; i++) { LAMP_CHECK(replicas[i], "NULL replica"); reps.push_back(&replicas[i]->model); }
  std::vector<Ten> xs, ts, as;
  for (int i = 0; i <= nreplicas; i++) {
    LAMP_CHECK(samples[i] && targets[i], "NULL batch");
    xs.push_back(borrow(samples[i])); ts.push_back(borrow(targets[i])); as.push_back(accs && accs[i] ? borrow(accs[i]) : Ten());
  }
  *num_examples = data_parallel_synchronous_step(main_model->model, *o->o, reps, xs, ts, as, zero_grad, step, schedule_factor);
  LAMP_API_END
}
int lamp_attention_fused_call_as_written(int on, int* previous) {
  LAMP_API_BEGIN
  bool& f = MultiheadAttention::fused_call_as_written();
  if (previous) *previous = f ? 1 : 0;
  f = on != 0;
  LAMP_API_END
}
int lamp_model_release(lamp_model* m) { LAMP_API_BEGIN delete m; LAMP_API_END }

}  // extern "C"
