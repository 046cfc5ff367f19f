// lamp.nn over the C ABI - see nn.h for the reference map.
#include "nn.h"
#include <thread>
#include <unordered_map>

namespace lamp {
namespace host {

// ---- Linear (nn/Linear.scala:19-33, init :45-66) -------------------------------------------------
Mod Linear::make(int64_t in, int64_t out, int dtype, int device, bool bias) {
  Var w = make_param(ops::normal(0.0, std::sqrt(2.0 / (double)(in + out)), {in, out}, dtype, device));
  Var b = bias ? make_param(ops::zeros({1, out}, dtype, device)) : nullptr;
  return std::make_shared<Linear>(w, b);
}
Var Linear::forward(const Var& x) {
  Var v;
  if (x->value.ndim() == 2) v = F::mm(x, weights);
  else {  // mm1: view(-1, last).mm(w).view(shape.dropRight(1) :+ -1)
    auto shape = x->shape();
    Var x2 = F::view(x, {-1, shape.back()});
    Var y = F::mm(x2, weights);
    shape.back() = -1;
    v = F::view(y, shape);
  }
  return bias ? F::add(bias, v) : v;   // bias.map(_ + v)
}

// ---- Conv2D (nn/Conv2D.scala:22-34, init :51-82) --------------------------------------------------
Mod Conv2D::make(int64_t inC, int64_t outC, int64_t k, int dtype, int device, bool bias, int64_t stride, int64_t padding, int64_t dilation,
                 int64_t groups) {
  auto m = std::make_shared<Conv2D>();
  m->weights = make_param(ops::normal(0.0, std::sqrt(2.0 / (double)(outC + inC)), {outC, inC / groups, k, k}, dtype, device));
  Ten b = ops::zeros({outC}, dtype, device);
  m->bias = bias ? make_param(b) : make_const(b);
  m->stride = stride; m->padding = padding; m->dilation = dilation; m->groups = groups;
  return m;
}
Var Conv2D::forward(const Var& x) {
  return F::convolution(x, weights, bias, {stride, stride}, {padding, padding}, {dilation, dilation}, false, {0, 0}, groups);
}

// ---- BatchNorm / BatchNorm2D (init: weight N(0, 0.01), bias 0, running mean 0, running var 0) ----
Mod BatchNorm::make(int64_t features, int dtype, int device, bool two_d) {
  auto m = std::make_shared<BatchNorm>();
  m->weight = make_param(ops::normal(0.0, 0.01, {features}, dtype, device));
  m->bias = make_param(ops::zeros({features}, dtype, device));
  m->runningMean = make_const(ops::zeros({features}, dtype, device));
  m->runningVar = make_const(ops::zeros({features}, dtype, device));
  m->two_d = two_d;
  return m;
}
Var BatchNorm::forward(const Var& x) {
  return two_d ? F::batch_norm_2d(x, weight, bias, runningMean->value, runningVar->value, training, momentum, eps)
               : F::batch_norm(x, weight, bias, runningMean->value, runningVar->value, training, momentum, eps);
}
bool BatchNorm::can_fuse_relu(const Var& x) const { return two_d && F::batch_norm_relu_2d_supported(x); }
Var BatchNorm::forward_relu(const Var& x) {
  return F::batch_norm_relu_2d(x, weight, bias, runningMean->value, runningVar->value, training, momentum, eps);
}
bool BatchNorm::can_fuse_add_relu(const Var& x) const { return training && can_fuse_relu(x); }
Var BatchNorm::forward_add_relu(const Var& x, const Var& addend) {
  return F::batch_norm_add_relu_2d(x, addend, weight, bias, runningMean->value, runningVar->value, training, momentum, eps);
}
Var Residual::forward_relu(const Var& x, int64_t pool_tail) {
  // (the tail behind the block's output y, where the one-node form does not apply)
  auto tail = [&](const Var& y) -> Var {
    if (!pool_tail) return y;
    if (y->value.ndim() == 4 && y->value.size(2) == pool_tail && y->value.size(3) == pool_tail && y->value.h()->is_device()) return F::global_avg_pool_log_softmax(y);
    return F::log_softmax(F::flatten(F::avg_pool2d(y, pool_tail, 1, 0), y->value.ndim() - 3, -1), 1);
  };
  auto* seq = dynamic_cast<Sequential*>(right.get());
  BatchNorm* bn = (seq && !seq->mods.empty()) ? dynamic_cast<BatchNorm*>(seq->mods.back().get()) : nullptr;
  if (bn) {
    // right branch up to (not including) its last batch norm, with the usual BatchNorm -> relu rewrite inside
    Sequential head(std::vector<Mod>(seq->mods.begin(), seq->mods.end() - 1));
    auto* lseq = left ? dynamic_cast<Sequential*>(left.get()) : nullptr;
    BatchNorm* lbn = (lseq && !lseq->mods.empty()) ? dynamic_cast<BatchNorm*>(lseq->mods.back().get()) : nullptr;
    // Both branches START with a Conv2D on x (every block of Cnn.resnet: cnn.scala:38-45 and 64-72): the two convolutions are one call, which
    // is one launch where a kernel keeps the staged input for both products (LAMP_CONV_SIBLING=0: two calls).  Same nodes, same values.
    static const bool pair_on = [] { const char* e = getenv("LAMP_CONV_SIBLING"); return !(e && e[0] == '0'); }();
    auto* c3 = head.mods.empty() ? nullptr : dynamic_cast<Conv2D*>(head.mods[0].get());
    auto* c1 = (lbn && lseq->mods.size() == 2) ? dynamic_cast<Conv2D*>(lseq->mods[0].get()) : nullptr;
    Var v, lconv;                                  // lconv: the left branch's convolution output when it came with the right one
    if (pair_on && c3 && c1 && c3->groups == c1->groups && x->value.h()->is_device()) {
      auto pr = F::convolution_pair(x, c3->weights, c3->bias, {c3->stride, c3->stride}, {c3->padding, c3->padding}, {c3->dilation, c3->dilation},
                                    c1->weights, c1->bias, {c1->stride, c1->stride}, {c1->padding, c1->padding}, {c1->dilation, c1->dilation}, c3->groups);
      Sequential rest(std::vector<Mod>(head.mods.begin() + 1, head.mods.end()));
      v = rest.forward(pr.first);
      lconv = pr.second;
    } else {
      v = head.forward(x);
    }
    auto left_branch = [&]() -> Var { return lconv ? lbn->forward(lconv) : (left ? left->forward(x) : x); };
    // both branches end in a batch norm (every block of Cnn.resnet: the left branch is Conv2D 1x1 -> BatchNorm2D): one op for
    // relu(bn(right) + bn(left)) - the left batch norm's output is never written (LAMP_FUSE_BN_PAIR=0: the chain)
    static const bool fuse_pair = [] { const char* e = getenv("LAMP_FUSE_BN_PAIR"); return !(e && e[0] == '0'); }();
    if (fuse_pair && lbn && bn->can_fuse_add_relu(v)) {
      Var lv = lconv;
      if (!lv) {
        Sequential lhead(std::vector<Mod>(lseq->mods.begin(), lseq->mods.end() - 1));
        lv = lhead.forward(x);
      }
      if (lbn->can_fuse_add_relu(lv) && lv->shape() == v->shape() && lv->value.dtype() == v->value.dtype()) {
        // ... and with the network's tail behind it, the block's output is never written (LAMP_FUSE_BLOCK_TAIL=0: the two nodes)
        if (pool_tail && v->value.ndim() == 4 && v->value.size(2) == pool_tail && v->value.size(3) == pool_tail)
          return F::batch_norm2_add_relu_pool_log_softmax_2d(v, bn->weight, bn->bias, bn->runningMean->value, bn->runningVar->value, bn->momentum, bn->eps,
                                                             lv, lbn->weight, lbn->bias, lbn->runningMean->value, lbn->runningVar->value, lbn->momentum, lbn->eps);
        return tail(F::batch_norm2_add_relu_2d(v, bn->weight, bn->bias, bn->runningMean->value, bn->runningVar->value, bn->momentum, bn->eps,
                                               lv, lbn->weight, lbn->bias, lbn->runningMean->value, lbn->runningVar->value, lbn->momentum, lbn->eps));
      }
      Var l = lbn->forward(lv);
      if (l->shape() == v->shape()) return tail(bn->forward_add_relu(v, l));
      return tail(F::relu(F::add(bn->forward(v), l)));
    }
    if (bn->can_fuse_add_relu(v)) {
      Var l = left_branch();
      if (l->shape() == v->shape()) return tail(bn->forward_add_relu(v, l));
      return tail(F::relu(F::add(bn->forward(v), l)));
    }
    Var l = left_branch();
    return tail(F::relu(F::add(bn->forward(v), l)));
  }
  return tail(F::relu(forward(x)));
}
// [..., Residual, relu, Dropout(p <= 0)*] or [..., Sequential that can]: everything in front of the last block as usual, the block with the tail
Var Sequential::forward_pool_tail(const Var& x, int64_t pool, bool probe) {
  size_t n = mods.size();
  while (n > 0) {
    auto* d = dynamic_cast<Dropout*>(mods[n - 1].get());
    if (d && d->prob <= 0) n--; else break;
  }
  if (n == 0) return nullptr;
  if (auto* inner = dynamic_cast<Sequential*>(mods[n - 1].get())) {
    if (!inner->forward_pool_tail(x, pool, true)) return nullptr;
    if (probe) return x;
    const Var v = n > 1 ? Sequential(std::vector<Mod>(mods.begin(), mods.begin() + (n - 1))).forward(x) : x;
    return inner->forward_pool_tail(v, pool, false);
  }
  auto* fn = dynamic_cast<Fun*>(mods[n - 1].get());
  auto* res = n >= 2 ? dynamic_cast<Residual*>(mods[n - 2].get()) : nullptr;
  if (!(fn && fn->tag == "relu" && res)) return nullptr;
  if (probe) return x;
  const Var v = n > 2 ? Sequential(std::vector<Mod>(mods.begin(), mods.begin() + (n - 2))).forward(x) : x;
  return res->forward_relu(v, pool);
}
// The reference's Sequential is a plain fold (nn/Sequential.scala).  The one rewrite done here: BatchNorm2D directly followed by
// Fun(relu) runs as the fused op (same values, three elementwise passes fewer).
Var Sequential::forward(const Var& x) {
  Var v = x;
  for (size_t i = 0; i < mods.size(); i++) {
    if (i + 3 < mods.size() && v->value.h()->is_device()) {
      // module -> Fun(avgpool2d, stride 1) -> Fun(flatten the last three dims) -> Fun(logsoftmax over dim 1) where the module ends in a residual
      // block under a relu (Cnn.resnet: the Sequential of the four blocks, cnn.scala:118-136): the last block and the tail as one node - the
      // block's output, which only the pool reads, is never written (LAMP_FUSE_BLOCK_TAIL=0: off)
      static const bool fuse_tail0 = [] { const char* e = getenv("LAMP_FUSE_POOL_LOGSOFTMAX"); return !(e && e[0] == '0'); }();
      static const bool fuse_block_tail = [] { const char* e = getenv("LAMP_FUSE_BLOCK_TAIL"); return !(e && e[0] == '0'); }();
      auto* pool = dynamic_cast<Fun*>(mods[i + 1].get());
      auto* flat = dynamic_cast<Fun*>(mods[i + 2].get());
      auto* lsm = dynamic_cast<Fun*>(mods[i + 3].get());
      if (fuse_tail0 && fuse_block_tail && pool && flat && lsm && pool->tag == "avgpool2d" && pool->b == 1 && pool->a >= 1 && flat->tag == "flatten_last" &&
          flat->a == 3 && lsm->tag == "logsoftmax" && lsm->a == 1 && mods[i]->forward_pool_tail(v, (int64_t)pool->a, true)) {
        v = mods[i]->forward_pool_tail(v, (int64_t)pool->a, false);
        i += 3;
        continue;
      }
    }
    if (i + 1 < mods.size()) {
      auto* bn = dynamic_cast<BatchNorm*>(mods[i].get());
      auto* fn = dynamic_cast<Fun*>(mods[i + 1].get());
      if (bn && fn && fn->tag == "relu" && bn->can_fuse_relu(v)) {
        // BatchNorm2D -> relu -> [Dropout(p <= 0)] -> Conv2D (the middle of every residual block, cnn.scala:38-60): the convolution applies
        // the batch norm + relu while staging its input, bitwise the chain's values (LAMP_FUSE_BN_CONV=0: the chain)
        static const bool fuse_conv = [] { const char* e = getenv("LAMP_FUSE_BN_CONV"); return !(e && e[0] == '0'); }();
        size_t j = i + 2;
        auto* drop = j < mods.size() ? dynamic_cast<Dropout*>(mods[j].get()) : nullptr;
        if (drop && drop->prob <= 0) j++;
        auto* conv = (!drop || drop->prob <= 0) && j < mods.size() ? dynamic_cast<Conv2D*>(mods[j].get()) : nullptr;
        if (fuse_conv && conv && bn->training && F::conv_of_batch_norm_relu_2d_pays(v, conv->weights, conv->stride, conv->padding, conv->dilation, conv->groups)) {
          v = F::conv_of_batch_norm_relu_2d(v, bn->weight, bn->bias, bn->runningMean->value, bn->runningVar->value, bn->momentum, bn->eps, conv->weights,
                                            conv->bias, {conv->stride, conv->stride}, {conv->padding, conv->padding}, {conv->dilation, conv->dilation},
                                            conv->groups);
          i = j;
          continue;
        }
        v = bn->forward_relu(v);
        i++;
        continue;
      }
      auto* res = dynamic_cast<Residual*>(mods[i].get());
      if (res && fn && fn->tag == "relu") {
        v = res->forward_relu(v);
        i++;
        continue;
      }
      // Fun(avgpool2d over the whole map) -> Fun(flatten the last three dims) -> Fun(logsoftmax over dim 1): one node, values of the chain
      auto* pool = dynamic_cast<Fun*>(mods[i].get());
      auto* lsm = i + 2 < mods.size() ? dynamic_cast<Fun*>(mods[i + 2].get()) : nullptr;
      static const bool fuse_tail = [] { const char* e = getenv("LAMP_FUSE_POOL_LOGSOFTMAX"); return !(e && e[0] == '0'); }();
      if (fuse_tail && pool && fn && lsm && pool->tag == "avgpool2d" && fn->tag == "flatten_last" && fn->a == 3 && lsm->tag == "logsoftmax" && lsm->a == 1 &&
          v->value.ndim() == 4 && v->value.size(2) == (int64_t)pool->a && v->value.size(3) == (int64_t)pool->a && v->value.h()->is_device()) {
        v = F::global_avg_pool_log_softmax(v);
        i += 2;
        continue;
      }
    }
    v = mods[i]->forward(v);
  }
  return v;
}
Mod LayerNorm::make(const std::vector<int64_t>& shape, int dtype, int device, bool scale, bool bias) {
  auto m = std::make_shared<LayerNorm>();
  m->normalizedShape = shape;
  if (scale) m->scale = make_param(ops::ones(shape, dtype, device));
  if (bias) m->bias = make_param(ops::zeros(shape, dtype, device));
  return m;
}

Mod make_fun(const std::string& name, double a, double b) {
  if (name == "relu") return std::make_shared<Fun>([](const Var& x) { return F::relu(x); }, "relu");
  if (name == "gelu") return std::make_shared<Fun>([](const Var& x) { return F::gelu(x); });
  if (name == "sigmoid") return std::make_shared<Fun>([](const Var& x) { return F::sigmoid(x); });
  if (name == "tanh") return std::make_shared<Fun>([](const Var& x) { return F::tanh(x); });
  if (name == "hardswish") return std::make_shared<Fun>([](const Var& x) { return F::hardswish(x); });
  if (name == "swish1") return std::make_shared<Fun>([](const Var& x) { return F::mult(x, F::sigmoid(x)); });
  if (name == "logsoftmax") return std::make_shared<Fun>([a](const Var& x) { return F::log_softmax(x, (int64_t)a); }, "logsoftmax", a);
  if (name == "avgpool2d") return std::make_shared<Fun>([a, b](const Var& x) { return F::avg_pool2d(x, (int64_t)a, (int64_t)b, 0); }, "avgpool2d", a, b);
  if (name == "maxpool2d") return std::make_shared<Fun>([a, b](const Var& x) { return F::max_pool2d(x, (int64_t)a, (int64_t)b, 0, 1); });
  if (name == "flatten_last") return std::make_shared<Fun>([a](const Var& x) { return F::flatten(x, x->value.ndim() - (int64_t)a, -1); }, "flatten_last", a);
  LAMP_CHECK(false, "unknown Fun module '" << name << "'");
  return nullptr;
}

// ---- Residual.make / Cnn.resnet (cnn.scala:33-137) -------------------------------------------------
Mod residual_make(int64_t inC, int64_t outC, int dtype, int device, double dropout, int64_t stride) {
  std::vector<Mod> right = {
      Conv2D::make(inC, outC, 3, dtype, device, false, stride, 1, 1, 1), BatchNorm::make(outC, dtype, device, true), make_fun("relu"),
      std::make_shared<Dropout>(dropout, true), Conv2D::make(outC, outC, 3, dtype, device, false, 1, 1, 1, 1),
      BatchNorm::make(outC, dtype, device, true)};
  Mod left;
  if (!(inC == outC && stride == 1)) {
    left = std::make_shared<Sequential>(std::vector<Mod>{Conv2D::make(inC, outC, 1, dtype, device, false, stride, 0, 1, 1),
                                                         BatchNorm::make(outC, dtype, device, true)});
  }
  return std::make_shared<Sequential>(std::vector<Mod>{std::make_shared<Residual>(std::make_shared<Sequential>(right), left),
                                                       make_fun("relu"), std::make_shared<Dropout>(dropout, true)});
}
Mod cnn_resnet(int64_t numClasses, double dropout, int dtype, int device) {
  return std::make_shared<Sequential>(std::vector<Mod>{
      Conv2D::make(3, 6, 5, dtype, device, false, 1, 2, 1, 1),
      std::make_shared<Sequential>(std::vector<Mod>{residual_make(6, 6, dtype, device, dropout, 2), residual_make(6, 16, dtype, device, dropout, 2),
                                                    residual_make(16, 128, dtype, device, dropout, 1),
                                                    residual_make(128, numClasses, dtype, device, dropout, 1)}),
      make_fun("avgpool2d", 8, 1), make_fun("flatten_last", 3), make_fun("logsoftmax", 1)});
}

// ---- MLP.apply (nn/MLP.scala:40-167) -------------------------------------------------------------
Mod mlp(int64_t in, int64_t out, const std::vector<int64_t>& hidden, int dtype, int device, double dropout, bool lastNonLinearity,
        const std::string& activation, int norm, bool bias) {
  // hasBias: LayerNorm(bias=true) and BatchNorm disable the Linear bias (MLP.scala:85-88)
  const bool hasBias = (norm == 1 || norm == 2) ? false : bias;
  auto make_norm = [&](int64_t dim) -> std::vector<Mod> {
    if (norm == 1) return {BatchNorm::make(dim, dtype, device, false)};
    if (norm == 2) return {LayerNorm::make({dim}, dtype, device, true, true)};
    return {};
  };
  auto block = [&](int64_t i, int64_t o, bool nonlin) {
    std::vector<Mod> m = {Linear::make(i, o, dtype, device, hasBias)};
    for (auto& n : make_norm(o)) m.push_back(n);
    if (nonlin) { m.push_back(make_fun(activation)); m.push_back(std::make_shared<Dropout>(dropout, true)); }
    return std::make_shared<Sequential>(m);
  };
  std::vector<Mod> layers;
  int64_t prev = in;
  for (int64_t h : hidden) { layers.push_back(block(prev, h, true)); prev = h; }
  layers.push_back(block(prev, out, lastNonLinearity));
  return std::make_shared<Sequential>(layers);
}

// ---- AdamW (nn/AdamW.scala) -----------------------------------------------------------------------
static bool is_low_precision(int dt) { return dt == kF16 || dt == kBF16; }
AdamW::AdamW(const std::vector<Ten>& params, double wd, double lr, double b1, double b2, double eps_, bool has_clip_, double clip_,
             bool debias_, bool mixed)
    : parameters(params), weightDecay(wd), learningRate(lr), beta1(b1), beta2(b2), eps(eps_), has_clip(has_clip_), clip(clip_),
      debias(debias_), mixedPrecision(mixed) {
  for (auto& p : parameters) {
    const bool up = mixedPrecision && is_low_precision(p.dtype());
    workingCopy.push_back(up ? ops::cast(p, kF32) : Ten());
    Ten z = ops::zeros_like(p);
    mt.push_back(up ? ops::cast(z, kF32) : z);
    vt.push_back(up ? ops::cast(z, kF32) : ops::zeros_like(p));
  }
  stepCountSTen = ops::scalar(0.0, kF64, parameters.empty() ? 0 : parameters[0].device());
}
std::vector<Ten> AdamW::state() {
  ops::fill_(stepCountSTen, (double)stepCount);
  std::vector<Ten> s = {stepCountSTen};
  for (auto& t : mt) s.push_back(t);
  for (auto& t : vt) s.push_back(t);
  for (auto& t : workingCopy) if (t.defined()) s.push_back(t);
  return s;
}
void AdamW::load(const std::vector<Ten>& tensors) {
  Optimizer::load(tensors);
  if (!tensors.empty()) {   // stepCount = stepCountSTen.toDoubleArray(0).toLong
    double v = 0;
    HCALL(lamp_item(stepCountSTen.h(), &v));
    stepCount = (int64_t)v;
  }
}
void AdamW::counters_from_state() {
  double v = 0;
  HCALL(lamp_item(stepCountSTen.h(), &v));
  stepCount = (int64_t)v;
}
void AdamW::step(const std::vector<Ten>& gradients, double scheduleFactor) {
  LAMP_CHECK(gradients.size() == parameters.size(), "AdamW.step: got " << gradients.size() << " gradients for " << parameters.size() << " parameters");
  std::vector<lamp_tensor*> p, g, m, v, w;
  std::vector<double> lr, wd, b1, b2;
  auto pick = [](const std::vector<double>& per, double all, size_t i) { return per.empty() ? all : per[i]; };
  for (size_t i = 0; i < parameters.size(); i++) {
    if (!gradients[i].defined()) continue;
    p.push_back(parameters[i].h()); g.push_back(gradients[i].h()); m.push_back(mt[i].h()); v.push_back(vt[i].h());
    w.push_back(workingCopy[i].h());
    lr.push_back(pick(learningRatePer, learningRate, i)); wd.push_back(pick(weightDecayPer, weightDecay, i));
    b1.push_back(pick(beta1Per, beta1, i)); b2.push_back(pick(beta2Per, beta2, i));
  }
  if (has_clip && !g.empty()) HCALL(lamp_gradient_clipping_(g.data(), (int)g.size(), clip));
  stepCount += 1;               // stepCountSTen (state()[0]) is brought up to date when the state is asked for: one launch less per step
  const int n = (int)p.size();
  HCALL(lamp_adamw_step_(p.data(), g.data(), m.data(), v.data(), w.data(), n, lr.data(), wd.data(), b1.data(), b2.data(), eps, scheduleFactor,
                         stepCount, debias));
}

SGDW::SGDW(const std::vector<Ten>& params, double lr, double wd, bool has_m, double mom, bool has_clip_, double clip_)
    : parameters(params), learningRate(lr), weightDecay(wd), momentum(mom), has_momentum(has_m), has_clip(has_clip_), clip(clip_) {
  if (has_momentum) for (auto& p : parameters) velocity.push_back(ops::zeros_like(p));
}
void SGDW::step(const std::vector<Ten>& gradients, double scheduleFactor) {
  LAMP_CHECK(gradients.size() == parameters.size(), "SGDW.step: gradient count mismatch");
  std::vector<lamp_tensor*> p, g, v;
  for (size_t i = 0; i < parameters.size(); i++) {
    if (!gradients[i].defined()) continue;
    p.push_back(parameters[i].h()); g.push_back(gradients[i].h());
    v.push_back(has_momentum ? velocity[i].h() : nullptr);
  }
  if (has_clip && !g.empty()) HCALL(lamp_gradient_clipping_(g.data(), (int)g.size(), clip));
  const int n = (int)p.size();
  std::vector<double> lr(n, learningRate), wd(n, weightDecay), mom(n, momentum);
  HCALL(lamp_sgdw_step_(p.data(), g.data(), v.data(), n, lr.data(), wd.data(), mom.data(), scheduleFactor));
}

// ---- SupervisedModel -------------------------------------------------------------------------------
// With LossFunctions.Identity the loss is computed inside the module (LanguageModelLoss, lm.scala:44-59): its input case class
// carries the target, which reaches the module as forward_multi's first extra tensor.  Single-input modules ignore it.
// acc += (loss.value * numInstances.toDouble) (SupervisedModel.scala:207): one launch.  The reference's accumulator is an f64 scalar
// (IOLoops.scala:715, distributed/package.scala:631) whatever the model's type; same-type accumulators keep the plain add.
static void accumulate_loss(const Ten& acc, const Ten& loss, int64_t n) {
  if (!acc.defined()) return;
  if (acc.dtype() == loss.dtype()) ops::add_(acc, ops::reshape(loss, acc.shape()), (double)n);
  else HCALL(lamp_add_scaled_mixed_(acc.h(), loss.h(), (double)n));
}
static Var run_module(SupervisedModel& m, const Ten& samples, const Ten& target) {
  if (m.loss_kind == 2) return m.module->forward_multi({make_const(samples)}, {target});
  return m.module->forward(make_const(samples));
}
std::pair<Var, int64_t> SupervisedModel::loss(const Var& output, const Ten& target) {
  if (loss_kind == 0) return {F::nll_loss(output, target, classWeights, reduction, ignore), output->value.size(0)};
  if (loss_kind == 1) return {F::mse_loss(output, target, 1), output->value.size(0)};
  return {output, target.size(0)};
}
// NLL on the device with an accumulator of the loss's dtype: `acc += n * loss` rides in the loss kernel (one launch less per step)
static bool loss_accumulates_in_kernel(const SupervisedModel& m, const Var& output, const Ten& acc) {
  static const bool on = [] { const char* e = getenv("LAMP_FUSE_LOSS_ACCUMULATE"); return !(e && e[0] == '0'); }();
  return on && m.loss_kind == 0 && m.reduction != 0 && acc.defined() && acc.h()->is_device() && output->value.h()->is_device() &&
         acc.dtype() == output->value.dtype() && acc.numel() == 1 && acc.device() == output->value.device() && output->value.ndim() == 2;
}
int64_t SupervisedModel::addTotalLossAndReturnGradientsAndNumExamples(const Ten& samples, const Ten& target, const Ten& acc, bool zeroGrad,
                                                                      std::vector<Ten>* gradients) {
  Var output = run_module(*this, samples, target);            // BatchStream emits const(features) (BatchStream.scala:562)
  if (loss_accumulates_in_kernel(*this, output, acc)) {
    const int64_t n = output->value.size(0);
    Var l = F::nll_loss_accumulate(output, target, classWeights, reduction, ignore, acc, (double)n);
    std::vector<Ten> g = module->gradients(l, zeroGrad);
    if (gradients) *gradients = g;
    return n;
  }
  auto ln = loss(output, target);
  std::vector<Ten> g = module->gradients(ln.first, zeroGrad);
  accumulate_loss(acc, ln.first->value, ln.second);
  if (gradients) *gradients = g;
  return ln.second;
}
int64_t SupervisedModel::addTotalLossAndReturnNumExamples(const Ten& samples, const Ten& target, const Ten& acc) {
  Var output = run_module(*this, samples, target);
  auto ln = loss(output, target);
  accumulate_loss(acc, ln.first->value, ln.second);
  return ln.second;
}

// ---- single-process data parallel (DataParallel.scala:195-311) ---------------------------------------------
int64_t data_parallel_synchronous_step(SupervisedModel& main, Optimizer& opt, const std::vector<SupervisedModel*>& replicas,
                                       const std::vector<Ten>& samples, const std::vector<Ten>& targets, const std::vector<Ten>& accs,
                                       bool zeroGrad, bool step, double scheduleFactor) {
  const size_t n = replicas.size() + 1;
  LAMP_CHECK(samples.size() == n && targets.size() == n && accs.size() == n, "assertion failed: batch.size == models.size + 1 (DataParallel.scala:207-208)");
  std::vector<SupervisedModel*> models = {&main};
  for (auto* r : replicas) models.push_back(r);
  std::vector<Var> mainState = main.module->state();
  LAMP_CHECK(!mainState.empty(), "data parallel step on a model without state");
  const int mainDevice = mainState[0]->value.device();
  int callerDevice = 0;
  HCALL(lamp_get_device(&callerDevice));
  HCALL(lamp_set_device(mainDevice));
  HCALL(lamp_device_synchronize());          // the previous optimiser step is complete before the replicas read the state

  std::vector<int64_t> examples(n, 0);
  std::vector<std::vector<Ten>> grads(n);
  std::vector<std::string> errors(n);
  auto work = [&](size_t i) {
    try {
      std::vector<Var> st = models[i]->module->state();
      LAMP_CHECK(st.size() == mainState.size(), "replica " << i << " has " << st.size() << " state tensors, the main model " << mainState.size());
      const int dev = st[0]->value.device();
      HCALL(lamp_set_device(dev));
      if (i > 0)                              // copyStateFromMain (:224-247)
        for (size_t k = 0; k < st.size(); k++) ops::copy_(st[k]->value, mainState[k]->value);
      examples[i] = models[i]->addTotalLossAndReturnGradientsAndNumExamples(samples[i], targets[i], accs[i], zeroGrad, &grads[i]);
      if (step)                               // gradTensor *= numExample (:273-281)
        for (auto& g : grads[i]) if (g.defined()) ops::mul_scalar_(g, (double)examples[i]);
      HCALL(lamp_device_synchronize());
    } catch (const std::exception& e) { errors[i] = e.what(); if (errors[i].empty()) errors[i] = "unknown error"; }
  };
  std::vector<std::thread> threads;
  for (size_t i = 1; i < n; i++) threads.emplace_back(work, i);
  work(0);
  for (auto& t : threads) t.join();
  HCALL(lamp_set_device(mainDevice));
  for (size_t i = 0; i < n; i++) LAMP_CHECK(errors[i].empty(), "data parallel model " << i << ": " << errors[i]);
  int64_t total = 0;
  for (auto e : examples) total += e;
  if (step) {                                 // averageGradientsIntoMain (:262-306) then stepOptimizer
    for (size_t i = 1; i < n; i++) {
      LAMP_CHECK(grads[i].size() == grads[0].size(), "assertion failed: grads.size == gradMain.size");
      for (size_t k = 0; k < grads[0].size(); k++) {
        LAMP_CHECK(grads[i][k].defined() == grads[0][k].defined(), "assertion failed: source.isEmpty == main.isEmpty");
        if (!grads[0][k].defined()) continue;
        lamp_tensor* onMain = nullptr;          // main.device.to(source)
        HCALL(lamp_to(&onMain, grads[i][k].h(), grads[i][k].dtype(), mainDevice, 1, 1));
        ops::add_(grads[0][k], Ten(onMain));
      }
    }
    for (auto& g : grads[0]) if (g.defined()) ops::mul_scalar_(g, 1.0 / (double)total);
    opt.step(grads[0], scheduleFactor);
  }
  HCALL(lamp_set_device(callerDevice));
  return total;
}

// ---- data parallel step ------------------------------------------------------------------------------
DataParallel::~DataParallel() { if (comm_stream) lamp_stream_release(comm_stream); }

namespace {
// the thread's current stream is switched to the exchange stream for a few calls: an error in between (RCCL failure, a gradient that
// is not contiguous) must not leave the thread on the high-priority stream, nor leak the handle of the compute stream
struct StreamScope {
  lamp_stream* prev = nullptr;
  explicit StreamScope(int device) { HCALL(lamp_stream_get_current(device, &prev)); }
  void enter(lamp_stream* s) { HCALL(lamp_stream_set_current(s)); }
  void leave() { HCALL(lamp_stream_set_current(prev)); }
  ~StreamScope() { if (prev) { lamp_stream_set_current(prev); lamp_stream_release(prev); } }
};
// while a collective runs on the exchange stream beside backward the device is not this thread's alone: kernels whose workgroups wait
// for each other (one-pass batch-norm backward) take their two-kernel form until the exchange has been joined (lamp_device_shared_hint)
struct SharedDeviceScope {
  int device, held = 0;
  explicit SharedDeviceScope(int d) : device(d) {}
  void acquire() { if (!held) { HCALL(lamp_device_shared_hint(device, +1)); held = 1; } }
  void release() { if (held) { lamp_device_shared_hint(device, -1); held = 0; } }
  ~SharedDeviceScope() { release(); }
};
}  // namespace

void DataParallel::sync_state(SupervisedModel& model, Optimizer& opt, int root) {
  LAMP_CHECK(comm, "sync_state needs a communicator");
  std::vector<lamp_tensor*> ts;
  std::vector<Ten> keep;
  for (auto& v : model.module->state()) { keep.push_back(v->value); }
  for (auto& t : opt.state()) if (t.defined()) keep.push_back(t);
  for (auto& t : keep) {
    LAMP_CHECK(t.h()->is_contiguous(), "state tensor " << t.h()->describe() << " is not contiguous");
    ts.push_back(t.h());
  }
  // groups of <= 64 tensors per ncclGroup (one communicator, many buffers)
  for (size_t lo = 0; lo < ts.size(); lo += 64) {
    const int n = (int)std::min<size_t>(64, ts.size() - lo);
    std::vector<lamp_comm*> cm(n, comm);
    HCALL(lamp_comm_broadcast(ts.data() + lo, cm.data(), n, root));
  }
  opt.counters_from_state();  // AdamW's step count follows the broadcast state()[0]
  synced_with = comm;
}

int64_t DataParallel::step(SupervisedModel& model, Optimizer& opt, const Ten& samples, const Ten& target, const Ten& acc, double scheduleFactor) {
  if (!comm) {
    std::vector<Ten> grads;
    const int64_t n = model.addTotalLossAndReturnGradientsAndNumExamples(samples, target, acc, true, &grads);
    opt.step(grads, scheduleFactor);
    return n;
  }
  // replicas must start from one state: ranks that initialised differently (checkpoint read on rank 0 only, different seeds) would
  // otherwise train divergent models without any error
  if (synced_with != comm) sync_state(model, opt, 0);
  // averageGradients (distributed/package.scala:690-719): g *= n ; reduce(n) ; reduce(g) ; g /= sum n.
  // Here: flat f32 buckets [n*g_i ... | n], one all-reduce each, every rank divides by the summed n.
  std::vector<Var> params = model.module->parameters();
  LAMP_CHECK(!params.empty(), "data-parallel step on a model without parameters");
  const int device = params[0]->value.device();
  int64_t total = 0;
  for (auto& p : params) total += p->value.numel();
  size_t split = params.size();                     // params[split..) = deep bucket
  for (int64_t tail = 0; split > 0 && tail * 10 < total * 9;) tail += params[--split]->value.numel();

  for (auto& p : params) p->zeroGrad();
  Var output = run_module(model, samples, target);
  auto ln = model.loss(output, target);
  const int64_t n = ln.second;

  StreamScope scope(device);
  SharedDeviceScope shared(device);
  lamp_stream* const cur = scope.prev;
  if (!comm_stream) HCALL(lamp_stream_get_from_pool(1, device, &comm_stream));

  auto exchange = [&](size_t lo, size_t hi, Ten& bucket, std::vector<Ten>& grads, std::vector<lamp_tensor*>& gh) {
    int64_t cnt = 0;
    for (size_t i = lo; i < hi; i++) {
      grads.push_back(params[i]->grad_inplace());  // materialised (zeros if nothing flowed)
      gh.push_back(grads.back().h());
      cnt += grads.back().numel();
    }
    if (!bucket.defined() || bucket.numel() != cnt + 1) bucket = ops::zeros({cnt + 1}, kF32, device);
    // gradients and bucket were allocated under the compute stream and are now used on the exchange stream
    HCALL(lamp_tensor_record_stream(bucket.h(), comm_stream));
    for (auto* g : gh) HCALL(lamp_tensor_record_stream(g, comm_stream));
    HCALL(lamp_stream_wait_stream(comm_stream, cur));              // the gradients are complete on the compute stream
    scope.enter(comm_stream);
    if (!gh.empty()) HCALL(lamp_flatten_into_(bucket.h(), gh.data(), (int)gh.size(), (double)n));
    ops::fill_(ops::slice(bucket, 0, cnt, cnt + 1, 1), (double)n);
    lamp_tensor* bt[1] = {bucket.h()};
    lamp_comm* cm[1] = {comm};
    shared.acquire();                                              // from here until the join below an RCCL kernel holds CUs of this device
    HCALL(lamp_comm_all_reduce(bt, cm, 1, 0));
    // averaged gradients back into the parameters' grad buffers, still on the exchange stream: for the deep bucket this overlaps
    // the rest of backward instead of queueing behind it
    if (!gh.empty()) HCALL(lamp_unflatten_from_(gh.data(), (int)gh.size(), bucket.h(), 1));
    scope.leave();
  };

  // gradients of a parameter are final once every op that consumes it has run its backward
  std::unordered_map<Variable*, int> uses;
  for (size_t i = split; i < params.size(); i++) uses[params[i].get()] = 0;
  for (Variable* v : topological_sort(ln.first.get()))
    if (v->op)
      for (auto& p : v->op->params) { auto it = uses.find(p.first.get()); if (it != uses.end()) it->second++; }
  int64_t pending = 0;
  for (auto& kv : uses) if (kv.second > 0) pending++;
  std::vector<Ten> g_deep, g_rest;
  std::vector<lamp_tensor*> h_deep, h_rest;
  bool deep_sent = false;
  auto send_deep = [&]() { exchange(split, params.size(), bucket_deep, g_deep, h_deep); deep_sent = true; };
  if (pending == 0 && split < params.size()) send_deep();
  backprop(ln.first, [&](Variable* v) {
    if (deep_sent || !v->op) return;
    for (auto& p : v->op->params) {
      auto it = uses.find(p.first.get());
      if (it != uses.end() && it->second > 0 && --it->second == 0 && --pending == 0) { send_deep(); return; }
    }
  });
  if (!deep_sent) send_deep();
  exchange(0, split, bucket_rest, g_rest, h_rest);
  accumulate_loss(acc, ln.first->value, n);
  HCALL(lamp_stream_wait_stream(cur, comm_stream));               // both averaged gradient sets are visible to the compute stream
  shared.release();
  std::vector<Ten> grads(g_rest);
  grads.insert(grads.end(), g_deep.begin(), g_deep.end());
  opt.step(grads, scheduleFactor);
  return n;
}

// averageGradients + optimizer.step on gradients that are already computed (distributed/package.scala:690-759) - the second half of
// `step` for callers that produce the gradients themselves, e.g. by replaying forward + backprop from a HIP graph: one flat f32 bucket
// [n * g_i ... | n], one all-reduce on the CURRENT stream (nothing is left to overlap with), division by the summed n, step.
void DataParallel::exchange_and_step(SupervisedModel& model, Optimizer& opt, const std::vector<Ten>& grads, int64_t n, double scheduleFactor) {
  LAMP_CHECK(comm, "exchange_and_step needs a communicator");
  if (synced_with != comm) sync_state(model, opt, 0);
  std::vector<lamp_tensor*> gh;
  int64_t cnt = 0;
  for (auto& g : grads) { LAMP_CHECK(g.defined(), "exchange_and_step: undefined gradient"); gh.push_back(g.h()); cnt += g.numel(); }
  const int device = grads.empty() ? 0 : grads[0].device();
  if (!bucket_all.defined() || bucket_all.numel() != cnt + 1) bucket_all = ops::zeros({cnt + 1}, kF32, device);
  if (!gh.empty()) HCALL(lamp_flatten_into_(bucket_all.h(), gh.data(), (int)gh.size(), (double)n));
  ops::fill_(ops::slice(bucket_all, 0, cnt, cnt + 1, 1), (double)n);
  lamp_tensor* bt[1] = {bucket_all.h()};
  lamp_comm* cm[1] = {comm};
  HCALL(lamp_comm_all_reduce(bt, cm, 1, 0));
  if (!gh.empty()) HCALL(lamp_unflatten_from_(gh.data(), (int)gh.size(), bucket_all.h(), 1));
  opt.step(grads, scheduleFactor);
}

}  // namespace host
}  // namespace lamp
