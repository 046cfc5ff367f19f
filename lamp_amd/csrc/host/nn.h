// lamp.nn restated over the C ABI: modules, loss functions, optimisers, the supervised training
// step and its data-parallel form.
//
// Reference: lamp-core/src/main/scala/lamp/nn/{Module,Linear,Conv2D,BatchNorm,BatchNorm2D,
// LayerNorm,Dropout,MLP,LossFunctions,SupervisedModel,AdamW,SGD,package}.scala;
// example-cifar100/src/main/scala/lamp/example/cifar/cnn.scala:11-137 (Residual, Cnn.resnet);
// lamp-data/src/main/scala/lamp/data/distributed/package.scala:690-759 (averageGradients/oneBatch).
#pragma once
#include "ops.h"

struct lamp_comm;

namespace lamp {
namespace host {

struct Module {
  virtual ~Module() = default;
  // every state tensor in lamp's order, parameters (needsGrad) and constants alike (Module.scala:272-318)
  virtual void collect_state(std::vector<Var>& out) = 0;
  virtual Var forward(const Var& x) = 0;
  // GenericModule[A, B] with a tuple / case-class input (Transformer.scala, lm.scala): the Variables and the plain tensors of A in
  // the reference's order; an Option that is None is an undefined Ten.  Single-input modules ignore the extras.
  virtual Var forward_multi(const std::vector<Var>& xs, const std::vector<Ten>& aux) { (void)aux; return forward(xs.at(0)); }
  // LogSoftMax(Flatten(AvgPool2D(forward(x), pool))) where this module can produce it WITHOUT writing forward(x) - the tail of Cnn.resnet behind
  // its last block (cnn.scala:129-136; Sequential::forward asks the module in front of that tail).  `probe`: only say whether it can (structure
  // alone: a non-null dummy), compute nothing.  nullptr: cannot - the caller runs forward and the tail as usual.
  virtual Var forward_pool_tail(const Var& x, int64_t pool, bool probe) { (void)x; (void)pool; (void)probe; return nullptr; }
  virtual void set_training(bool) {}   // TrainingMode.asEval / asTraining
  std::vector<Var> state() { std::vector<Var> s; collect_state(s); return s; }
  std::vector<Var> parameters() {
    std::vector<Var> p;
    for (auto& v : state()) if (v->needsGrad()) p.push_back(v);
    return p;
  }
  void zeroGrad() { for (auto& p : parameters()) p->zeroGrad(); }
  // Module.gradients (Module.scala:300-314)
  std::vector<Ten> gradients(const Var& loss, bool zeroGradFirst = true) {
    if (zeroGradFirst) zeroGrad();
    backprop(loss);
    std::vector<Ten> g;
    for (auto& p : parameters()) g.push_back(p->grad_inplace());   // exclusive, materialised (zeros if nothing flowed)
    return g;
  }
};
using Mod = std::shared_ptr<Module>;

struct Linear : Module {        // nn/Linear.scala:7-67 (2-D weights; bias shape [1, out])
  Var weights, bias;
  Linear(Var w, Var b) : weights(std::move(w)), bias(std::move(b)) {}
  static Mod make(int64_t in, int64_t out, int dtype, int device, bool bias);
  void collect_state(std::vector<Var>& o) override { o.push_back(weights); if (bias) o.push_back(bias); }
  Var forward(const Var& x) override;
};
struct Conv2D : Module {        // nn/Conv2D.scala:8-83 (bias is always a tensor; const zeros when bias=false)
  Var weights, bias;
  int64_t stride, padding, dilation, groups;
  static Mod make(int64_t inC, int64_t outC, int64_t k, int dtype, int device, bool bias, int64_t stride, int64_t padding,
                  int64_t dilation, int64_t groups);
  void collect_state(std::vector<Var>& o) override { o.push_back(weights); o.push_back(bias); }
  Var forward(const Var& x) override;
};
struct BatchNorm : Module {     // nn/BatchNorm.scala:7-88 and nn/BatchNorm2D.scala:8-70
  Var weight, bias, runningMean, runningVar;
  bool training = true, two_d = false;
  double momentum = 0.1, eps = 1e-5;
  static Mod make(int64_t features, int dtype, int device, bool two_d);
  void collect_state(std::vector<Var>& o) override { o.push_back(weight); o.push_back(bias); o.push_back(runningMean); o.push_back(runningVar); }
  Var forward(const Var& x) override;
  bool can_fuse_relu(const Var& x) const;       // BatchNorm2D on maps of >= 64 elements
  Var forward_relu(const Var& x);                // relu(forward(x)) as one fused op, identical values
  bool can_fuse_add_relu(const Var& x) const;   // additionally: training mode
  Var forward_add_relu(const Var& x, const Var& addend);   // relu(forward(x) + addend) as one fused op
  void set_training(bool t) override { training = t; }
};
struct LayerNorm : Module {     // nn/LayerNorm.scala:8-57
  Var scale, bias;              // either may be null
  std::vector<int64_t> normalizedShape;
  double eps = 1e-5;
  static Mod make(const std::vector<int64_t>& shape, int dtype, int device, bool scale, bool bias);
  void collect_state(std::vector<Var>& o) override { if (scale) o.push_back(scale); if (bias) o.push_back(bias); }
  Var forward(const Var& x) override { return F::layer_norm(x, scale, bias, normalizedShape, eps); }
};
struct Dropout : Module {       // nn/Dropout.scala:6-8 (skips the op when p <= 0)
  double prob; bool training;
  Dropout(double p, bool t) : prob(p), training(t) {}
  void collect_state(std::vector<Var>&) override {}
  Var forward(const Var& x) override { return prob > 0 ? F::dropout(x, prob, training) : x; }
  void set_training(bool t) override { training = t; }
};
struct Fun : Module {           // nn `Fun(scope => input => ...)`
  std::function<Var(const Var&)> f;
  std::string tag;              // "relu" for the plain relu: lets Sequential fuse BatchNorm2D -> relu; "avgpool2d" / "flatten_last" /
                                // "logsoftmax" (with their arguments in a, b): the tail of Cnn.resnet runs as one kernel
  double a = 0, b = 0;
  explicit Fun(std::function<Var(const Var&)> f_, std::string tag_ = "", double a_ = 0, double b_ = 0)
      : f(std::move(f_)), tag(std::move(tag_)), a(a_), b(b_) {}
  void collect_state(std::vector<Var>&) override {}
  Var forward(const Var& x) override { return f(x); }
};
struct Sequential : Module {
  std::vector<Mod> mods;
  explicit Sequential(std::vector<Mod> m) : mods(std::move(m)) {}
  void collect_state(std::vector<Var>& o) override { for (auto& m : mods) m->collect_state(o); }
  Var forward(const Var& x) override;
  Var forward_pool_tail(const Var& x, int64_t pool, bool probe) override;
  void set_training(bool t) override { for (auto& m : mods) m->set_training(t); }
};
struct Residual : Module {      // cnn.scala:11-21
  Mod right, left;              // left may be null (identity)
  Residual(Mod r, Mod l) : right(std::move(r)), left(std::move(l)) {}
  void collect_state(std::vector<Var>& o) override { right->collect_state(o); if (left) left->collect_state(o); }
  Var forward(const Var& x) override {
    Var r = right->forward(x);
    Var l = left ? left->forward(x) : x;
    return F::add(r, l);
  }
  // relu(forward(x)); when the right branch is a Sequential ending in a fusable BatchNorm2D the add and the relu run inside
  // its normalise kernel (identical values)
  // pool_tail > 0: LogSoftMax(Flatten(AvgPool2D(relu(forward(x)), pool_tail))) - one node with the two batch norms where that form applies
  Var forward_relu(const Var& x, int64_t pool_tail = 0);
  void set_training(bool t) override { right->set_training(t); if (left) left->set_training(t); }
};

Mod make_fun(const std::string& name, double a = 0, double b = 0);
Mod residual_make(int64_t inC, int64_t outC, int dtype, int device, double dropout, int64_t stride);   // cnn.scala:33-87
Mod cnn_resnet(int64_t numClasses, double dropout, int dtype, int device);                              // cnn.scala:89-137
// MLP.apply (nn/MLP.scala:40-167): activation relu|gelu|sigmoid|hardswish|swish1, norm 0 none / 1 batch / 2 layer(bias,scale)
Mod mlp(int64_t in, int64_t out, const std::vector<int64_t>& hidden, int dtype, int device, double dropout, bool lastNonLinearity,
        const std::string& activation, int norm, bool bias);

// ---- optimisers ------------------------------------------------------------------------------
struct Optimizer {
  virtual ~Optimizer() = default;
  virtual void step(const std::vector<Ten>& gradients, double scheduleFactor) = 0;   // gradient may be undefined (None)
  virtual std::vector<Ten> state() = 0;
  // Optimizer.load: copyFrom into every state tensor in order (AdamW.scala:87-93, SGD.scala:38-42)
  virtual void load(const std::vector<Ten>& tensors) {
    std::vector<Ten> st = state();
    for (size_t i = 0; i < st.size() && i < tensors.size(); i++) ops::copy_(st[i], tensors[i]);
  }
  // host-side counters that mirror a state tensor (AdamW's step count) re-read from it after the state tensors were overwritten
  // in place (broadcast from another rank)
  virtual void counters_from_state() {}
};
struct AdamW : Optimizer {      // nn/AdamW.scala:29-177
  std::vector<Ten> parameters, mt, vt, workingCopy;   // workingCopy[i] undefined when not mixed precision
  double weightDecay, learningRate, beta1, beta2, eps;
  bool has_clip; double clip; bool debias, mixedPrecision;
  // OptimizedHyperparameter = PTag => Double (Optimizer.scala / AdamW.scala:29-47): when non-empty, one value per parameter
  std::vector<double> weightDecayPer, learningRatePer, beta1Per, beta2Per;
  int64_t stepCount = 0;
  Ten stepCountSTen;            // f64 scalar, state()[0] (AdamW.scala:97)
  AdamW(const std::vector<Ten>& params, double wd, double lr, double b1, double b2, double eps, bool has_clip, double clip,
        bool debias, bool mixed);
  void step(const std::vector<Ten>& gradients, double scheduleFactor) override;
  std::vector<Ten> state() override;
  void load(const std::vector<Ten>& tensors) override;   // also restores stepCount from state()[0]
  void counters_from_state() override;
};
struct SGDW : Optimizer {       // nn/SGD.scala:19-99
  std::vector<Ten> parameters, velocity;
  double learningRate, weightDecay, momentum; bool has_momentum, has_clip; double clip;
  SGDW(const std::vector<Ten>& params, double lr, double wd, bool has_momentum, double momentum, bool has_clip, double clip);
  void step(const std::vector<Ten>& gradients, double scheduleFactor) override;
  std::vector<Ten> state() override { return velocity; }
};

// ---- supervised model (nn/SupervisedModel.scala:151-211, LossFunctions.scala:39-55) -------------
struct SupervisedModel {
  Mod module;
  int loss_kind = 0;            // 0 NLL, 1 MSE, 2 identity
  Ten classWeights; int64_t reduction = 1, ignore = -100;
  std::pair<Var, int64_t> loss(const Var& output, const Ten& target);
  // addTotalLossAndReturnGradientsAndNumExamples: acc += loss * n ; returns (n, gradients)
  int64_t addTotalLossAndReturnGradientsAndNumExamples(const Ten& samples, const Ten& target, const Ten& acc, bool zeroGrad,
                                                       std::vector<Ten>* gradients);
  int64_t addTotalLossAndReturnNumExamples(const Ten& samples, const Ten& target, const Ten& acc);
};

// one data-parallel step: local gradients, example-weighted all-reduce of ONE flat f32 bucket
// (numExamples appended as the last element), identical optimiser step on every rank.
// The exchange runs on a second stream in TWO buckets: the parameters of the deep layers (the tail of the parameter list
// that holds >= 90 % of the elements; their gradients are complete first) are packed and all-reduced while backward
// continues through the shallow layers, the rest follows after backward.  Every rank issues the same two collectives in the
// same order.
struct DataParallel {
  lamp_comm* comm = nullptr;    // null => single process (no exchange)
  Ten bucket_deep, bucket_rest; // f32 [sum numel + 1] each, last element = numExamples
  Ten bucket_all;               // the single bucket of exchange_and_step
  lamp_stream* comm_stream = nullptr;
  lamp_comm* synced_with = nullptr;   // communicator the replicas were last made identical over
  int64_t step(SupervisedModel& model, Optimizer& opt, const Ten& samples, const Ten& target, const Ten& acc, double scheduleFactor = 1.0);
  // broadcast(root): module.state (parameters AND batch-norm running statistics) and the optimiser state from rank `root` to every
  // rank (distributed/package.scala:683-688 does this before every batch; here the replicas step identically, so once before the
  // first step - and whenever the caller wants the non-parameter state of rank 0 everywhere: validation, checkpoints)
  void sync_state(SupervisedModel& model, Optimizer& opt, int root = 0);
  // the exchange + optimiser half of `step` on gradients computed elsewhere (a replayed HIP graph): example-weighted mean over the ranks
  void exchange_and_step(SupervisedModel& model, Optimizer& opt, const std::vector<Ten>& grads, int64_t numExamples, double scheduleFactor = 1.0);
  ~DataParallel();
};

// Single-process data parallel (lamp-data DataParallel.scala:195-311, `synchronousStep`): the main model + optimiser live on one
// GPU, every other GPU holds a replica.  Per step: the main state is copied to the replicas, every model computes its gradients on
// its own batch from its own host thread (current device and stream are per thread), the gradients are weighted by their example
// counts, copied to the main GPU and summed there, divided by the total, and the optimiser steps the main model.
// samples / targets / accs are ordered main first.  Returns the number of examples of all models.
int64_t data_parallel_synchronous_step(SupervisedModel& main, Optimizer& opt, const std::vector<SupervisedModel*>& replicas,
                                       const std::vector<Ten>& samples, const std::vector<Ten>& targets, const std::vector<Ten>& accs,
                                       bool zeroGrad, bool step, double scheduleFactor);

}  // namespace host
}  // namespace lamp

struct lamp_module { lamp::host::Mod m; };   // the C handle of lamp_host.h
