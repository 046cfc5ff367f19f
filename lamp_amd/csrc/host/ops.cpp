// lamp's autograd operators, restated over the C ABI.
//
// Reference: lamp-core/src/main/scala/lamp/autograd/ops.scala - each function below cites the
// case class it mirrors.  The forward runs at construction, every backward closure ADDS its
// partial derivative to the gradient of its input (`out += ...`), exactly as in the reference,
// including its quirks (relu gradient at 0 is 1; IndexSelect's `out += out.indexAdd(...)`).
//
// Gradient buffers are created by the first accumulation (see Variable in autograd.h): where the
// reference does zeros_like + `out += v`, the first closure to run simply installs v (0 + v == v
// exactly), later closures add in place.  In-place native forms (addmm_out_transposed*, the fused
// relu backward) take their beta = 0 / out-of-place variant for that first accumulation.
#include "ops.h"
#include <cstring>
#include <map>
#include <tuple>

#include <unordered_set>

namespace lamp {
namespace host {

Var make_const(const Ten& t) {
  auto v = std::make_shared<Variable>();
  v->value = t;
  return v;
}
Var make_param(const Ten& t) {
  auto v = std::make_shared<Variable>();
  v->value = t;
  v->wants_grad = true;
  return v;
}
Var make_result(const std::shared_ptr<Op>& op, const Ten& value) {
  auto v = std::make_shared<Variable>();
  v->op = op;
  v->value = value;
  v->wants_grad = true;
  return v;
}

std::vector<Variable*> topological_sort(Variable* root) {
  // children first; reversed at the end => root first.  Marks in a hash set and an explicit stack: the language model's graph has a few
  // thousand nodes and chains deeper than a recursion should go (a linear mark search made the sort quadratic).
  std::vector<Variable*> order;
  std::unordered_set<Variable*> marks;
  struct Frame { Variable* n; size_t next; };
  std::vector<Frame> stack;
  if (marks.insert(root).second) stack.push_back({root, 0});
  while (!stack.empty()) {
    Frame& f = stack.back();
    Variable* n = f.n;
    const size_t np = n->op ? n->op->params.size() : 0;
    if (f.next < np) {
      Variable* c = n->op->params[f.next++].first.get();
      if (marks.insert(c).second) stack.push_back({c, 0});   // (may invalidate f: not used afterwards)
      continue;
    }
    order.push_back(n);
    stack.pop_back();
  }
  return std::vector<Variable*>(order.rbegin(), order.rend());
}

void backprop(const Var& root) { backprop(root, nullptr); }
// `after` runs once per visited node, after the node's backward closures (also for nodes nothing flowed into): the
// data-parallel step uses it to start the gradient exchange of the deep layers while the shallow ones are still in backward
// true while a backprop with an `after` hook runs on this thread: a parameter's gradient must be final when the hook sees the parameter
// (the data-parallel step starts its exchange there), so nothing may put a weight gradient off
static thread_local bool tls_backprop_hooked = false;
// the handle of the ones the running backprop seeded its one-element root with (nullptr otherwise): a closure that sees exactly this tensor
// as its incoming derivative knows the derivative IS one without reading it (the loss tail's gradient is computed ahead for that case)
static thread_local const lamp_tensor* tls_seed_ones = nullptr;
void backprop(const Var& root, const std::function<void(Variable*)>& after) {
  if (!root->needsGrad()) return;
  struct Hooked { bool prev; Hooked(bool h) : prev(tls_backprop_hooked) { tls_backprop_hooked = h; } ~Hooked() { tls_backprop_hooked = prev; } } hooked((bool)after);
  // partialDerivative.get.fill_(1d).  A one-element root (every loss) takes a constant kept per (thread, device, dtype, stream)
  // instead of a fill launch per step; it is marked shared, so anything that wanted to modify it in place copies it first.
  if (root->value.numel() == 1) {
    struct Key { int device, dtype, ndim; void* st; bool operator<(const Key& o) const { return std::tie(device, dtype, ndim, st) < std::tie(o.device, o.dtype, o.ndim, o.st); } };
    static thread_local std::map<Key, Ten> ones;
    void* native = nullptr;                                 // the stream the constant was filled on (a handle is a fresh object per call)
    if (root->value.h()->is_device()) {
      lamp_stream* cur = nullptr;
      HCALL(lamp_stream_get_current(root->value.device(), &cur));
      const int rc = lamp_stream_native(cur, &native);
      lamp_stream_release(cur);
      if (rc != 0) throw ::lamp::Error(lamp_last_error());
    }
    const Key k{root->value.device(), root->value.dtype(), root->value.ndim(), native};
    auto it = ones.find(k);
    int capturing = 0;
    if (it == ones.end()) HCALL(lamp_graph_is_capturing(&capturing));
    if (it == ones.end() && capturing) {
      // a fill recorded into a graph has not run yet: a constant created here would be handed to later eager passes unfilled
      root->grad = ops::ones_like(root->value);
      root->grad_shared = false;
    } else {
      if (it == ones.end()) it = ones.emplace(k, ops::ones_like(root->value)).first;
      root->grad = it->second;
      root->grad_shared = true;
    }
  } else {
    root->grad = ops::ones_like(root->value);
    root->grad_shared = false;
  }
  struct Seed { const lamp_tensor* prev; Seed(const lamp_tensor* h) : prev(tls_seed_ones) { tls_seed_ones = h; } ~Seed() { tls_seed_ones = prev; } }
      seed(root->value.numel() == 1 ? root->grad.h() : nullptr);
  const std::vector<Variable*> order = topological_sort(root.get());
  for (Variable* v : order) {                 // what an earlier, abandoned pass over this graph may have left parked
    v->pending.clear();
    if (v->op && v->op->reset) v->op->reset();
  }
  for (Variable* v : order) {
    if (!v->pending.empty()) {                // (all consumers are done: nothing will complete a pair now)
      auto fs = std::move(v->pending);
      v->pending.clear();
      for (auto& f : fs) f();
    }
    if (v->op && v->has_grad()) {             // a node nothing flowed into contributes exact zeros
      for (auto& p : v->op->params)
        if (p.first->needsGrad()) p.second(v->grad, *p.first);
    }
    if (after) after(v);
  }
  HCALL(lamp_flush_deferred());               // the convolutions' weight-gradient reductions of this pass, in one launch
}

namespace {
std::shared_ptr<Op> new_op(const char* name) {
  auto o = std::make_shared<Op>();
  o->name = name;
  return o;
}
// out (+/-)= p.unbroadcast(shape)
void acc_unbroadcast(const Ten& p, Variable& out, const std::vector<int64_t>& shape, bool subtract = false) {
  const bool same = p.shape() == shape;
  Ten u = same ? p : ops::unbroadcast(p, shape);
  if (subtract) out.subtract(u);
  else out.accumulate(u, !same);
}
// a dense copy of p broadcast to `shape`
Ten broadcast_copy(const Ten& p, const std::vector<int64_t>& shape) {
  lamp_tensor *e = nullptr, *c = nullptr;
  HCALL(lamp_expand(&e, p.h(), shape.data(), (int)shape.size()));
  Ten eh(e);
  HCALL(lamp_clone(&c, e));
  return Ten(c);
}
}  // namespace

namespace F {

// ---- shape ops (ops.scala:15-49, 1827-1843) -----------------------------------------------------
Var transpose(const Var& a, int64_t d1, int64_t d2) {
  auto op = new_op("Transpose");
  op->params.push_back({a, [d1, d2](const Ten& p, Variable& out) { out.accumulate(ops::transpose(p, d1, d2), false); }});
  return make_result(op, ops::transpose(a->value, d1, d2));
}
Var view(const Var& a, const std::vector<int64_t>& shape) {
  auto op = new_op("View");
  op->params.push_back({a, [](const Ten& p, Variable& out) { out.accumulate(ops::reshape(p, out.shape()), false); }});
  return make_result(op, ops::view(a->value, shape));
}
Var reshape(const Var& a, const std::vector<int64_t>& shape) {
  auto op = new_op("Reshape");
  op->params.push_back({a, [](const Ten& p, Variable& out) { out.accumulate(ops::reshape(p, out.shape()), false); }});
  return make_result(op, ops::reshape(a->value, shape));
}
Var flatten(const Var& a, int64_t start, int64_t end) {
  auto op = new_op("Flatten");
  op->params.push_back({a, [](const Ten& p, Variable& out) { out.accumulate(ops::reshape(p, out.shape()), false); }});
  return make_result(op, ops::flatten(a->value, start, end));
}
Var concatenate(const std::vector<Var>& as, int64_t dim) {   // ops.scala:51-62
  auto op = new_op("Concatenate");
  std::vector<Ten> vals;
  int64_t from = 0;
  for (auto& a : as) {
    const int64_t to = from + a->value.size((int)dim);
    op->params.push_back({a, [dim, from, to](const Ten& p, Variable& out) { out.accumulate(ops::slice(p, dim, from, to, 1), false); }});
    vals.push_back(a->value);
    from = to;
  }
  return make_result(op, ops::cat(vals, dim));
}

// ---- arithmetic (ops.scala:511-621) -------------------------------------------------------------
Var add(const Var& a, const Var& b) {
  auto op = new_op("Add");
  auto as = a->shape(), bs = b->shape();
  op->params.push_back({a, [as](const Ten& p, Variable& out) { acc_unbroadcast(p, out, as); }});
  op->params.push_back({b, [bs](const Ten& p, Variable& out) { acc_unbroadcast(p, out, bs); }});
  return make_result(op, ops::add(a->value, b->value));
}
Var const_add(const Var& a, double b) {
  auto op = new_op("ConstAdd");
  auto as = a->shape();
  op->params.push_back({a, [as](const Ten& p, Variable& out) { acc_unbroadcast(p, out, as); }});
  return make_result(op, ops::add_scalar(a->value, b));
}
Var minus(const Var& a, const Var& b) {
  auto op = new_op("Minus");
  auto as = a->shape(), bs = b->shape();
  op->params.push_back({a, [as](const Ten& p, Variable& out) { acc_unbroadcast(p, out, as); }});
  op->params.push_back({b, [bs](const Ten& p, Variable& out) { acc_unbroadcast(p, out, bs, true); }});
  return make_result(op, ops::sub(a->value, b->value));
}
Var const_mult(const Var& a, double b) {
  auto op = new_op("ConstMult");
  auto as = a->shape();
  op->params.push_back({a, [as, b](const Ten& p, Variable& out) {
    Ten t = ops::mul_scalar(p, b);
    if (t.shape() == as) out.accumulate(t, true); else out.accumulate(ops::unbroadcast(t, as), true);
  }});
  return make_result(op, ops::mul_scalar(a->value, b));
}
Var mult(const Var& a, const Var& b) {
  auto op = new_op("Mult");
  auto as = a->shape(), bs = b->shape();
  Ten av = a->value, bv = b->value;
  op->params.push_back({a, [as, bv](const Ten& p, Variable& out) {
    Ten t = ops::mul(p, bv);
    out.accumulate(t.shape() == as ? t : ops::unbroadcast(t, as), true);
  }});
  op->params.push_back({b, [bs, av](const Ten& p, Variable& out) {
    Ten t = ops::mul(p, av);
    out.accumulate(t.shape() == bs ? t : ops::unbroadcast(t, bs), true);
  }});
  return make_result(op, ops::mul(a->value, b->value));
}
// (a * b) + c as one kernel with the values of the two-operator chain; the closures are Mult's and Add's
Var mult_add(const Var& a, const Var& b, const Var& c) {
  auto op = new_op("MultAdd");
  auto as = a->shape(), bs = b->shape(), cs = c->shape();
  Ten av = a->value, bv = b->value;
  op->params.push_back({a, [as, bv](const Ten& p, Variable& out) {
    Ten t = ops::mul(p, bv);
    out.accumulate(t.shape() == as ? t : ops::unbroadcast(t, as), true);
  }});
  op->params.push_back({b, [bs, av](const Ten& p, Variable& out) {
    Ten t = ops::mul(p, av);
    out.accumulate(t.shape() == bs ? t : ops::unbroadcast(t, bs), true);
  }});
  op->params.push_back({c, [cs](const Ten& p, Variable& out) { out.accumulate(p.shape() == cs ? p : ops::unbroadcast(p, cs), p.shape() != cs); }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_mul_add(&o, av.h(), bv.h(), c->value.h()));
  return make_result(op, Ten(o));
}
Var div(const Var& a, const Var& b) {
  auto op = new_op("Div");
  auto as = a->shape(), bs = b->shape();
  Ten bv = b->value;
  Ten val = ops::div(a->value, b->value);
  op->params.push_back({a, [as, bv](const Ten& p, Variable& out) {
    Ten t = ops::div(p, bv);
    out.accumulate(t.shape() == as ? t : ops::unbroadcast(t, as), true);
  }});
  op->params.push_back({b, [bs, bv, val](const Ten& p, Variable& out) {
    Ten tmp = ops::div(val, bv);
    ops::mul_(tmp, p);
    out.subtract(tmp.shape() == bs ? tmp : ops::unbroadcast(tmp, bs));
  }});
  return make_result(op, val);
}
Var sum(const Var& a, const std::vector<int64_t>& dim, bool keepDim) {   // ops.scala:623-630: out += p (broadcast)
  auto op = new_op("Sum");
  auto as = a->shape();
  op->params.push_back({a, [as, dim, keepDim](const Ten& p, Variable& out) {
    if (p.shape() == as) { out.accumulate(p, false); return; }
    // `out += p` broadcasts p; a reduced dim that was dropped (keepDim = false) only broadcasts when ATen's
    // rules allow it, which is the case lamp uses (full sums, or keepDim = true)
    if (!out.has_grad()) out.accumulate(broadcast_copy(p, as), true);
    else ops::add_(out.grad_inplace(), p);
    (void)dim; (void)keepDim;
  }});
  return make_result(op, dim.empty() ? ops::sum_all(a->value) : ops::sum_dims(a->value, dim, keepDim));
}
Var mean(const Var& a, const std::vector<int64_t>& dim, bool keepDim) {  // ops.scala:1034-1054
  auto op = new_op("Mean");
  int64_t n = 1;
  for (auto d : dim) n *= a->value.size((int)d);
  auto as = a->shape();
  op->params.push_back({a, [n, as](const Ten& p, Variable& out) {
    if (!out.has_grad()) {
      Ten e = broadcast_copy(p, as);
      ops::mul_scalar_(e, 1.0 / (double)n);
      out.accumulate(e, true);
    } else ops::add_(out.grad_inplace(), p, 1.0 / (double)n);
  }});
  return make_result(op, ops::mean_dims(a->value, dim, keepDim));
}
Var norm2(const Var& a, const std::vector<int64_t>& dim, bool keepDim) {  // ops.scala:632-645
  auto op = new_op("Norm2");
  Ten av = a->value;
  Ten val = ops::norm2_dims(a->value, dim, keepDim);
  op->params.push_back({a, [av, val](const Ten& p, Variable& out) {
    Ten pa = ops::mul(p, av);
    ops::div_(pa, val);
    out.accumulate(pa, true);
  }});
  return make_result(op, val);
}

// ---- GEMM (ops.scala:665-724) -------------------------------------------------------------------
namespace {
// out = beta * out + op(a) op(b) through the transposed natives; first accumulation uses beta = 0 into a new buffer
template <class Fn> void gemm_accumulate(Variable& out, Fn call) {
  if (!out.has_grad()) {
    lamp_tensor* fresh = nullptr;
    HCALL(lamp_empty(&fresh, out.value.h()->sizes, out.value.ndim(), out.value.dtype(), out.value.device()));
    Ten t(fresh);
    call(t, 0.0);
    out.grad = t;
    out.grad_shared = false;
  } else {
    call(out.grad_inplace(), 1.0);
  }
}
}  // namespace
Var mm(const Var& a, const Var& b) {
  auto op = new_op("MatMul");
  Ten av = a->value, bv = b->value;
  op->params.push_back({a, [bv](const Ten& p, Variable& out) {     // dA += p . B^T   (Tensor.addmm_out_transposed2)
    gemm_accumulate(out, [&](const Ten& o, double beta) { HCALL(lamp_addmm_out_transposed2(o.h(), o.h(), p.h(), bv.h(), beta, 1.0)); });
  }});
  op->params.push_back({b, [av](const Ten& p, Variable& out) {     // dB += A^T . p   (Tensor.addmm_out_transposed1)
    gemm_accumulate(out, [&](const Ten& o, double beta) { HCALL(lamp_addmm_out_transposed1(o.h(), o.h(), av.h(), p.h(), beta, 1.0)); });
  }});
  return make_result(op, ops::mm(a->value, b->value));
}
// x.mm(w) + bias as one GEMM whose epilogue adds the row vector (bf16: the product is rounded before the addition, so the values are
// bitwise those of the two-operator chain); the three closures are those of MatMul and of Add's broadcast operand
Var linear_bias(const Var& x, const Var& w, const Var& bias) {
  LAMP_CHECK(x->value.ndim() == 2 && w->value.ndim() == 2, "linear_bias: 2-D operands");
  auto op = new_op("MatMulAdd");
  Ten xv = x->value, wv = w->value;
  auto bs = bias->shape();
  op->params.push_back({x, [wv](const Ten& p, Variable& out) {
    gemm_accumulate(out, [&](const Ten& o, double beta) { HCALL(lamp_addmm_out_transposed2(o.h(), o.h(), p.h(), wv.h(), beta, 1.0)); });
  }});
  op->params.push_back({w, [xv](const Ten& p, Variable& out) {
    gemm_accumulate(out, [&](const Ten& o, double beta) { HCALL(lamp_addmm_out_transposed1(o.h(), o.h(), xv.h(), p.h(), beta, 1.0)); });
  }});
  op->params.push_back({bias, [bs](const Ten& p, Variable& out) { out.accumulate(p.shape() == bs ? p : ops::unbroadcast(p, bs), p.shape() != bs); }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_linear_bias(&o, xv.h(), wv.h(), bias->value.h()));
  return make_result(op, Ten(o));
}
Var bmm(const Var& a, const Var& b) {
  auto op = new_op("BatchedMatMul");
  Ten av = a->value, bv = b->value;
  op->params.push_back({a, [bv](const Ten& p, Variable& out) {
    gemm_accumulate(out, [&](const Ten& o, double beta) { HCALL(lamp_baddbmm_out_transposed2(o.h(), o.h(), p.h(), bv.h(), beta, 1.0)); });
  }});
  op->params.push_back({b, [av](const Ten& p, Variable& out) {
    gemm_accumulate(out, [&](const Ten& o, double beta) { HCALL(lamp_baddbmm_out_transposed1(o.h(), o.h(), av.h(), p.h(), beta, 1.0)); });
  }});
  return make_result(op, ops::bmm(a->value, b->value));
}

// ---- element-wise functions (ops.scala:754-1032) ------------------------------------------------
Var exp(const Var& a) {
  auto op = new_op("Exp");
  Ten val = ops::exp(a->value);
  op->params.push_back({a, [val](const Ten& p, Variable& out) { out.addcmul(p, val, 1.0); }});
  return make_result(op, val);
}
Var log(const Var& a) {
  auto op = new_op("Log");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) { out.addcmul(p, ops::reciprocal(av), 1.0); }});
  return make_result(op, ops::log(a->value));
}
Var log1p(const Var& a) {
  auto op = new_op("Log1p");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) {
    Ten tmp = ops::add_scalar(av, 1.0);
    HCALL(lamp_reciprocal_(tmp.h()));
    out.addcmul(p, tmp, 1.0);
  }});
  return make_result(op, ops::log1p(a->value));
}
Var sin(const Var& a) {
  auto op = new_op("Sin");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) { out.addcmul(p, ops::cos(av), 1.0); }});
  return make_result(op, ops::sin(a->value));
}
Var cos(const Var& a) {
  auto op = new_op("Cos");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) { out.addcmul(p, ops::sin(av), -1.0); }});
  return make_result(op, ops::cos(a->value));
}
Var tanh(const Var& a) {
  auto op = new_op("Tanh");
  Ten val = ops::tanh(a->value);
  op->params.push_back({a, [val](const Ten& p, Variable& out) { out.accumulate(ops::tanh_backward(p, val), true); }});
  return make_result(op, val);
}
Var pow_const(const Var& a, double e) {
  auto op = new_op("PowConst");
  Ten av = a->value;
  op->params.push_back({a, [av, e](const Ten& p, Variable& out) { out.addcmul(p, ops::pow_scalar(av, e - 1), e); }});
  return make_result(op, ops::pow_scalar(a->value, e));
}
// Relu (ops.scala:918-935): out += p * where(a < 0, 0, 1) - five ATen calls and three temporaries in
// the reference; here the same arithmetic in one fused kernel.
namespace {
void relu_like_backward(const Ten& p, Variable& out, const Ten& x, double slope) {
  if (!out.has_grad()) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_relu_backward(&t, p.h(), x.h(), slope));
    out.accumulate(Ten(t), true);
  } else {
    HCALL(lamp_relu_backward_accumulate_(out.grad_inplace().h(), p.h(), x.h(), slope));
  }
}
}  // namespace
Var relu(const Var& a) {
  auto op = new_op("Relu");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) { relu_like_backward(p, out, av, 0.0); }});
  return make_result(op, ops::relu(a->value));
}
Var leaky_relu(const Var& a, double slope) {
  auto op = new_op("LeakyRelu");
  Ten av = a->value;
  op->params.push_back({a, [av, slope](const Ten& p, Variable& out) { relu_like_backward(p, out, av, slope); }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_leaky_relu(&o, a->value.h(), slope));
  return make_result(op, Ten(o));
}
Var gelu(const Var& a) {
  auto op = new_op("Gelu");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) { out.accumulate(ops::gelu_backward(p, av), true); }});
  return make_result(op, ops::gelu(a->value));
}
Var sigmoid(const Var& a) {
  auto op = new_op("Sigmoid");
  Ten val = ops::sigmoid(a->value);
  op->params.push_back({a, [val](const Ten& p, Variable& out) { out.accumulate(ops::sigmoid_backward(p, val), true); }});
  return make_result(op, val);
}
Var hardswish(const Var& a) {
  auto op = new_op("HardSwish");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) { out.accumulate(ops::hardswish_backward(p, av), true); }});
  return make_result(op, ops::hardswish(a->value));
}
Var softplus(const Var& a, double beta, double threshold) {
  auto op = new_op("Softplus");
  Ten av = a->value;
  op->params.push_back({a, [av, beta, threshold](const Ten& p, Variable& out) {
    lamp_tensor* o = nullptr;
    HCALL(lamp_softplus_backward(&o, p.h(), av.h(), beta, threshold));
    out.accumulate(Ten(o), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_softplus(&o, a->value.h(), beta, threshold));
  return make_result(op, Ten(o));
}
Var log_softmax(const Var& a, int64_t dim) {   // ops.scala:955-975
  auto op = new_op("LogSoftMax");
  lamp_tensor* o = nullptr;
  HCALL(lamp_log_softmax(&o, a->value.h(), dim));
  Ten val(o);
  op->params.push_back({a, [val, dim](const Ten& p, Variable& out) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_log_softmax_backward_data(&t, p.h(), val.h(), dim));
    out.accumulate(Ten(t), true);
  }});
  return make_result(op, val);
}
Var dropout(const Var& a, double prob, bool train) {   // ops.scala:1079-1100
  auto op = new_op("Dropout");
  if (prob <= 0.0) {
    op->params.push_back({a, [](const Ten& p, Variable& out) { out.accumulate(p, false); }});
    return make_result(op, a->value);
  }
  Ten mask = ops::ones_like(a->value);
  HCALL(lamp_dropout_(mask.h(), prob, train));
  op->params.push_back({a, [mask](const Ten& p, Variable& out) { out.addcmul(p, mask, 1.0); }});
  return make_result(op, ops::mul(a->value, mask));
}

// ---- losses (ops.scala:1176-1304) ---------------------------------------------------------------
Var nll_loss(const Var& input, const Ten& target, const Ten& weights, int64_t reduction, int64_t ignore) {
  return nll_loss_accumulate(input, target, weights, reduction, ignore, Ten(), 0.0);
}
// acc (optional): acc += scale * loss in the loss kernel's launch (the caller's epoch-loss bookkeeping, IOLoops.scala:714)
Var nll_loss_accumulate(const Var& input, const Ten& target, const Ten& weights, int64_t reduction, int64_t ignore, const Ten& acc, double scale) {
  LAMP_CHECK(input->value.ndim() == 2, "Nll Loss assumes 2D input (samples x classes). Higher dimensions not implemented.");
  LAMP_CHECK(target.ndim() == 1, "Target should be a 1D tensor with [0,C-1] integers, C number of classes.");
  auto op = new_op("NllLoss");
  lamp_tensor *v = nullptr, *tw = nullptr, *pgh = nullptr;
  static const bool fuse_tail = [] { const char* e = getenv("LAMP_FUSE_LOSS_BACKWARD"); return !(e && e[0] == '0'); }();
  // Cnn.resnet's tail with the loss as the root (SupervisedModel.scala:190-211): the loss launch also produces, for a derivative of one, the
  // pooled LogSoftMax's input gradient as one value per plane (LAMP_FUSE_LOSS_TAIL=0: the separate backward launch)
  static const bool fuse_ahead = [] { const char* e = getenv("LAMP_FUSE_LOSS_TAIL"); return !(e && e[0] == '0'); }();
  const bool from_tail = input->op && input->op->params.size() == 1 && std::strcmp(input->op->name, "GlobalAvgPoolLogSoftMax") == 0 &&
                         input->value.h()->is_device() && input->op->params[0].first->needsGrad() && input->op->params[0].first->value.ndim() == 4;
  // ... or from the node that holds the last block too (batch_norm2_add_relu_pool_log_softmax_2d): its first input has the pooled tensor's shape
  const bool from_block_tail = input->op && (bool)input->op->pooled_input_grad && input->op->params.size() == 6 && input->value.h()->is_device() &&
                               input->op->params[0].first->value.ndim() == 4;
  // (bf16: the dtype whose batch-norm backward reads the plane values in place; in f32 / f64 the expanded view would be written out again)
  if (fuse_tail && fuse_ahead && (from_tail || from_block_tail) && reduction != 0 && input->value.dtype() == kBF16) {
    const Ten& xin = input->op->params[0].first->value;
    HCALL(lamp_nll_loss_forward_pooled_gradient_(&v, &tw, &pgh, input->value.h(), target.h(), weights.h(), reduction, ignore, acc.defined() ? acc.h() : nullptr, scale,
                                                 xin.size(2) * xin.size(3)));
  } else if (acc.defined()) HCALL(lamp_nll_loss_forward_accumulate_(&v, &tw, input->value.h(), target.h(), weights.h(), reduction, ignore, acc.h(), scale));
  else HCALL(lamp_nll_loss_forward(&v, &tw, input->value.h(), target.h(), weights.h(), reduction, ignore));
  Ten val(v), total_weight(tw), iv = input->value, plane_grad = pgh ? Ten(pgh) : Ten();
  op->params.push_back({input, [=](const Ten& p, Variable& out) {
    if (plane_grad.defined() && p.h() == tls_seed_ones && !tls_backprop_hooked && out.op && out.op->pooled_input_grad) {
      out.op->pooled_input_grad(plane_grad);                // the node of the last block + tail: its inputs' gradients at once
      return;
    }
    if (plane_grad.defined() && p.h() == tls_seed_ones && !tls_backprop_hooked && out.op && out.op->params.size() == 1 &&
        std::strcmp(out.op->name, "GlobalAvgPoolLogSoftMax") == 0) {
      // the derivative is the seed itself: the gradient computed ahead is the one the launch below would write, as an expanded view
      Variable& xin = *out.op->params[0].first;
      const std::vector<int64_t> xs = xin.value.shape();
      lamp_tensor *u2 = nullptr, *u3 = nullptr, *e = nullptr;
      HCALL(lamp_unsqueeze(&u2, plane_grad.h(), 2));
      const Ten t2(u2);
      HCALL(lamp_unsqueeze(&u3, t2.h(), 3));
      const Ten pv(u3);
      HCALL(lamp_expand(&e, pv.h(), xs.data(), 4));
      xin.accumulate(Ten(e), false);                        // (shared: nothing may add into a stride-0 view in place)
      return;
    }
    // Cnn.resnet's tail: the log-probabilities come from the pooled LogSoftMax node.  Its input gradient is linear in this contribution, so
    // the contribution goes straight to THAT node's input in one launch (the [N, C] gradient row is never written); anything else that
    // flows into `out` takes the node's own closure as before
    // (under a backprop with an `after` hook every visited variable must carry its derivative, as in autograd.scala: no short cut then)
    if (fuse_tail && !tls_backprop_hooked && out.op && out.op->params.size() == 1 && std::strcmp(out.op->name, "GlobalAvgPoolLogSoftMax") == 0 && p.h()->is_device() &&
        out.op->params[0].first->needsGrad()) {
      Variable& xin = *out.op->params[0].first;
      lamp_tensor* t = nullptr;
      HCALL(lamp_global_avg_pool_log_softmax_nll_backward(&t, p.h(), target.h(), weights.h(), reduction, ignore, total_weight.h(), iv.h(), xin.value.h()));
      xin.accumulate(Ten(t), true);
      return;
    }
    lamp_tensor* t = nullptr;
    HCALL(lamp_nll_loss_backward(&t, p.h(), iv.h(), target.h(), weights.h(), reduction, ignore, total_weight.h()));
    out.accumulate(Ten(t), true);
  }});
  return make_result(op, val);
}
Var mse_loss(const Var& input, const Ten& target, int64_t reduction) {
  LAMP_CHECK(input->value.numel() == target.numel(), "mse loss: input/target size mismatch");
  auto op = new_op("MseLoss");
  Ten tv = ops::view(target, input->shape()), iv = input->value;
  op->params.push_back({input, [=](const Ten& p, Variable& out) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_mse_loss_backward(&t, p.h(), iv.h(), tv.h(), reduction));
    out.accumulate(Ten(t), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_mse_loss(&o, iv.h(), tv.h(), reduction));
  return make_result(op, Ten(o));
}

// ---- index / distance (ops.scala:179-197, 725-786) ----------------------------------------------
Var index_select(const Var& input, int64_t dim, const Var& index) {
  auto op = new_op("IndexSelect");
  Ten idx = index->value;
  op->params.push_back({input, [dim, idx](const Ten& p, Variable& out) {
    // val tmp = out.indexAdd(dim, index, p); out += tmp      => out = 2 * out + scatter(p)   (sic)
    if (!out.has_grad()) {
      out.accumulate(ops::index_add(ops::zeros_like(out.value), dim, idx, p), true);
    } else {
      Ten tmp = ops::index_add(out.grad, dim, idx, p);
      out.accumulate(tmp, true);
    }
  }});
  return make_result(op, ops::index_select(input->value, dim, idx));
}
Var mask_fill(const Var& input, const Ten& mask, double fill) {   // ops.scala:148-159
  auto op = new_op("MaskFill");
  op->params.push_back({input, [mask](const Ten& p, Variable& out) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_masked_fill(&t, p.h(), mask.h(), 0.0));
    out.accumulate(Ten(t), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_masked_fill(&o, input->value.h(), mask.h(), fill));
  return make_result(op, Ten(o));
}
Var euclidean_distance(const Var& a, const Var& b, int64_t dim) {
  auto op = new_op("EuclideanDistance");
  Ten diff = ops::sub(a->value, b->value);
  Ten norm = ops::norm2_dims(diff, {dim}, true);
  op->params.push_back({a, [diff, norm](const Ten& p, Variable& out) { out.addcmul(p, ops::div(diff, norm), 1.0); }});
  op->params.push_back({b, [diff, norm](const Ten& p, Variable& out) { out.addcmul(p, ops::div(diff, norm), -1.0); }});
  return make_result(op, norm);
}
Var capped_shifted_negative_exponential(const Var& a, double shift) {
  auto op = new_op("CappedShiftedNegativeExponential");
  const int dt = a->value.dtype(), dev = a->value.device();
  Ten pred = ops::le_scalar(a->value, shift);
  Ten ones = ops::ones({1}, dt, dev);
  Ten above = ops::sub(ops::scalar(shift, dt, dev), a->value);
  HCALL(lamp_exp_(above.h()));
  Ten result = ops::where(pred, ones, above);
  op->params.push_back({a, [=](const Ten& p, Variable& out) {
    Ten zeros = ops::zeros({1}, dt, dev);
    Ten nonzeros = ops::mul_scalar(result, -1.0);
    out.addcmul(p, ops::where(pred, zeros, nonzeros), 1.0);
  }});
  return make_result(op, result);
}

// ScaledDotProductAttention (ops.scala:2342-2390): joinedBackward - one backward call yields the three gradients.  Here the first
// closure that runs makes the call and parks the other two results for its siblings (same p), as batch norm does above.
// Self-attention's three projections and the fused attention as ONE node (round 3; MultiheadAttention with query, keys and values the
// same Variable, Transformer.scala:889-962): q | k | v = x . [Wq | Wk | Wv] is one product (24576 x 768 x 2304 in the language model:
// 3.4 rounds of the 256 x 256 kernel instead of 3 x 2), the attention kernels read the three column blocks in place, and the backward
// gets dq | dk | dv as the column blocks of one buffer (lamp_scaled_dot_product_attention_backward on packed operands), so that
// dX = [dq | dk | dv] . [Wq | Wk | Wv]^T and d[Wq | Wk | Wv] = x^T . [dq | dk | dv] are one product each.  Values: the projections are
// bitwise those of three products; dX is rounded once instead of three times.  Returns (batch, sequence, heads x d).
static Ten dense_copy_if_needed(const Ten& t) {
  if (t.h()->is_contiguous()) return t;
  lamp_tensor* o = nullptr;
  HCALL(lamp_contiguous(&o, t.h()));
  return Ten(o);
}
Var packed_self_attention(const Var& x, const Var& wq, const Var& wk, const Var& wv, int64_t numHeads, bool isCausal) {
  auto op = new_op("PackedSelfAttention");
  const Ten xv = x->value;
  const int64_t nB = xv.size(0), nS = xv.size(1), in = xv.size(2), HD = wq->value.size(1), D = HD / numHeads;
  LAMP_CHECK(wk->value.shape() == wq->value.shape() && wv->value.shape() == wq->value.shape() && wq->value.size(0) == in && HD % numHeads == 0,
             "packed_self_attention: the three projections must have one shape");
  const Ten x2 = ops::reshape(xv, {nB * nS, in});
  const Ten wcat = ops::cat({wq->value, wk->value, wv->value}, 1);                 // [in, 3 HD]
  const Ten qkv = ops::mm(x2, wcat);                                                // [tokens, 3 HD]
  auto block = [&](const Ten& base, int64_t which) {                                // (B, heads, S, d) view of column block `which`
    const int64_t sz[4] = {nB, numHeads, nS, D}, st[4] = {nS * 3 * HD, D, 3 * HD, 1};
    lamp_tensor* o = nullptr;
    HCALL(lamp_as_strided(&o, base.h(), sz, st, 4, base.h()->offset + which * HD));
    return Ten(o);
  };
  const Ten q4 = block(qkv, 0), k4 = block(qkv, 1), v4 = block(qkv, 2);
  lamp_tensor *o = nullptr, *l = nullptr;
  HCALL(lamp_scaled_dot_product_attention_bias(&o, &l, q4.h(), k4.h(), v4.h(), nullptr, isCausal, 0.0));
  const Ten out(o), lse(l);
  const Ten value = ops::flatten(ops::transpose(out, 1, 2), 2, 3);                  // (B, S, heads x d): a view when the kernel wrote (B, S, heads, d) storage
  struct Cache { Ten dqkv, dw[3], p; };
  auto cache = std::make_shared<Cache>();
  auto ensure = [=](const Ten& p) {
    if (cache->p.defined() && cache->p.h() == p.h()) return;
    const Ten g4 = ops::transpose(ops::view(dense_copy_if_needed(p), {nB, nS, numHeads, D}), 1, 2);
    lamp_tensor* r3[3] = {nullptr, nullptr, nullptr};
    HCALL(lamp_scaled_dot_product_attention_bias_backward(r3, g4.h(), q4.h(), k4.h(), v4.h(), out.h(), lse.h(), nullptr, isCausal, 0.0));
    const Ten dq(r3[0]), dk(r3[1]), dv(r3[2]);
    const int64_t sz[2] = {nB * nS, 3 * HD}, st[2] = {3 * HD, 1};
    if (dq.h()->st == dk.h()->st && dk.h()->st == dv.h()->st && dq.h()->strides[2] == 3 * HD && dk.h()->offset == dq.h()->offset + HD) {
      lamp_tensor* packed = nullptr;                                                // the kernels wrote the three column blocks of one buffer
      HCALL(lamp_as_strided(&packed, dq.h(), sz, st, 2, dq.h()->offset));
      cache->dqkv = Ten(packed);
    } else {                                                                        // composed attention (other dtypes / head widths): assemble
      auto flat = [&](const Ten& t) { return ops::reshape(ops::transpose(t, 1, 2), {nB * nS, HD}); };
      cache->dqkv = ops::cat({flat(dq), flat(dk), flat(dv)}, 1);
    }
    lamp_tensor* dw = nullptr;
    const int64_t ws[2] = {in, 3 * HD};
    HCALL(lamp_empty(&dw, ws, 2, xv.dtype(), xv.device()));
    const Ten dwcat(dw);
    HCALL(lamp_addmm_out_transposed1(dw, dw, x2.h(), cache->dqkv.h(), 0.0, 1.0));     // d[Wq | Wk | Wv] = x^T . [dq | dk | dv]
    lamp_tensor* parts[3] = {nullptr, nullptr, nullptr};
    HCALL(lamp_chunk_contiguous(parts, dw, 3, 1));                                    // the three weights' gradients, dense, in one launch
    for (int i = 0; i < 3; i++) cache->dw[i] = Ten(parts[i]);
    cache->p = p;
  };
  op->params.push_back({x, [=](const Ten& p, Variable& o_) {                          // dX += [dq | dk | dv] . [Wq | Wk | Wv]^T
    ensure(p);
    gemm_accumulate(o_, [&](const Ten& g, double beta) {
      const Ten g2 = ops::view(g, {nB * nS, in});
      HCALL(lamp_addmm_out_transposed2(g2.h(), g2.h(), cache->dqkv.h(), wcat.h(), beta, 1.0));
    });
  }});
  auto wback = [=](int which) {
    return [=](const Ten& p, Variable& o_) {
      ensure(p);
      o_.accumulate(cache->dw[which], true);
      cache->dw[which] = Ten();
      if (which == 2) { cache->dqkv = Ten(); cache->p = Ten(); }
    };
  };
  op->params.push_back({wq, wback(0)});
  op->params.push_back({wk, wback(1)});
  op->params.push_back({wv, wback(2)});
  return make_result(op, value);
}
Var scaled_dot_product_attention(const Var& query, const Var& key, const Var& value, bool isCausal, const Ten& attentionBias) {
  auto op = new_op("ScaledDotProductAttention");
  lamp_tensor *o = nullptr, *l = nullptr;
  lamp_tensor* const bias = attentionBias.defined() ? attentionBias.h() : nullptr;      // Option[STen]: no gradient flows into it
  HCALL(lamp_scaled_dot_product_attention_bias(&o, &l, query->value.h(), key->value.h(), value->value.h(), bias, isCausal, 0.0));
  Ten out(o), lse(l);
  struct Cache { Ten g[3]; Ten p; };
  auto cache = std::make_shared<Cache>();
  const Ten qv = query->value, kv = key->value, vv = value->value;
  auto back = [=](int which) {
    return [=](const Ten& p, Variable& acc) {
      if (!(cache->p.defined() && cache->p.h() == p.h() && cache->g[which].defined())) {
        lamp_tensor* r3[3] = {nullptr, nullptr, nullptr};
        HCALL(lamp_scaled_dot_product_attention_bias_backward(r3, p.h(), qv.h(), kv.h(), vv.h(), out.h(), lse.h(), attentionBias.defined() ? attentionBias.h() : nullptr,
                                                              isCausal, 0.0));
        for (int i = 0; i < 3; i++) cache->g[i] = Ten(r3[i]);
        cache->p = p;
      }
      acc.accumulate(cache->g[which], true);
      cache->g[which] = Ten();
      if (!cache->g[0].defined() && !cache->g[1].defined() && !cache->g[2].defined()) cache->p = Ten();
    };
  };
  op->params.push_back({query, back(0)});
  op->params.push_back({key, back(1)});
  op->params.push_back({value, back(2)});
  return make_result(op, out);
}

// ---- convolution / pooling (ops.scala:1547-1825) ------------------------------------------------
// the Convolution node (ops.scala:1547-1651); `computed`, when given, is the forward value a fused launch has already produced
// The two Convolution nodes of convolution_pair share this: the first of the two input-gradient closures to run leaves its incoming
// derivative here (and a fallback on the input variable), the second one computes both contributions with lamp_convolution_backward_input_pair.
struct ConvPairGrad {
  struct Side { Ten w; std::vector<int64_t> stride, padding, dilation; Ten p; Ten pw; Variable* wout = nullptr; };
  Side side[2];
  int64_t groups = 1;
  int waiting = -1;                            // which side's derivative is held for the input gradient
  int wwaiting = -1;                           // ... and for the weight gradients (pw, wout: the derivative and the parameter it belongs to)
};
static Var convolution_node(const Var& input, const Var& weight, const Var& bias, const std::vector<int64_t>& stride,
                            const std::vector<int64_t>& padding, const std::vector<int64_t>& dilation, bool transposed,
                            const std::vector<int64_t>& outputPadding, int64_t groups, const Ten* computed,
                            const std::shared_ptr<ConvPairGrad>& pair = nullptr, int pair_side = 0) {
  auto op = new_op("Convolution");
  const int ns = (int)stride.size();
  Ten iv = input->value, wv = weight->value;
  static const bool fuse_accumulate = [] { const char* e = getenv("LAMP_CONV_DGRAD_ACCUMULATE"); return !(e && e[0] == '0'); }();
  static const bool fuse_pair = [] { const char* e = getenv("LAMP_CONV_DGRAD_PAIR"); return !(e && e[0] == '0'); }();
  auto back = [=](int which) {
    // (a plain function object: the pair's deferred form calls it again)
    std::function<void(const Ten&, Variable&)> single;
    single = [=](const Ten& p, Variable& out) {
      if (which == 0 && !transposed && fuse_accumulate && out.has_grad() && out.grad.h()->is_device() && p.h()->is_device() && out.grad.dtype() == p.dtype()) {
        // the input already holds another consumer's contribution (a residual block): `out += dgrad` inside the dgrad kernel
        lamp_tensor* r = nullptr;
        HCALL(lamp_convolution_backward_input_add(&r, p.h(), iv.h(), wv.h(), stride.data(), padding.data(), dilation.data(), ns,
                                                  outputPadding.data(), groups, out.grad.h()));
        out.grad = Ten(r);
        out.grad_shared = false;
        return;
      }
      lamp_tensor* o3[3] = {nullptr, nullptr, nullptr};
      uint8_t mask[3] = {(uint8_t)(which == 0), (uint8_t)(which == 1), (uint8_t)(which == 2)};
      HCALL(lamp_convolution_backward(o3, p.h(), iv.h(), wv.h(), stride.data(), padding.data(), dilation.data(), ns, transposed,
                                      outputPadding.data(), groups, mask));
      out.accumulate(Ten(o3[which]), true);
    };
    static const bool fuse_wpair = [] { const char* e = getenv("LAMP_CONV_WGRAD_PAIR"); return !(e && e[0] == '0'); }();
    if (which == 1 && pair && fuse_wpair && !transposed) {
      // the two weight gradients from one call (x is staged once).  The fallback sits on the INPUT variable, which backprop reaches after
      // both convolutions (a parameter is reached right after its own node, before the sibling has run)
      Variable* inp = input.get();
      return Backward([=](const Ten& p, Variable& out) {
        // only a LEAF filter's gradient may arrive late: a filter computed by other nodes (weight norm) is visited right after this node
        // and its own closures must find the complete derivative
        if (!p.h()->is_device() || tls_backprop_hooked || out.op) { single(p, out); return; }
        ConvPairGrad& st = *pair;
        if (st.wwaiting < 0) {
          st.wwaiting = pair_side;
          st.side[pair_side].pw = p;
          st.side[pair_side].wout = &out;
          inp->pending.push_back([=] {
            ConvPairGrad& s2 = *pair;
            if (s2.wwaiting != pair_side) return;
            const Ten held = s2.side[pair_side].pw;
            Variable* wo = s2.side[pair_side].wout;
            s2.side[pair_side].pw = Ten(); s2.side[pair_side].wout = nullptr; s2.wwaiting = -1;
            single(held, *wo);
          });
          return;
        }
        if (st.wwaiting == pair_side) { single(p, out); return; }
        const int other = st.wwaiting;
        const Ten po = st.side[other].pw;
        Variable* wo = st.side[other].wout;
        st.side[other].pw = Ten(); st.side[other].wout = nullptr; st.wwaiting = -1;
        const Ten& pa = pair_side == 0 ? p : po;
        const Ten& pb = pair_side == 0 ? po : p;
        lamp_tensor* o2[2] = {nullptr, nullptr};
        HCALL(lamp_convolution_backward_weight_pair(o2, iv.h(), pa.h(), st.side[0].w.h(), st.side[0].stride.data(), st.side[0].padding.data(),
                                                    st.side[0].dilation.data(), pb.h(), st.side[1].w.h(), st.side[1].stride.data(),
                                                    st.side[1].padding.data(), st.side[1].dilation.data(), ns, st.groups));
        const Ten da(o2[0]), db(o2[1]);
        (pair_side == 0 ? out : *wo).accumulate(da, true);
        (pair_side == 0 ? *wo : out).accumulate(db, true);
      });
    }
    if (which != 0 || !pair || !fuse_pair || transposed) return Backward(single);
    return Backward([=](const Ten& p, Variable& out) {
      if (!(p.h()->is_device() && fuse_accumulate)) { single(p, out); return; }
      ConvPairGrad& st = *pair;
      if (st.waiting < 0) {
        // first of the two: hold the derivative; should the sibling's never arrive, the input's own turn in backprop computes this one alone
        st.waiting = pair_side;
        st.side[pair_side].p = p;
        Variable* outp = &out;
        out.pending.push_back([=] {
          ConvPairGrad& s2 = *pair;
          if (s2.waiting != pair_side) return;
          const Ten held = s2.side[pair_side].p;
          s2.side[pair_side].p = Ten(); s2.waiting = -1;
          single(held, *outp);
        });
        return;
      }
      if (st.waiting == pair_side) { single(p, out); return; }                       // (a second pass over the same node: not a pair)
      const int other = st.waiting;
      const Ten po = st.side[other].p;
      st.side[other].p = Ten(); st.waiting = -1;             // (its fallback on `out` finds nothing waiting)
      const Ten& pa = pair_side == 0 ? p : po;
      const Ten& pb = pair_side == 0 ? po : p;
      const bool have = out.has_grad() && out.grad.h()->is_device() && out.grad.dtype() == p.dtype();
      if (out.has_grad() && !have) { single(po, out); single(p, out); return; }
      lamp_tensor* r = nullptr;
      HCALL(lamp_convolution_backward_input_pair(&r, iv.h(), pa.h(), st.side[0].w.h(), st.side[0].stride.data(), st.side[0].padding.data(),
                                                 st.side[0].dilation.data(), pb.h(), st.side[1].w.h(), st.side[1].stride.data(),
                                                 st.side[1].padding.data(), st.side[1].dilation.data(), ns, st.groups, have ? out.grad.h() : nullptr));
      out.grad = Ten(r);
      out.grad_shared = false;
    });
  };
  op->params.push_back({input, back(0)});
  op->params.push_back({weight, back(1)});
  op->params.push_back({bias, back(2)});
  if (pair) op->reset = [pair] {
    pair->waiting = pair->wwaiting = -1;
    for (auto& sd : pair->side) { sd.p = Ten(); sd.pw = Ten(); sd.wout = nullptr; }
  };
  if (computed) return make_result(op, *computed);
  lamp_tensor* o = nullptr;
  HCALL(lamp_convolution(&o, iv.h(), wv.h(), bias->value.h(), stride.data(), padding.data(), dilation.data(), ns, transposed,
                         outputPadding.data(), groups));
  return make_result(op, Ten(o));
}
Var convolution(const Var& input, const Var& weight, const Var& bias, const std::vector<int64_t>& stride,
                const std::vector<int64_t>& padding, const std::vector<int64_t>& dilation, bool transposed,
                const std::vector<int64_t>& outputPadding, int64_t groups) {
  return convolution_node(input, weight, bias, stride, padding, dilation, transposed, outputPadding, groups, nullptr);
}
// Two Convolution nodes on ONE input (the two branches of Residual, cnn.scala:16-20) whose forward values come from one call: the graph, the
// closures and every value are those of two `convolution` calls
std::pair<Var, Var> convolution_pair(const Var& input, const Var& weight_a, const Var& bias_a, const std::vector<int64_t>& stride_a,
                                     const std::vector<int64_t>& padding_a, const std::vector<int64_t>& dilation_a, const Var& weight_b,
                                     const Var& bias_b, const std::vector<int64_t>& stride_b, const std::vector<int64_t>& padding_b,
                                     const std::vector<int64_t>& dilation_b, int64_t groups) {
  const int ns = (int)stride_a.size();
  LAMP_CHECK(ns == (int)stride_b.size(), "convolution_pair: the two convolutions have different numbers of spatial dimensions");
  lamp_tensor* o2[2] = {nullptr, nullptr};
  HCALL(lamp_convolution_pair(o2, input->value.h(), weight_a->value.h(), bias_a->value.h(), stride_a.data(), padding_a.data(), dilation_a.data(),
                              weight_b->value.h(), bias_b->value.h(), stride_b.data(), padding_b.data(), dilation_b.data(), ns, groups));
  const Ten ya(o2[0]), yb(o2[1]);
  const std::vector<int64_t> zero(ns, 0);
  auto pg = std::make_shared<ConvPairGrad>();
  pg->side[0] = {weight_a->value, stride_a, padding_a, dilation_a, Ten()};
  pg->side[1] = {weight_b->value, stride_b, padding_b, dilation_b, Ten()};
  pg->groups = groups;
  Var a = convolution_node(input, weight_a, bias_a, stride_a, padding_a, dilation_a, false, zero, groups, &ya, pg, 0);
  Var b = convolution_node(input, weight_b, bias_b, stride_b, padding_b, dilation_b, false, zero, groups, &yb, pg, 1);
  return {a, b};
}
Var avg_pool2d(const Var& input, int64_t k, int64_t stride, int64_t padding) {
  LAMP_CHECK(input->value.ndim() == 4, "Input dimensions must be 4");
  auto op = new_op("AvgPool2D");
  Ten iv = input->value;
  op->params.push_back({input, [=](const Ten& p, Variable& out) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_avg_pool2d_backward(&t, p.h(), iv.h(), k, stride, padding, 0, 1));
    out.accumulate(Ten(t), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_avg_pool2d(&o, iv.h(), k, stride, padding, 0, 1));
  return make_result(op, Ten(o));
}
// AvgPool2D over the whole map -> Flatten -> LogSoftMax as one node (the closures of the three, composed)
Var global_avg_pool_log_softmax(const Var& input) {
  LAMP_CHECK(input->value.ndim() == 4, "Input dimensions must be 4");
  auto op = new_op("GlobalAvgPoolLogSoftMax");
  Ten iv = input->value;
  lamp_tensor* o = nullptr;
  HCALL(lamp_global_avg_pool_log_softmax(&o, iv.h()));
  Ten val(o);
  op->params.push_back({input, [iv, val](const Ten& p, Variable& out) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_global_avg_pool_log_softmax_backward(&t, p.h(), val.h(), iv.h()));
    out.accumulate(Ten(t), true);
  }});
  return make_result(op, val);
}
Var max_pool2d(const Var& input, int64_t k, int64_t stride, int64_t padding, int64_t dilation) {
  LAMP_CHECK(input->value.ndim() == 4, "Input dimensions must be 4");
  auto op = new_op("MaxPool2D");
  Ten iv = input->value;
  lamp_tensor *o = nullptr, *m = nullptr;
  HCALL(lamp_max_pool2d_with_indices(&o, &m, iv.h(), k, stride, padding, dilation, 0));
  Ten mask(m);
  op->params.push_back({input, [=](const Ten& p, Variable& out) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_max_pool2d_with_indices_backward(&t, p.h(), iv.h(), k, stride, padding, dilation, 0, mask.h()));
    out.accumulate(Ten(t), true);
  }});
  return make_result(op, Ten(o));
}

// ---- normalisation (ops.scala:1846-2140) --------------------------------------------------------
static Var batch_norm_impl(const char* name, const Var& input, const Ten& x, const Var& weight, const Var& bias, const Ten& runningMean,
                           const Ten& runningVar, bool training, double momentum, double eps, bool two_d, bool relu = false) {
  auto op = new_op(name);
  const std::vector<int64_t> expected = {x.size(1)};
  LAMP_CHECK(weight->shape() == expected, "Expected [" << expected[0] << "] got weight shape of " << weight->value.h()->describe());
  LAMP_CHECK(bias->shape() == expected, "Expected [" << expected[0] << "] got bias shape of " << bias->value.h()->describe());
  LAMP_CHECK(runningMean.shape() == expected && runningVar.shape() == expected, "running statistics have the wrong shape");
  lamp_tensor* o3[3] = {nullptr, nullptr, nullptr};
  if (relu) HCALL(lamp_native_batch_norm_relu(o3, x.h(), weight->value.h(), bias->value.h(), runningMean.h(), runningVar.h(), training, momentum, eps));
  else HCALL(lamp_native_batch_norm(o3, x.h(), weight->value.h(), bias->value.h(), runningMean.h(), runningVar.h(), training, momentum, eps));
  Ten out(o3[0]), saveMean(o3[1]), saveInvstd(o3[2]), wv = weight->value, bv = bias->value;
  // one backward call; with relu fused the gradient mask is recomputed from x inside the kernels
  auto bn_backward = [=](lamp_tensor* r3[3], const Ten& fp, const uint8_t mask[3]) {
    if (relu) HCALL(lamp_native_batch_norm_relu_backward(r3, fp.h(), x.h(), wv.h(), bv.h(), runningMean.h(), runningVar.h(), saveMean.h(), saveInvstd.h(),
                                                         training, eps, mask));
    else HCALL(lamp_native_batch_norm_backward(r3, fp.h(), x.h(), wv.h(), runningMean.h(), runningVar.h(), saveMean.h(), saveInvstd.h(), training, eps, mask));
  };
  // The reference calls native_batch_norm_backward once per requested derivative (ops.scala:1901,1924,
  // 2086,2107) and sums p again for the bias; all three reduce the same per-channel sums.  Here the input
  // closure (it runs first) asks for dx, dweight AND dbias in one call and parks the latter two for the
  // sibling closures - identical arithmetic, one reduction pass over (dy, x) instead of three.
  struct Cache { Ten dweight, dbias, p; };                 // results parked for the sibling closures + the gradient they belong to
  auto cache = std::make_shared<Cache>();
  const bool want_w = weight->needsGrad(), want_b = bias->needsGrad();
  auto back = [=](int which) {
    return [=](const Ten& p, Variable& o) {
      if (which == 1 && cache->dweight.defined() && cache->p.h() == p.h()) {
        o.accumulate(ops::reshape(cache->dweight, o.shape()), true);
        cache->dweight = Ten();
        return;
      }
      Ten fp = two_d ? p : ops::flatten(p, 1, p.ndim() - 1);
      lamp_tensor* r3[3] = {nullptr, nullptr, nullptr};
      const bool first = which == 0;   // the input closure runs first and shares its reduction pass with weight and bias
      uint8_t mask[3] = {(uint8_t)(which == 0), (uint8_t)(which == 1 || (first && want_w)), (uint8_t)(first && want_b)};
      bn_backward(r3, fp, mask);
      Ten r0(r3[0]), r1(r3[1]), r2(r3[2]);
      if (first) { cache->dweight = r1; cache->dbias = r2; cache->p = p; }
      o.accumulate(ops::reshape(which == 0 ? r0 : r1, o.shape()), true);
    };
  };
  op->params.push_back({input, back(0)});
  op->params.push_back({weight, back(1)});
  // bias gradient = sum of p over every dim but the channel one (ops.scala:1944-1953, 2126-2138: unbroadcast + view);
  // that sum is also a by-product of native_batch_norm_backward's reduction, so it is taken from there when available
  op->params.push_back({bias, [=](const Ten& p, Variable& o) {
    if (cache->dbias.defined() && cache->p.h() == p.h()) {
      o.accumulate(ops::reshape(cache->dbias, o.shape()), true);
      cache->dbias = Ten(); cache->p = Ten();
      return;
    }
    if (relu) {                                              // the bias gradient is the sum of the MASKED p
      lamp_tensor* r3[3] = {nullptr, nullptr, nullptr};
      const uint8_t mask[3] = {0, 0, 1};
      bn_backward(r3, p, mask);
      Ten r0(r3[0]), r1(r3[1]), r2(r3[2]);
      o.accumulate(ops::reshape(r2, o.shape()), true);
      return;
    }
    if (two_d) {
      std::vector<int64_t> tgt = o.shape();
      for (int i = 0; i < p.ndim() - 2; i++) tgt.push_back(1);
      o.accumulate(ops::reshape(ops::unbroadcast(p, tgt), o.shape()), p.shape() != tgt);   // equal shapes: a view of p, not ours to modify
    } else {
      Ten fp = ops::flatten(p, 1, p.ndim() - 1);
      o.accumulate(ops::unbroadcast(fp, o.shape()), fp.shape() != o.shape());
    }
  }});
  return make_result(op, two_d ? out : ops::reshape(out, input->shape()));
}
Var batch_norm(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, bool training,
               double momentum, double eps) {
  Ten x = ops::flatten(input->value, 1, input->value.ndim() - 1);
  return batch_norm_impl("BatchNorm", input, x, weight, bias, runningMean, runningVar, training, momentum, eps, false);
}
Var batch_norm_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, bool training,
                  double momentum, double eps) {
  LAMP_CHECK(input->value.ndim() >= 3, "Expected 3D or 4D tensor");
  return batch_norm_impl("BatchNorm2D", input, input->value, weight, bias, runningMean, runningVar, training, momentum, eps, true);
}
// relu(batch_norm_2d(x)) as ONE op (the BatchNorm2D -> Fun(relu) pair of cnn.scala:36-40): identical values, the relu and
// relu-backward passes folded into the normalisation kernels.  Maps of fewer than 64 elements take the unfused pair.
bool batch_norm_relu_2d_supported(const Var& input) {
  const int nd = input->value.ndim();
  if (nd < 3) return false;
  int64_t hw = 1;
  for (int d = 2; d < nd; d++) hw *= input->value.size(d);
  return hw >= 64;
}
Var batch_norm_relu_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, bool training,
                       double momentum, double eps) {
  LAMP_CHECK(input->value.ndim() >= 3, "Expected 3D or 4D tensor");
  return batch_norm_impl("BatchNorm2DRelu", input, input->value, weight, bias, runningMean, runningVar, training, momentum, eps, true, true);
}
// relu(batch_norm_2d(x) + addend) as ONE op: the tail of lamp's residual block (cnn.scala:11-21: `right(x) + left(x)` where
// right ends in BatchNorm2D, followed by Fun(relu) in Cnn.residual, cnn.scala:36-46).  Values are those of the three ops
// (every intermediate is rounded as they round it); the backward hands the relu-masked gradient to the addend and the batch
// norm gradients to x / weight / bias from one fused pass.
Var batch_norm_add_relu_2d(const Var& input, const Var& addend, const Var& weight, const Var& bias, const Ten& runningMean,
                           const Ten& runningVar, bool training, double momentum, double eps) {
  auto op = new_op("BatchNorm2DAddRelu");
  const Ten x = input->value, av = addend->value, wv = weight->value, bv = bias->value;
  const std::vector<int64_t> expected = {x.size(1)};
  LAMP_CHECK(weight->shape() == expected && bias->shape() == expected, "batch norm weight / bias have the wrong shape");
  LAMP_CHECK(runningMean.shape() == expected && runningVar.shape() == expected, "running statistics have the wrong shape");
  lamp_tensor* o3[3] = {nullptr, nullptr, nullptr};
  HCALL(lamp_native_batch_norm_add_relu(o3, x.h(), av.h(), wv.h(), bv.h(), runningMean.h(), runningVar.h(), training, momentum, eps));
  Ten out(o3[0]), saveMean(o3[1]), saveInvstd(o3[2]);
  struct Cache { Ten g[4]; Ten p; };                       // dx, dweight, dbias, daddend of one backward call, and the p they belong to
  auto cache = std::make_shared<Cache>();
  const bool want[4] = {input->needsGrad(), weight->needsGrad(), bias->needsGrad(), addend->needsGrad()};
  auto back = [=](int which) {
    return [=](const Ten& p, Variable& o) {
      if (!(cache->p.defined() && cache->p.h() == p.h() && cache->g[which].defined())) {
        // first closure of this backward pass: one call for every gradient that will be asked for
        uint8_t mask[4] = {(uint8_t)want[0], (uint8_t)want[1], (uint8_t)want[2], (uint8_t)want[3]};
        mask[which] = 1;
        lamp_tensor* r4[4] = {nullptr, nullptr, nullptr, nullptr};
        HCALL(lamp_native_batch_norm_add_relu_backward(r4, p.h(), x.h(), av.h(), wv.h(), bv.h(), runningMean.h(), runningVar.h(), saveMean.h(),
                                                       saveInvstd.h(), training, eps, mask));
        for (int i = 0; i < 4; i++) cache->g[i] = r4[i] ? Ten(r4[i]) : Ten();
        cache->p = p;
      }
      Ten g = cache->g[which];
      cache->g[which] = Ten();
      bool any = false;
      for (int i = 0; i < 4; i++) any = any || cache->g[i].defined();
      if (!any) cache->p = Ten();
      o.accumulate(ops::reshape(g, o.shape()), true);
    };
  };
  op->params.push_back({input, back(0)});
  op->params.push_back({weight, back(1)});
  op->params.push_back({bias, back(2)});
  op->params.push_back({addend, back(3)});
  return make_result(op, out);
}
// relu(batch_norm_2d(x) + batch_norm_2d(x2)) as ONE op: the tail of lamp's residual block when the LEFT branch ends in a batch norm too
// (Conv2D 1x1 -> BatchNorm2D, cnn.scala:62-78; every block of Cnn.resnet).  Forward values are bitwise those of the chain (left batch
// norm, then batch_norm_add_relu_2d); neither the left branch's normalised output nor the relu-masked gradient is ever written
// (lamp_native_batch_norm2_add_relu).  One backward call serves the six closures.
Var batch_norm2_add_relu_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, double momentum, double eps,
                            const Var& input2, const Var& weight2, const Var& bias2, const Ten& runningMean2, const Ten& runningVar2, double momentum2,
                            double eps2) {
  auto op = new_op("BatchNorm2DPairAddRelu");
  const Ten x = input->value, wv = weight->value, bv = bias->value, x2 = input2->value, wv2 = weight2->value, bv2 = bias2->value;
  const std::vector<int64_t> expected = {x.size(1)};
  LAMP_CHECK(weight->shape() == expected && bias->shape() == expected && weight2->shape() == expected && bias2->shape() == expected,
             "batch norm weight / bias have the wrong shape");
  LAMP_CHECK(runningMean.shape() == expected && runningVar.shape() == expected && runningMean2.shape() == expected && runningVar2.shape() == expected,
             "running statistics have the wrong shape");
  lamp_tensor* o5[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  HCALL(lamp_native_batch_norm2_add_relu(o5, x.h(), wv.h(), bv.h(), runningMean.h(), runningVar.h(), x2.h(), wv2.h(), bv2.h(), runningMean2.h(),
                                         runningVar2.h(), momentum, momentum2, eps, eps2));
  Ten out(o5[0]), saveMean(o5[1]), saveInvstd(o5[2]), saveMean2(o5[3]), saveInvstd2(o5[4]);
  struct Cache { Ten g[6]; Ten p; };                       // dx, dweight, dbias, dx2, dweight2, dbias2 of one backward call, and the p they belong to
  auto cache = std::make_shared<Cache>();
  const bool want[6] = {input->needsGrad(), weight->needsGrad(), bias->needsGrad(), input2->needsGrad(), weight2->needsGrad(), bias2->needsGrad()};
  auto back = [=](int which) {
    return [=](const Ten& p, Variable& o) {
      if (!(cache->p.defined() && cache->p.h() == p.h() && cache->g[which].defined())) {
        uint8_t mask[6];
        for (int i = 0; i < 6; i++) mask[i] = (uint8_t)want[i];
        mask[which] = 1;
        lamp_tensor* r6[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        HCALL(lamp_native_batch_norm2_add_relu_backward(r6, p.h(), x.h(), wv.h(), bv.h(), saveMean.h(), saveInvstd.h(), x2.h(), wv2.h(), bv2.h(),
                                                        saveMean2.h(), saveInvstd2.h(), eps, eps2, mask));
        for (int i = 0; i < 6; i++) cache->g[i] = r6[i] ? Ten(r6[i]) : Ten();
        cache->p = p;
      }
      Ten g = cache->g[which];
      cache->g[which] = Ten();
      bool any = false;
      for (int i = 0; i < 6; i++) any = any || cache->g[i].defined();
      if (!any) cache->p = Ten();
      o.accumulate(ops::reshape(g, o.shape()), true);
    };
  };
  op->params.push_back({input, back(0)});
  op->params.push_back({weight, back(1)});
  op->params.push_back({bias, back(2)});
  op->params.push_back({input2, back(3)});
  op->params.push_back({weight2, back(4)});
  op->params.push_back({bias2, back(5)});
  return make_result(op, out);
}
// ... and, as the LAST block of Cnn.resnet, with the network's tail behind it: LogSoftMax(Flatten(AvgPool2D(relu(bn(x) + bn2(x2))))) as ONE node
// (cnn.scala:129-136; lamp_native_batch_norm2_add_relu_pool_log_softmax).  The block's output y has one reader, the pool: it is never written.
// Backward: y's gradient is one value per plane - the pooled LogSoftMax's input gradient, (p - exp(out) sum p) / HW with the roundings of
// lamp_global_avg_pool_log_softmax_backward - handed to lamp_native_batch_norm2_add_relu_backward as a view expanded over the map.
static Ten expand_planes(const Ten& plane_grad, const std::vector<int64_t>& xs) {
  lamp_tensor *u2 = nullptr, *u3 = nullptr, *e = nullptr;
  HCALL(lamp_unsqueeze(&u2, plane_grad.h(), 2));
  const Ten t2(u2);
  HCALL(lamp_unsqueeze(&u3, t2.h(), 3));
  const Ten pv(u3);
  HCALL(lamp_expand(&e, pv.h(), xs.data(), 4));
  return Ten(e);
}
Var batch_norm2_add_relu_pool_log_softmax_2d(const Var& input, const Var& weight, const Var& bias, const Ten& runningMean, const Ten& runningVar, double momentum,
                                             double eps, const Var& input2, const Var& weight2, const Var& bias2, const Ten& runningMean2,
                                             const Ten& runningVar2, double momentum2, double eps2) {
  auto op = new_op("BatchNorm2DPairAddReluPoolLogSoftMax");
  const Ten x = input->value, wv = weight->value, bv = bias->value, x2 = input2->value, wv2 = weight2->value, bv2 = bias2->value;
  const std::vector<int64_t> expected = {x.size(1)};
  LAMP_CHECK(x.ndim() == 4, "Input dimensions must be 4");
  LAMP_CHECK(weight->shape() == expected && bias->shape() == expected && weight2->shape() == expected && bias2->shape() == expected,
             "batch norm weight / bias have the wrong shape");
  LAMP_CHECK(runningMean.shape() == expected && runningVar.shape() == expected && runningMean2.shape() == expected && runningVar2.shape() == expected,
             "running statistics have the wrong shape");
  lamp_tensor* o5[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  HCALL(lamp_native_batch_norm2_add_relu_pool_log_softmax(o5, x.h(), wv.h(), bv.h(), runningMean.h(), runningVar.h(), x2.h(), wv2.h(), bv2.h(), runningMean2.h(),
                                                          runningVar2.h(), momentum, momentum2, eps, eps2));
  Ten out(o5[0]), saveMean(o5[1]), saveInvstd(o5[2]), saveMean2(o5[3]), saveInvstd2(o5[4]);
  const std::vector<int64_t> xs = x.shape();
  const double hw = (double)(xs[2] * xs[3]);
  const bool want[6] = {input->needsGrad(), weight->needsGrad(), bias->needsGrad(), input2->needsGrad(), weight2->needsGrad(), bias2->needsGrad()};
  // the six gradients for y's gradient given as plane values [N, C]
  auto six = [=](const Ten& plane_grad, const bool (&need)[6], Ten (&g)[6]) {
    const Ten gy = expand_planes(plane_grad, xs);
    uint8_t mask[6];
    for (int i = 0; i < 6; i++) mask[i] = (uint8_t)need[i];
    lamp_tensor* r6[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    HCALL(lamp_native_batch_norm2_add_relu_backward(r6, gy.h(), x.h(), wv.h(), bv.h(), saveMean.h(), saveInvstd.h(), x2.h(), wv2.h(), bv2.h(), saveMean2.h(),
                                                    saveInvstd2.h(), eps, eps2, mask));
    for (int i = 0; i < 6; i++) g[i] = r6[i] ? Ten(r6[i]) : Ten();
  };
  // p -> plane values: _log_softmax_backward_data, then / HW (each rounded to the dtype, as the pooled LogSoftMax's own backward rounds them)
  auto planes_of = [=](const Ten& p) {
    lamp_tensor *gi = nullptr, *pg = nullptr;
    HCALL(lamp_log_softmax_backward_data(&gi, p.h(), out.h(), 1));
    const Ten gih(gi);
    HCALL(lamp_div_scalar(&pg, gih.h(), hw));
    return Ten(pg);
  };
  struct Cache { Ten g[6]; Ten p; };
  auto cache = std::make_shared<Cache>();
  auto back = [=](int which) {
    return [=](const Ten& p, Variable& o) {
      if (!(cache->p.defined() && cache->p.h() == p.h() && cache->g[which].defined())) {
        bool need[6];
        for (int i = 0; i < 6; i++) need[i] = want[i];
        need[which] = true;
        six(planes_of(p), need, cache->g);
        cache->p = p;
      }
      Ten g = cache->g[which];
      cache->g[which] = Ten();
      bool any = false;
      for (int i = 0; i < 6; i++) any = any || cache->g[i].defined();
      if (!any) cache->p = Ten();
      o.accumulate(ops::reshape(g, o.shape()), true);
    };
  };
  op->params.push_back({input, back(0)});
  op->params.push_back({weight, back(1)});
  op->params.push_back({bias, back(2)});
  op->params.push_back({input2, back(3)});
  op->params.push_back({weight2, back(4)});
  op->params.push_back({bias2, back(5)});
  op->reset = [cache] { for (auto& t : cache->g) t = Ten(); cache->p = Ten(); };
  // (raw pointers: the node's params hold the six variables for as long as the node lives)
  Variable* vars[6] = {input.get(), weight.get(), bias.get(), input2.get(), weight2.get(), bias2.get()};
  op->pooled_input_grad = [=](const Ten& plane_grad) {
    Ten g[6];
    six(plane_grad, want, g);
    for (int i = 0; i < 6; i++)
      if (want[i] && g[i].defined()) vars[i]->accumulate(ops::reshape(g[i], vars[i]->shape()), true);
  };
  return make_result(op, out);
}
bool conv_of_batch_norm_relu_2d_pays(const Var& input, const Var& weight, int64_t stride, int64_t padding, int64_t dilation, int64_t groups) {
  if (input->value.ndim() != 4 || weight->value.ndim() != 4) return false;
  const int64_t s2[2] = {stride, stride}, p2[2] = {padding, padding}, d2[2] = {dilation, dilation};
  int folds = 0;
  HCALL(lamp_convolution_bn_relu_input_folds(&folds, input->value.h(), weight->value.h(), s2, p2, d2, 2, groups));
  return folds != 0;
}
// Convolution(relu(BatchNorm2D(x))) as ONE node: the middle of lamp's residual block (Conv2D -> BatchNorm2D -> relu -> Dropout(0) -> Conv2D,
// cnn.scala:38-60).  The batch norm becomes a per-channel table (lamp_batch_norm_affine) that the convolution applies while staging its
// input; values are bitwise those of BatchNorm2D -> relu -> Convolution and the normalised tensor is never written.  Backward: the
// convolution's gradients (the weight gradient rebuilds relu(bn(x)) the same way), then the fused batch-norm-relu backward on x.
Var conv_of_batch_norm_relu_2d(const Var& input, const Var& bnWeight, const Var& bnBias, const Ten& runningMean, const Ten& runningVar, double momentum,
                               double eps, const Var& weight, const Var& bias, const std::vector<int64_t>& stride, const std::vector<int64_t>& padding,
                               const std::vector<int64_t>& dilation, int64_t groups) {
  auto op = new_op("ConvolutionOfBatchNormRelu");
  const Ten x = input->value, gv = bnWeight->value, bv = bnBias->value, wv = weight->value;
  const std::vector<int64_t> expected = {x.size(1)};
  LAMP_CHECK(x.ndim() == 4, "Input dimensions must be 4");
  LAMP_CHECK(bnWeight->shape() == expected && bnBias->shape() == expected, "batch norm weight / bias have the wrong shape");
  LAMP_CHECK(runningMean.shape() == expected && runningVar.shape() == expected, "running statistics have the wrong shape");
  lamp_tensor* o3[3] = {nullptr, nullptr, nullptr};
  HCALL(lamp_batch_norm_affine(o3, x.h(), gv.h(), bv.h(), runningMean.h(), runningVar.h(), momentum, eps));
  Ten affine(o3[0]), saveMean(o3[1]), saveInvstd(o3[2]);
  const int ns = (int)stride.size();
  lamp_tensor* o = nullptr;
  HCALL(lamp_convolution_bn_relu_input(&o, x.h(), affine.h(), wv.h(), bias->value.h(), stride.data(), padding.data(), dilation.data(), ns, groups));
  struct Cache { Ten g[5]; Ten p; };                       // dx, d(bn weight), d(bn bias), d(conv weight), d(conv bias) of one backward pass
  auto cache = std::make_shared<Cache>();
  const bool want[5] = {input->needsGrad(), bnWeight->needsGrad(), bnBias->needsGrad(), weight->needsGrad(), bias->needsGrad()};
  auto back = [=](int which) {
    return [=](const Ten& p, Variable& out) {
      if (!(cache->p.defined() && cache->p.h() == p.h() && cache->g[which].defined())) {
        bool w[5];
        for (int i = 0; i < 5; i++) w[i] = want[i];
        w[which] = true;
        const bool through_bn = w[0] || w[1] || w[2];
        const uint8_t cmask[3] = {(uint8_t)through_bn, (uint8_t)w[3], (uint8_t)w[4]};
        lamp_tensor* c3[3] = {nullptr, nullptr, nullptr};
        HCALL(lamp_convolution_bn_relu_input_backward(c3, p.h(), x.h(), affine.h(), wv.h(), stride.data(), padding.data(), dilation.data(), ns, groups, cmask));
        Ten dact = c3[0] ? Ten(c3[0]) : Ten();
        cache->g[3] = c3[1] ? Ten(c3[1]) : Ten();
        cache->g[4] = c3[2] ? Ten(c3[2]) : Ten();
        if (through_bn) {
          const uint8_t bmask[3] = {(uint8_t)w[0], (uint8_t)w[1], (uint8_t)w[2]};
          lamp_tensor* b3[3] = {nullptr, nullptr, nullptr};
          HCALL(lamp_native_batch_norm_relu_backward(b3, dact.h(), x.h(), gv.h(), bv.h(), runningMean.h(), runningVar.h(), saveMean.h(), saveInvstd.h(), 1, eps, bmask));
          for (int i = 0; i < 3; i++) cache->g[i] = b3[i] ? Ten(b3[i]) : Ten();
        }
        cache->p = p;
      }
      Ten g = cache->g[which];
      cache->g[which] = Ten();
      bool any = false;
      for (int i = 0; i < 5; i++) any = any || cache->g[i].defined();
      if (!any) cache->p = Ten();
      out.accumulate(ops::reshape(g, out.shape()), true);
    };
  };
  op->params.push_back({input, back(0)});
  op->params.push_back({bnWeight, back(1)});
  op->params.push_back({bnBias, back(2)});
  op->params.push_back({weight, back(3)});
  op->params.push_back({bias, back(4)});
  return make_result(op, Ten(o));
}
Var layer_norm(const Var& input, const Var& weight, const Var& bias, const std::vector<int64_t>& normalizedShape, double eps) {
  auto op = new_op("LayerNormOp");
  lamp_tensor* o3[3] = {nullptr, nullptr, nullptr};
  const lamp_tensor* w = weight ? weight->value.h() : nullptr;
  const lamp_tensor* b = bias ? bias->value.h() : nullptr;
  HCALL(lamp_native_layer_norm(o3, input->value.h(), normalizedShape.data(), (int)normalizedShape.size(), w, b, eps));
  Ten out(o3[0]), mean(o3[1]), rstd(o3[2]), iv = input->value;
  Ten wv = weight ? weight->value : Ten(), bv = bias ? bias->value : Ten();
  auto back = [=](int which) {
    return [=](const Ten& p, Variable& o) {
      lamp_tensor* r3[3] = {nullptr, nullptr, nullptr};
      uint8_t mask[3] = {(uint8_t)(which == 0), (uint8_t)(which == 1), (uint8_t)(which == 2)};
      HCALL(lamp_native_layer_norm_backward(r3, p.h(), iv.h(), normalizedShape.data(), (int)normalizedShape.size(), mean.h(), rstd.h(),
                                            wv.h(), bv.h(), mask));
      o.accumulate(Ten(r3[which]), true);
    };
  };
  op->params.push_back({input, back(0)});
  if (weight) op->params.push_back({weight, back(1)});
  if (bias) op->params.push_back({bias, back(2)});
  return make_result(op, out);
}
Var embedding(const Var& input, const Var& weight) {   // ops.scala:2141-2170
  auto op = new_op("Embedding");
  Ten idx = input->value;
  const int64_t nw = weight->value.size(0);
  op->params.push_back({weight, [=](const Ten& p, Variable& out) {
    lamp_tensor* t = nullptr;
    HCALL(lamp_embedding_backward(&t, p.h(), idx.h(), nw, /*padding_idx, as the reference passes it*/ 0));
    out.accumulate(Ten(t), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_embedding(&o, weight->value.h(), idx.h()));
  return make_result(op, Ten(o));
}

}  // namespace F
}  // namespace host
}  // namespace lamp
