// Host-side mirror of lamp-core's autograd + nn layer, written against the C ABI only
// (include/lamp_hip.h) - the same calls a JVM would make through JNI.
//
// Reference: lamp-core/src/main/scala/lamp/autograd/autograd.scala:63-282 (Op, Variable,
// backprop), :488-518 (topologicalSort); autograd/package.scala:60-78 (const/param).
// In lamp these run on the JVM; no JVM exists in this environment, so the sequencing logic is
// restated in C++ (same names, same order of ATen calls, same accumulate-into-grad contract).
#pragma once
#include <cmath>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../core/tensor.h"

namespace lamp {
namespace host {

// ---- owned tensor handle (what STen is on the JVM: one aten.Tensor + a Scope that frees it) ----
struct Box {
  lamp_tensor* h;
  explicit Box(lamp_tensor* h_) : h(h_) {}
  ~Box() { if (h) lamp_tensor_release(h); }
  Box(const Box&) = delete;
  Box& operator=(const Box&) = delete;
};
struct Ten {
  std::shared_ptr<Box> b;
  Ten() = default;
  explicit Ten(lamp_tensor* h) : b(h ? std::make_shared<Box>(h) : nullptr) {}
  lamp_tensor* h() const { return b ? b->h : nullptr; }
  bool defined() const { return b && b->h; }
  std::vector<int64_t> shape() const { return b->h->shape(); }
  int64_t size(int i) const { return b->h->sizes[i < 0 ? i + b->h->ndim : i]; }
  int ndim() const { return b->h->ndim; }
  int dtype() const { return b->h->dtype; }
  int device() const { return b->h->device(); }
  int64_t numel() const { return b->h->numel(); }
};
// borrow a caller-owned handle without taking ownership (+1 handle on the same view)
inline Ten borrow(const lamp_tensor* t) {
  lamp_tensor* r = nullptr;
  if (lamp_tensor_retain(t, &r) != 0) throw Error(lamp_last_error());
  return Ten(r);
}

#define HCALL(expr)                                                      \
  do {                                                                   \
    if ((expr) != 0) throw ::lamp::Error(lamp_last_error());             \
  } while (0)

// thin wrappers: one C-ABI call each (the STen methods lamp uses)
namespace ops {
#define H_UN(NAME, CALL) inline Ten NAME(const Ten& a) { lamp_tensor* o = nullptr; HCALL(CALL(&o, a.h())); return Ten(o); }
H_UN(zeros_like, lamp_zeros_like)
H_UN(ones_like, lamp_ones_like)
H_UN(clone, lamp_clone)
H_UN(relu, lamp_relu)
H_UN(gelu, lamp_gelu)
H_UN(sigmoid, lamp_sigmoid)
H_UN(tanh, lamp_tanh)
H_UN(hardswish, lamp_hardswish)
H_UN(exp, lamp_exp)
H_UN(log, lamp_log)
H_UN(log1p, lamp_log1p)
H_UN(sin, lamp_sin)
H_UN(cos, lamp_cos)
H_UN(reciprocal, lamp_reciprocal)
H_UN(sqrt, lamp_sqrt)
H_UN(t, lamp_t)
H_UN(sum_all, lamp_sum_all)
#undef H_UN
#define H_BIN(NAME, CALL) inline Ten NAME(const Ten& a, const Ten& b) { lamp_tensor* o = nullptr; HCALL(CALL(&o, a.h(), b.h())); return Ten(o); }
H_BIN(mul, lamp_mul)
H_BIN(div, lamp_div)
H_BIN(mm, lamp_mm)
H_BIN(bmm, lamp_bmm)
H_BIN(gelu_backward, lamp_gelu_backward)
H_BIN(sigmoid_backward, lamp_sigmoid_backward)
H_BIN(tanh_backward, lamp_tanh_backward)
H_BIN(hardswish_backward, lamp_hardswish_backward)
H_BIN(minimum, lamp_minimum)
H_BIN(maximum, lamp_maximum)
#undef H_BIN
inline Ten add(const Ten& a, const Ten& b, double alpha = 1.0) { lamp_tensor* o = nullptr; HCALL(lamp_add(&o, a.h(), b.h(), alpha)); return Ten(o); }
inline Ten sub(const Ten& a, const Ten& b, double alpha = 1.0) { lamp_tensor* o = nullptr; HCALL(lamp_sub(&o, a.h(), b.h(), alpha)); return Ten(o); }
inline Ten add_scalar(const Ten& a, double s) { lamp_tensor* o = nullptr; HCALL(lamp_add_scalar(&o, a.h(), s, 1.0)); return Ten(o); }
inline Ten mul_scalar(const Ten& a, double s) { lamp_tensor* o = nullptr; HCALL(lamp_mul_scalar(&o, a.h(), s)); return Ten(o); }
inline Ten pow_scalar(const Ten& a, double e) { lamp_tensor* o = nullptr; HCALL(lamp_pow_scalar(&o, a.h(), e)); return Ten(o); }
inline void add_(const Ten& self, const Ten& b, double alpha = 1.0) { HCALL(lamp_add_(self.h(), b.h(), alpha)); }      // +=
inline void sub_(const Ten& self, const Ten& b, double alpha = 1.0) { HCALL(lamp_sub_(self.h(), b.h(), alpha)); }      // -=
inline void mul_(const Ten& self, const Ten& b) { HCALL(lamp_mul_(self.h(), b.h())); }
inline void div_(const Ten& self, const Ten& b) { HCALL(lamp_div_(self.h(), b.h())); }
inline void mul_scalar_(const Ten& self, double s) { HCALL(lamp_mul_scalar_(self.h(), s)); }
inline void addcmul_(const Ten& self, const Ten& t1, const Ten& t2, double v) { HCALL(lamp_addcmul_out(self.h(), self.h(), t1.h(), t2.h(), v)); }
inline void fill_(const Ten& self, double v) { HCALL(lamp_fill_(self.h(), v)); }
inline void zero_(const Ten& self) { HCALL(lamp_zero_(self.h())); }
inline void copy_(const Ten& dst, const Ten& src) { HCALL(lamp_copy_(dst.h(), src.h(), 1)); }
inline Ten view(const Ten& a, const std::vector<int64_t>& s) { lamp_tensor* o = nullptr; HCALL(lamp_view(&o, a.h(), s.data(), (int)s.size())); return Ten(o); }
inline Ten reshape(const Ten& a, const std::vector<int64_t>& s) { lamp_tensor* o = nullptr; HCALL(lamp_reshape(&o, a.h(), s.data(), (int)s.size())); return Ten(o); }
inline Ten flatten(const Ten& a, int64_t s, int64_t e) { lamp_tensor* o = nullptr; HCALL(lamp_flatten(&o, a.h(), s, e)); return Ten(o); }
inline Ten transpose(const Ten& a, int64_t d0, int64_t d1) { lamp_tensor* o = nullptr; HCALL(lamp_transpose(&o, a.h(), d0, d1)); return Ten(o); }
inline Ten slice(const Ten& a, int64_t d, int64_t s, int64_t e, int64_t st) { lamp_tensor* o = nullptr; HCALL(lamp_slice(&o, a.h(), d, s, e, st)); return Ten(o); }
inline Ten select(const Ten& a, int64_t d, int64_t i) { lamp_tensor* o = nullptr; HCALL(lamp_select(&o, a.h(), d, i)); return Ten(o); }
inline Ten unbroadcast(const Ten& p, const std::vector<int64_t>& s) { lamp_tensor* o = nullptr; HCALL(lamp_unbroadcast(&o, p.h(), s.data(), (int)s.size())); return Ten(o); }
inline Ten sum_dims(const Ten& a, const std::vector<int64_t>& d, bool keep) { lamp_tensor* o = nullptr; HCALL(lamp_sum_dims(&o, a.h(), d.data(), (int)d.size(), keep)); return Ten(o); }
inline Ten mean_dims(const Ten& a, const std::vector<int64_t>& d, bool keep) { lamp_tensor* o = nullptr; HCALL(lamp_mean_dims(&o, a.h(), d.data(), (int)d.size(), keep)); return Ten(o); }
inline Ten norm2_dims(const Ten& a, const std::vector<int64_t>& d, bool keep) { lamp_tensor* o = nullptr; HCALL(lamp_norm2_dims(&o, a.h(), d.data(), (int)d.size(), keep)); return Ten(o); }
inline Ten zeros(const std::vector<int64_t>& s, int dtype, int device) { lamp_tensor* o = nullptr; HCALL(lamp_zeros(&o, s.data(), (int)s.size(), dtype, device)); return Ten(o); }
inline Ten ones(const std::vector<int64_t>& s, int dtype, int device) { lamp_tensor* o = nullptr; HCALL(lamp_ones(&o, s.data(), (int)s.size(), dtype, device)); return Ten(o); }
inline Ten scalar(double v, int dtype, int device) { lamp_tensor* o = nullptr; HCALL(lamp_scalar_tensor(&o, v, dtype, device)); return Ten(o); }
inline Ten normal(double mean, double std, const std::vector<int64_t>& s, int dtype, int device) { lamp_tensor* o = nullptr; HCALL(lamp_normal(&o, mean, std, s.data(), (int)s.size(), dtype, device)); return Ten(o); }
inline Ten cast(const Ten& a, int dtype) { lamp_tensor* o = nullptr; HCALL(lamp_cast(&o, a.h(), dtype)); return Ten(o); }
inline Ten index_select(const Ten& a, int64_t d, const Ten& idx) { lamp_tensor* o = nullptr; HCALL(lamp_index_select(&o, a.h(), d, idx.h())); return Ten(o); }
inline Ten index_add(const Ten& a, int64_t d, const Ten& idx, const Ten& src) { lamp_tensor* o = nullptr; HCALL(lamp_index_add(&o, a.h(), d, idx.h(), src.h())); return Ten(o); }
inline Ten where(const Ten& c, const Ten& a, const Ten& b) { lamp_tensor* o = nullptr; HCALL(lamp_where(&o, c.h(), a.h(), b.h())); return Ten(o); }
inline Ten lt_scalar(const Ten& a, double s) { lamp_tensor* o = nullptr; HCALL(lamp_lt_scalar(&o, a.h(), s)); return Ten(o); }
inline Ten le_scalar(const Ten& a, double s) { lamp_tensor* o = nullptr; HCALL(lamp_le_scalar(&o, a.h(), s)); return Ten(o); }
inline Ten cat(const std::vector<Ten>& ts, int64_t dim) {
  std::vector<lamp_tensor*> hs;
  for (auto& t : ts) hs.push_back(t.h());
  lamp_tensor* o = nullptr;
  HCALL(lamp_cat(&o, hs.data(), (int)hs.size(), dim));
  return Ten(o);
}
}  // namespace ops

// ---- autograd --------------------------------------------------------------------------------
struct Variable;
using Var = std::shared_ptr<Variable>;
// a backward closure receives the incoming partial derivative p and the INPUT variable whose gradient it
// must add to (autograd.scala:66-84: "the result is accumulated (added)")
using Backward = std::function<void(const Ten& p, Variable& out)>;

struct Op {
  virtual ~Op() = default;
  std::vector<std::pair<Var, Backward>> params;
  const char* name = "op";
  // state a node's closures keep BETWEEN each other during one backprop (the convolution pair's held derivatives): backprop calls this
  // for every node before it starts, so a pass that was abandoned half-way (an exception between two siblings) leaves nothing behind
  std::function<void()> reset;
  // a node whose value is LogSoftMax(pool(y)) of a tensor y it never wrote (batch_norm2_add_relu_pool_log_softmax_2d): takes y's gradient as one
  // value per plane ([N, C], any strides) and accumulates the gradients of its inputs at once.  The loss's closure calls it INSTEAD of
  // accumulating into the node's output when it holds that gradient already (nll_loss_accumulate); the node's own closures then never run.
  std::function<void(const Ten& plane_grad)> pooled_input_grad;
};

// Gradient buffers.  lamp allocates a zeros_like buffer for every op output up front (autograd.scala:89-96) and
// every closure does `out += ...`.  Here the buffer is created by the FIRST accumulation instead (0 + v == v,
// bit for bit): no memset, no read-modify-write for the common single-consumer case, same values.
//   grad undefined  <=> mathematically zero (nothing accumulated yet)
//   grad_shared     <=> the tensor was adopted from somewhere else (e.g. `out += p` with equal shapes adopts p):
//                       it must not be modified in place; the next accumulation writes a fresh tensor.
struct Variable {
  std::shared_ptr<Op> op;   // empty for constants / parameters
  Ten value;
  bool wants_grad = false;  // lamp's needsGrad
  Ten grad;
  bool grad_shared = false;
  // a contribution one of this variable's consumers has put off until another consumer's arrives (the two first convolutions of a residual
  // block: their input gradients come from one launch); backprop runs it when it reaches this variable and it is still there
  std::vector<std::function<void()>> pending;
  bool needsGrad() const { return wants_grad; }
  std::vector<int64_t> shape() const { return value.shape(); }
  void zeroGrad() { grad = Ten(); grad_shared = false; }            // lazily zero
  // the gradient as a tensor (materialises zeros if nothing was accumulated)
  Ten grad_tensor() {
    if (!wants_grad) return Ten();
    if (!grad.defined()) { grad = ops::zeros_like(value); grad_shared = false; }
    return grad;
  }
  // an exclusively owned buffer that kernels may update in place
  Ten grad_inplace() {
    if (!grad.defined()) { grad = ops::zeros_like(value); grad_shared = false; }
    else if (grad_shared) { grad = ops::clone(grad); grad_shared = false; }
    return grad;
  }
  // out += t.  `fresh`: t is a temporary nobody else will read or write, so it can become the buffer itself.
  void accumulate(const Ten& t, bool fresh) {
    if (!grad.defined()) { grad = t; grad_shared = !fresh; return; }
    if (grad_shared) { grad = ops::add(grad, t); grad_shared = false; return; }
    ops::add_(grad, t);
  }
  void accumulate_scaled(const Ten& t, double alpha) {   // out += alpha * t
    if (!grad.defined()) { grad = ops::mul_scalar(t, alpha); grad_shared = false; return; }
    if (grad_shared) { grad = ops::add(grad, t, alpha); grad_shared = false; return; }
    ops::add_(grad, t, alpha);
  }
  void subtract(const Ten& t) { accumulate_scaled(t, -1.0); }
  // out += value * t1 * t2   (ATen addcmul)
  void addcmul(const Ten& t1, const Ten& t2, double v) {
    if (!grad.defined()) {
      Ten m = ops::mul(t1, t2);
      if (v != 1.0) ops::mul_scalar_(m, v);
      if (m.shape() != value.shape()) { grad = ops::zeros_like(value); ops::add_(grad, m); }   // broadcast product
      else grad = m;
      grad_shared = false;
      return;
    }
    ops::addcmul_(grad_inplace(), t1, t2, v);
  }
  bool has_grad() const { return grad.defined(); }
};

Var make_const(const Ten& t);   // package.scala:60-68
Var make_param(const Ten& t);   // package.scala:70-78
Var make_result(const std::shared_ptr<Op>& op, const Ten& value);   // Variable.apply (autograd.scala:88-96)
std::vector<Variable*> topological_sort(Variable* root);   // autograd.scala:490-518
void backprop(const Var& root);                            // autograd.scala:264-282
void backprop(const Var& root, const std::function<void(Variable*)>& after_node);

}  // namespace host
}  // namespace lamp

// C-ABI handles of the host layer
struct lamp_var { lamp::host::Var v; };
