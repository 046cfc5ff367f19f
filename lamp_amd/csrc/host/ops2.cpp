// The rest of lamp's differentiable operators (lamp-core/src/main/scala/lamp/autograd/ops.scala) that the reference's own gradient
// suite exercises (autograd.test.scala) and that sit one step off the training hot path: shape / index operators, the remaining
// element-wise functions and losses, Variance, WeightNorm, MaxPool1D.  Same rules as ops.cpp: the forward value and every backward
// closure issue the ATen calls the Scala closures issue, in their order, through the C ABI - including the reference's quirks
// (named at each operator).  Linear algebra (Inv, PInv, LogDet, Cholesky*, Diag), sparse tensors and Cross are not mirrored: they are
// outside SURVEY section 8.
#include "ops.h"

namespace lamp {
namespace host {

namespace {
std::shared_ptr<Op> new_op(const char* name) {
  auto o = std::make_shared<Op>();
  o->name = name;
  return o;
}
Ten call1(int (*f)(lamp_tensor**, const lamp_tensor*), const Ten& a) { lamp_tensor* o = nullptr; HCALL(f(&o, a.h())); return Ten(o); }
Ten masked_scatter(const Ten& self, const Ten& mask, const Ten& src) { lamp_tensor* o = nullptr; HCALL(lamp_masked_scatter(&o, self.h(), mask.h(), src.h())); return Ten(o); }
Ten arange(int64_t start, int64_t end, int64_t step, int device) { lamp_tensor* o = nullptr; HCALL(lamp_arange(&o, (double)start, (double)end, (double)step, kI64, device)); return Ten(o); }
}  // namespace

namespace F {

// ---- shape / index operators (ops.scala:64-260, 410-509, 647-663) -----------------------------------------------------------------
Var stack(const std::vector<Var>& as, int64_t dim) {                 // Stack: out += p.select(dim, idx)
  auto op = new_op("Stack");
  std::vector<lamp_tensor*> hs;
  int64_t idx = 0;
  for (auto& a : as) {
    op->params.push_back({a, [dim, idx](const Ten& p, Variable& out) { out.accumulate(ops::select(p, dim, idx), false); }});
    hs.push_back(a->value.h());
    idx++;
  }
  lamp_tensor* o = nullptr;
  HCALL(lamp_stack(&o, hs.data(), (int)hs.size(), dim));
  return make_result(op, Ten(o));
}
Var select(const Var& a, int64_t dim, int64_t index) {               // Select: tmp = zeros.indexAdd(dim, [index], p.view(.. 1 ..)); out += tmp
  auto op = new_op("Select");
  op->params.push_back({a, [dim, index](const Ten& p, Variable& out) {
    Ten tmp = ops::zeros(out.shape(), out.value.dtype(), out.value.device());
    std::vector<int64_t> ps = p.shape();
    ps.insert(ps.begin() + dim, 1);
    lamp_tensor* sc = nullptr;
    HCALL(lamp_scalar_tensor_l(&sc, index, kI64, out.value.device()));
    out.accumulate(ops::index_add(tmp, dim, Ten(sc), ops::view(p, ps)), true);
  }});
  return make_result(op, ops::select(a->value, dim, index));
}
Var slice(const Var& a, int64_t dim, int64_t start, int64_t end, int64_t step) {   // Slice: zeros.indexAdd(dim, arange(start, end, step), p)
  auto op = new_op("Slice");
  op->params.push_back({a, [dim, start, end, step](const Ten& p, Variable& out) {
    Ten tmp = ops::zeros(out.shape(), out.value.dtype(), out.value.device());
    out.accumulate(ops::index_add(tmp, dim, arange(start, end, step, out.value.device()), p), true);
  }});
  return make_result(op, ops::slice(a->value, dim, start, end, step));
}
Var mask_select(const Var& input, const Var& mask) {                 // MaskSelect: out += zerosLike(out).maskedScatter(mask, p)
  auto op = new_op("MaskSelect");
  Ten mv = mask->value;
  op->params.push_back({input, [mv](const Ten& p, Variable& out) { out.accumulate(masked_scatter(ops::zeros_like(out.value), mv, p), true); }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_masked_select(&o, input->value.h(), mv.h()));
  return make_result(op, Ten(o));
}
Var index_fill(const Var& input, int64_t dim, const Var& index, double fill) {     // IndexFill: out += p.indexFill(dim, index, 0)
  auto op = new_op("IndexFill");
  Ten iv = index->value;
  op->params.push_back({input, [dim, iv](const Ten& p, Variable& out) {
    lamp_tensor* o = nullptr;
    HCALL(lamp_index_fill(&o, p.h(), dim, iv.h(), 0.0));
    out.accumulate(Ten(o), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_index_fill(&o, input->value.h(), dim, iv.h(), fill));
  return make_result(op, Ten(o));
}
Var where(const Ten& condition, const Var& trueBranch, const Var& falseBranch) {   // Where: out.addcmulSelf(p, where(c, 1, 0), 1)
  auto op = new_op("Where");
  Ten tv = trueBranch->value, fv = falseBranch->value;
  op->params.push_back({trueBranch, [condition, tv, fv](const Ten& p, Variable& out) {
    out.addcmul(p, ops::where(condition, ops::ones_like(tv), ops::zeros_like(fv)), 1.0);
  }});
  op->params.push_back({falseBranch, [condition, tv, fv](const Ten& p, Variable& out) {
    out.addcmul(p, ops::where(condition, ops::zeros_like(tv), ops::ones_like(fv)), 1.0);
  }});
  return make_result(op, ops::where(condition, tv, fv));
}
Var assign(const Var& abandon, const Var& keep) {                    // Assign: the value of `keep`; `abandon` receives nothing
  auto op = new_op("Assign");
  op->params.push_back({abandon, [](const Ten&, Variable&) {}});
  op->params.push_back({keep, [](const Ten& p, Variable& out) { out.accumulate(p, false); }});
  return make_result(op, keep->value);
}
Var cast_to_precision(const Var& a, int dtype) {                      // CastToPrecision: same type -> the variable itself
  if (a->value.dtype() == dtype) return a;
  auto op = new_op("CastToPrecision");
  const int from = a->value.dtype();
  op->params.push_back({a, [from](const Ten& p, Variable& out) { out.accumulate(ops::cast(p, from), p.dtype() != from); }});
  return make_result(op, ops::cast(a->value, dtype));
}
Var scatter_add(const Var& src, const Var& index, int64_t dim, int64_t maxIndex) { // ScatterAdd: out += p.gather(dim, index)
  LAMP_CHECK(src->value.size((int)dim) == index->value.size((int)dim), "assertion failed: src.shape(dim) == index.shape(dim)");
  auto op = new_op("ScatterAdd");
  Ten iv = index->value;
  op->params.push_back({src, [dim, iv](const Ten& p, Variable& out) {
    lamp_tensor* o = nullptr;
    HCALL(lamp_gather(&o, p.h(), dim, iv.h()));
    out.accumulate(Ten(o), true);
  }});
  std::vector<int64_t> shape = src->shape();
  shape[dim] = maxIndex;
  Ten zeros = ops::zeros(shape, src->value.dtype(), src->value.device());
  lamp_tensor* o = nullptr;
  HCALL(lamp_scatter_add(&o, zeros.h(), dim, iv.h(), src->value.h()));
  return make_result(op, Ten(o));
}
Var index_add(const Var& src, const Var& index, int64_t dim, int64_t maxIndex) {   // IndexAdd: out += p.indexSelect(dim, index)
  auto op = new_op("IndexAdd");
  Ten iv = index->value;
  op->params.push_back({src, [dim, iv](const Ten& p, Variable& out) { out.accumulate(ops::index_select(p, dim, iv), true); }});
  std::vector<int64_t> shape = src->shape();
  shape[dim] = maxIndex;
  return make_result(op, ops::index_add(ops::zeros(shape, src->value.dtype(), src->value.device()), dim, iv, src->value));
}
Var index_add_to_target(const Var& target, const Var& src, const Var& index, int64_t dim) {   // IndexAddToTarget
  auto op = new_op("IndexAddToTarget");
  Ten iv = index->value;
  op->params.push_back({src, [dim, iv](const Ten& p, Variable& out) { out.accumulate(ops::index_select(p, dim, iv), true); }});
  op->params.push_back({target, [](const Ten& p, Variable& out) { out.accumulate(p, false); }});
  return make_result(op, ops::index_add(target->value, dim, iv, src->value));
}
Var repeat_interleave(const Var& self, const Var& repeats, int64_t dim) {
  // RepeatInterleave (ops.scala:484-509): the closure scatters back along dimension 0 whatever `dim` was (as written)
  auto op = new_op("RepeatInterleave");
  Ten rv = repeats->value;
  const int64_t n0 = self->value.size(0);
  op->params.push_back({self, [rv, n0](const Ten& p, Variable& out) {
    lamp_tensor* ri = nullptr;
    HCALL(lamp_repeat_interleave_tensor(&ri, arange(0, n0, 1, out.value.device()).h(), rv.h(), 0));
    out.accumulate(ops::index_add(ops::zeros_like(out.value), 0, Ten(ri), p), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_repeat_interleave_tensor(&o, self->value.h(), rv.h(), dim));
  return make_result(op, Ten(o));
}
Var expand_as(const Var& a, const Ten& as) {                          // ExpandAs / Expand: out += p.unbroadcast(a.shape)
  auto op = new_op("ExpandAs");
  auto sh = a->shape();
  op->params.push_back({a, [sh](const Ten& p, Variable& out) { out.accumulate(ops::unbroadcast(p, sh), true); }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_expand_as(&o, a->value.h(), as.h()));
  return make_result(op, Ten(o));
}
Var expand(const Var& a, const std::vector<int64_t>& shape) {
  auto op = new_op("Expand");
  auto sh = a->shape();
  op->params.push_back({a, [sh](const Ten& p, Variable& out) { out.accumulate(ops::unbroadcast(p, sh), true); }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_expand(&o, a->value.h(), shape.data(), (int)shape.size()));
  return make_result(op, Ten(o));
}

// ---- Diag, Cross and the non-differentiable ArgMax / OneHot / EqWhere (ops.scala:120-132, 230-259, 333-350, 581-601) -----------------------
Var diag(const Var& a, int64_t diagonal) {                            // backward: out += p.diag(diagonal)
  auto op = new_op("Diag");
  op->params.push_back({a, [diagonal](const Ten& p, Variable& out) {
    lamp_tensor* d = nullptr; HCALL(lamp_diag(&d, p.h(), diagonal));
    out.accumulate(Ten(d), true);
  }});
  lamp_tensor* v = nullptr; HCALL(lamp_diag(&v, a->value.h(), diagonal));
  return make_result(op, Ten(v));
}
Var cross(const Var& a, const Var& b, int64_t dim) {
  // as the reference writes it: a: out -= p * ones(a.shape).cross(b, dim) ; b: out += p * ones(b.shape).cross(a, dim)
  auto op = new_op("Cross");
  Ten av = a->value, bv = b->value;
  auto ones_cross = [dim](const Ten& like, const Ten& other) {
    lamp_tensor* c = nullptr; HCALL(lamp_cross(&c, ops::ones_like(like).h(), other.h(), dim));
    return Ten(c);
  };
  op->params.push_back({a, [av, bv, ones_cross](const Ten& p, Variable& out) { out.subtract(ops::mul(p, ones_cross(av, bv))); }});
  op->params.push_back({b, [av, bv, ones_cross](const Ten& p, Variable& out) { out.accumulate(ops::mul(p, ones_cross(bv, av)), true); }});
  lamp_tensor* v = nullptr; HCALL(lamp_cross(&v, av.h(), bv.h(), dim));
  return make_result(op, Ten(v));
}
namespace {
Var not_differentiable(const char* name, const Var& a, const Ten& value) {
  auto op = new_op(name);
  std::string what = std::string(name) + " is not differentiable";
  op->params.push_back({a, [what](const Ten&, Variable&) { throw Error(what); }});
  return make_result(op, value);
}
}  // namespace
Var argmax(const Var& a, int64_t dim, bool keepDim) {
  lamp_tensor* v = nullptr; HCALL(lamp_argmax(&v, a->value.h(), dim, keepDim ? 1 : 0));
  return not_differentiable("ArgMax", a, Ten(v));
}
Var one_hot(const Var& a, int64_t numClasses) {
  lamp_tensor* v = nullptr; HCALL(lamp_one_hot(&v, a->value.h(), numClasses));
  return not_differentiable("OneHot", a, Ten(v));
}
Var eq_where(const Var& a, int64_t b) {                               // no parameters: a constant of the graph
  auto op = new_op("EqWhere");
  lamp_tensor* v = nullptr; HCALL(lamp_eq_scalar(&v, a->value.h(), (double)b));
  return make_result(op, Ten(v));
}

// ---- element-wise (ops.scala:841-916, 2287-2340) -------------------------------------------------------------------------------------
Var tan(const Var& a) {                                               // Tan: tmp = value^2 ; tmp += ones(1) ; out.addcmulSelf(p, tmp, 1)
  auto op = new_op("Tan");
  Ten val = call1(lamp_tan, a->value);
  op->params.push_back({a, [val](const Ten& p, Variable& out) { out.addcmul(p, ops::add_scalar(ops::pow_scalar(val, 2.0), 1.0), 1.0); }});
  return make_result(op, val);
}
Var atan(const Var& a) {                                              // ArcTan: tmp = a^2 ; += 1 ; reciprocal_
  auto op = new_op("ArcTan");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) {
    Ten t = ops::add_scalar(ops::pow_scalar(av, 2.0), 1.0);
    HCALL(lamp_reciprocal_(t.h()));
    out.addcmul(p, t, 1.0);
  }});
  return make_result(op, call1(lamp_atan, av));
}
Var pow(const Var& a, const Var& exponent) {
  // Pow (ops.scala:890-916).  Both closures read the exponent as ONE host number (exponent.toDoubleArray(0)); the exponent's
  // gradient is p unbroadcast to [out.sizes.head or 1, 1] times the SUM of a^e log a - kept as written.
  auto op = new_op("Pow");
  Ten av = a->value, ev = exponent->value;
  auto exp0 = [ev]() { double v = 0; HCALL(lamp_item(ops::slice(ops::reshape(ev, {-1}), 0, 0, 1, 1).h(), &v)); return v; };
  op->params.push_back({a, [av, exp0](const Ten& p, Variable& out) { const double e = exp0(); out.addcmul(p, ops::pow_scalar(av, e - 1.0), e); }});
  op->params.push_back({exponent, [av, exp0](const Ten& p, Variable& out) {
    const double e = exp0();
    Ten t3 = ops::mul(ops::pow_scalar(av, e), ops::log(av));
    auto os = out.shape();
    Ten p2 = ops::unbroadcast(p, {os.empty() ? 1 : os[0], 1});
    out.addcmul(p2, ops::sum_all(t3), 1.0);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_pow_tensor(&o, av.h(), ev.h()));
  return make_result(op, Ten(o));
}
static Var elementwise_min_max(const Var& a, const Var& b, bool is_min) {
  // ElementWiseMinimum / Maximum: mask = a == value; each closure masked_scatters p into zeros - masked_scatter consumes p's elements
  // IN ORDER, it does not pick the elements under the mask (as written in the reference; with a constant p the two coincide)
  auto op = new_op(is_min ? "ElementWiseMinimum" : "ElementWiseMaximum");
  Ten val = is_min ? ops::minimum(a->value, b->value) : ops::maximum(a->value, b->value);
  lamp_tensor *m = nullptr, *mn = nullptr;
  HCALL(lamp_eq(&m, a->value.h(), val.h()));
  Ten mask(m);
  HCALL(lamp_logical_not(&mn, mask.h()));
  Ten maskneg(mn);
  op->params.push_back({a, [mask](const Ten& p, Variable& out) { out.accumulate(masked_scatter(ops::zeros_like(out.value), mask, p), true); }});
  op->params.push_back({b, [maskneg](const Ten& p, Variable& out) { out.accumulate(masked_scatter(ops::zeros_like(out.value), maskneg, p), true); }});
  return make_result(op, val);
}
Var minimum(const Var& a, const Var& b) { return elementwise_min_max(a, b, true); }
Var maximum(const Var& a, const Var& b) { return elementwise_min_max(a, b, false); }

// ---- reductions / norms (ops.scala:1055-1174, 1369-1383) ------------------------------------------------------------------------------
Var variance(const Var& a, const std::vector<int64_t>& dim) {
  // Variance: varAndMean(dim, unbiased, keepDim); out.addcmulSelf(p, a - mean, 2 / (SUM of the reduced sizes - 1)) (as written)
  auto op = new_op("Variance");
  lamp_tensor *v = nullptr, *m = nullptr;
  HCALL(lamp_var_mean_dims(&v, &m, a->value.h(), dim.data(), (int)dim.size(), 1, 1));
  Ten var(v), mean(m), av = a->value;
  int64_t ssum = 0;
  for (auto d : dim) ssum += av.size((int)d);
  op->params.push_back({a, [av, mean, ssum](const Ten& p, Variable& out) { out.addcmul(p, ops::sub(av, mean), 2.0 / (double)(ssum - 1)); }});
  return make_result(op, var);
}
Var squared_frobenius(const Var& a) {                                 // SquaredFrobeniusMatrixNorm: frobeniusNorm([-2, -1]).pow_(2)
  auto op = new_op("SquaredFrobeniusMatrixNorm");
  Ten av = a->value;
  op->params.push_back({a, [av](const Ten& p, Variable& out) { out.addcmul(p, av, 2.0); }});
  const int nd = av.ndim();
  Ten fr = ops::norm2_dims(av, {(int64_t)nd - 2, (int64_t)nd - 1}, false);
  return make_result(op, ops::pow_scalar(fr, 2.0));
}
Var weight_norm(const Var& v, const Var& g, int64_t dim) {            // WeightNorm (ops.scala:1103-1160), arXiv 1602.07868 eq. 2, 3
  LAMP_CHECK(v->value.ndim() == 2, "assertion failed: WeightNorm: v should have 2 dimensions");
  LAMP_CHECK(g->shape() == (std::vector<int64_t>{1, v->value.size(1)}), "assertion failed: WeightNorm: g should have dimensions 1 x a where a is the second dimension of v.");
  auto op = new_op("WeightNorm");
  Ten vv = v->value, gv = g->value;
  Ten norm = ops::norm2_dims(vv, {dim}, false);
  auto gradg = [vv, norm](const Ten& p) { Ten t = ops::sum_dims(ops::mul(p, vv), {0}, false); ops::div_(t, norm); return t; };
  op->params.push_back({v, [vv, gv, norm, gradg](const Ten& p, Variable& out) {
    Ten tmp3 = ops::mul(ops::div(gv, norm), p);
    Ten tmp2 = ops::mul(gv, gradg(p));
    ops::div_(tmp2, norm);
    ops::div_(tmp2, norm);
    Ten tmp4 = ops::mul(tmp2, vv);
    out.accumulate(ops::add(tmp3, tmp4, -1.0), true);
  }});
  op->params.push_back({g, [gradg](const Ten& p, Variable& out) {
    Ten t = gradg(p);
    out.accumulate(t.shape() == out.shape() ? t : ops::reshape(t, out.shape()), true);
  }});
  Ten w = ops::mul(vv, gv);
  ops::div_(w, norm);
  return make_result(op, w);
}

// ---- losses (ops.scala:1207-1247, 1309-1367) -------------------------------------------------------------------------------------------
Var smooth_l1_loss(const Var& input, const Ten& target, int64_t reduction, double beta) {
  LAMP_CHECK(input->value.numel() == target.numel(), "assertion failed: input.value.numel == target.numel");
  auto op = new_op("SmoothL1Loss");
  Ten xv = input->value, tv = ops::view(target, input->shape());
  op->params.push_back({input, [xv, tv, reduction, beta](const Ten& p, Variable& out) {
    lamp_tensor* o = nullptr;
    HCALL(lamp_smooth_l1_loss_backward(&o, p.h(), xv.h(), tv.h(), reduction, beta));
    out.accumulate(Ten(o), true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_smooth_l1_loss(&o, xv.h(), tv.h(), reduction, beta));
  return make_result(op, Ten(o));
}
Var binary_cross_entropy_with_logits(const Var& input, const Ten& target, const Ten& posWeights, int64_t reduction) {
  LAMP_CHECK(input->shape() == target.shape(), "assertion failed: BinaryCrossEntropyWithLogitsLoss input and target have the same shape.");
  auto op = new_op("BinaryCrossEntropyWithLogitsLoss");
  Ten xv = input->value;
  op->params.push_back({input, [xv, target, posWeights, reduction](const Ten& p, Variable& out) {
    // -[pos y (1 - sigmoid(x)) - (1 - y) sigmoid(x)] * grad, composed as the reference composes it
    Ten t;
    if (posWeights.defined()) {
      Ten pt = ops::mul(posWeights, target);
      t = ops::add_scalar(pt, 1.0);
      ops::sub_(t, target);
      ops::mul_(t, ops::sigmoid(xv));
      ops::sub_(t, pt);
    } else {
      t = ops::sigmoid(xv);
      ops::sub_(t, target);
    }
    ops::mul_(t, p);
    if (reduction == 1) ops::mul_scalar_(t, 1.0 / (double)xv.numel());
    out.accumulate(t, true);
  }});
  lamp_tensor* o = nullptr;
  HCALL(lamp_binary_cross_entropy_with_logits(&o, xv.h(), target.h(), posWeights.defined() ? posWeights.h() : nullptr, reduction));
  return make_result(op, Ten(o));
}

// ---- MaxPool1D (ops.scala:1658-1715) ---------------------------------------------------------------------------------------------------
Var max_pool1d(const Var& input, int64_t k, int64_t stride, int64_t padding, int64_t dilation) {
  LAMP_CHECK(input->value.ndim() == 3, "assertion failed: Input dimensions must be 3");
  auto op = new_op("MaxPool1D");
  lamp_tensor *o = nullptr, *idx = nullptr;
  HCALL(lamp_max_pool1d_with_indices(&o, &idx, input->value.h(), k, stride, padding, dilation, 0));
  Ten mask(idx), xv = input->value;
  // the reference index_adds p into zeros row by row with the positions of the maxima; one gather-form kernel gives the same sums
  op->params.push_back({input, [xv, mask, k, stride, padding, dilation](const Ten& p, Variable& out) {
    lamp_tensor* dx = nullptr;
    HCALL(lamp_max_pool1d_with_indices_backward(&dx, p.h(), xv.h(), k, stride, padding, dilation, 0, mask.h()));
    out.accumulate(Ten(dx), true);
  }});
  return make_result(op, Ten(o));
}

}  // namespace F
}  // namespace host
}  // namespace lamp
