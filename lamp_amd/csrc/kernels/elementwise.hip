// Broadcasting element-wise kernels (HBM-bound: one pass, 16-byte loads/stores per lane on the
// contiguous path, a generic strided path for views and broadcasts).
//
// Replaces the ATen calls behind lamp's arithmetic (reference call sites:
// lamp-sten/src/main/scala/lamp/STen.scala:365-468 (out variants), 1110-1217 (+,-,*,/ and in
// place), 1242-1264 (addcmul), 1266-1320 (unary); backward closures in
// lamp-core/src/main/scala/lamp/autograd/ops.scala:511-621, 754-1032).
//
// bf16 math is done in f32 and rounded once on store, f64 stays f64 - the same evaluation
// ATen's CPU kernels use, so results agree to rounding with the oracle.
#include "device_utils.h"
#include "../core/strided.h"

namespace lamp {

// ---- kernels -----------------------------------------------------------------------------------
// contiguous path: every operand has stride 1, or stride 0 (a broadcast one-element operand)
template <class TO, class TI, int NIN, class F>
__global__ __launch_bounds__(256) void ew_vec_kernel(TO* __restrict__ out, const TI* in0, const TI* in1, const TI* in2,
                                                     int64_t n, int bcast_mask, F f) {
  using A = acc_t<TI>;
  constexpr int W = 16 / sizeof(TI);
  const int64_t nvec = n / W;
  A s0 = A(0), s1 = A(0), s2 = A(0);
  if (bcast_mask & 1) s0 = load_as<A>(in0[0]);
  if (NIN > 1 && (bcast_mask & 2)) s1 = load_as<A>(in1[0]);
  if (NIN > 2 && (bcast_mask & 4)) s2 = load_as<A>(in2[0]);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    Vec<TI, W> v0, v1, v2;
    if (!(bcast_mask & 1)) v0 = *reinterpret_cast<const Vec<TI, W>*>(in0 + i * W);
    if (NIN > 1 && !(bcast_mask & 2)) v1 = *reinterpret_cast<const Vec<TI, W>*>(in1 + i * W);
    if (NIN > 2 && !(bcast_mask & 4)) v2 = *reinterpret_cast<const Vec<TI, W>*>(in2 + i * W);
    Vec<TO, W> r;
#pragma unroll
    for (int k = 0; k < W; k++) {
      A a = (bcast_mask & 1) ? s0 : load_as<A>(v0.v[k]);
      A b = (NIN > 1) ? ((bcast_mask & 2) ? s1 : load_as<A>(v1.v[k])) : A(0);
      A c = (NIN > 2) ? ((bcast_mask & 4) ? s2 : load_as<A>(v2.v[k])) : A(0);
      r.v[k] = f.template apply<TO>(a, b, c);
    }
    *reinterpret_cast<Vec<TO, W>*>(out + i * W) = r;
  }
  // tail
  const int64_t tail0 = nvec * W;
  const int64_t t = tail0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) {
    A a = (bcast_mask & 1) ? s0 : load_as<A>(in0[t]);
    A b = (NIN > 1) ? ((bcast_mask & 2) ? s1 : load_as<A>(in1[t])) : A(0);
    A c = (NIN > 2) ? ((bcast_mask & 4) ? s2 : load_as<A>(in2[t])) : A(0);
    out[t] = f.template apply<TO>(a, b, c);
  }
}

// [R, N] iteration with a contiguous output where every input is dense, a row vector broadcast over the rows (bias / scale of a
// token batch) or one element: 16-byte packets, the row vector re-read from L2.  kind: 2 bits per input (0 dense, 1 row vector, 2 scalar)
template <class TO, class TI, int NIN, class F>
__global__ __launch_bounds__(256) void ew_rowvec_kernel(TO* __restrict__ out, const TI* in0, const TI* in1, const TI* in2, int64_t npk, int ppr,
                                                        int kind, F f) {
  using A = acc_t<TI>;
  constexpr int W = 16 / sizeof(TI);
  const int k0 = kind & 3, k1 = (kind >> 2) & 3, k2 = (kind >> 4) & 3;
  A s0 = A(0), s1 = A(0), s2 = A(0);
  if (k0 == 2) s0 = load_as<A>(in0[0]);
  if (NIN > 1 && k1 == 2) s1 = load_as<A>(in1[0]);
  if (NIN > 2 && k2 == 2) s2 = load_as<A>(in2[0]);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npk; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cp = i % ppr;
    Vec<TI, W> v0, v1, v2;
    if (k0 != 2) v0 = *reinterpret_cast<const Vec<TI, W>*>(in0 + (k0 == 0 ? i : cp) * W);
    if (NIN > 1 && k1 != 2) v1 = *reinterpret_cast<const Vec<TI, W>*>(in1 + (k1 == 0 ? i : cp) * W);
    if (NIN > 2 && k2 != 2) v2 = *reinterpret_cast<const Vec<TI, W>*>(in2 + (k2 == 0 ? i : cp) * W);
    Vec<TO, W> r;
#pragma unroll
    for (int k = 0; k < W; k++) {
      A a = k0 == 2 ? s0 : load_as<A>(v0.v[k]);
      A b = (NIN > 1) ? (k1 == 2 ? s1 : load_as<A>(v1.v[k])) : A(0);
      A c = (NIN > 2) ? (k2 == 2 ? s2 : load_as<A>(v2.v[k])) : A(0);
      r.v[k] = f.template apply<TO>(a, b, c);
    }
    *reinterpret_cast<Vec<TO, W>*>(out + i * W) = r;
  }
}

template <class TO, class TI, int NIN, class F>
__global__ __launch_bounds__(256) void ew_strided_kernel(TO* __restrict__ out, const TI* in0, const TI* in1, const TI* in2,
                                                         int64_t n, IterArgs it, F f) {
  using A = acc_t<TI>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t off[4];
    iter_offsets<4>(it, i, off);
    A a = load_as<A>(in0[off[1]]);
    A b = (NIN > 1) ? load_as<A>(in1[off[2]]) : A(0);
    A c = (NIN > 2) ? load_as<A>(in2[off[3]]) : A(0);
    out[off[0]] = f.template apply<TO>(a, b, c);
  }
}

// ---- launcher ----------------------------------------------------------------------------------
template <class TO, class TI, int NIN, class F>
void launch_ew(Tensor* out, const Tensor* a, const Tensor* b, const Tensor* c, F f) {
  const Tensor* ops[4] = {out, a, b ? b : a, c ? c : a};
  IterSpace it = make_iter(out->shape(), ops, 4);
  int64_t n = it.numel;
  if (n == 0) return;
  if (!out->is_device()) {
    // lamp's CPU device (device.scala:138): tensors that live in host memory compute where they live, with the functors the kernels
    // use - one scalar loop over the broadcast index space.  (A staging-speed path for the small host tensors lamp creates - scalars,
    // class weights, index lists, test fixtures; models run on the GPU.)
    TO* po = out->ptr<TO>();
    const TI* h0 = a->ptr<TI>();
    const TI* h1 = b ? b->ptr<TI>() : nullptr;
    const TI* h2 = c ? c->ptr<TI>() : nullptr;
    const IterArgs ia = to_args(it);
    using A = acc_t<TI>;
    for (int64_t i = 0; i < n; i++) {
      int64_t off[4];
      iter_offsets<4>(ia, i, off);
      const A av = load_as<A>(h0[off[1]]);
      const A bv = (NIN > 1) ? load_as<A>(h1[off[2]]) : A(0);
      const A cv = (NIN > 2) ? load_as<A>(h2[off[3]]) : A(0);
      po[off[0]] = f.template apply<TO>(av, bv, cv);
    }
    return;
  }
  hipStream_t st = current_stream(out->device());
  const TI* p0 = a->ptr<TI>();
  const TI* p1 = b ? b->ptr<TI>() : nullptr;
  const TI* p2 = c ? c->ptr<TI>() : nullptr;
  // vector path eligibility
  bool vec_ok = it.ndim == 1 && it.strides[0][0] == 1;
  int mask = 0;
  if (vec_ok) {
    for (int o = 1; o <= NIN; o++) {
      int64_t s = it.strides[o][0];
      if (s == 0) mask |= 1 << (o - 1);
      else if (s != 1) vec_ok = false;
    }
    auto aligned = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!aligned(out->data())) vec_ok = false;
    if (!(mask & 1) && !aligned(p0)) vec_ok = false;
    if (NIN > 1 && !(mask & 2) && !aligned(p1)) vec_ok = false;
    if (NIN > 2 && !(mask & 4) && !aligned(p2)) vec_ok = false;
  }
  if (n == 1) vec_ok = false;
  // row-vector broadcast over a dense [R, N] output
  bool row_ok = false;
  int kind = 0;
  constexpr int WV = 16 / sizeof(TI);
  if (!vec_ok && sizeof(TO) == sizeof(TI) && it.ndim == 2 && it.strides[0][1] == 1 && it.strides[0][0] == it.sizes[1] && it.sizes[1] % WV == 0 &&
      it.sizes[1] / WV < (1ll << 31)) {
    row_ok = ((uintptr_t)out->data() & 15) == 0;
    const void* ps[3] = {p0, p1, p2};
    for (int o = 1; o <= NIN && row_ok; o++) {
      const int64_t s0 = it.strides[o][0], s1 = it.strides[o][1];
      int kd;
      if (s0 == it.sizes[1] && s1 == 1) kd = 0;
      else if (s0 == 0 && s1 == 1) kd = 1;
      else if (s0 == 0 && s1 == 0) kd = 2;
      else { row_ok = false; break; }
      if (kd != 2 && ((uintptr_t)ps[o - 1] & 15) != 0) row_ok = false;
      kind |= kd << (2 * (o - 1));
    }
  }
  KernelTimer kt("elementwise", 0, (double)n * (NIN * sizeof(TI) + sizeof(TO)), st);
  if (row_ok) {
    const int64_t npk = n / WV;
    hipLaunchKernelGGL((ew_rowvec_kernel<TO, TI, NIN, F>), dim3(grid_for(npk, 256)), dim3(256), 0, st, out->ptr<TO>(), p0, p1, p2, npk,
                       (int)(it.sizes[1] / WV), kind, f);
  } else if (vec_ok) {
    constexpr int W = 16 / sizeof(TI);
    int grid = grid_for((n + W - 1) / W, 256);
    hipLaunchKernelGGL((ew_vec_kernel<TO, TI, NIN, F>), dim3(grid), dim3(256), 0, st, out->ptr<TO>(), p0, p1, p2, n, mask, f);
  } else {
    int grid = grid_for(n, 256);
    hipLaunchKernelGGL((ew_strided_kernel<TO, TI, NIN, F>), dim3(grid), dim3(256), 0, st, out->ptr<TO>(), p0, p1, p2, n,
                       to_args(it), f);
  }
  LAMP_LAUNCH_CHECK();
}

// ---- functors ----------------------------------------------------------------------------------
#define FUNCTOR_BEGIN(NAME) struct NAME { double p0 = 0, p1 = 0; template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A a, A b, A c) const {
#define FUNCTOR_END } };

template <class A> __host__ __device__ __forceinline__ A dexp(A x);
template <> __host__ __device__ __forceinline__ float dexp(float x) { return expf(x); }
template <> __host__ __device__ __forceinline__ double dexp(double x) { return exp(x); }
template <> __host__ __device__ __forceinline__ int64_t dexp(int64_t x) { return (int64_t)exp((double)x); }

#define MATH1(NAME, FF, DF)                                                                       \
  template <class A> __host__ __device__ __forceinline__ A NAME(A x) { return (A)DF((double)x); }          \
  template <> __host__ __device__ __forceinline__ float NAME(float x) { return FF(x); }                    \
  template <> __host__ __device__ __forceinline__ double NAME(double x) { return DF(x); }
MATH1(m_log, logf, log)
MATH1(m_log1p, log1pf, log1p)
MATH1(m_sqrt, sqrtf, sqrt)
MATH1(m_sin, sinf, sin)
MATH1(m_cos, cosf, cos)
MATH1(m_tan, tanf, tan)
MATH1(m_atan, atanf, atan)
MATH1(m_tanh, tanhf, tanh)
MATH1(m_erf, erff, erf)
MATH1(m_fabs, fabsf, fabs)
MATH1(m_acos, acosf, acos)
MATH1(m_asin, asinf, asin)
MATH1(m_ceil, ceilf, ceil)
MATH1(m_floor, floorf, floor)
MATH1(m_rint, rintf, rint)
MATH1(m_expm1, expm1f, expm1)
MATH1(m_log10, log10f, log10)
template <class A> __host__ __device__ __forceinline__ A m_atan2(A y, A x) { return (A)atan2((double)y, (double)x); }
template <> __host__ __device__ __forceinline__ float m_atan2(float y, float x) { return atan2f(y, x); }
template <class A> __host__ __device__ __forceinline__ A m_fmod(A x, A y) { return (A)fmod((double)x, (double)y); }
template <> __host__ __device__ __forceinline__ float m_fmod(float x, float y) { return fmodf(x, y); }
// ATen remainder: the result takes the sign of the divisor (Python's %)
template <class A> __host__ __device__ __forceinline__ A m_remainder(A a, A b) { A r = m_fmod<A>(a, b); if (r != A(0) && ((r < A(0)) != (b < A(0)))) r += b; return r; }
template <class A> __host__ __device__ __forceinline__ A m_pow(A x, A y) { return (A)pow((double)x, (double)y); }
template <> __host__ __device__ __forceinline__ float m_pow(float x, float y) { return powf(x, y); }

FUNCTOR_BEGIN(FAdd) return store_as<TO>((A)(a + (A)p0 * b)); FUNCTOR_END
FUNCTOR_BEGIN(FSub) return store_as<TO>((A)(a - (A)p0 * b)); FUNCTOR_END
FUNCTOR_BEGIN(FMul) return store_as<TO>((A)(a * b)); FUNCTOR_END
FUNCTOR_BEGIN(FDiv) return store_as<TO>((A)(a / b)); FUNCTOR_END
FUNCTOR_BEGIN(FMax) return store_as<TO>((A)((a > b || a != a) ? a : b)); FUNCTOR_END
FUNCTOR_BEGIN(FMin) return store_as<TO>((A)((a < b || a != a) ? a : b)); FUNCTOR_END
FUNCTOR_BEGIN(FPow) return store_as<TO>(m_pow<A>(a, b)); FUNCTOR_END
FUNCTOR_BEGIN(FAddS) return store_as<TO>((A)(a + (A)(p0 * p1))); FUNCTOR_END       // a + alpha*scalar (p0 scalar, p1 alpha)
FUNCTOR_BEGIN(FMulS) return store_as<TO>((A)(a * (A)p0)); FUNCTOR_END
FUNCTOR_BEGIN(FDivS) return store_as<TO>((A)(a / (A)p0)); FUNCTOR_END
struct FPowS {
  double p0 = 0, p1 = 0;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A a, A b, A c) const {
    // the special cases ATen's CPU pow kernel takes (exact for 2, 3, 0.5, -1, -2, -0.5)
    if (p0 == 2.0) return store_as<TO>((A)(a * a));
    if (p0 == 3.0) return store_as<TO>((A)(a * a * a));
    if (p0 == 0.5) return store_as<TO>(m_sqrt<A>(a));
    if (p0 == 1.0) return store_as<TO>(a);
    if (p0 == -1.0) return store_as<TO>((A)(A(1) / a));
    if (p0 == -2.0) return store_as<TO>((A)(A(1) / (a * a)));
    if (p0 == -0.5) return store_as<TO>((A)(A(1) / m_sqrt<A>(a)));
    return store_as<TO>(m_pow<A>(a, (A)p0));
  }
};
FUNCTOR_BEGIN(FAddcmul) return store_as<TO>((A)(a + (A)p0 * b * c)); FUNCTOR_END
// (a * b) + c with the product rounded to the tensor type first: bit for bit the chain Mult then Add (no fma contraction)
struct FMulAdd {
  double p0 = 0, p1 = 0;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A a, A b, A c) const {
#pragma clang fp contract(off)
    const A prod = load_as<A>(store_as<TO>((A)(a * b)));
    return store_as<TO>((A)(prod + c));
  }
};
FUNCTOR_BEGIN(FAddcdiv) return store_as<TO>((A)(a + (A)p0 * b / c)); FUNCTOR_END
// out(a) += p(b) * (x(c) < 0 ? slope : 1)      ops.scala:918-953
FUNCTOR_BEGIN(FReluBwdAcc) return store_as<TO>((A)(a + b * ((c < A(0)) ? (A)p0 : A(1)))); FUNCTOR_END

FUNCTOR_BEGIN(FReluBwd) return store_as<TO>((A)(a * ((b < A(0)) ? (A)p0 : A(1)))); FUNCTOR_END   // p * (x < 0 ? slope : 1)
FUNCTOR_BEGIN(FRelu) return store_as<TO>((A)((a < A(0)) ? A(0) : a)); FUNCTOR_END   // NaN propagates like ATen's relu
FUNCTOR_BEGIN(FLeakyRelu) return store_as<TO>((A)((a > A(0)) ? a : a * (A)p0)); FUNCTOR_END
FUNCTOR_BEGIN(FGelu) return store_as<TO>((A)(a * A(0.5) * (A(1) + m_erf<A>(a * A(0.70710678118654752440))))); FUNCTOR_END
struct FGeluBwd {  // (grad, self)
  double p0 = 0, p1 = 0;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A g, A x, A c) const {
    const A kAlpha = A(0.70710678118654752440), kBeta = A(0.39894228040143267794);  // 1/sqrt(2), 1/sqrt(2 pi)
    A cdf = A(0.5) * (A(1) + m_erf<A>(x * kAlpha));
    A pdf = kBeta * dexp<A>(x * x * A(-0.5));
    return store_as<TO>((A)(g * (cdf + x * pdf)));
  }
};
FUNCTOR_BEGIN(FSigmoid) return store_as<TO>((A)(A(1) / (A(1) + dexp<A>(-a)))); FUNCTOR_END
FUNCTOR_BEGIN(FSigmoidBwd) return store_as<TO>((A)(a * (A(1) - b) * b)); FUNCTOR_END   // (grad, output)
FUNCTOR_BEGIN(FTanh) return store_as<TO>(m_tanh<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FTanhBwd) return store_as<TO>((A)(a * (A(1) - b * b))); FUNCTOR_END       // (grad, output)
struct FHardswish {
  double p0 = 0, p1 = 0;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A x, A b, A c) const {
    A t = x + A(3);
    t = t < A(0) ? A(0) : (t > A(6) ? A(6) : t);
    return store_as<TO>((A)(x * t / A(6)));
  }
};
struct FHardswishBwd {  // (grad, self)
  double p0 = 0, p1 = 0;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A g, A x, A c) const {
    if (x < A(-3)) return store_as<TO>(A(0));
    if (x <= A(3)) return store_as<TO>((A)(g * ((x / A(3)) + A(0.5))));
    return store_as<TO>(g);
  }
};
struct FSoftplus {  // p0 beta, p1 threshold
  double p0 = 1, p1 = 20;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A x, A b, A c) const {
    A xb = x * (A)p0;
    return store_as<TO>((A)((xb > (A)p1) ? x : m_log1p<A>(dexp<A>(xb)) / (A)p0));
  }
};
struct FSoftplusBwd {  // (grad, self)
  double p0 = 1, p1 = 20;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A g, A x, A c) const {
    A xb = x * (A)p0;
    A z = dexp<A>(xb);
    return store_as<TO>((A)((xb > (A)p1) ? g : g * z / (z + A(1))));
  }
};
FUNCTOR_BEGIN(FExp) return store_as<TO>(dexp<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FLog) return store_as<TO>(m_log<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FLog1p) return store_as<TO>(m_log1p<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FSqrt) return store_as<TO>(m_sqrt<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FSquare) return store_as<TO>((A)(a * a)); FUNCTOR_END
FUNCTOR_BEGIN(FRecip) return store_as<TO>((A)(A(1) / a)); FUNCTOR_END
FUNCTOR_BEGIN(FNeg) return store_as<TO>((A)(-a)); FUNCTOR_END
FUNCTOR_BEGIN(FAbs) return store_as<TO>((A)(a < A(0) ? -a : a)); FUNCTOR_END
FUNCTOR_BEGIN(FSign) return store_as<TO>((A)((A(0) < a) - (a < A(0)))); FUNCTOR_END
FUNCTOR_BEGIN(FSin) return store_as<TO>(m_sin<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FCos) return store_as<TO>(m_cos<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FTan) return store_as<TO>(m_tan<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FAtan) return store_as<TO>(m_atan<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FLogicalNot) return (TO)(a == A(0)); FUNCTOR_END
FUNCTOR_BEGIN(FAcos) return store_as<TO>(m_acos<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FAsin) return store_as<TO>(m_asin<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FCeil) return store_as<TO>(m_ceil<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FFloor) return store_as<TO>(m_floor<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FRound) return store_as<TO>(m_rint<A>(a)); FUNCTOR_END                // ATen round: half to even
FUNCTOR_BEGIN(FExpm1) return store_as<TO>(m_expm1<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FLog10) return store_as<TO>(m_log10<A>(a)); FUNCTOR_END
FUNCTOR_BEGIN(FAtan2) return store_as<TO>(m_atan2<A>(a, b)); FUNCTOR_END
FUNCTOR_BEGIN(FRemainder) return store_as<TO>(m_remainder<A>(a, b)); FUNCTOR_END
FUNCTOR_BEGIN(FRemainderS) return store_as<TO>(m_remainder<A>(a, (A)p0)); FUNCTOR_END
FUNCTOR_BEGIN(FLogicalAnd) return (TO)((a != A(0)) && (b != A(0))); FUNCTOR_END
FUNCTOR_BEGIN(FLogicalOr) return (TO)((a != A(0)) || (b != A(0))); FUNCTOR_END
FUNCTOR_BEGIN(FLogicalXor) return (TO)((a != A(0)) != (b != A(0))); FUNCTOR_END
FUNCTOR_BEGIN(FIsNan) return (TO)(a != a); FUNCTOR_END
FUNCTOR_BEGIN(FIsFinite) return (TO)((a - a) == A(0)); FUNCTOR_END
struct FNanToNum {   // nan_to_num(nan, posinf, neginf): p0 = nan replacement; infinities -> largest finite values of the type (ATen defaults)
  double p0 = 0, p1 = 0;
  template <class TO, class A> __host__ __device__ __forceinline__ TO apply(A a, A b, A c) const {
    if (a != a) return store_as<TO>((A)p0);
    if ((a - a) != A(0)) { const A big = sizeof(TO) == 8 ? (A)1.7976931348623157e308 : (sizeof(TO) == 4 ? (A)3.4028234663852886e38 : (A)p1); return store_as<TO>(a > A(0) ? big : -big); }
    return store_as<TO>(a);
  }
};
// comparisons: TO = uint8_t
FUNCTOR_BEGIN(FLt) return (TO)(a < b); FUNCTOR_END
FUNCTOR_BEGIN(FLe) return (TO)(a <= b); FUNCTOR_END
FUNCTOR_BEGIN(FGt) return (TO)(a > b); FUNCTOR_END
FUNCTOR_BEGIN(FGe) return (TO)(a >= b); FUNCTOR_END
FUNCTOR_BEGIN(FEq) return (TO)(a == b); FUNCTOR_END
FUNCTOR_BEGIN(FNe) return (TO)(a != b); FUNCTOR_END
FUNCTOR_BEGIN(FLtS) return (TO)(a < (A)p0); FUNCTOR_END
FUNCTOR_BEGIN(FLeS) return (TO)(a <= (A)p0); FUNCTOR_END
FUNCTOR_BEGIN(FGtS) return (TO)(a > (A)p0); FUNCTOR_END
FUNCTOR_BEGIN(FGeS) return (TO)(a >= (A)p0); FUNCTOR_END
FUNCTOR_BEGIN(FEqS) return (TO)(a == (A)p0); FUNCTOR_END
FUNCTOR_BEGIN(FNeS) return (TO)(a != (A)p0); FUNCTOR_END

// ---- host helpers ------------------------------------------------------------------------------
static void check_inputs(const Tensor* a, const Tensor* b, const Tensor* c) {
  LAMP_CHECK(a != nullptr, "input is null");
  if (!a->is_device()) {     // all operands on the host: lamp's CPU device (see launch_ew)
    if (b) { LAMP_CHECK(!b->is_device(), "tensors on different devices: " << a->describe() << " vs " << b->describe());
             LAMP_CHECK(a->dtype == b->dtype, "dtype mismatch: " << a->describe() << " vs " << b->describe()); }
    if (c) { LAMP_CHECK(!c->is_device(), "tensors on different devices: " << a->describe() << " vs " << c->describe());
             LAMP_CHECK(a->dtype == c->dtype, "dtype mismatch: " << a->describe() << " vs " << c->describe()); }
    return;
  }
  check_device_tensor(a, "input");
  if (b) { check_device_tensor(b, "input"); check_same_device(a, b);
           LAMP_CHECK(a->dtype == b->dtype, "dtype mismatch: " << a->describe() << " vs " << b->describe()); }
  if (c) { check_device_tensor(c, "input"); check_same_device(a, c);
           LAMP_CHECK(a->dtype == c->dtype, "dtype mismatch: " << a->describe() << " vs " << c->describe()); }
}
static std::vector<int64_t> out_shape(const Tensor* a, const Tensor* b, const Tensor* c) {
  std::vector<int64_t> s = a->shape();
  if (b) s = broadcast_shapes(s, b->shape());
  if (c) s = broadcast_shapes(s, c->shape());
  return s;
}

// same-dtype output; `out` == nullptr allocates
template <int NIN, class F, bool FLOAT_ONLY>
Tensor* run_same(Tensor* out, const Tensor* a, const Tensor* b, const Tensor* c, F f) {
  check_inputs(a, b, c);
  auto shape = out_shape(a, b, c);
  Hold owned;
  if (!out) { owned = Hold(new_tensor(shape, a->dtype, a->device())); out = owned.get(); }
  else {
    LAMP_CHECK(out->is_device() == a->is_device() && (!out->is_device() || out->device() == a->device()), "out " << out->describe() << " is not on the device of the inputs");
    LAMP_CHECK(out->shape() == shape, "out " << out->describe() << " does not match the broadcast shape of the inputs");
    LAMP_CHECK(out->dtype == a->dtype, "out dtype mismatch: " << out->describe() << " vs " << a->describe());
  }
  if (FLOAT_ONLY) {
    LAMP_DISPATCH_FLOAT(a->dtype, T, (launch_ew<T, T, NIN, F>(out, a, b, c, f)));
  } else {
    LAMP_DISPATCH_ALL(a->dtype, T, (launch_ew<T, T, NIN, F>(out, a, b, c, f)));
  }
  return owned.get() ? owned.take() : out;
}
template <int NIN, class F> Tensor* run_bool(const Tensor* a, const Tensor* b, F f) {
  check_inputs(a, b, nullptr);
  auto shape = out_shape(a, b, nullptr);
  Hold out(new_tensor(shape, kBool, a->device()));
  LAMP_DISPATCH_ALL(a->dtype, T, (launch_ew<uint8_t, T, NIN, F>(out.get(), a, b, nullptr, f)));
  return out.take();
}

// ---- where / masked_fill (mixed dtypes: bool condition) ----------------------------------------
template <class T>
__global__ void where_kernel(T* out, const uint8_t* cond, const T* a, const T* b, int64_t n, IterArgs it) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t off[4];
    iter_offsets<4>(it, i, off);
    out[off[0]] = cond[off[1]] ? a[off[2]] : b[off[3]];
  }
}
template <class T>
__global__ void masked_fill_kernel(T* out, const T* a, const uint8_t* mask, T value, int64_t n, IterArgs it) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t off[3];
    iter_offsets<3>(it, i, off);
    out[off[0]] = mask[off[2]] ? value : a[off[1]];
  }
}

// acc[i] += scale * x[i] with acc and x of DIFFERENT floating types, the product and sum taken in f64: the epoch-loss accumulator of
// the training loops is an f64 scalar whatever the model's type (`STen.scalarDouble(0, options)`, IOLoops.scala:715;
// `acc += loss * numInstances`, SupervisedModel.scala:207).  A bf16 accumulator would round every batch to 8 bits.
template <class TA, class TX>
__global__ void add_scaled_mixed_kernel(TA* __restrict__ acc, const TX* __restrict__ x, double scale, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc[i] = store_as<TA>((acc_t<TA>)(load_as<double>(acc[i]) + scale * load_as<double>(x[i])));
}

}  // namespace lamp

using namespace lamp;

#define API1(NAME, F, FLOAT_ONLY)                                                                  \
  int lamp_##NAME(lamp_tensor** out, const lamp_tensor* a) {                                       \
    LAMP_API_BEGIN *out = run_same<1, F, FLOAT_ONLY>(nullptr, a, nullptr, nullptr, F{}); LAMP_API_END \
  }
#define API1_INPLACE(NAME, F, FLOAT_ONLY)                                                          \
  int lamp_##NAME(lamp_tensor* a) {                                                                \
    LAMP_API_BEGIN run_same<1, F, FLOAT_ONLY>(a, a, nullptr, nullptr, F{}); LAMP_API_END           \
  }
#define API2(NAME, F, FLOAT_ONLY)                                                                  \
  int lamp_##NAME(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b) {                 \
    LAMP_API_BEGIN *out = run_same<2, F, FLOAT_ONLY>(nullptr, a, b, nullptr, F{}); LAMP_API_END    \
  }
#define API_CMP(NAME, F)                                                                           \
  int lamp_##NAME(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b) {                 \
    LAMP_API_BEGIN *out = run_bool<2, F>(a, b, F{}); LAMP_API_END                                  \
  }
#define API_CMP_S(NAME, F)                                                                         \
  int lamp_##NAME(lamp_tensor** out, const lamp_tensor* a, double b) {                             \
    LAMP_API_BEGIN *out = run_bool<1, F>(a, nullptr, F{b, 0}); LAMP_API_END                        \
  }

extern "C" {

int lamp_add(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, double alpha) {
  LAMP_API_BEGIN *out = run_same<2, FAdd, false>(nullptr, a, b, nullptr, FAdd{alpha, 0}); LAMP_API_END
}
int lamp_sub(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, double alpha) {
  LAMP_API_BEGIN *out = run_same<2, FSub, false>(nullptr, a, b, nullptr, FSub{alpha, 0}); LAMP_API_END
}
API2(mul, FMul, false)
API2(div, FDiv, false)
API2(maximum, FMax, false)
API2(minimum, FMin, false)
API2(pow_tensor, FPow, true)
int lamp_add_scalar(lamp_tensor** out, const lamp_tensor* a, double b, double alpha) {
  LAMP_API_BEGIN *out = run_same<1, FAddS, false>(nullptr, a, nullptr, nullptr, FAddS{b, alpha}); LAMP_API_END
}
int lamp_sub_scalar(lamp_tensor** out, const lamp_tensor* a, double b, double alpha) {
  LAMP_API_BEGIN *out = run_same<1, FAddS, false>(nullptr, a, nullptr, nullptr, FAddS{-b, alpha}); LAMP_API_END
}
int lamp_mul_scalar(lamp_tensor** out, const lamp_tensor* a, double b) {
  LAMP_API_BEGIN *out = run_same<1, FMulS, false>(nullptr, a, nullptr, nullptr, FMulS{b, 0}); LAMP_API_END
}
int lamp_div_scalar(lamp_tensor** out, const lamp_tensor* a, double b) {
  LAMP_API_BEGIN *out = run_same<1, FDivS, false>(nullptr, a, nullptr, nullptr, FDivS{b, 0}); LAMP_API_END
}
int lamp_pow_scalar(lamp_tensor** out, const lamp_tensor* a, double e) {
  LAMP_API_BEGIN *out = run_same<1, FPowS, true>(nullptr, a, nullptr, nullptr, FPowS{e, 0}); LAMP_API_END
}
int lamp_add_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b, double alpha) {
  LAMP_API_BEGIN run_same<2, FAdd, false>(out, a, b, nullptr, FAdd{alpha, 0}); LAMP_API_END
}
int lamp_sub_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b, double alpha) {
  LAMP_API_BEGIN run_same<2, FSub, false>(out, a, b, nullptr, FSub{alpha, 0}); LAMP_API_END
}
int lamp_mul_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN run_same<2, FMul, false>(out, a, b, nullptr, FMul{}); LAMP_API_END
}
int lamp_div_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN run_same<2, FDiv, false>(out, a, b, nullptr, FDiv{}); LAMP_API_END
}
int lamp_add_(lamp_tensor* self, const lamp_tensor* b, double alpha) { return lamp_add_out(self, self, b, alpha); }
int lamp_add_scaled_mixed_(lamp_tensor* acc, const lamp_tensor* x, double scale) {
  LAMP_API_BEGIN
  check_device_tensor(acc, "acc"); check_device_tensor(x, "x"); check_same_device(acc, x);
  LAMP_CHECK(acc->numel() == x->numel() && acc->is_contiguous() && x->is_contiguous(),
             "acc " << acc->describe() << " and x " << x->describe() << " must be contiguous with the same number of elements");
  const int64_t n = acc->numel();
  if (n == 0) return 0;
  hipStream_t st = current_stream(acc->device());
  LAMP_DISPATCH_FLOAT(acc->dtype, TA, LAMP_DISPATCH_FLOAT(x->dtype, TX,
      hipLaunchKernelGGL((add_scaled_mixed_kernel<TA, TX>), dim3(grid_for(n, 256)), dim3(256), 0, st, acc->ptr<TA>(), x->ptr<TX>(), scale, n)));
  LAMP_LAUNCH_CHECK();
  LAMP_API_END
}
int lamp_sub_(lamp_tensor* self, const lamp_tensor* b, double alpha) { return lamp_sub_out(self, self, b, alpha); }
int lamp_mul_(lamp_tensor* self, const lamp_tensor* b) { return lamp_mul_out(self, self, b); }
int lamp_div_(lamp_tensor* self, const lamp_tensor* b) { return lamp_div_out(self, self, b); }
int lamp_add_scalar_(lamp_tensor* self, double b, double alpha) {
  LAMP_API_BEGIN run_same<1, FAddS, false>(self, self, nullptr, nullptr, FAddS{b, alpha}); LAMP_API_END
}
int lamp_mul_scalar_(lamp_tensor* self, double b) {
  LAMP_API_BEGIN run_same<1, FMulS, false>(self, self, nullptr, nullptr, FMulS{b, 0}); LAMP_API_END
}
int lamp_addcmul_out(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* t1, const lamp_tensor* t2, double value) {
  LAMP_API_BEGIN run_same<3, FAddcmul, false>(out, self, t1, t2, FAddcmul{value, 0}); LAMP_API_END
}
int lamp_mul_add(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, const lamp_tensor* c) {
  LAMP_API_BEGIN *out = run_same<3, FMulAdd, true>(nullptr, a, b, c, FMulAdd{}); LAMP_API_END
}
int lamp_addcdiv_out(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* t1, const lamp_tensor* t2, double value) {
  LAMP_API_BEGIN run_same<3, FAddcdiv, true>(out, self, t1, t2, FAddcdiv{value, 0}); LAMP_API_END
}
int lamp_relu_backward_accumulate_(lamp_tensor* out, const lamp_tensor* p, const lamp_tensor* x, double negative_slope) {
  LAMP_API_BEGIN run_same<3, FReluBwdAcc, true>(out, out, p, x, FReluBwdAcc{negative_slope, 0}); LAMP_API_END
}

int lamp_relu_backward(lamp_tensor** out, const lamp_tensor* p, const lamp_tensor* x, double negative_slope) {
  LAMP_API_BEGIN *out = run_same<2, FReluBwd, true>(nullptr, p, x, nullptr, FReluBwd{negative_slope, 0}); LAMP_API_END
}
API1(relu, FRelu, false)
API1_INPLACE(relu_, FRelu, false)
int lamp_leaky_relu(lamp_tensor** out, const lamp_tensor* a, double slope) {
  LAMP_API_BEGIN *out = run_same<1, FLeakyRelu, true>(nullptr, a, nullptr, nullptr, FLeakyRelu{slope, 0}); LAMP_API_END
}
API1(gelu, FGelu, true)
API2(gelu_backward, FGeluBwd, true)
API1(sigmoid, FSigmoid, true)
API2(sigmoid_backward, FSigmoidBwd, true)
API1(tanh, FTanh, true)
API2(tanh_backward, FTanhBwd, true)
API1(hardswish, FHardswish, true)
API2(hardswish_backward, FHardswishBwd, true)
int lamp_softplus(lamp_tensor** out, const lamp_tensor* a, double beta, double threshold) {
  LAMP_API_BEGIN *out = run_same<1, FSoftplus, true>(nullptr, a, nullptr, nullptr, FSoftplus{beta, threshold}); LAMP_API_END
}
int lamp_softplus_backward(lamp_tensor** out, const lamp_tensor* g, const lamp_tensor* x, double beta, double threshold) {
  LAMP_API_BEGIN *out = run_same<2, FSoftplusBwd, true>(nullptr, g, x, nullptr, FSoftplusBwd{beta, threshold}); LAMP_API_END
}
API1(exp, FExp, true)
API1_INPLACE(exp_, FExp, true)
API1(log, FLog, true)
API1(log1p, FLog1p, true)
API1(sqrt, FSqrt, true)
API1_INPLACE(sqrt_, FSqrt, true)
API1(square, FSquare, false)
API1(reciprocal, FRecip, true)
API1_INPLACE(reciprocal_, FRecip, true)
API1(neg, FNeg, false)
API1(abs, FAbs, false)
API1(sign, FSign, false)
API1(sin, FSin, true)
API1(cos, FCos, true)
API1(tan, FTan, true)
API1(atan, FAtan, true)
API1(acos, FAcos, true)
API1(asin, FAsin, true)
API1(ceil, FCeil, true)
API1(floor, FFloor, true)
API1(round, FRound, true)
API1(expm1, FExpm1, true)
API1(log10, FLog10, true)
API2(atan2, FAtan2, true)
API2(remainder, FRemainder, false)
int lamp_remainder_scalar(lamp_tensor** out, const lamp_tensor* a, double b) {
  LAMP_API_BEGIN *out = run_same<1, FRemainderS, false>(nullptr, a, nullptr, nullptr, FRemainderS{b, 0}); LAMP_API_END
}
int lamp_nan_to_num(lamp_tensor** out, const lamp_tensor* a, double nan) {
  LAMP_API_BEGIN
  const double big16 = a->dtype == kBF16 ? 3.3895313892515355e38 : 65504.0;   // largest finite bf16 / f16
  *out = run_same<1, FNanToNum, true>(nullptr, a, nullptr, nullptr, FNanToNum{nan, big16});
  LAMP_API_END
}
// in-place forms (ATen's trailing underscore: STen's `_`-suffixed methods, STen.scala:1200-1700)
API1_INPLACE(abs_, FAbs, false)
API1_INPLACE(acos_, FAcos, true)
API1_INPLACE(asin_, FAsin, true)
API1_INPLACE(atan_, FAtan, true)
API1_INPLACE(ceil_, FCeil, true)
API1_INPLACE(floor_, FFloor, true)
API1_INPLACE(cos_, FCos, true)
API1_INPLACE(sin_, FSin, true)
API1_INPLACE(tan_, FTan, true)
API1_INPLACE(tanh_, FTanh, true)
API1_INPLACE(sigmoid_, FSigmoid, true)
API1_INPLACE(log_, FLog, true)
API1_INPLACE(log1p_, FLog1p, true)
API1_INPLACE(square_, FSquare, false)
int lamp_leaky_relu_(lamp_tensor* a, double slope) {
  LAMP_API_BEGIN run_same<1, FLeakyRelu, true>(a, a, nullptr, nullptr, FLeakyRelu{slope, 0}); LAMP_API_END
}
API_CMP(logical_and, FLogicalAnd)
API_CMP(logical_or, FLogicalOr)
API_CMP(logical_xor, FLogicalXor)
int lamp_isnan(lamp_tensor** out, const lamp_tensor* a) { LAMP_API_BEGIN *out = run_bool<1, FIsNan>(a, nullptr, FIsNan{}); LAMP_API_END }
int lamp_isfinite(lamp_tensor** out, const lamp_tensor* a) { LAMP_API_BEGIN *out = run_bool<1, FIsFinite>(a, nullptr, FIsFinite{}); LAMP_API_END }

API_CMP(lt, FLt)
API_CMP(le, FLe)
API_CMP(gt, FGt)
API_CMP(ge, FGe)
API_CMP(eq, FEq)
API_CMP(ne, FNe)
API_CMP_S(lt_scalar, FLtS)
API_CMP_S(le_scalar, FLeS)
API_CMP_S(gt_scalar, FGtS)
API_CMP_S(ge_scalar, FGeS)
API_CMP_S(eq_scalar, FEqS)
API_CMP_S(ne_scalar, FNeS)
int lamp_logical_not(lamp_tensor** out, const lamp_tensor* a) {
  LAMP_API_BEGIN *out = run_bool<1, FLogicalNot>(a, nullptr, FLogicalNot{}); LAMP_API_END
}

int lamp_where(lamp_tensor** out, const lamp_tensor* cond, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN
  check_device_tensor(cond, "condition"); check_device_tensor(a, "self"); check_device_tensor(b, "other");
  LAMP_CHECK(cond->dtype == kBool || cond->dtype == kU8, "where expects a bool condition, got " << cond->describe());
  LAMP_CHECK(a->dtype == b->dtype, "where: dtype mismatch " << a->describe() << " vs " << b->describe());
  auto shape = broadcast_shapes(broadcast_shapes(cond->shape(), a->shape()), b->shape());
  Hold r(new_tensor(shape, a->dtype, a->device()));
  const Tensor* ops[4] = {r.get(), cond, a, b};
  IterSpace it = make_iter(shape, ops, 4);
  if (it.numel > 0) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((where_kernel<T>), dim3(grid_for(it.numel, 256)), dim3(256), 0,
                                                      current_stream(a->device()), r->ptr<T>(), cond->ptr<uint8_t>(),
                                                      a->ptr<T>(), b->ptr<T>(), it.numel, to_args(it)));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_masked_fill(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* mask, double value) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_device_tensor(mask, "mask");
  LAMP_CHECK(mask->dtype == kBool || mask->dtype == kU8, "masked_fill expects a bool mask");
  auto shape = broadcast_shapes(a->shape(), mask->shape());
  LAMP_CHECK(shape == a->shape(), "masked_fill: mask must broadcast to self");
  Hold r(new_tensor(shape, a->dtype, a->device()));
  const Tensor* ops[3] = {r.get(), a, mask};
  IterSpace it = make_iter(shape, ops, 3);
  if (it.numel > 0) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((masked_fill_kernel<T>), dim3(grid_for(it.numel, 256)), dim3(256), 0,
                                                      current_stream(a->device()), r->ptr<T>(), a->ptr<T>(),
                                                      mask->ptr<uint8_t>(), store_as<T>((acc_t<T>)value), it.numel, to_args(it)));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}

}  // extern "C"
