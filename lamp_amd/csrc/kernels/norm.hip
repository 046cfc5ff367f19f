// Batch norm and layer norm, forward and backward (HBM-bound).
//
// Replaces ATen.native_batch_norm / native_batch_norm_backward / native_layer_norm /
// native_layer_norm_backward as lamp calls them (reference:
// lamp-core/src/main/scala/lamp/autograd/ops.scala:1846-1955 (BatchNorm, input flattened to
// [N, F]), 2037-2140 (BatchNorm2D, [N, C, *]), 1956-2032 (LayerNormOp)).
//
// Semantics kept from ATen: biased variance for normalisation, running_var updated with the
// UNBIASED estimate, running stats updated in place with `momentum`, save_invstd = 1/sqrt(var+eps);
// statistics in f32 for bf16/f32 inputs (f64 for f64) using Welford/Chan merging so that
// variance never suffers catastrophic cancellation.
//
// Layout: x is [N, C, HW] contiguous (HW = 1 for the 2-D case). Pass 1 reduces each channel with
// one or more workgroups (64-lane shuffle merge + LDS), pass 2 is a coalesced normalise.
#include "device_utils.h"
#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>
#include "../core/strided.h"

namespace lamp {

template <class A> struct Welford {
  A n, mean, m2;
};
template <class A> __device__ __forceinline__ void wf_add(Welford<A>& w, A x) {
  w.n += A(1);
  A d = x - w.mean;
  w.mean += d / w.n;
  w.m2 += d * (x - w.mean);
}
template <class A> __device__ __forceinline__ Welford<A> wf_merge(const Welford<A>& a, const Welford<A>& b) {
  Welford<A> r;
  r.n = a.n + b.n;
  if (r.n == A(0)) { r.mean = 0; r.m2 = 0; return r; }
  A d = b.mean - a.mean;
  A f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}
template <class A> __device__ __forceinline__ Welford<A> wf_wave(Welford<A> w) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Welford<A> o;
    o.n = __shfl_xor(w.n, off, 64);
    o.mean = __shfl_xor(w.mean, off, 64);
    o.m2 = __shfl_xor(w.m2, off, 64);
    w = wf_merge(w, o);
  }
  return w;
}

// ---- batch norm statistics: partial[split][c] = Welford over a slice of (n, hw) -----------------
// big-HW variant: block (c, split). The channel's N*HW elements are walked as 16-byte packets
// (HW % W == 0) by all threads of all splits; each thread keeps a SHIFTED sum / sum of squares
// (shift = its first element, so no catastrophic cancellation) which is converted to a Welford
// triple and Chan-merged across lanes, waves and splits.
template <class T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, acc_t<T>* __restrict__ partial, int64_t N, int64_t C,
                                                       int64_t HW, int nsplit, int vec) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  __shared__ A sm[3][4];
  const int64_t c = blockIdx.x;
  const int split = blockIdx.y;
  A cnt = 0, s1 = 0, s2 = 0, shift = 0;
  bool first = true;
  if (vec) {
    const int64_t vpp = HW / W;                 // packets per (n, c) plane
    const int64_t total = N * vpp;
    for (int64_t i = (int64_t)split * blockDim.x + threadIdx.x; i < total; i += (int64_t)nsplit * blockDim.x) {
      const int64_t n = i / vpp, v = i - n * vpp;
      const Vec<T, W> pk = *reinterpret_cast<const Vec<T, W>*>(x + (n * C + c) * HW + v * W);
      if (first) { shift = load_as<A>(pk.v[0]); first = false; }
#pragma unroll
      for (int k = 0; k < W; k++) { const A d = load_as<A>(pk.v[k]) - shift; s1 += d; s2 += d * d; }
      cnt += A(W);
    }
  } else {
    const int64_t total = N * HW;
    for (int64_t i = (int64_t)split * blockDim.x + threadIdx.x; i < total; i += (int64_t)nsplit * blockDim.x) {
      const int64_t n = i / HW, v = i - n * HW;
      const A val = load_as<A>(x[(n * C + c) * HW + v]);
      if (first) { shift = val; first = false; }
      const A d = val - shift;
      s1 += d; s2 += d * d; cnt += A(1);
    }
  }
  Welford<A> w{cnt, cnt > A(0) ? shift + s1 / cnt : A(0), cnt > A(0) ? s2 - s1 * s1 / cnt : A(0)};
  w = wf_wave(w);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[0][wid] = w.n; sm[1][wid] = w.mean; sm[2][wid] = w.m2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    Welford<A> r{sm[0][0], sm[1][0], sm[2][0]};
    for (int k = 1; k < (int)(blockDim.x >> 6); k++) r = wf_merge(r, Welford<A>{sm[0][k], sm[1][k], sm[2][k]});
    A* o = partial + ((int64_t)split * C + c) * 3;
    o[0] = r.n; o[1] = r.mean; o[2] = r.m2;
  }
}
// small-HW variant (HW < 64, typically 1): thread per channel, coalesced along c when HW == 1
template <class T>
__global__ __launch_bounds__(256) void bn_stats_col_kernel(const T* __restrict__ x, acc_t<T>* __restrict__ partial, int64_t N,
                                                           int64_t C, int64_t HW, int nsplit) {
  using A = acc_t<T>;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int split = blockIdx.y;
  Welford<A> w{0, 0, 0};
  for (int64_t n = split; n < N; n += nsplit) {
    const T* p = x + (n * C + c) * HW;
    for (int64_t i = 0; i < HW; i++) wf_add(w, load_as<A>(p[i]));
  }
  A* o = partial + ((int64_t)split * C + c) * 3;
  o[0] = w.n; o[1] = w.mean; o[2] = w.m2;
}
// finalize: one wavefront per channel merges the split partials (Chan), lane 0 writes save_mean / save_invstd and
// updates the running statistics
template <class T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const acc_t<T>* __restrict__ partial, int64_t C, int nsplit, T* __restrict__ save_mean,
                                                          T* __restrict__ save_invstd, T* running_mean, T* running_var, double momentum, double eps) {
  using A = acc_t<T>;
  const int lane = threadIdx.x & 63;
  const int64_t c = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (c >= C) return;
  Welford<A> r{0, 0, 0};
  for (int s = lane; s < nsplit; s += 64) {
    const A* p = partial + ((int64_t)s * C + c) * 3;
    r = wf_merge(r, Welford<A>{p[0], p[1], p[2]});
  }
  r = wf_wave(r);
  if (lane != 0) return;
  const A var_biased = r.m2 / r.n;
  const A invstd = A(1) / (A)sqrt((double)(var_biased + (A)eps));
  save_mean[c] = store_as<T>(r.mean);
  save_invstd[c] = store_as<T>(invstd);
  if (running_mean) running_mean[c] = store_as<T>((A)((A)momentum * r.mean + (A)(1 - momentum) * load_as<A>(running_mean[c])));
  if (running_var) {
    const A unbiased = r.m2 / (r.n - A(1));
    running_var[c] = store_as<T>((A)((A)momentum * unbiased + (A)(1 - momentum) * load_as<A>(running_var[c])));
  }
}
// eval mode: mean/invstd from the running statistics
template <class T>
__global__ void bn_eval_stats_kernel(const T* running_mean, const T* running_var, T* mean, T* invstd, int64_t C, double eps) {
  using A = acc_t<T>;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean ? running_mean[c] : store_as<T>(A(0));
  const A v = running_var ? load_as<A>(running_var[c]) : A(1);
  invstd[c] = store_as<T>((A)(A(1) / (A)sqrt((double)(v + (A)eps))));
}
// the normalised value, ONE expression shared by the forward kernel and by the backward kernels that recompute it for
// the fused relu mask (an explicit fma so that every kernel rounds identically)
template <class A> __device__ __forceinline__ A bn_affine(A x, A mu, A scale, A bb);
template <> __device__ __forceinline__ float bn_affine<float>(float x, float mu, float scale, float bb) { return __builtin_fmaf(x - mu, scale, bb); }
template <> __device__ __forceinline__ double bn_affine<double>(double x, double mu, double scale, double bb) { return __builtin_fma(x - mu, scale, bb); }
// y = (x - mean) * invstd * w + b      (16-byte packets when HW % W == 0: one channel per packet); relu: y = max(y, 0)
// applied to the ROUNDED value, as the separate relu kernel would
template <class T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ mean,
                                                       const T* __restrict__ invstd, const T* __restrict__ w, const T* __restrict__ b,
                                                       int64_t total, int64_t C, int64_t HW, int vec, int relu) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  if (vec) {
    const int64_t vpp = HW / W, nv = total / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t c = (i / vpp) % C;
      const A scale = load_as<A>(invstd[c]) * (w ? load_as<A>(w[c]) : A(1));
      const A mu = load_as<A>(mean[c]);
      const A bb = b ? load_as<A>(b[c]) : A(0);
      Vec<T, W> pk = *reinterpret_cast<const Vec<T, W>*>(x + i * W);
#pragma unroll
      for (int k = 0; k < W; k++) {
        T v = store_as<T>(bn_affine<A>(load_as<A>(pk.v[k]), mu, scale, bb));
        if (relu && load_as<A>(v) < A(0)) v = store_as<T>(A(0));
        pk.v[k] = v;
      }
      *reinterpret_cast<Vec<T, W>*>(y + i * W) = pk;
    }
    return;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = (i / HW) % C;
    const A scale = load_as<A>(invstd[c]) * (w ? load_as<A>(w[c]) : A(1));
    const A bb = b ? load_as<A>(b[c]) : A(0);
    T v = store_as<T>(bn_affine<A>(load_as<A>(x[i]), load_as<A>(mean[c]), scale, bb));
    if (relu && load_as<A>(v) < A(0)) v = store_as<T>(A(0));
    y[i] = v;
  }
}

// Channel c's partial statistics -> (mean, invstd) as they are SAVED (rounded to T) in stat[0..1]; called by every thread of the
// workgroup.  nsplit > 0: partial[s][c] (the statistics kernel), merged by the first wavefront; nsplit < 0: partial[c][s] with -nsplit
// EQUAL-count triples (a convolution's per-image statistics), merged by the whole workgroup in a fixed order:
// with shift = mean_0,  mean = shift + S1 / np,  m2 = sum m2_s + n0 (S2 - S1^2 / np)  where S1 = sum (mean_s - shift), S2 = sum (mean_s - shift)^2
// (no division per partial).  Every workgroup of the channel gets bitwise the same values; slice 0 writes save_mean / save_invstd and
// updates the running statistics.
template <class T>
__device__ __forceinline__ void bn_merge_channel(const acc_t<T>* __restrict__ partial, int nsplit, int64_t c, int64_t C, int slice, T* __restrict__ save_mean,
                                                 T* __restrict__ save_invstd, T* running_mean, T* running_var, double momentum, double eps,
                                                 acc_t<T>* stat, acc_t<T> (*wpart)[3]) {
  using A = acc_t<T>;
  const int np = nsplit < 0 ? -nsplit : nsplit;
  if (nsplit < 0) {
    const A* pc = partial + (int64_t)c * np * 3;
    const A shift = pc[1];
    A s1 = 0, s2 = 0, sm = 0;
    for (int s = threadIdx.x; s < np; s += blockDim.x) { const A d = pc[s * 3 + 1] - shift; s1 += d; s2 += d * d; sm += pc[s * 3 + 2]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2); sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) { wpart[threadIdx.x >> 6][0] = s1; wpart[threadIdx.x >> 6][1] = s2; wpart[threadIdx.x >> 6][2] = sm; }
    __syncthreads();
  }
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    Welford<A> r{0, 0, 0};
    if (nsplit < 0) {
      A s1 = 0, s2 = 0, sm = 0;
      for (int k = 0; k < (int)(blockDim.x >> 6); k++) { s1 += wpart[k][0]; s2 += wpart[k][1]; sm += wpart[k][2]; }
      const A n0 = partial[(int64_t)c * np * 3], shift = partial[(int64_t)c * np * 3 + 1];
      r.n = n0 * (A)np;
      r.mean = shift + s1 / (A)np;
      r.m2 = sm + n0 * (s2 - s1 * s1 / (A)np);
    } else {
      for (int s = lane; s < np; s += 64) {
        const A* p = partial + ((int64_t)s * C + c) * 3;
        r = wf_merge(r, Welford<A>{p[0], p[1], p[2]});
      }
      r = wf_wave(r);
    }
    if (lane == 0) {
      const A var_biased = r.m2 / r.n;
      const A invstd = A(1) / (A)sqrt((double)(var_biased + (A)eps));
      const T m_t = store_as<T>(r.mean), is_t = store_as<T>(invstd);
      stat[0] = load_as<A>(m_t);                       // the normalisation uses the values as they are saved (rounded to T)
      stat[1] = load_as<A>(is_t);
      if (slice == 0) {
        save_mean[c] = m_t;
        save_invstd[c] = is_t;
        if (running_mean) running_mean[c] = store_as<T>((A)((A)momentum * r.mean + (A)(1 - momentum) * load_as<A>(running_mean[c])));
        if (running_var) {
          const A unbiased = r.m2 / (r.n - A(1));
          running_var[c] = store_as<T>((A)((A)momentum * unbiased + (A)(1 - momentum) * load_as<A>(running_var[c])));
        }
      }
    }
  }
  __syncthreads();
}
// The batch norm as a per-channel table instead of a normalised tensor (lamp_batch_norm_affine): affine[c] = (mean, invstd * weight, bias, 0),
// the three operands of bn_affine<float>, for a convolution that applies batch norm + relu while it stages its input (conv_igemm.hip).
template <class T>
__global__ __launch_bounds__(256) void bn_affine_table_kernel(const acc_t<T>* __restrict__ partial, int nsplit, T* __restrict__ save_mean, T* __restrict__ save_invstd,
                                                              T* running_mean, T* running_var, double momentum, double eps, const T* __restrict__ w,
                                                              const T* __restrict__ b, int64_t C, float4* __restrict__ affine) {
  using A = acc_t<T>;
  __shared__ A stat[2];
  __shared__ A wpart[4][3];
  const int64_t c = blockIdx.x;
  bn_merge_channel<T>(partial, nsplit, c, C, 0, save_mean, save_invstd, running_mean, running_var, momentum, eps, stat, wpart);
  if (threadIdx.x == 0) affine[c] = make_float4((float)stat[0], (float)(stat[1] * (w ? load_as<A>(w[c]) : A(1))), b ? (float)load_as<A>(b[c]) : 0.f, 0.f);
}
template <class T>
__global__ __launch_bounds__(256) void bn_affine_from_saved_kernel(const T* __restrict__ mean, const T* __restrict__ invstd, const T* __restrict__ w,
                                                                   const T* __restrict__ b, int64_t C, float4* __restrict__ affine) {
  using A = acc_t<T>;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c < C) affine[c] = make_float4((float)load_as<A>(mean[c]), (float)(load_as<A>(invstd[c]) * (w ? load_as<A>(w[c]) : A(1))), b ? (float)load_as<A>(b[c]) : 0.f, 0.f);
}
// A second batch norm whose OUTPUT is the addend (lamp_native_batch_norm2_add_relu: the tail of lamp's residual block when the left
// branch is convolution -> batch norm too, cnn.scala:62-78): x == nullptr means none.  Passed by value.
template <class T> struct BnSecond {
  const T* x; const acc_t<T>* partial; int nsplit; T* save_mean; T* save_invstd; T* running_mean; T* running_var; double momentum, eps; const T* w; const T* b;
};

// Channel-aligned normalise with the finalize folded in (training mode, maps of >= 64 elements): workgroup (c, slice).
// bn_merge_channel gives every workgroup of the channel bitwise the same mean / invstd; then the workgroup normalises its slice of the
// channel.  One launch less per batch norm, and no cross-workgroup hand-off (the partials are complete at the kernel boundary).
// With `second`: y = relu(round(round(bn(x)) + round(bn2(x2)))) - the three-kernel chain bn2 -> (bn + add + relu) with every intermediate
// rounded as the chain rounds it, without writing and re-reading bn2's output.
template <class T>
__global__ __launch_bounds__(256) void bn_apply2_kernel(const T* __restrict__ x, T* __restrict__ y, const acc_t<T>* __restrict__ partial, int nsplit,
                                                        T* __restrict__ save_mean, T* __restrict__ save_invstd, T* running_mean, T* running_var,
                                                        double momentum, double eps, const T* __restrict__ w, const T* __restrict__ b, int64_t N,
                                                        int64_t C, int64_t HW, int nblk, int vec, int relu, const T* __restrict__ addend,
                                                        BnSecond<T> second, T* __restrict__ pooled = nullptr) {
  // pooled (round 6, vec path, HW / W a power of two <= 64): the output is consumed by a global average pool and by nothing else (the last
  // block of Cnn.resnet in front of AvgPool2D -> Flatten -> LogSoftMax, cnn.scala:129-136): y is NOT written; the HW / W lanes that hold a
  // plane sum their rounded outputs in gap_lsm_fwd_kernel's order (eight values per lane, then the butterfly) and pooled[n][c] gets that
  // kernel's rounded mean - bitwise what the pool would have read back from y
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  __shared__ A stat[2], stat2[2];
  __shared__ A wpart[4][3];
  const int64_t c = blockIdx.x;
  const int slice = blockIdx.y;
  bn_merge_channel<T>(partial, nsplit, c, C, slice, save_mean, save_invstd, running_mean, running_var, momentum, eps, stat, wpart);
  if (second.x)
    bn_merge_channel<T>(second.partial, second.nsplit, c, C, slice, second.save_mean, second.save_invstd, second.running_mean, second.running_var,
                        second.momentum, second.eps, stat2, wpart);
  const A mu = stat[0];
  const A scale = stat[1] * (w ? load_as<A>(w[c]) : A(1));
  const A bb = b ? load_as<A>(b[c]) : A(0);
  const A mu2 = second.x ? stat2[0] : A(0);
  const A scale2 = second.x ? stat2[1] * (second.w ? load_as<A>(second.w[c]) : A(1)) : A(0);
  const A bb2 = (second.x && second.b) ? load_as<A>(second.b[c]) : A(0);
  const T* __restrict__ adsrc = second.x ? second.x : addend;
  if (vec) {
    const int64_t vpp = HW / W, total = N * vpp;
    for (int64_t i = (int64_t)slice * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
      const int64_t n = i / vpp, v = i - n * vpp;
      const int64_t base = (n * C + c) * HW + v * W;
      // x (and the addend) are not read again before the backward pass: streamed, so that y - which the next convolution reads - keeps the caches
      Vec<T, W> pk, ad;
      { const uint4 t = nt_load16(reinterpret_cast<const uint4*>(x + base)); pk = *reinterpret_cast<const Vec<T, W>*>(&t); }
      if (adsrc) { const uint4 t = nt_load16(reinterpret_cast<const uint4*>(adsrc + base)); ad = *reinterpret_cast<const Vec<T, W>*>(&t); }
#pragma unroll
      for (int k = 0; k < W; k++) {
        T o = store_as<T>(bn_affine<A>(load_as<A>(pk.v[k]), mu, scale, bb));
        if (adsrc) {
          T a = ad.v[k];
          if (second.x) a = store_as<T>(bn_affine<A>(load_as<A>(a), mu2, scale2, bb2));   // the left branch's batch norm, rounded as its own kernel rounds
          o = store_as<T>((A)(load_as<A>(o) + load_as<A>(a)));                           // the residual add, rounded as the add kernel rounds
        }
        if (relu && load_as<A>(o) < A(0)) o = store_as<T>(A(0));
        pk.v[k] = o;
      }
      if (pooled) {
        A sum = 0;
#pragma unroll
        for (int k = 0; k < W; k++) sum += load_as<A>(pk.v[k]);
        for (int off = (int)vpp >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (v == 0) pooled[n * C + c] = store_as<T>((A)(sum / (A)HW));
      } else *reinterpret_cast<Vec<T, W>*>(y + base) = pk;
    }
  } else {
    const int64_t total = N * HW;
    for (int64_t i = (int64_t)slice * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
      const int64_t n = i / HW, v = i - n * HW;
      const int64_t base = (n * C + c) * HW + v;
      T o = store_as<T>(bn_affine<A>(load_as<A>(x[base]), mu, scale, bb));
      if (adsrc) {
        T a = adsrc[base];
        if (second.x) a = store_as<T>(bn_affine<A>(load_as<A>(a), mu2, scale2, bb2));
        o = store_as<T>((A)(load_as<A>(o) + load_as<A>(a)));
      }
      if (relu && load_as<A>(o) < A(0)) o = store_as<T>(A(0));
      y[base] = o;
    }
  }
}

// ---- batch norm backward -------------------------------------------------------------------------
// partial[split][c] = (sum dy, sum dy * (x - mean))
// relu != 0: dy is the gradient of relu(bn(x)); it is masked where the (recomputed, rounded) normalised value is < 0
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                            acc_t<T>* __restrict__ partial, int64_t N, int64_t C, int64_t HW, int nsplit, int vec,
                                                            int relu, const T* __restrict__ invstd, const T* __restrict__ w, const T* __restrict__ b,
                                                            const T* __restrict__ addend) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  __shared__ A sm[2][4];
  const int64_t c = blockIdx.x;
  const int split = blockIdx.y;
  const A mu = load_as<A>(mean[c]);
  const A scale = relu ? load_as<A>(invstd[c]) * (w ? load_as<A>(w[c]) : A(1)) : A(0);
  const A bb = (relu && b) ? load_as<A>(b[c]) : A(0);
  A s1 = 0, s2 = 0;
  if (vec) {
    const int64_t vpp = HW / W, total = N * vpp;
    for (int64_t i = (int64_t)split * blockDim.x + threadIdx.x; i < total; i += (int64_t)nsplit * blockDim.x) {
      const int64_t n = i / vpp, v = i - n * vpp;
      const int64_t base = (n * C + c) * HW + v * W;
      const Vec<T, W> g = *reinterpret_cast<const Vec<T, W>*>(dy + base);
      const Vec<T, W> xv = *reinterpret_cast<const Vec<T, W>*>(x + base);
      Vec<T, W> ad;
      if (addend) ad = *reinterpret_cast<const Vec<T, W>*>(addend + base);
#pragma unroll
      for (int k = 0; k < W; k++) {
        A gg = load_as<A>(g.v[k]);
        const A xx = load_as<A>(xv.v[k]);
        if (relu) {
          T pre = store_as<T>(bn_affine<A>(xx, mu, scale, bb));
          if (addend) pre = store_as<T>((A)(load_as<A>(pre) + load_as<A>(ad.v[k])));
          if (load_as<A>(pre) < A(0)) gg = A(0);
        }
        s1 += gg; s2 += gg * (xx - mu);
      }
    }
  } else {
    const int64_t total = N * HW;
    for (int64_t i = (int64_t)split * blockDim.x + threadIdx.x; i < total; i += (int64_t)nsplit * blockDim.x) {
      const int64_t n = i / HW, v = i - n * HW;
      const int64_t base = (n * C + c) * HW + v;
      A gg = load_as<A>(dy[base]);
      const A xx = load_as<A>(x[base]);
      if (relu) {
        T pre = store_as<T>(bn_affine<A>(xx, mu, scale, bb));
        if (addend) pre = store_as<T>((A)(load_as<A>(pre) + load_as<A>(addend[base])));
        if (load_as<A>(pre) < A(0)) gg = A(0);
      }
      s1 += gg; s2 += gg * (xx - mu);
    }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[0][wid] = s1; sm[1][wid] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    A a = 0, bsum = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) { a += sm[0][k]; bsum += sm[1][k]; }
    partial[((int64_t)split * C + c) * 2] = a;
    partial[((int64_t)split * C + c) * 2 + 1] = bsum;
  }
}
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_col_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                                acc_t<T>* __restrict__ partial, int64_t N, int64_t C, int64_t HW, int nsplit) {
  using A = acc_t<T>;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int split = blockIdx.y;
  const A mu = load_as<A>(mean[c]);
  A s1 = 0, s2 = 0;
  for (int64_t n = split; n < N; n += nsplit) {
    const int64_t base = (n * C + c) * HW;
    for (int64_t i = 0; i < HW; i++) {
      const A g = load_as<A>(dy[base + i]);
      s1 += g;
      s2 += g * (load_as<A>(x[base + i]) - mu);
    }
  }
  partial[((int64_t)split * C + c) * 2] = s1;
  partial[((int64_t)split * C + c) * 2 + 1] = s2;
}
// sums[c] = merged partials (one wavefront per channel); also writes dweight / dbias if requested
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const acc_t<T>* __restrict__ partial, acc_t<T>* __restrict__ sums, int64_t C, int nsplit,
                                                              const T* __restrict__ invstd, T* dweight, T* dbias) {
  using A = acc_t<T>;
  const int lane = threadIdx.x & 63;
  const int64_t c = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (c >= C) return;
  A a = 0, b = 0;
  for (int s = lane; s < nsplit; s += 64) { a += partial[((int64_t)s * C + c) * 2]; b += partial[((int64_t)s * C + c) * 2 + 1]; }
  a = wave_sum(a); b = wave_sum(b);
  if (lane != 0) return;
  sums[c * 2] = a; sums[c * 2 + 1] = b;
  if (dweight) dweight[c] = store_as<T>((A)(b * load_as<A>(invstd[c])));
  if (dbias) dbias[c] = store_as<T>(a);
}
// training: dx = (dy - sum_dy/M - (x-mean)*invstd^2*sum_dy_xmu/M) * invstd * w ; eval: dx = dy * invstd * w
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                           const T* __restrict__ invstd, const T* __restrict__ w,
                                                           const acc_t<T>* __restrict__ sums, T* __restrict__ dx, int64_t total, int64_t C,
                                                           int64_t HW, double inv_m, int training, int vec, int relu, const T* __restrict__ b) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  if (vec) {
    const int64_t vpp = HW / W, nv = total / W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t c = (i / vpp) % C;
      const A is = load_as<A>(invstd[c]);
      const A wc = w ? load_as<A>(w[c]) : A(1);
      const A mu = load_as<A>(mean[c]);
      const A k = training ? sums[c * 2 + 1] * is * is * (A)inv_m : A(0);
      const A gm = training ? sums[c * 2] * (A)inv_m : A(0);
      const A scale = is * wc, bb = (relu && b) ? load_as<A>(b[c]) : A(0);
      const Vec<T, W> g = *reinterpret_cast<const Vec<T, W>*>(dy + i * W);
      const Vec<T, W> xv = *reinterpret_cast<const Vec<T, W>*>(x + i * W);
      Vec<T, W> r;
#pragma unroll
      for (int q = 0; q < W; q++) {
        A gg = load_as<A>(g.v[q]);
        const A xx = load_as<A>(xv.v[q]);
        if (relu && load_as<A>(store_as<T>(bn_affine<A>(xx, mu, scale, bb))) < A(0)) gg = A(0);
        r.v[q] = store_as<T>(training ? (A)((gg - gm - (xx - mu) * k) * is * wc) : (A)(gg * is * wc));
      }
      *reinterpret_cast<Vec<T, W>*>(dx + i * W) = r;
    }
    return;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = (i / HW) % C;
    const A is = load_as<A>(invstd[c]);
    const A wc = w ? load_as<A>(w[c]) : A(1);
    const A mu = load_as<A>(mean[c]);
    A g = load_as<A>(dy[i]);
    const A xx = load_as<A>(x[i]);
    if (relu && load_as<A>(store_as<T>(bn_affine<A>(xx, mu, is * wc, b ? load_as<A>(b[c]) : A(0)))) < A(0)) g = A(0);
    A r;
    if (training) {
      const A k = sums[c * 2 + 1] * is * is * (A)inv_m;
      const A gm = sums[c * 2] * (A)inv_m;
      r = (g - gm - (xx - mu) * k) * is * wc;
    } else {
      r = g * is * wc;
    }
    dx[i] = store_as<T>(r);
  }
}

// Channel-aligned dx with the finalize folded in: workgroup (c, slice) sums the channel's split partials in the fixed order of
// bn_bwd_finalize_kernel, slice 0 writes dweight / dbias, then (do_apply) the workgroup writes its slice of dx.
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                            const T* __restrict__ invstd, const T* __restrict__ w, const T* __restrict__ b,
                                                            const acc_t<T>* __restrict__ partial, int nsplit, T* dweight, T* dbias,
                                                            T* __restrict__ dx, int64_t N, int64_t C, int64_t HW, int nblk, double inv_m,
                                                            int training, int vec, int relu, int do_apply, const T* __restrict__ addend,
                                                            T* __restrict__ dadd) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  __shared__ A stat[2];
  const int64_t c = blockIdx.x;
  const int slice = blockIdx.y;
  const A is = load_as<A>(invstd[c]);
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    A a = 0, bs = 0;
    for (int s = lane; s < nsplit; s += 64) { a += partial[((int64_t)s * C + c) * 2]; bs += partial[((int64_t)s * C + c) * 2 + 1]; }
    a = wave_sum(a); bs = wave_sum(bs);
    if (lane == 0) {
      stat[0] = a; stat[1] = bs;
      if (slice == 0) {
        if (dweight) dweight[c] = store_as<T>((A)(bs * is));
        if (dbias) dbias[c] = store_as<T>(a);
      }
    }
  }
  if (!do_apply) return;
  __syncthreads();
  const A wc = w ? load_as<A>(w[c]) : A(1);
  const A mu = load_as<A>(mean[c]);
  const A k = training ? stat[1] * is * is * (A)inv_m : A(0);
  const A gm = training ? stat[0] * (A)inv_m : A(0);
  const A scale = is * wc, bb = (relu && b) ? load_as<A>(b[c]) : A(0);
  if (vec) {
    const int64_t vpp = HW / W, total = N * vpp;
    for (int64_t i = (int64_t)slice * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
      const int64_t n = i / vpp, v = i - n * vpp;
      const int64_t base = (n * C + c) * HW + v * W;
      const Vec<T, W> g = *reinterpret_cast<const Vec<T, W>*>(dy + base);
      const Vec<T, W> xv = *reinterpret_cast<const Vec<T, W>*>(x + base);
      Vec<T, W> ad, r, gm_out;
      if (addend) ad = *reinterpret_cast<const Vec<T, W>*>(addend + base);
#pragma unroll
      for (int q = 0; q < W; q++) {
        A gg = load_as<A>(g.v[q]);
        const A xx = load_as<A>(xv.v[q]);
        if (relu) {
          T pre = store_as<T>(bn_affine<A>(xx, mu, scale, bb));
          if (addend) pre = store_as<T>((A)(load_as<A>(pre) + load_as<A>(ad.v[q])));
          if (load_as<A>(pre) < A(0)) gg = A(0);
        }
        gm_out.v[q] = store_as<T>(gg);                      // dy or 0: exact
        r.v[q] = store_as<T>(training ? (A)((gg - gm - (xx - mu) * k) * is * wc) : (A)(gg * is * wc));
      }
      if (dx) *reinterpret_cast<Vec<T, W>*>(dx + base) = r;
      if (dadd) *reinterpret_cast<Vec<T, W>*>(dadd + base) = gm_out;
    }
  } else {
    const int64_t total = N * HW;
    for (int64_t i = (int64_t)slice * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
      const int64_t n = i / HW, v = i - n * HW;
      const int64_t base = (n * C + c) * HW + v;
      A gg = load_as<A>(dy[base]);
      const A xx = load_as<A>(x[base]);
      if (relu) {
        T pre = store_as<T>(bn_affine<A>(xx, mu, scale, bb));
        if (addend) pre = store_as<T>((A)(load_as<A>(pre) + load_as<A>(addend[base])));
        if (load_as<A>(pre) < A(0)) gg = A(0);
      }
      if (dx) dx[base] = store_as<T>(training ? (A)((gg - gm - (xx - mu) * k) * is * wc) : (A)(gg * is * wc));
      if (dadd) dadd[base] = store_as<T>(gg);
    }
  }
}

// ---- batch-norm backward in ONE pass over (dy, x) ------------------------------------------------------------------------------
// The two kernels above read dy and x (and the addend) twice: once for the channel sums, once for dx.  A bf16 activation of the
// ResNet step is at most 33.5 MB, and the register files of the chip hold 128 MB: here workgroup (c, s) - 512 threads, slice s of
// channel c's 16-byte packets - loads its packets ONCE, keeps the (masked) gradient and x in registers (NP packets of each per
// thread), publishes its partial sums, waits for the S - 1 other workgroups of its channel, sums the S partials in a fixed order
// (every workgroup of the channel gets the same bits; nothing is atomically accumulated) and writes dx (and the addend's gradient)
// from the registers.  HBM traffic 5 -> 3 passes (7 -> 5 with an addend), one launch instead of two.
// Waiting is safe because workgroups are handed out in launch order and a channel's S workgroups are consecutive (b = c * S + s):
// the oldest incomplete channel always gets the next free slots.  Two SUCH kernels running at once on one device (different
// streams) could in principle starve each other, so the host orders them by an event when the stream changes; a wait that is
// never satisfied (~seconds) traps instead of hanging the device.
// slots[c * S + s] = workgroup (c, s)'s (s1, s2), all-ones between launches; depart[c] counts the workgroups that have read the channel's
// slots, zero between launches: the last one to leave resets both.
// DUAL (lamp_native_batch_norm2_add_relu_backward): the addend is itself a batch norm's output, round(bn2(x2)), which the forward never
// wrote: `addend` points at x2, the mask is recomputed from (x, x2), a third sum (sum g (x2 - mean2)) travels through a second slot, and
// instead of the masked gradient the kernel writes the SECOND batch norm's input gradient (x2 is read once more for that - by then it
// comes from the Infinity Cache) and its dweight / dbias: one launch and 6 passes instead of two launches and 8.
struct BnFusedDual { const bf16_t* mean2; const bf16_t* invstd2; const bf16_t* w2; const bf16_t* b2; bf16_t* dweight2; bf16_t* dbias2; unsigned long long* slots2; };
template <int NP, bool RELU, bool ADD, bool DUAL = false, bool PLANES = false>  // compile-time: run-time branches in the element loop let the compiler sink the sums
__global__ __launch_bounds__(512) void bn_bwd_fused_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ mean,
                                                           const bf16_t* __restrict__ invstd, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                           unsigned long long* slots, unsigned* depart, bf16_t* dweight, bf16_t* dbias, bf16_t* __restrict__ dx,
                                                           int64_t N, int C, int HW, int S, double inv_m, int relu,
                                                           const bf16_t* __restrict__ addend, bf16_t* __restrict__ dadd, int vshift, int* assert_word,
                                                           BnFusedDual dual, int dy_sc) {
  // PLANES: dy holds one value per PLANE, class-major - (n, c) at dy[c * dy_sc + n] - for 8 x 8 maps: the gradient arrives as a view expanded
  // over the map (the pooled LogSoftMax's input gradient, lamp_nll_loss_forward_pooled_gradient_); a packet is that value eight times, and the
  // [N, C, H, W] tensor does not exist.  A wave's 64 packets are the eight planes n8 .. n8 + 7 of channel c, whose values are 16 consecutive
  // bytes at a wave-uniform address: ONE scalar load per round (inline asm: the sixteen of a thread are all in flight together and waited for
  // once, behind the vector loads of x - as an ordinary load hipcc sank each into the lane-select branches with a wait of its own)
  __shared__ float sm[3][8];
  __shared__ float stat[3];
  const int c = blockIdx.x / S, s = blockIdx.x - c * S;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int vpp = HW >> 3;
  const int64_t total = N * vpp;                            // packets of this channel
  const float mu = (float)mean[c], is = (float)invstd[c];
  const float wc = w ? (float)w[c] : 1.f;
  const float scale = is * wc, bb = (RELU && b) ? (float)b[c] : 0.f;
  const float mu2 = DUAL ? (float)dual.mean2[c] : 0.f, is2 = DUAL ? (float)dual.invstd2[c] : 0.f;
  const float wc2 = (DUAL && dual.w2) ? (float)dual.w2[c] : 1.f;
  const float scale2 = is2 * wc2, bb2 = (DUAL && dual.b2) ? (float)dual.b2[c] : 0.f;
  uint4 gv[NP], xv[NP];
  typedef unsigned int bn_u4s __attribute__((ext_vector_type(4)));
  bn_u4s pq[PLANES ? NP : 1];                               // PLANES: the eight plane values of each round, in scalar registers
  int base[NP];                                             // packet (16-byte) index into the tensors, -1: none (host: numel < 2^34)
  const uint4* dy4 = reinterpret_cast<const uint4*>(dy);
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  const uint4* ad4 = reinterpret_cast<const uint4*>(addend);
  // every load is issued unconditionally (a packet past the end re-reads the channel's first one and is zeroed): no branch and no
  // 64-bit division between the loads, so all 2 NP of a thread are in flight together
#pragma unroll
  for (int k = 0; k < NP; k++) {
    const unsigned i = (unsigned)(k * S + s) * 512u + (unsigned)tid;
    const bool valid = i < (unsigned)total;
    const unsigned ii = valid ? i : 0u;
    const unsigned n = vshift >= 0 ? (ii >> vshift) : (ii / (unsigned)vpp);
    const int idx = (int)((n * (unsigned)C + (unsigned)c) * (unsigned)vpp + (ii - n * (unsigned)vpp));
    base[k] = valid ? idx : -1;
    if constexpr (PLANES) {
      const unsigned ibase = (unsigned)(k * S + s) * 512u + (unsigned)(tid & ~63);
      const unsigned n8 = __builtin_amdgcn_readfirstlane(ibase < (unsigned)total ? (ibase >> 3) : 0u);
      const unsigned short* a = reinterpret_cast<const unsigned short*>(dy) + (size_t)c * (unsigned)dy_sc + n8;
      asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(pq[k]) : "s"(a));
    } else gv[k] = nt_load16(dy4 + idx);                    // the gradient's last reader
    xv[k] = x4[idx];
  }
  if constexpr (PLANES) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NP; k++) {
      asm volatile("" : "+s"(pq[k]));                       // (produced by the wait above)
      const int d = lane >> 4;
      const unsigned w32 = d == 0 ? pq[k][0] : d == 1 ? pq[k][1] : d == 2 ? pq[k][2] : pq[k][3];
      const unsigned u = ((lane >> 3) & 1) ? (w32 >> 16) : (w32 & 0xffffu);
      const unsigned uu = u | (u << 16);
      gv[k] = make_uint4(uu, uu, uu, uu);
    }
  }
#pragma unroll
  for (int k = 0; k < NP; k++)
    if (base[k] < 0) gv[k] = make_uint4(0, 0, 0, 0);        // contributes nothing to the sums, never stored
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  constexpr int HALF = NP > 4 ? 4 : NP;                     // the addend is only needed for the mask: loaded in groups of <= 4 packets
#pragma unroll
  for (int h = 0; h < NP; h += HALF) {
    uint4 av[HALF];
    if (ADD) {
#pragma unroll
      for (int k = 0; k < HALF; k++) av[k] = ad4[base[h + k] >= 0 ? base[h + k] : c * vpp];
    }
#pragma unroll
    for (int k = 0; k < HALF; k++) {
      unsigned* g32 = &gv[h + k].x;
      const unsigned* x32 = &xv[h + k].x;
      const unsigned* a32 = &av[k].x;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        unsigned gw = g32[q];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float xx = __uint_as_float(e ? (x32[q] & 0xffff0000u) : (x32[q] << 16));
          float gg = __uint_as_float(e ? (gw & 0xffff0000u) : (gw << 16));
          float x2v = 0.f;
          if (RELU) {
            bf16_t pre(bn_affine<float>(xx, mu, scale, bb));
            if (ADD) {
              float av_ = __uint_as_float(e ? (a32[q] & 0xffff0000u) : (a32[q] << 16));
              if (DUAL) { x2v = av_; av_ = (float)bf16_t(bn_affine<float>(av_, mu2, scale2, bb2)); }   // the left branch's output as its kernel rounds it
              pre = bf16_t((float)pre + av_);
            }
            if ((float)pre < 0.f) { gg = 0.f; gw &= e ? 0x0000ffffu : 0xffff0000u; }
          }
          s1 += gg; s2 = __builtin_fmaf(gg, xx - mu, s2);   // explicit: every instantiation rounds alike
          if (DUAL) s3 = __builtin_fmaf(gg, x2v - mu2, s3);
        }
        g32[q] = gw;                                        // dy or 0: what dx and the addend's gradient are computed from
      }
      // the sums are one serial chain: without this the scheduler unpacks every packet ahead of it and keeps ~128 more values live
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // what crosses the wait is the PACKED data (128 registers at NP = 16); without this the compiler keeps the unpacked f32 values of
  // the first phase for the second one (twice the registers, spilled)
#pragma unroll
  for (int k = 0; k < NP; k++) {
    asm volatile("" : "+v"(gv[k].x), "+v"(gv[k].y), "+v"(gv[k].z), "+v"(gv[k].w));
    asm volatile("" : "+v"(xv[k].x), "+v"(xv[k].y), "+v"(xv[k].z), "+v"(xv[k].w));
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (DUAL) s3 = wave_sum(s3);
  if (lane == 0) { sm[0][wid] = s1; sm[1][wid] = s2; if (DUAL) sm[2][wid] = s3; }
  __syncthreads();
  if (wid == 0) {
    // Exchange of the partial sums between the S workgroups of the channel.  No fences (a release / acquire pair at agent scope writes
    // back and invalidates the XCD's whole L2: the step got 35 % slower) and no counter on the critical path: a workgroup publishes
    // (s1, s2) as ONE 8-byte agent-scope atomic store into its slot, and everybody polls the S slots (agent-scope atomic loads bypass
    // the per-XCD L2) until none holds the all-ones pattern the slots rest at between launches - two memory round trips in all.
    float a = 0.f, bs = 0.f, cs = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) { a += sm[0][k]; bs += sm[1][k]; if (DUAL) cs += sm[2][k]; }
    if (S > 1) {
      unsigned long long* slot = slots + (int64_t)c * S;
      unsigned long long* slot2 = DUAL ? dual.slots2 + (int64_t)c * S : nullptr;
      if (lane == 0) {
        unsigned lo = __float_as_uint(a), hi = __float_as_uint(bs);
        if (lo == 0xffffffffu) lo = 0x7fc00000u;            // a NaN either way; the all-ones pattern means "not written yet"
        if (hi == 0xffffffffu) hi = 0x7fc00000u;
        // DUAL: the third sum first (upper word 0: never the all-ones pattern); whoever then sees the first slot written ...
        if (DUAL) __hip_atomic_store(slot2 + s, (unsigned long long)__float_as_uint(cs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + s, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      a = 0.f; bs = 0.f; cs = 0.f;
      // The wait ends when the channel's other workgroups have run, and they run as soon as compute units are free: workgroups are
      // handed out in launch order, so a foreign kernel that holds CUs (an RCCL collective waiting for a slow peer) only delays this
      // one.  No trap (a straggling rank must not abort the process): after two minutes of the 100 MHz clock the lane
      // reports a device-side assertion - raised by the host's next wait - and goes on, so the slots still return to rest.
      unsigned spins = 0;
      unsigned long long t0 = 0;
      bool gave_up = false;
      for (int k = lane; k < S; k += 64) {
        unsigned long long v;
        while ((v = __hip_atomic_load(slot + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull && !gave_up) {
          __builtin_amdgcn_s_sleep(2);
          if ((++spins & 0xfffu) == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 12000000000ull) { gave_up = true; if (assert_word) *assert_word = kAssertBnExchangeTimeout; }
          }
        }
        a += __uint_as_float((unsigned)v); bs += __uint_as_float((unsigned)(v >> 32));
        if (DUAL) {
          // ... polls the second one by itself (two independent relaxed stores are not ordered for the reader)
          unsigned long long v2;
          while ((v2 = __hip_atomic_load(slot2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull && !gave_up) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 0xfffu) == 0) {
              const unsigned long long now = __builtin_amdgcn_s_memrealtime();
              if (t0 == 0) t0 = now;
              else if (now - t0 > 12000000000ull) { gave_up = true; if (assert_word) *assert_word = kAssertBnExchangeTimeout; }
            }
          }
          cs += __uint_as_float((unsigned)v2);
        }
      }
      a = wave_sum(a); bs = wave_sum(bs);
      if (DUAL) cs = wave_sum(cs);
      // every slot of the channel has been read by this workgroup; the last one to say so puts the slots back to rest
      if (lane == 0) {
        const unsigned left = __hip_atomic_fetch_add(depart + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned)S - 1) {
          for (int k = 0; k < S; k++) __hip_atomic_store(slot + k, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (DUAL) for (int k = 0; k < S; k++) __hip_atomic_store(slot2 + k, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(depart + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (lane == 0) {
      stat[0] = a; stat[1] = bs; if (DUAL) stat[2] = cs;
      if (s == 0) {
        if (dweight) dweight[c] = bf16_t(bs * is);
        if (dbias) dbias[c] = bf16_t(a);
        if (DUAL) {
          if (dual.dweight2) dual.dweight2[c] = bf16_t(cs * is2);
          if (dual.dbias2) dual.dbias2[c] = bf16_t(a);
        }
      }
    }
  }
  __syncthreads();
  if (!dx && !dadd) return;
  const float kk = stat[1] * is * is * (float)inv_m, gm = stat[0] * (float)inv_m;
  const float kk2 = DUAL ? stat[2] * is2 * is2 * (float)inv_m : 0.f;
  if (DUAL && dadd) {
    // the second batch norm's input gradient: dx2 = (g - mean(g) - (x2 - mean2) k2) invstd2 w2 from the masked gradient in registers and
    // x2 read once more (groups of <= 4 packets, as the mask's reads above)
#pragma unroll
    for (int h = 0; h < NP; h += HALF) {
      uint4 av[HALF];
#pragma unroll
      for (int k = 0; k < HALF; k++) av[k] = ad4[base[h + k] >= 0 ? base[h + k] : c * vpp];
#pragma unroll
      for (int k = 0; k < HALF; k++) {
        if (base[h + k] < 0) continue;
        uint4 r;
        unsigned* r32 = &r.x;
        const unsigned* g32 = &gv[h + k].x;
        const unsigned* x32 = &av[k].x;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float g0 = __uint_as_float(g32[q] << 16), g1 = __uint_as_float(g32[q] & 0xffff0000u);
          const float x0 = __uint_as_float(x32[q] << 16), x1 = __uint_as_float(x32[q] & 0xffff0000u);
          const bf16_t lo(__builtin_fmaf(-(x0 - mu2), kk2, g0 - gm) * is2 * wc2), hi(__builtin_fmaf(-(x1 - mu2), kk2, g1 - gm) * is2 * wc2);
          r32[q] = (unsigned)lo.bits | ((unsigned)hi.bits << 16);
        }
        reinterpret_cast<uint4*>(dadd)[base[h + k]] = r;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NP; k++) {
    if (base[k] < 0) continue;
    if (!DUAL && dadd) reinterpret_cast<uint4*>(dadd)[base[k]] = gv[k];
    if (dx) {
      uint4 r;
      unsigned* r32 = &r.x;
      const unsigned* g32 = &gv[k].x;
      const unsigned* x32 = &xv[k].x;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float g0 = __uint_as_float(g32[q] << 16), g1 = __uint_as_float(g32[q] & 0xffff0000u);
        const float x0 = __uint_as_float(x32[q] << 16), x1 = __uint_as_float(x32[q] & 0xffff0000u);
        const bf16_t lo(__builtin_fmaf(-(x0 - mu), kk, g0 - gm) * is * wc), hi(__builtin_fmaf(-(x1 - mu), kk, g1 - gm) * is * wc);
        r32[q] = (unsigned)lo.bits | ((unsigned)hi.bits << 16);
      }
      reinterpret_cast<uint4*>(dx)[base[k]] = r;
    }
  }
}

// ---- the one-pass backward in f32 and f64 (round 5): the precisions the reference's CIFAR example runs in ---------------
// (example-cifar100 cifar100.scala:127-129; op ops.scala:2037-2140).  The same protocol as bn_bwd_fused_kernel on 16-byte packets of 4 floats /
// 2 doubles: workgroup (c, s) keeps its NP packets of (masked) dy and of x in registers, publishes its two partial sums, polls the S slots
// of its channel, sums them in a fixed order and writes dx (and the addend's gradient) from the registers: 5 -> 3 passes (7 -> 5 with an
// addend), one launch instead of two.  Sums in the accumulation type of the two-kernel form (f32 / f64).  An f32 activation of the step is
// 67 MB - more than 256 co-resident workgroups hold (33.5 MB per operand) - so the host launches the channels in CHUNKS that are each fully
// co-resident (64 channels x 4 slices for 128 x 2048 x 8 x 8 in f32): c0 is the chunk's first channel, slots are indexed inside the chunk.
// f64 sums need 16 bytes: the second sum travels through a second slot array (each slot is written by one 8-byte atomic store; the reader
// polls both, as the dual bf16 form does).
template <class A> __device__ __forceinline__ unsigned long long bn_slot_bits(A v);
template <> __device__ __forceinline__ unsigned long long bn_slot_bits<double>(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return b == ~0ull ? 0x7ff8000000000000ull : b;            // a NaN either way; the all-ones pattern means "not written yet"
}
template <class T, int NP, bool RELU, bool ADD>
__global__ __launch_bounds__(512) void bn_bwd_fused_fp_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean, const T* __restrict__ invstd,
                                                              const T* __restrict__ w, const T* __restrict__ b, unsigned long long* slots, unsigned long long* slots2,
                                                              unsigned* depart, T* dweight, T* dbias, T* __restrict__ dx, int64_t N, int C, int HW, int S, double inv_m,
                                                              const T* __restrict__ addend, T* __restrict__ dadd, int vshift, int* assert_word, int c0) {
  using A = acc_t<T>;
  constexpr int W = 16 / (int)sizeof(T);
  constexpr bool F64 = sizeof(A) == 8;
  __shared__ A sm[2][8];
  __shared__ A stat[2];
  const int cl = blockIdx.x / S, s = blockIdx.x - cl * S, c = c0 + cl;      // cl: channel inside this launch's chunk
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int vpp = HW / W;
  const int64_t total = N * vpp;                            // packets of this channel
  const A mu = load_as<A>(mean[c]), is = load_as<A>(invstd[c]);
  const A wc = w ? load_as<A>(w[c]) : A(1);
  const A scale = is * wc, bb = (RELU && b) ? load_as<A>(b[c]) : A(0);
  uint4 gv[NP], xv[NP];
  int base[NP];                                             // packet (16-byte) index into the tensors, -1: none (host: packets < 2^31)
  const uint4* dy4 = reinterpret_cast<const uint4*>(dy);
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  const uint4* ad4 = reinterpret_cast<const uint4*>(addend);
#pragma unroll
  for (int k = 0; k < NP; k++) {
    const unsigned i = (unsigned)(k * S + s) * 512u + (unsigned)tid;
    const bool valid = i < (unsigned)total;
    const unsigned ii = valid ? i : 0u;
    const unsigned n = vshift >= 0 ? (ii >> vshift) : (ii / (unsigned)vpp);
    const int idx = (int)((n * (unsigned)C + (unsigned)c) * (unsigned)vpp + (ii - n * (unsigned)vpp));
    base[k] = valid ? idx : -1;
    gv[k] = nt_load16(dy4 + idx);                           // the gradient's last reader
    xv[k] = x4[idx];
  }
#pragma unroll
  for (int k = 0; k < NP; k++)
    if (base[k] < 0) gv[k] = make_uint4(0, 0, 0, 0);        // contributes nothing to the sums, never stored
  A s1 = 0, s2 = 0;
  constexpr int GRP = NP > 4 ? 4 : NP;                      // the addend is only needed for the mask: loaded in groups of <= 4 packets
#pragma unroll
  for (int h = 0; h < NP; h += GRP) {
    uint4 av[GRP];
    if (ADD) {
#pragma unroll
      for (int k = 0; k < GRP; k++) av[k] = ad4[base[h + k] >= 0 ? base[h + k] : c * vpp];
    }
#pragma unroll
    for (int k = 0; k < GRP; k++) {
      T* gp = reinterpret_cast<T*>(&gv[h + k]);
      const T* xp = reinterpret_cast<const T*>(&xv[h + k]);
      const T* ap = reinterpret_cast<const T*>(&av[k]);
#pragma unroll
      for (int q = 0; q < W; q++) {
        const A xx = load_as<A>(xp[q]);
        A gg = load_as<A>(gp[q]);
        if (RELU) {
          T pre = store_as<T>(bn_affine<A>(xx, mu, scale, bb));
          if (ADD) pre = store_as<T>((A)(load_as<A>(pre) + load_as<A>(ap[q])));
          if (load_as<A>(pre) < A(0)) { gg = A(0); gp[q] = store_as<T>(A(0)); }
        }
        s1 += gg;
        if (F64) s2 = (A)__builtin_fma((double)gg, (double)(xx - mu), (double)s2); else s2 = (A)__builtin_fmaf((float)gg, (float)(xx - mu), (float)s2);
      }
      __builtin_amdgcn_sched_barrier(0);                    // the sums are one serial chain (as in the bf16 kernel)
    }
  }
#pragma unroll
  for (int k = 0; k < NP; k++) {
    asm volatile("" : "+v"(gv[k].x), "+v"(gv[k].y), "+v"(gv[k].z), "+v"(gv[k].w));
    asm volatile("" : "+v"(xv[k].x), "+v"(xv[k].y), "+v"(xv[k].z), "+v"(xv[k].w));
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { sm[0][wid] = s1; sm[1][wid] = s2; }
  __syncthreads();
  if (wid == 0) {
    A a = 0, bs = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { a += sm[0][k]; bs += sm[1][k]; }
    if (S > 1) {
      unsigned long long* slot = slots + (int64_t)cl * S;
      unsigned long long* slot2 = slots2 + (int64_t)cl * S;
      if (lane == 0) {
        if (F64) {
          __hip_atomic_store(slot2 + s, bn_slot_bits<double>((double)bs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(slot + s, bn_slot_bits<double>((double)a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          unsigned lo = __float_as_uint((float)a), hi = __float_as_uint((float)bs);
          if (lo == 0xffffffffu) lo = 0x7fc00000u;
          if (hi == 0xffffffffu) hi = 0x7fc00000u;
          __hip_atomic_store(slot + s, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      a = 0; bs = 0;
      unsigned spins = 0;
      unsigned long long t0 = 0;
      bool gave_up = false;
      auto poll = [&](unsigned long long* p) {
        unsigned long long v;
        while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull && !gave_up) {
          __builtin_amdgcn_s_sleep(2);
          if ((++spins & 0xfffu) == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            else if (now - t0 > 12000000000ull) { gave_up = true; if (assert_word) *assert_word = kAssertBnExchangeTimeout; }
          }
        }
        return v;
      };
      for (int k = lane; k < S; k += 64) {
        const unsigned long long v = poll(slot + k);
        if (F64) {
          const unsigned long long v2 = poll(slot2 + k);    // (two independent relaxed stores are not ordered for the reader)
          a += (A)__longlong_as_double((long long)v); bs += (A)__longlong_as_double((long long)v2);
        } else { a += (A)__uint_as_float((unsigned)v); bs += (A)__uint_as_float((unsigned)(v >> 32)); }
      }
      a = wave_sum(a); bs = wave_sum(bs);
      if (lane == 0) {
        const unsigned left = __hip_atomic_fetch_add(depart + cl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned)S - 1) {
          for (int k = 0; k < S; k++) __hip_atomic_store(slot + k, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (F64) for (int k = 0; k < S; k++) __hip_atomic_store(slot2 + k, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(depart + cl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (lane == 0) {
      stat[0] = a; stat[1] = bs;
      if (s == 0) {
        if (dweight) dweight[c] = store_as<T>((A)(bs * is));
        if (dbias) dbias[c] = store_as<T>(a);
      }
    }
  }
  __syncthreads();
  if (!dx && !dadd) return;
  // dx = (g - mean(g) - (x - mean) k) invstd w, k = sum(g (x - mean)) invstd^2 / m: the expression of bn_bwd_apply2_kernel
  const A kk = stat[1] * is * is * (A)inv_m, gm = stat[0] * (A)inv_m;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    if (base[k] < 0) continue;
    if (dadd) reinterpret_cast<uint4*>(dadd)[base[k]] = gv[k];
    if (dx) {
      uint4 r;
      T* rp = reinterpret_cast<T*>(&r);
      const T* gp = reinterpret_cast<const T*>(&gv[k]);
      const T* xp = reinterpret_cast<const T*>(&xv[k]);
#pragma unroll
      for (int q = 0; q < W; q++) rp[q] = store_as<T>((A)((load_as<A>(gp[q]) - gm - (load_as<A>(xp[q]) - mu) * kk) * is * wc));
      reinterpret_cast<uint4*>(dx)[base[k]] = r;
    }
  }
}

// ---- layer norm: rows [M, D] ---------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, T* __restrict__ mean_out,
                                                     T* __restrict__ rstd_out, const T* __restrict__ w, const T* __restrict__ b, int64_t M,
                                                     int64_t D, double eps) {
  using A = acc_t<T>;
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= M) return;
  const T* xp = x + row * D;
  Welford<A> wf{0, 0, 0};
  for (int64_t d = lane; d < D; d += 64) wf_add(wf, load_as<A>(xp[d]));
  wf = wf_wave(wf);
  const A mu = wf.mean;
  const A rstd = A(1) / (A)sqrt((double)(wf.m2 / wf.n + (A)eps));
  if (lane == 0) { mean_out[row] = store_as<T>(mu); rstd_out[row] = store_as<T>(rstd); }
  T* yp = y + row * D;
  for (int64_t d = lane; d < D; d += 64) {
    A v = (load_as<A>(xp[d]) - mu) * rstd;
    if (w) v *= load_as<A>(w[d]);
    if (b) v += load_as<A>(b[d]);
    yp[d] = store_as<T>(v);
  }
}
// dx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat))
template <class T>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                        const T* __restrict__ rstd, const T* __restrict__ w, T* __restrict__ dx, int64_t M,
                                                        int64_t D) {
  using A = acc_t<T>;
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= M) return;
  const A mu = load_as<A>(mean[row]), rs = load_as<A>(rstd[row]);
  const T* xp = x + row * D;
  const T* gp = dy + row * D;
  A s1 = 0, s2 = 0;
  for (int64_t d = lane; d < D; d += 64) {
    const A gw = load_as<A>(gp[d]) * (w ? load_as<A>(w[d]) : A(1));
    const A xh = (load_as<A>(xp[d]) - mu) * rs;
    s1 += gw;
    s2 += gw * xh;
  }
  s1 = wave_sum(s1) / (A)D;
  s2 = wave_sum(s2) / (A)D;
  T* op = dx + row * D;
  for (int64_t d = lane; d < D; d += 64) {
    const A gw = load_as<A>(gp[d]) * (w ? load_as<A>(w[d]) : A(1));
    const A xh = (load_as<A>(xp[d]) - mu) * rs;
    op[d] = store_as<T>((A)(rs * (gw - s1 - xh * s2)));
  }
}
// column sums over rows: dweight[d] = sum_rows dy * xhat ; dbias[d] = sum_rows dy. partial [nsplit][D][2]
template <class T>
__global__ __launch_bounds__(256) void ln_bwd_dwdb_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                          const T* __restrict__ rstd, acc_t<T>* __restrict__ partial, int64_t M, int64_t D,
                                                          int nsplit) {
  using A = acc_t<T>;
  const int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (d >= D) return;
  const int split = blockIdx.y;
  A sw = 0, sb = 0;
  for (int64_t r = split; r < M; r += nsplit) {
    const A g = load_as<A>(dy[r * D + d]);
    sw += g * (load_as<A>(x[r * D + d]) - load_as<A>(mean[r])) * load_as<A>(rstd[r]);
    sb += g;
  }
  partial[((int64_t)split * D + d) * 2] = sw;
  partial[((int64_t)split * D + d) * 2 + 1] = sb;
}
template <class T>
__global__ void ln_bwd_finalize_kernel(const acc_t<T>* __restrict__ partial, T* dweight, T* dbias, int64_t D, int nsplit) {
  using A = acc_t<T>;
  const int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (d >= D) return;
  A a = 0, b = 0;
  for (int s = 0; s < nsplit; s++) { a += partial[((int64_t)s * D + d) * 2]; b += partial[((int64_t)s * D + d) * 2 + 1]; }
  if (dweight) dweight[d] = store_as<T>(a);
  if (dbias) dbias[d] = store_as<T>(b);
}

// ---- vectorised layer norm: a wavefront per row, the row held in registers as up to MAXP 16-byte packets per lane -------------
// (the scalar kernels above re-read the row and pay a division per element in their Welford update: 1.1-1.7 TB/s on 16384 x 4096
// bf16).  Two passes over the REGISTERS: mean, then the centred sum of squares.
template <class T, int MAXP>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, T* __restrict__ mean_out,
                                                         T* __restrict__ rstd_out, const T* __restrict__ w, const T* __restrict__ b, int64_t M,
                                                         int64_t D, double eps) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= M) return;
  const int npk = (int)(D / W);
  const Vec<T, W>* xp = reinterpret_cast<const Vec<T, W>*>(x + row * D);
  Vec<T, W> pk[MAXP];
  A s = 0;
#pragma unroll
  for (int i = 0; i < MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < npk) {
      pk[i] = xp[p];
#pragma unroll
      for (int k = 0; k < W; k++) s += load_as<A>(pk[i].v[k]);
    }
  }
  const A mu = wave_sum(s) / (A)D;
  A q = 0;
#pragma unroll
  for (int i = 0; i < MAXP; i++)
    if (lane + 64 * i < npk) {
#pragma unroll
      for (int k = 0; k < W; k++) { const A d = load_as<A>(pk[i].v[k]) - mu; q += d * d; }
    }
  const A rstd = A(1) / (A)sqrt((double)(wave_sum(q) / (A)D + (A)eps));
  if (lane == 0) { mean_out[row] = store_as<T>(mu); rstd_out[row] = store_as<T>(rstd); }
  Vec<T, W>* yp = reinterpret_cast<Vec<T, W>*>(y + row * D);
#pragma unroll
  for (int i = 0; i < MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < npk) {
      Vec<T, W> wv, bv, o;
      if (w) wv = reinterpret_cast<const Vec<T, W>*>(w)[p];
      if (b) bv = reinterpret_cast<const Vec<T, W>*>(b)[p];
#pragma unroll
      for (int k = 0; k < W; k++) {
        A v = (load_as<A>(pk[i].v[k]) - mu) * rstd;
        if (w) v *= load_as<A>(wv.v[k]);
        if (b) v += load_as<A>(bv.v[k]);
        o.v[k] = store_as<T>(v);
      }
      yp[p] = o;
    }
  }
}
template <class T, int MAXP>
__global__ __launch_bounds__(256) void ln_bwd_dx_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                            const T* __restrict__ rstd, const T* __restrict__ w, T* __restrict__ dx, int64_t M,
                                                            int64_t D) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= M) return;
  const int npk = (int)(D / W);
  const A mu = load_as<A>(mean[row]), rs = load_as<A>(rstd[row]);
  const Vec<T, W>* xp = reinterpret_cast<const Vec<T, W>*>(x + row * D);
  const Vec<T, W>* gp = reinterpret_cast<const Vec<T, W>*>(dy + row * D);
  Vec<T, W> px[MAXP], pg[MAXP];                // x, then x-hat is recomputed; g * w kept in pg's slots as T would lose bits: recompute
  A s1 = 0, s2 = 0;
#pragma unroll
  for (int i = 0; i < MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < npk) {
      px[i] = xp[p]; pg[i] = gp[p];
      Vec<T, W> wv;
      if (w) wv = reinterpret_cast<const Vec<T, W>*>(w)[p];
#pragma unroll
      for (int k = 0; k < W; k++) {
        const A gw = load_as<A>(pg[i].v[k]) * (w ? load_as<A>(wv.v[k]) : A(1));
        const A xh = (load_as<A>(px[i].v[k]) - mu) * rs;
        s1 += gw; s2 += gw * xh;
      }
    }
  }
  s1 = wave_sum(s1) / (A)D;
  s2 = wave_sum(s2) / (A)D;
  Vec<T, W>* op = reinterpret_cast<Vec<T, W>*>(dx + row * D);
#pragma unroll
  for (int i = 0; i < MAXP; i++) {
    const int p = lane + 64 * i;
    if (p < npk) {
      Vec<T, W> wv, o;
      if (w) wv = reinterpret_cast<const Vec<T, W>*>(w)[p];
#pragma unroll
      for (int k = 0; k < W; k++) {
        const A gw = load_as<A>(pg[i].v[k]) * (w ? load_as<A>(wv.v[k]) : A(1));
        const A xh = (load_as<A>(px[i].v[k]) - mu) * rs;
        o.v[k] = store_as<T>((A)(rs * (gw - s1 - xh * s2)));
      }
      op[p] = o;
    }
  }
}
// dweight / dbias partial sums: a thread owns W consecutive columns (16-byte loads), a block row-split `split` of nsplit
template <class T>
__global__ __launch_bounds__(256) void ln_bwd_dwdb_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mean,
                                                              const T* __restrict__ rstd, acc_t<T>* __restrict__ partial, int64_t M, int64_t D,
                                                              int nsplit) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  const int64_t pcol = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;     // packet column
  if (pcol * W >= D) return;
  const int split = blockIdx.y;
  A sw[W], sb[W];
#pragma unroll
  for (int k = 0; k < W; k++) { sw[k] = 0; sb[k] = 0; }
  for (int64_t r = split; r < M; r += nsplit) {
    const Vec<T, W> g = *reinterpret_cast<const Vec<T, W>*>(dy + r * D + pcol * W);
    const Vec<T, W> xv = *reinterpret_cast<const Vec<T, W>*>(x + r * D + pcol * W);
    const A mu = load_as<A>(mean[r]), rs = load_as<A>(rstd[r]);
#pragma unroll
    for (int k = 0; k < W; k++) { const A gg = load_as<A>(g.v[k]); sw[k] += gg * (load_as<A>(xv.v[k]) - mu) * rs; sb[k] += gg; }
  }
#pragma unroll
  for (int k = 0; k < W; k++) {
    partial[((int64_t)split * D + pcol * W + k) * 2] = sw[k];
    partial[((int64_t)split * D + pcol * W + k) * 2 + 1] = sb[k];
  }
}

template <class A> constexpr int acc_dtype() { return std::is_same<A, double>::value ? kF64 : kF32; }

struct BnGeom { int64_t N, C, HW; };
static BnGeom bn_geom(const Tensor* x) {
  LAMP_CHECK(x->ndim >= 2, "batch norm expects at least 2 dims, got " << x->describe());
  BnGeom g{x->sizes[0], x->sizes[1], 1};
  for (int i = 2; i < x->ndim; i++) g.HW *= x->sizes[i];
  return g;
}
static int pick_split(int64_t outputs_blocks, int64_t N) {
  int64_t target = (int64_t)num_cus() * 4;
  int64_t s = target / std::max<int64_t>(outputs_blocks, 1);
  s = std::min<int64_t>(s, N);
  s = std::min<int64_t>(s, 256);
  return (int)std::max<int64_t>(s, 1);
}
// The one-pass backward (bn_bwd_fused_kernel).  Returns false when the geometry does not qualify (the caller runs the two kernels).
static std::atomic<int> g_bn_bwd_mode{-1};
struct BnFusedState { unsigned* sync = nullptr; hipStream_t last = nullptr; bool has_last = false; hipEvent_t ev = nullptr; };
constexpr int BN_FUSED_MAXC = 4096;                         // depart[BN_FUSED_MAXC] (4-byte counters), then slots[BN_FUSED_SLOTS] (8 bytes each)
constexpr int BN_FUSED_SLOTS = 4096;
// the second batch norm of the dual form (bn_bwd_fused_kernel<.., DUAL>): `addc` is then ITS input x2 and `dadd` receives ITS input gradient
struct BnDualHost { const Tensor* mean2; const Tensor* invstd2; const Tensor* w2; const Tensor* b2; Tensor* dw2; Tensor* db2; };
// the per-device counters / slots of the waiting kernels, created on first use; orders this launch behind the previous waiting kernel of
// another stream (two of them must not overlap).  nullptr: the buffer cannot be created from this thread (the tensor's device is not current).
static unsigned* bn_fused_sync_state(int device, hipStream_t st) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  const bool capturing = cap == hipStreamCaptureStatusActive;
  static std::mutex mu;
  static std::map<int, BnFusedState> states;
  std::lock_guard<std::mutex> lk(mu);
  BnFusedState& stt = states[device];
  if (!stt.sync) {
    if (current_device() != device) return nullptr;
    HIP_CHECK(hipMalloc((void**)&stt.sync, BN_FUSED_MAXC * sizeof(unsigned) + 2 * BN_FUSED_SLOTS * sizeof(unsigned long long)));
    hipStream_t side = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIP_CHECK(hipMemsetAsync(stt.sync, 0, BN_FUSED_MAXC * sizeof(unsigned), side));
    HIP_CHECK(hipMemsetAsync(stt.sync + BN_FUSED_MAXC, 0xff, 2 * BN_FUSED_SLOTS * sizeof(unsigned long long), side));
    HIP_CHECK(hipStreamSynchronize(side));
    HIP_CHECK(hipStreamDestroy(side));
    HIP_CHECK(hipEventCreateWithFlags(&stt.ev, hipEventDisableTiming));
  }
  if (!capturing) {
    if (stt.has_last && stt.last != st) {
      HIP_CHECK(hipEventRecord(stt.ev, stt.last));
      HIP_CHECK(hipStreamWaitEvent(st, stt.ev, 0));
    }
    stt.last = st; stt.has_last = true;
  }
  return stt.sync;
}
// f32 / f64 (bn_bwd_fused_fp_kernel): the channels are launched in chunks whose workgroups are all co-resident
template <class T>
static bool bn_bwd_fused_fp_launch(const Tensor* gc, const Tensor* xc, const Tensor* mean_t, const Tensor* invstd_t, const Tensor* weight, const Tensor* bias,
                                   Tensor* dw, Tensor* db, Tensor* dx, Tensor* dadd, const Tensor* addc, const BnGeom& g, int relu, hipStream_t st) {
  static const bool env_on = [] { const char* e = getenv("LAMP_BN_FUSED_BWD"); return !(e && e[0] == '0'); }();
  static const bool fp_on = [] { const char* e = getenv("LAMP_BN_FUSED_FP"); return !(e && e[0] == '0'); }();
  const int mode = g_bn_bwd_mode.load(std::memory_order_relaxed);
  const bool on = mode < 0 ? (env_on && fp_on) : mode >= 1;
  if (on && mode != 2 && device_shared(xc->device()) > 0) return false;
  constexpr int W = 16 / (int)sizeof(T);
  if (!on || g.HW % W != 0) return false;
  const int64_t packets = g.N * (g.HW / W);                 // per channel
  if (packets <= 0 || packets >= (int64_t)1 << 30 || xc->numel() / W >= (int64_t)1 << 31) return false;
  if (addc && !relu) return false;
  const int cus = num_cus();
  auto per_thread = [&](int64_t s) { return (packets + s * 512 - 1) / (s * 512); };
  int64_t s = std::max<int64_t>(1, (int64_t)cus / g.C);
  s = std::min<int64_t>(s, (packets + 511) / 512);
  if (per_thread(s) > 16) s = (packets + 16 * 512 - 1) / (16 * 512);
  const int ppt = (int)per_thread(s);
  const int NP = ppt <= 1 ? 1 : ppt <= 2 ? 2 : ppt <= 4 ? 4 : ppt <= 8 ? 8 : 16;
  static const int np_mask = [] { const char* e = getenv("LAMP_BN_FUSED_NP_MASK"); return e ? atoi(e) : 24; }();   // as the bf16 form: the large activations
  if (!(np_mask & NP)) return false;
#define BN_FP_K(NPv) (addc ? (const void*)bn_bwd_fused_fp_kernel<T, NPv, true, true> : relu ? (const void*)bn_bwd_fused_fp_kernel<T, NPv, true, false> \
                           : (const void*)bn_bwd_fused_fp_kernel<T, NPv, false, false>)
  const void* kfn = NP == 1 ? BN_FP_K(1) : NP == 2 ? BN_FP_K(2) : NP == 4 ? BN_FP_K(4) : NP == 8 ? BN_FP_K(8) : BN_FP_K(16);
#undef BN_FP_K
  const int64_t resident = std::min<int64_t>(BN_FUSED_SLOTS, (int64_t)cus * std::max(1, kernel_occupancy(kfn, 512, 0)));
  if (s > resident) return false;                           // one channel's slices alone do not fit the chip
  const int64_t chunk = std::min<int64_t>(std::min<int64_t>(g.C, resident / s), BN_FUSED_MAXC);
  unsigned* sync = bn_fused_sync_state(xc->device(), st);
  if (!sync) return false;
  unsigned* departp = sync;
  unsigned long long* slotp = reinterpret_cast<unsigned long long*>(sync + BN_FUSED_MAXC);
  unsigned long long* slot2p = slotp + BN_FUSED_SLOTS;
  const double passes = (dx ? 3.0 : 2.0) + (addc ? 1.0 : 0.0) + (dadd ? 1.0 : 0.0);
  KernelTimer kt("bn_bwd_fused", 0, passes * (double)xc->numel() * sizeof(T), st);
  const T* dyp = gc->ptr<T>(); const T* xp = xc->ptr<T>();
  const T* mp = mean_t->ptr<T>(); const T* ip = invstd_t->ptr<T>();
  const T* wp = weight ? weight->ptr<T>() : (const T*)nullptr;
  const T* bp = bias ? bias->ptr<T>() : (const T*)nullptr;
  T* dwp = dw ? dw->ptr<T>() : (T*)nullptr; T* dbp = db ? db->ptr<T>() : (T*)nullptr;
  T* dxp = dx ? dx->ptr<T>() : (T*)nullptr;
  int64_t a_N = g.N; int a_C = (int)g.C, a_HW = (int)g.HW, a_S = (int)s;
  double inv_m = 1.0 / (double)(g.N * g.HW);
  const int vppi = (int)(g.HW / W);
  int a_vshift = -1;
  for (int b = 0; b < 31; b++) if (vppi == (1 << b)) a_vshift = b;
  const T* adp = addc ? addc->ptr<T>() : (const T*)nullptr;
  T* dap = dadd ? dadd->ptr<T>() : (T*)nullptr;
  int* awp = device_assert_word(xc->device());
  for (int64_t c0 = 0; c0 < g.C; c0 += chunk) {
    int a_c0 = (int)c0;
    const int64_t nc = std::min<int64_t>(chunk, g.C - c0);
    void* args[] = {(void*)&dyp, (void*)&xp, (void*)&mp, (void*)&ip, (void*)&wp, (void*)&bp, (void*)&slotp, (void*)&slot2p, (void*)&departp, (void*)&dwp, (void*)&dbp,
                    (void*)&dxp, (void*)&a_N, (void*)&a_C, (void*)&a_HW, (void*)&a_S, (void*)&inv_m, (void*)&adp, (void*)&dap, (void*)&a_vshift, (void*)&awp, (void*)&a_c0};
    HIP_CHECK(hipLaunchKernel(kfn, dim3((unsigned)(nc * s)), dim3(512), args, 0, st));
  }
  return true;
}
// gplanes (instead of gc): the gradient as one value per plane (see plane_broadcast_base)
struct BnPlaneGrad { const bf16_t* p = nullptr; int sn = 0, sc = 0; };
static bool bn_bwd_fused_launch(const Tensor* gc, const Tensor* xc, const Tensor* mean_t, const Tensor* invstd_t, const Tensor* weight, const Tensor* bias,
                                Tensor* dw, Tensor* db, Tensor* dx, Tensor* dadd, const Tensor* addc, const BnGeom& g, int relu, hipStream_t st,
                                const BnDualHost* dualh = nullptr, BnPlaneGrad gplanes = BnPlaneGrad()) {
  static const bool env_on = [] { const char* e = getenv("LAMP_BN_FUSED_BWD"); return !(e && e[0] == '0'); }();
  const int mode = g_bn_bwd_mode.load(std::memory_order_relaxed);
  const bool on = mode < 0 ? env_on : mode >= 1;
  // a device that is also running kernels this library does not schedule (the overlapped RCCL all-reduce of the eager data-parallel
  // step): the waiting workgroups would sit on their CUs until the foreign kernel lets the rest of their channel in - two kernels then
  if (on && mode != 2 && device_shared(xc->device()) > 0) return false;
  if (!on || g.C > BN_FUSED_MAXC || g.HW % 8 != 0 || xc->numel() >= (int64_t)1 << 34) return false;
  const int64_t packets = g.N * (g.HW / 8);                 // per channel
  if (packets <= 0 || packets >= (int64_t)1 << 30) return false;
  if (addc && !relu) return false;
  const int cus = num_cus();
  static const int per_cu = [] { const char* e = getenv("LAMP_BN_FUSED_PER_CU"); return e ? std::max(1, atoi(e)) : 1; }();   // measured: 2 is slower
  auto per_thread = [&](int64_t s) { return (packets + s * 512 - 1) / (s * 512); };
  // the scalar-load form of a plane gradient: 8 x 8 maps, class-major values in whole 16-byte groups (a wave's 64 packets are eight whole planes), the
  // dual form (the only consumer of the loss tail's gradient in Cnn.resnet); anything else gets the materialised tensor from the caller
  const bool planes = gplanes.p != nullptr;
  if (planes && !(dualh && g.HW == 64 && gplanes.sn == 1 && g.N % 8 == 0 && gplanes.sc % 8 == 0 && ((uintptr_t)gplanes.p & 15) == 0)) return false;
  const void* kfn = nullptr;
  int64_t S = 0;
  for (int wgs = per_cu; wgs >= 1 && !kfn; wgs--) {        // workgroups per CU aimed at: more, smaller ones first
    int64_t s = std::max<int64_t>(1, (int64_t)cus * wgs / g.C);
    s = std::min<int64_t>(s, (packets + 511) / 512);
    if (per_thread(s) > 16) s = (packets + 16 * 512 - 1) / (16 * 512);
    const int ppt = (int)per_thread(s);
    const int NP = ppt <= 1 ? 1 : ppt <= 2 ? 2 : ppt <= 4 ? 4 : ppt <= 8 ? 8 : 16;
    // Which sizes take this path (bit = packets per thread).  Default: 8 and 16, i.e. activations of some 10 MB and more.  Below that the
    // exchange between the workgroups (two memory round trips, ~3 us) costs what the second pass over an L2 / MALL-resident tensor
    // costs: the ResNet step's six small layers were 4 us SLOWER in total with it, the six large ones 50 us faster.
    // The dual form replaces FOUR kernels (two reductions, two applies) and two passes more: taken at every size.
    static const int np_mask = [] { const char* e = getenv("LAMP_BN_FUSED_NP_MASK"); return e ? atoi(e) : 24; }();
    // Activations of at most 4 MiB (the B <= 256 steps): both kernels of the two-pass form sit at the launch floor there, and one launch with an
    // exchange beats two (A/B: B = 256 0.5433 -> 0.5359 ms per step, B = 32 0.4671 -> 0.4545; the 4 MiB layer itself 0.5137 -> 0.5115); above
    // (B = 2048's small maps, 4 - 8 MB) it is even.
    static const int64_t small_bytes = [] { const char* e = getenv("LAMP_BN_FUSED_SMALL_BYTES"); return e ? (int64_t)atoll(e) : ((int64_t)4 << 20) + 1; }();
    if (!dualh && !(np_mask & NP) && !(xc->numel() * 2 < small_bytes)) continue;
#define BN_FUSED_K(NPv) (dualh ? (planes ? (const void*)bn_bwd_fused_kernel<NPv, true, true, true, true> : (const void*)bn_bwd_fused_kernel<NPv, true, true, true>) \
                               : addc ? (const void*)bn_bwd_fused_kernel<NPv, true, true> \
                               : relu ? (const void*)bn_bwd_fused_kernel<NPv, true, false> : (const void*)bn_bwd_fused_kernel<NPv, false, false>)
    const void* k = NP == 1 ? BN_FUSED_K(1) : NP == 2 ? BN_FUSED_K(2) : NP == 4 ? BN_FUSED_K(4) : NP == 8 ? BN_FUSED_K(8) : BN_FUSED_K(16);
#undef BN_FUSED_K
    // every workgroup co-resident: nobody waits for a workgroup that has no slot yet
    if (g.C * s <= std::min<int64_t>(BN_FUSED_SLOTS, (int64_t)cus * std::max(1, kernel_occupancy(k, 512, 0)))) { kfn = k; S = s; }
  }
  if (!kfn) return false;
  // one counter set per device and workgroups that wait for each other: two of these kernels must not overlap.  Same stream: ordered anyway.
  // Another stream: this launch waits for everything queued there so far.  (A graph replayed on one stream while another thread runs eagerly
  // on a second one is not covered: LAMP_BN_FUSED_BWD=0 for such a program.)  The buffer is zeroed once, on a stream of its own and waited
  // for: st may be capturing, and the counters must be zero in memory before the first launch really runs.
  unsigned* sync = bn_fused_sync_state(xc->device(), st);
  if (!sync) return false;                                  // (callers run with the tensor's device current; the buffer must live there)
  unsigned* departp = sync;
  unsigned long long* slotp = reinterpret_cast<unsigned long long*>(sync + BN_FUSED_MAXC);
  const double passes = (dx ? 3.0 : 2.0) + (addc ? 1.0 : 0.0) + (dadd ? 1.0 : 0.0) - (gplanes.p ? 1.0 : 0.0);   // (the dual form's second read of x2 is served by the caches)
  KernelTimer kt("bn_bwd_fused", 0, passes * (double)xc->numel() * 2.0, st);
  const bf16_t* dyp = gplanes.p ? gplanes.p : gc->ptr<bf16_t>(); const bf16_t* xp = xc->ptr<bf16_t>();
  const bf16_t* mp = mean_t->ptr<bf16_t>(); const bf16_t* ip = invstd_t->ptr<bf16_t>();
  const bf16_t* wp = weight ? weight->ptr<bf16_t>() : (const bf16_t*)nullptr;
  const bf16_t* bp = bias ? bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
  bf16_t* dwp = dw ? dw->ptr<bf16_t>() : (bf16_t*)nullptr; bf16_t* dbp = db ? db->ptr<bf16_t>() : (bf16_t*)nullptr;
  bf16_t* dxp = dx ? dx->ptr<bf16_t>() : (bf16_t*)nullptr;
  int64_t a_N = g.N; int a_C = (int)g.C, a_HW = (int)g.HW, a_S = (int)S, a_relu = relu, a_sc = gplanes.sc;
  double inv_m = 1.0 / (double)(g.N * g.HW);
  const int vppi = (int)(g.HW / 8);
  int a_vshift = -1;                                        // packets per image row a power of two: a shift instead of a division
  for (int b = 0; b < 31; b++) if (vppi == (1 << b)) a_vshift = b;
  const bf16_t* adp = addc ? addc->ptr<bf16_t>() : (const bf16_t*)nullptr;
  bf16_t* dap = dadd ? dadd->ptr<bf16_t>() : (bf16_t*)nullptr;
  int* awp = device_assert_word(xc->device());
  BnFusedDual dual{};
  if (dualh) {
    dual.mean2 = dualh->mean2->ptr<bf16_t>(); dual.invstd2 = dualh->invstd2->ptr<bf16_t>();
    dual.w2 = dualh->w2 ? dualh->w2->ptr<bf16_t>() : (const bf16_t*)nullptr; dual.b2 = dualh->b2 ? dualh->b2->ptr<bf16_t>() : (const bf16_t*)nullptr;
    dual.dweight2 = dualh->dw2 ? dualh->dw2->ptr<bf16_t>() : (bf16_t*)nullptr; dual.dbias2 = dualh->db2 ? dualh->db2->ptr<bf16_t>() : (bf16_t*)nullptr;
    dual.slots2 = slotp + BN_FUSED_SLOTS;
  }
  void* args[] = {(void*)&dyp, (void*)&xp, (void*)&mp, (void*)&ip, (void*)&wp, (void*)&bp, (void*)&slotp, (void*)&departp, (void*)&dwp, (void*)&dbp, (void*)&dxp,
                  (void*)&a_N, (void*)&a_C, (void*)&a_HW, (void*)&a_S, (void*)&inv_m, (void*)&a_relu, (void*)&adp, (void*)&dap, (void*)&a_vshift, (void*)&awp,
                  (void*)&dual, (void*)&a_sc};
  HIP_CHECK(hipLaunchKernel(kfn, dim3((unsigned)(g.C * S)), dim3(512), args, 0, st));
  return true;
}
// A gradient that is constant over every plane, handed over as a view expanded over the map (strides [sn, sc, 0, 0]) of one value per (n, c):
// where those values start and how they are laid out, or {nullptr}.  (Produced by the loss tail, lamp_nll_loss_forward_pooled_gradient_.)
static BnPlaneGrad plane_broadcast_base(const Tensor* g) {
  BnPlaneGrad r;
  if (!g || g->dtype != kBF16 || g->ndim != 4 || !g->is_device()) return r;
  if (g->sizes[2] * g->sizes[3] <= 1 || g->strides[2] != 0 || g->strides[3] != 0) return r;
  const int64_t sn = g->sizes[0] > 1 ? g->strides[0] : 0, sc = g->sizes[1] > 1 ? g->strides[1] : 0;
  if (sn < 0 || sc < 0 || sn >= ((int64_t)1 << 24) || sc >= ((int64_t)1 << 24) || (g->sizes[0] - 1) * sn + (g->sizes[1] - 1) * sc >= ((int64_t)1 << 31)) return r;
  r.p = g->ptr<bf16_t>(); r.sn = (int)sn; r.sc = (int)sc;
  return r;
}
static void check_cvec(const Tensor* t, int64_t C, int dtype, const char* what) {
  if (!t) return;
  check_device_tensor(t, what);
  LAMP_CHECK(t->numel() == C && t->dtype == dtype && t->is_contiguous(), what << " must be a contiguous [" << C << "] tensor of the input dtype, got " << t->describe());
}

}  // namespace lamp

using namespace lamp;

extern "C" {

int lamp_bn_backward_mode(int mode) {
  LAMP_API_BEGIN
  LAMP_CHECK(mode >= -1 && mode <= 2, "lamp_bn_backward_mode: mode must be -1, 0, 1 or 2");
  g_bn_bwd_mode.store(mode, std::memory_order_relaxed);
  LAMP_API_END
}

// the batch norm whose output is the addend of lamp_native_batch_norm2_add_relu (statistics, running statistics and saved tensors of
// its own; normalised inside the first one's kernel)
struct BnSecondArgs {
  const lamp_tensor* x; const lamp_tensor* weight; const lamp_tensor* bias; lamp_tensor* running_mean; lamp_tensor* running_var; double momentum, eps;
  lamp_tensor* save_mean = nullptr; lamp_tensor* save_invstd = nullptr;        // results
};
static int bn_forward_impl(lamp_tensor* out3[3], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias,
                           lamp_tensor* running_mean, lamp_tensor* running_var, int training, double momentum, double eps, int relu,
                           const lamp_tensor* addend = nullptr, BnSecondArgs* second = nullptr, bool table_only = false, bool* pooled = nullptr) {
  // pooled (in: asked for, out: done): out3[0] = the [N, C] plane means of the output instead of the output (bn_apply2_kernel), where the
  // two-batch-norm form runs its vector path over planes of a power-of-two number of packets; otherwise *pooled = false and out3[0] is the output
  LAMP_API_BEGIN
  check_device_tensor(x, "input");
  BnGeom g = bn_geom(x);
  const bool want_pooled = pooled && *pooled;
  if (pooled) *pooled = false;
  LAMP_CHECK(!table_only || (training && !addend && !second && x->dtype != kF64), "the batch norm table exists in training mode, as an f32 table for f32 / f16 / bf16 inputs");
  Hold addc, x2c, mean2, invstd2;
  if (second) {
    check_device_tensor(second->x, "second input");
    LAMP_CHECK(!addend && relu && training && g.HW >= 64, "the two-batch-norm residual tail exists in training mode for maps of at least 64 elements");
    LAMP_CHECK(second->x->shape() == x->shape() && second->x->dtype == x->dtype, "second input " << second->x->describe() << " does not match input " << x->describe());
    check_cvec(second->weight, g.C, x->dtype, "second weight"); check_cvec(second->bias, g.C, x->dtype, "second bias");
    check_cvec(second->running_mean, g.C, x->dtype, "second running_mean"); check_cvec(second->running_var, g.C, x->dtype, "second running_var");
    x2c = Hold(contiguous(second->x));
    int64_t cs2[1] = {g.C};
    mean2 = Hold(new_tensor(cs2, 1, x->dtype, x->device())); invstd2 = Hold(new_tensor(cs2, 1, x->dtype, x->device()));
  }
  if (addend) {
    check_device_tensor(addend, "addend");
    LAMP_CHECK(relu && training && g.HW >= 64, "the fused batch-norm-add-relu exists in training mode for maps of at least 64 elements");
    LAMP_CHECK(addend->shape() == x->shape() && addend->dtype == x->dtype, "addend " << addend->describe() << " does not match input " << x->describe());
    addc = Hold(contiguous(addend));
  }
  check_cvec(weight, g.C, x->dtype, "weight"); check_cvec(bias, g.C, x->dtype, "bias");
  check_cvec(running_mean, g.C, x->dtype, "running_mean"); check_cvec(running_var, g.C, x->dtype, "running_var");
  Hold xc(contiguous(x));
  int64_t cs[1] = {g.C}, ts[2] = {g.C, 4};
  Hold y(table_only ? new_tensor(ts, 2, kF32, x->device()) : new_like(xc.get()));   // table_only: out3[0] is the [C, 4] table, x is never normalised here
  Hold pooled_t;
  Hold mean(new_tensor(cs, 1, x->dtype, x->device())), invstd(new_tensor(cs, 1, x->dtype, x->device()));
  hipStream_t st = current_stream(x->device());
  const int64_t total = x->numel();
  LAMP_DISPATCH_FLOAT(x->dtype, T, {
    using A = acc_t<T>;
    const int vec = (g.HW % (16 / sizeof(T)) == 0) && (((uintptr_t)xc->raw() | (uintptr_t)y->raw() | (uintptr_t)(x2c.get() ? x2c->raw() : nullptr)) & 15) == 0;
    float4* table = table_only ? reinterpret_cast<float4*>(y->raw()) : nullptr;
    if (training) {
      LAMP_CHECK(g.N * g.HW > 0, "batch norm over an empty batch");
      const bool col = g.HW < 64;
      const int64_t blocks = col ? (g.C + 255) / 256 : g.C;
      const int nsplit = pick_split(blocks, g.N);
      // statistics handed over by the convolution that produced x (its epilogue computed per-image Welford triples): no stats pass
      int npart = nsplit;
      Hold partial;
      if (!col && std::is_same<A, float>::value) partial = Hold(conv_stats_lookup(xc.get(), g.C, &npart));
      const bool have_stats = partial.get() != nullptr;
      if (!have_stats) {
        npart = nsplit;
        int64_t ps[1] = {(int64_t)nsplit * g.C * 3};
        partial = Hold(new_tensor(ps, 1, acc_dtype<A>(), x->device()));
      }
      T* rm = running_mean ? running_mean->ptr<T>() : (T*)nullptr;
      T* rv = running_var ? running_var->ptr<T>() : (T*)nullptr;
      const T* wp = weight ? weight->ptr<T>() : (const T*)nullptr;
      const T* bp = bias ? bias->ptr<T>() : (const T*)nullptr;
      if (!have_stats) {
        KernelTimer kt("bn_fwd_stats", 0, (double)total * sizeof(T), st);
        if (col) hipLaunchKernelGGL((bn_stats_col_kernel<T>), dim3((unsigned)blocks, nsplit), dim3(256), 0, st, static_cast<const Tensor*>(xc.get())->ptr<T>(), partial->ptr<A>(), g.N, g.C, g.HW, nsplit);
        else hipLaunchKernelGGL((bn_stats_kernel<T>), dim3((unsigned)g.C, nsplit), dim3(256), 0, st, static_cast<const Tensor*>(xc.get())->ptr<T>(), partial->ptr<A>(), g.N, g.C, g.HW, nsplit, vec);
        LAMP_LAUNCH_CHECK();
      }
      // the second batch norm's statistics: handed over by ITS convolution, or one statistics pass over its input
      BnSecond<T> sec{};
      Hold partial2;
      if (second) {
        LAMP_CHECK(!col && total > 0, "the two-batch-norm residual tail needs maps of at least 64 elements");
        int npart2 = nsplit;
        if (std::is_same<A, float>::value) partial2 = Hold(conv_stats_lookup(x2c.get(), g.C, &npart2));
        const bool have2 = partial2.get() != nullptr;
        if (!have2) {
          npart2 = nsplit;
          int64_t ps[1] = {(int64_t)nsplit * g.C * 3};
          partial2 = Hold(new_tensor(ps, 1, acc_dtype<A>(), x->device()));
          KernelTimer kt("bn_fwd_stats", 0, (double)total * sizeof(T), st);
          hipLaunchKernelGGL((bn_stats_kernel<T>), dim3((unsigned)g.C, nsplit), dim3(256), 0, st, static_cast<const Tensor*>(x2c.get())->ptr<T>(), partial2->ptr<A>(), g.N, g.C, g.HW, nsplit, vec);
          LAMP_LAUNCH_CHECK();
        }
        sec.x = static_cast<const Tensor*>(x2c.get())->ptr<T>();
        sec.partial = static_cast<const Tensor*>(partial2.get())->ptr<A>(); sec.nsplit = have2 ? -npart2 : npart2;
        sec.save_mean = mean2->ptr<T>(); sec.save_invstd = invstd2->ptr<T>();
        sec.running_mean = second->running_mean ? second->running_mean->ptr<T>() : (T*)nullptr;
        sec.running_var = second->running_var ? second->running_var->ptr<T>() : (T*)nullptr;
        sec.momentum = second->momentum; sec.eps = second->eps;
        sec.w = second->weight ? second->weight->ptr<T>() : (const T*)nullptr; sec.b = second->bias ? second->bias->ptr<T>() : (const T*)nullptr;
      }
      if (table_only && !col) {
        hipLaunchKernelGGL((bn_affine_table_kernel<T>), dim3((unsigned)g.C), dim3(256), 0, st, static_cast<const Tensor*>(partial.get())->ptr<A>(),
                           have_stats ? -npart : npart, mean->ptr<T>(), invstd->ptr<T>(), rm, rv, momentum, eps, wp, bp, g.C, table);
        LAMP_LAUNCH_CHECK();
      } else if (!col && total > 0) {
        // finalize folded into the channel-aligned normalise
        const int vec2 = (vec && (!addc.get() || ((uintptr_t)addc->data() & 15) == 0)) ? 1 : 0;
        const int64_t vpp = g.HW / (int64_t)(16 / sizeof(T));
        T* poolp = nullptr;
        if (want_pooled && second && vec2 && vpp >= 1 && vpp <= 64 && (vpp & (vpp - 1)) == 0) {
          int64_t ps[2] = {g.N, g.C};
          pooled_t = Hold(new_tensor(ps, 2, x->dtype, x->device()));
          poolp = pooled_t->ptr<T>();
          *pooled = true;
        }
        KernelTimer kt("bn_fwd_apply", 0, ((second ? 3.0 : addc.get() ? 3.0 : 2.0) - (poolp ? 1.0 : 0.0)) * (double)total * sizeof(T), st);
        hipLaunchKernelGGL((bn_apply2_kernel<T>), dim3((unsigned)g.C, nsplit), dim3(256), 0, st, static_cast<const Tensor*>(xc.get())->ptr<T>(), y->ptr<T>(),
                           static_cast<const Tensor*>(partial.get())->ptr<A>(), have_stats ? -npart : npart,
                           mean->ptr<T>(), invstd->ptr<T>(), rm, rv, momentum, eps, wp, bp, g.N, g.C, g.HW, nsplit,
                           vec2, relu, addc.get() ? addc->ptr<T>() : (const T*)nullptr, sec, poolp);
        LAMP_LAUNCH_CHECK();
      } else {
        hipLaunchKernelGGL((bn_finalize_kernel<T>), dim3((unsigned)((g.C * 64 + 255) / 256)), dim3(256), 0, st, partial->ptr<A>(), g.C, nsplit,
                           mean->ptr<T>(), invstd->ptr<T>(), rm, rv, momentum, eps);
        LAMP_LAUNCH_CHECK();
        if (table_only) {
          hipLaunchKernelGGL((bn_affine_from_saved_kernel<T>), dim3((unsigned)((g.C + 255) / 256)), dim3(256), 0, st, mean->ptr<T>(), invstd->ptr<T>(), wp, bp, g.C, table);
          LAMP_LAUNCH_CHECK();
        } else if (total > 0) {
          KernelTimer kt("bn_fwd_apply", 0, 2.0 * (double)total * sizeof(T), st);
          hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, xc->ptr<T>(), y->ptr<T>(), mean->ptr<T>(), invstd->ptr<T>(),
                             wp, bp, total, g.C, g.HW, vec, relu);
          LAMP_LAUNCH_CHECK();
        }
      }
    } else {
      hipLaunchKernelGGL((bn_eval_stats_kernel<T>), dim3((unsigned)((g.C + 255) / 256)), dim3(256), 0, st,
                         running_mean ? running_mean->ptr<T>() : (const T*)nullptr, running_var ? running_var->ptr<T>() : (const T*)nullptr,
                         mean->ptr<T>(), invstd->ptr<T>(), g.C, eps);
      LAMP_LAUNCH_CHECK();
      if (total > 0) {
        KernelTimer kt("bn_fwd_apply", 0, 2.0 * (double)total * sizeof(T), st);
        hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, xc->ptr<T>(), y->ptr<T>(), mean->ptr<T>(),
                           invstd->ptr<T>(), weight ? weight->ptr<T>() : (const T*)nullptr, bias ? bias->ptr<T>() : (const T*)nullptr, total,
                           g.C, g.HW, vec, relu);
        LAMP_LAUNCH_CHECK();
      }
    }
  });
  if (!training) {
    // ATen returns empty save tensors in eval mode; keep handles valid but zero-sized semantics are not needed by lamp
  }
  out3[0] = pooled_t.get() ? pooled_t.take() : y.take(); out3[1] = mean.take(); out3[2] = invstd.take();
  if (second) { second->save_mean = mean2.take(); second->save_invstd = invstd2.take(); }
  LAMP_API_END
}
int lamp_native_batch_norm(lamp_tensor* out3[3], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias,
                           lamp_tensor* running_mean, lamp_tensor* running_var, int training, double momentum, double eps) {
  return bn_forward_impl(out3, x, weight, bias, running_mean, running_var, training, momentum, eps, 0);
}
int lamp_native_batch_norm_relu(lamp_tensor* out3[3], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias,
                                lamp_tensor* running_mean, lamp_tensor* running_var, int training, double momentum, double eps) {
  return bn_forward_impl(out3, x, weight, bias, running_mean, running_var, training, momentum, eps, 1);
}

// out3 = (affine f32 [C, 4], save_mean, save_invstd): the training-mode batch norm of x as the table a consumer convolution applies
// (lamp_convolution_bn_relu_input); running statistics are updated exactly as lamp_native_batch_norm updates them
int lamp_batch_norm_affine(lamp_tensor* out3[3], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias, lamp_tensor* running_mean,
                           lamp_tensor* running_var, double momentum, double eps) {
  return bn_forward_impl(out3, x, weight, bias, running_mean, running_var, 1, momentum, eps, 1, nullptr, nullptr, true);
}

static int bn_backward_impl(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* weight,
                            const lamp_tensor* bias, const lamp_tensor* running_mean, const lamp_tensor* running_var,
                            const lamp_tensor* save_mean, const lamp_tensor* save_invstd, int training, double eps, const uint8_t mask[3],
                            int relu, const lamp_tensor* addend = nullptr, lamp_tensor** daddend_out = nullptr) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(grad_out, "grad_out");
  LAMP_CHECK(grad_out->shape() == x->shape() && grad_out->dtype == x->dtype, "grad_out " << grad_out->describe() << " does not match input " << x->describe());
  BnGeom g = bn_geom(x);
  check_cvec(weight, g.C, x->dtype, "weight");
  if (relu) {
    check_cvec(bias, g.C, x->dtype, "bias");
    LAMP_CHECK(g.HW >= 64, "the fused batch-norm-relu backward exists for maps of at least 64 elements, got " << x->describe());
  }
  Hold xc(contiguous(x)), gc(contiguous(grad_out));
  Hold addc, dadd;
  if (addend) {
    check_device_tensor(addend, "addend");
    LAMP_CHECK(relu && training, "the fused batch-norm-add-relu backward exists in training mode");
    LAMP_CHECK(addend->shape() == x->shape() && addend->dtype == x->dtype, "addend " << addend->describe() << " does not match input " << x->describe());
    addc = Hold(contiguous(addend));
    if (daddend_out) dadd = Hold(new_like(xc.get()));
  }
  hipStream_t st = current_stream(x->device());
  int64_t cs[1] = {g.C};
  Hold mean_h, invstd_h;
  const Tensor* mean_t = save_mean;
  const Tensor* invstd_t = save_invstd;
  if (training) {
    LAMP_CHECK(save_mean && save_invstd, "training-mode batch norm backward needs save_mean and save_invstd");
    check_cvec(save_mean, g.C, x->dtype, "save_mean"); check_cvec(save_invstd, g.C, x->dtype, "save_invstd");
  } else {
    check_cvec(running_mean, g.C, x->dtype, "running_mean"); check_cvec(running_var, g.C, x->dtype, "running_var");
    mean_h = Hold(new_tensor(cs, 1, x->dtype, x->device()));
    invstd_h = Hold(new_tensor(cs, 1, x->dtype, x->device()));
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((bn_eval_stats_kernel<T>), dim3((unsigned)((g.C + 255) / 256)), dim3(256), 0, st,
                                                        running_mean ? running_mean->ptr<T>() : (const T*)nullptr,
                                                        running_var ? running_var->ptr<T>() : (const T*)nullptr, mean_h->ptr<T>(),
                                                        invstd_h->ptr<T>(), g.C, eps));
    LAMP_LAUNCH_CHECK();
    mean_t = mean_h.get(); invstd_t = invstd_h.get();
  }
  Hold dx(mask[0] ? new_like(xc.get()) : nullptr);
  Hold dw(mask[1] ? new_tensor(cs, 1, x->dtype, x->device()) : nullptr);
  Hold db(mask[2] ? new_tensor(cs, 1, x->dtype, x->device()) : nullptr);
  const int64_t total = x->numel();
  LAMP_DISPATCH_FLOAT(x->dtype, T, {
    using A = acc_t<T>;
    const int vec = (g.HW % (16 / sizeof(T)) == 0) && (((uintptr_t)xc->data() | (uintptr_t)gc->data() | (uintptr_t)(dx.get() ? dx->data() : nullptr) |
                                                          (uintptr_t)(addc.get() ? addc->data() : nullptr) | (uintptr_t)(dadd.get() ? dadd->data() : nullptr)) & 15) == 0;
    const bool col = g.HW < 64;
    bool fused_done = false;
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (training && vec && !col && total > 0 && (dx.get() || dadd.get()))
        fused_done = bn_bwd_fused_launch(gc.get(), xc.get(), mean_t, invstd_t, weight, bias, dw.get(), db.get(), dx.get(), dadd.get(), addc.get(), g, relu, st);
    } else if constexpr (std::is_same<T, float>::value || std::is_same<T, double>::value) {
      if (training && vec && !col && total > 0 && (dx.get() || dadd.get()))
        fused_done = bn_bwd_fused_fp_launch<T>(gc.get(), xc.get(), mean_t, invstd_t, weight, bias, dw.get(), db.get(), dx.get(), dadd.get(), addc.get(), g, relu, st);
    }
    if (!fused_done) {
    const int64_t blocks = col ? (g.C + 255) / 256 : g.C;
    const int nsplit = pick_split(blocks, g.N);
    int64_t ps[1] = {(int64_t)nsplit * g.C * 2};
    int64_t ss[1] = {g.C * 2};
    Hold partial(new_tensor(ps, 1, acc_dtype<A>(), x->device())), sums(new_tensor(ss, 1, acc_dtype<A>(), x->device()));
    {
      KernelTimer kt1("bn_bwd_reduce", 0, 2.0 * (double)total * sizeof(T), st);
      if (col) hipLaunchKernelGGL((bn_bwd_reduce_col_kernel<T>), dim3((unsigned)blocks, nsplit), dim3(256), 0, st, gc->ptr<T>(), xc->ptr<T>(), mean_t->ptr<T>(), partial->ptr<A>(), g.N, g.C, g.HW, nsplit);
      else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3((unsigned)g.C, nsplit), dim3(256), 0, st, gc->ptr<T>(), xc->ptr<T>(), mean_t->ptr<T>(), partial->ptr<A>(), g.N, g.C, g.HW, nsplit, vec,
                              relu, invstd_t->ptr<T>(), weight ? weight->ptr<T>() : (const T*)nullptr, bias ? bias->ptr<T>() : (const T*)nullptr,
                              addc.get() ? addc->ptr<T>() : (const T*)nullptr);
    }
    LAMP_LAUNCH_CHECK();
    T* dwp = dw.get() ? dw->ptr<T>() : (T*)nullptr;
    T* dbp = db.get() ? db->ptr<T>() : (T*)nullptr;
    const T* wp = weight ? weight->ptr<T>() : (const T*)nullptr;
    const T* bp = bias ? bias->ptr<T>() : (const T*)nullptr;
    if (!col) {
      // finalize folded into the channel-aligned dx kernel (one slice and no element loop when only dweight / dbias are wanted)
      const int do_apply = ((dx.get() || dadd.get()) && total > 0) ? 1 : 0;
      const int nblk = do_apply ? nsplit : 1;
      KernelTimer kt2("bn_bwd_apply", 0, 3.0 * (double)total * sizeof(T), st);
      hipLaunchKernelGGL((bn_bwd_apply2_kernel<T>), dim3((unsigned)g.C, nblk), dim3(256), 0, st, gc->ptr<T>(), xc->ptr<T>(), mean_t->ptr<T>(), invstd_t->ptr<T>(),
                         wp, bp, partial->ptr<A>(), nsplit, dwp, dbp, dx.get() ? dx->ptr<T>() : (T*)nullptr, g.N, g.C, g.HW, nblk,
                         1.0 / (double)(g.N * g.HW), training, vec, relu, do_apply, addc.get() ? addc->ptr<T>() : (const T*)nullptr,
                         dadd.get() ? dadd->ptr<T>() : (T*)nullptr);
      LAMP_LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL((bn_bwd_finalize_kernel<T>), dim3((unsigned)((g.C * 64 + 255) / 256)), dim3(256), 0, st, partial->ptr<A>(), sums->ptr<A>(), g.C,
                         nsplit, invstd_t->ptr<T>(), dwp, dbp);
      LAMP_LAUNCH_CHECK();
      if (dx.get() && total > 0) {
        KernelTimer kt2("bn_bwd_apply", 0, 3.0 * (double)total * sizeof(T), st);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, gc->ptr<T>(), xc->ptr<T>(), mean_t->ptr<T>(),
                           invstd_t->ptr<T>(), wp, sums->ptr<A>(), dx->ptr<T>(), total, g.C, g.HW, 1.0 / (double)(g.N * g.HW), training, vec, relu, bp);
        LAMP_LAUNCH_CHECK();
      }
    }
    }
  });
  out3[0] = dx.take(); out3[1] = dw.take(); out3[2] = db.take();
  if (daddend_out) *daddend_out = dadd.take();
  LAMP_API_END
}
int lamp_native_batch_norm_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* weight,
                                    const lamp_tensor* running_mean, const lamp_tensor* running_var, const lamp_tensor* save_mean,
                                    const lamp_tensor* save_invstd, int training, double eps, const uint8_t mask[3]) {
  return bn_backward_impl(out3, grad_out, x, weight, nullptr, running_mean, running_var, save_mean, save_invstd, training, eps, mask, 0);
}
int lamp_native_batch_norm_relu_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* weight,
                                         const lamp_tensor* bias, const lamp_tensor* running_mean, const lamp_tensor* running_var,
                                         const lamp_tensor* save_mean, const lamp_tensor* save_invstd, int training, double eps,
                                         const uint8_t mask[3]) {
  return bn_backward_impl(out3, grad_out, x, weight, bias, running_mean, running_var, save_mean, save_invstd, training, eps, mask, 1);
}

int lamp_native_batch_norm_add_relu(lamp_tensor* out3[3], const lamp_tensor* x, const lamp_tensor* addend, const lamp_tensor* weight,
                                    const lamp_tensor* bias, lamp_tensor* running_mean, lamp_tensor* running_var, int training, double momentum,
                                    double eps) {
  if (!addend) { ::lamp::set_last_error("lamp_native_batch_norm_add_relu: addend is null"); return 1; }
  return bn_forward_impl(out3, x, weight, bias, running_mean, running_var, training, momentum, eps, 1, addend);
}
int lamp_native_batch_norm_add_relu_backward(lamp_tensor* out4[4], const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* addend,
                                             const lamp_tensor* weight, const lamp_tensor* bias, const lamp_tensor* running_mean,
                                             const lamp_tensor* running_var, const lamp_tensor* save_mean, const lamp_tensor* save_invstd,
                                             int training, double eps, const uint8_t mask[4]) {
  if (!addend) { ::lamp::set_last_error("lamp_native_batch_norm_add_relu_backward: addend is null"); return 1; }
  out4[3] = nullptr;
  return bn_backward_impl(out4, grad_out, x, weight, bias, running_mean, running_var, save_mean, save_invstd, training, eps, mask, 1, addend,
                          mask[3] ? &out4[3] : nullptr);
}

// relu(bn(x) + bn2(x2)) - the tail of lamp's residual block when both branches end in a batch norm (cnn.scala:62-78 under Fun(relu),
// :36-46) - as ONE kernel per direction.  Values: those of the chain bn2 -> (bn + add + relu) with every intermediate rounded as the
// chain rounds it (forward: bitwise); the forward never writes bn2's output, the backward never writes the masked gradient.
int lamp_native_batch_norm2_add_relu(lamp_tensor* out5[5], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias,
                                     lamp_tensor* running_mean, lamp_tensor* running_var, const lamp_tensor* x2, const lamp_tensor* weight2,
                                     const lamp_tensor* bias2, lamp_tensor* running_mean2, lamp_tensor* running_var2, double momentum,
                                     double momentum2, double eps, double eps2) {
  for (int i = 0; i < 5; i++) out5[i] = nullptr;
  if (!x2) { ::lamp::set_last_error("lamp_native_batch_norm2_add_relu: the second input is null"); return 1; }
  BnSecondArgs sec{x2, weight2, bias2, running_mean2, running_var2, momentum2, eps2};
  lamp_tensor* o3[3] = {nullptr, nullptr, nullptr};
  const int rc = bn_forward_impl(o3, x, weight, bias, running_mean, running_var, 1, momentum, eps, 1, nullptr, &sec);
  if (rc != 0) return rc;
  out5[0] = o3[0]; out5[1] = o3[1]; out5[2] = o3[2]; out5[3] = sec.save_mean; out5[4] = sec.save_invstd;
  return 0;
}
// ... -> AvgPool2D over the whole map -> Flatten -> LogSoftMax: the LAST block of Cnn.resnet with the network's tail (cnn.scala:129-136).  The
// block's output has one reader, the pool - it is never written: bn_apply2_kernel leaves its plane means [N, C] (bitwise the pool's), the
// LogSoftMax runs on those.  out5[0] = the log-probabilities [N, C]; the other four as lamp_native_batch_norm2_add_relu.  Where the pooled
// form does not apply (planes that are no power-of-two number of 16-byte packets, unaligned views) the two calls run one after the other.
int lamp_native_batch_norm2_add_relu_pool_log_softmax(lamp_tensor* out5[5], const lamp_tensor* x, const lamp_tensor* weight, const lamp_tensor* bias,
                                                      lamp_tensor* running_mean, lamp_tensor* running_var, const lamp_tensor* x2, const lamp_tensor* weight2,
                                                      const lamp_tensor* bias2, lamp_tensor* running_mean2, lamp_tensor* running_var2, double momentum,
                                                      double momentum2, double eps, double eps2) {
  for (int i = 0; i < 5; i++) out5[i] = nullptr;
  if (!x2) { ::lamp::set_last_error("lamp_native_batch_norm2_add_relu_pool_log_softmax: the second input is null"); return 1; }
  BnSecondArgs sec{x2, weight2, bias2, running_mean2, running_var2, momentum2, eps2};
  lamp_tensor* o3[3] = {nullptr, nullptr, nullptr};
  bool pooled = true;
  const int rc = bn_forward_impl(o3, x, weight, bias, running_mean, running_var, 1, momentum, eps, 1, nullptr, &sec, false, &pooled);
  if (rc != 0) return rc;
  lamp_tensor* o = nullptr;
  const int rc2 = pooled ? lamp_log_softmax(&o, o3[0], 1) : lamp_global_avg_pool_log_softmax(&o, o3[0]);
  (void)lamp_tensor_release(o3[0]);
  if (rc2 != 0) {
    (void)lamp_tensor_release(o3[1]); (void)lamp_tensor_release(o3[2]); (void)lamp_tensor_release(sec.save_mean); (void)lamp_tensor_release(sec.save_invstd);
    return rc2;
  }
  out5[0] = o; out5[1] = o3[1]; out5[2] = o3[2]; out5[3] = sec.save_mean; out5[4] = sec.save_invstd;
  return 0;
}
int lamp_native_batch_norm2_add_relu_backward(lamp_tensor* out6[6], const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* weight,
                                              const lamp_tensor* bias, const lamp_tensor* save_mean, const lamp_tensor* save_invstd,
                                              const lamp_tensor* x2, const lamp_tensor* weight2, const lamp_tensor* bias2,
                                              const lamp_tensor* save_mean2, const lamp_tensor* save_invstd2, double eps, double eps2,
                                              const uint8_t mask[6]) {
  LAMP_API_BEGIN
  for (int i = 0; i < 6; i++) out6[i] = nullptr;
  check_device_tensor(x, "input"); check_device_tensor(grad_out, "grad_out"); check_device_tensor(x2, "second input");
  LAMP_CHECK(grad_out->shape() == x->shape() && grad_out->dtype == x->dtype, "grad_out " << grad_out->describe() << " does not match input " << x->describe());
  LAMP_CHECK(x2->shape() == x->shape() && x2->dtype == x->dtype, "second input " << x2->describe() << " does not match input " << x->describe());
  BnGeom g = bn_geom(x);
  LAMP_CHECK(g.HW >= 64, "the two-batch-norm residual tail exists for maps of at least 64 elements, got " << x->describe());
  check_cvec(weight, g.C, x->dtype, "weight"); check_cvec(bias, g.C, x->dtype, "bias");
  check_cvec(weight2, g.C, x->dtype, "second weight"); check_cvec(bias2, g.C, x->dtype, "second bias");
  LAMP_CHECK(save_mean && save_invstd && save_mean2 && save_invstd2, "the backward needs the saved statistics of both batch norms");
  check_cvec(save_mean, g.C, x->dtype, "save_mean"); check_cvec(save_invstd, g.C, x->dtype, "save_invstd");
  check_cvec(save_mean2, g.C, x->dtype, "second save_mean"); check_cvec(save_invstd2, g.C, x->dtype, "second save_invstd");
  // (a plane-broadcast gradient - the loss tail's - is read as its [N, C] values by the one-pass kernel; materialised only if that kernel does not run)
  const BnPlaneGrad gplanes = plane_broadcast_base(grad_out);
  Hold xc(contiguous(x)), gc(gplanes.p ? nullptr : contiguous(grad_out)), x2c(contiguous(x2));
  hipStream_t st = current_stream(x->device());
  const int64_t total = x->numel();
  if (x->dtype == kBF16 && total > 0 && (mask[0] || mask[3])) {
    int64_t cs[1] = {g.C};
    Hold dx(mask[0] ? new_like(xc.get()) : nullptr), dx2(mask[3] ? new_like(xc.get()) : nullptr);
    Hold dw(mask[1] ? new_tensor(cs, 1, x->dtype, x->device()) : nullptr), db(mask[2] ? new_tensor(cs, 1, x->dtype, x->device()) : nullptr);
    Hold dw2(mask[4] ? new_tensor(cs, 1, x->dtype, x->device()) : nullptr), db2(mask[5] ? new_tensor(cs, 1, x->dtype, x->device()) : nullptr);
    const bool aligned = g.HW % 8 == 0 && (((uintptr_t)xc->data() | (uintptr_t)(gc.get() ? gc->data() : nullptr) | (uintptr_t)x2c->data() |
                                            (uintptr_t)(dx.get() ? dx->data() : nullptr) | (uintptr_t)(dx2.get() ? dx2->data() : nullptr)) & 15) == 0;
    BnDualHost dh{save_mean2, save_invstd2, weight2, bias2, dw2.get(), db2.get()};
    bool done = aligned && bn_bwd_fused_launch(gc.get(), xc.get(), save_mean, save_invstd, weight, bias, dw.get(), db.get(), dx.get(), dx2.get(), x2c.get(), g, 1, st, &dh, gplanes);
    if (!done && !gc.get()) {                               // the plane form did not take it: the same kernel on the materialised gradient
      gc = Hold(contiguous(grad_out));
      done = (((uintptr_t)gc->data()) & 15) == 0 && g.HW % 8 == 0 &&
             bn_bwd_fused_launch(gc.get(), xc.get(), save_mean, save_invstd, weight, bias, dw.get(), db.get(), dx.get(), dx2.get(), x2c.get(), g, 1, st, &dh);
    }
    if (done) {
      out6[0] = dx.take(); out6[1] = dw.take(); out6[2] = db.take(); out6[3] = dx2.take(); out6[4] = dw2.take(); out6[5] = db2.take();
      return 0;
    }
  }
  if (!gc.get()) gc = Hold(contiguous(grad_out));
  // every other case (f32 / f64, small batches, a shared device): the chain itself - the second batch norm's output from its saved
  // statistics, the one-addend backward, then the second batch norm's backward on the masked gradient
  Hold l(new_like(x2c.get()));
  if (total > 0) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, {
      const int vec = (g.HW % (16 / sizeof(T)) == 0) && (((uintptr_t)x2c->raw() | (uintptr_t)l->raw()) & 15) == 0;
      KernelTimer kt("bn_fwd_apply", 0, 2.0 * (double)total * sizeof(T), st);
      hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, static_cast<const Tensor*>(x2c.get())->ptr<T>(), l->ptr<T>(),
                         save_mean2->ptr<T>(), save_invstd2->ptr<T>(), weight2 ? weight2->ptr<T>() : (const T*)nullptr,
                         bias2 ? bias2->ptr<T>() : (const T*)nullptr, total, g.C, g.HW, vec, 0);
    });
    LAMP_LAUNCH_CHECK();
  }
  const bool second_wanted = mask[3] || mask[4] || mask[5];
  lamp_tensor* r4[4] = {nullptr, nullptr, nullptr, nullptr};
  const uint8_t m4[4] = {mask[0], mask[1], mask[2], (uint8_t)second_wanted};
  if (lamp_native_batch_norm_add_relu_backward(r4, gc.get(), xc.get(), l.get(), weight, bias, nullptr, nullptr, save_mean, save_invstd, 1, eps, m4) != 0)
    throw Error(lamp_last_error());
  Hold h0(r4[0]), h1(r4[1]), h2(r4[2]), h3(r4[3]);
  lamp_tensor* r3[3] = {nullptr, nullptr, nullptr};
  if (second_wanted) {
    const uint8_t m3[3] = {mask[3], mask[4], mask[5]};
    if (lamp_native_batch_norm_backward(r3, h3.get(), x2c.get(), weight2, nullptr, nullptr, save_mean2, save_invstd2, 1, eps2, m3) != 0)
      throw Error(lamp_last_error());
  }
  out6[0] = h0.take(); out6[1] = h1.take(); out6[2] = h2.take(); out6[3] = r3[0]; out6[4] = r3[1]; out6[5] = r3[2];
  LAMP_API_END
}

int lamp_native_layer_norm(lamp_tensor* out3[3], const lamp_tensor* x, const int64_t* normalized_shape, int nnorm,
                           const lamp_tensor* weight, const lamp_tensor* bias, double eps) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input");
  LAMP_CHECK(nnorm >= 1 && nnorm <= x->ndim, "bad normalized_shape");
  int64_t D = 1;
  for (int i = 0; i < nnorm; i++) {
    LAMP_CHECK(normalized_shape[i] == x->sizes[x->ndim - nnorm + i], "normalized_shape does not match the trailing input dims");
    D *= normalized_shape[i];
  }
  const int64_t M = D ? x->numel() / D : 0;
  if (weight) { check_device_tensor(weight, "weight"); LAMP_CHECK(weight->numel() == D && weight->dtype == x->dtype && weight->is_contiguous(), "bad layer norm weight"); }
  if (bias) { check_device_tensor(bias, "bias"); LAMP_CHECK(bias->numel() == D && bias->dtype == x->dtype && bias->is_contiguous(), "bad layer norm bias"); }
  Hold xc(contiguous(x));
  Hold y(new_like(xc.get()));
  std::vector<int64_t> sshape;
  for (int i = 0; i < x->ndim; i++) sshape.push_back(i < x->ndim - nnorm ? x->sizes[i] : 1);
  Hold mean(new_tensor(sshape, x->dtype, x->device())), rstd(new_tensor(sshape, x->dtype, x->device()));
  if (M > 0) {
    const int64_t blocks = (M * 64 + 255) / 256;
    hipStream_t st = current_stream(x->device());
    LAMP_DISPATCH_FLOAT(x->dtype, T, {
      constexpr int W = 16 / sizeof(T);
      const T* wp = weight ? weight->ptr<T>() : (const T*)nullptr;
      const T* bp = bias ? bias->ptr<T>() : (const T*)nullptr;
      const int64_t npk = D / W;
      const bool vec = D % W == 0 && npk <= 64 * 8 && (((uintptr_t)xc->raw() | (uintptr_t)y->raw() | (uintptr_t)wp | (uintptr_t)bp) & 15) == 0;
      const T* xp = static_cast<const Tensor*>(xc.get())->ptr<T>();
      if (!vec) hipLaunchKernelGGL((ln_fwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), mean->ptr<T>(), rstd->ptr<T>(), wp, bp, M, D, eps);
      else if (npk <= 64) hipLaunchKernelGGL((ln_fwd_vec_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), mean->ptr<T>(), rstd->ptr<T>(), wp, bp, M, D, eps);
      else if (npk <= 128) hipLaunchKernelGGL((ln_fwd_vec_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), mean->ptr<T>(), rstd->ptr<T>(), wp, bp, M, D, eps);
      else if (npk <= 256) hipLaunchKernelGGL((ln_fwd_vec_kernel<T, 4>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), mean->ptr<T>(), rstd->ptr<T>(), wp, bp, M, D, eps);
      else hipLaunchKernelGGL((ln_fwd_vec_kernel<T, 8>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), mean->ptr<T>(), rstd->ptr<T>(), wp, bp, M, D, eps);
    });
    LAMP_LAUNCH_CHECK();
  }
  out3[0] = y.take(); out3[1] = mean.take(); out3[2] = rstd.take();
  LAMP_API_END
}

int lamp_native_layer_norm_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* x, const int64_t* normalized_shape,
                                    int nnorm, const lamp_tensor* mean, const lamp_tensor* rstd, const lamp_tensor* weight,
                                    const lamp_tensor* bias, const uint8_t mask[3]) {
  LAMP_API_BEGIN
  (void)bias;
  check_device_tensor(x, "input"); check_device_tensor(grad_out, "grad_out"); check_device_tensor(mean, "mean"); check_device_tensor(rstd, "rstd");
  LAMP_CHECK(grad_out->shape() == x->shape() && grad_out->dtype == x->dtype, "grad_out does not match input");
  int64_t D = 1;
  for (int i = 0; i < nnorm; i++) D *= normalized_shape[i];
  const int64_t M = D ? x->numel() / D : 0;
  LAMP_CHECK(mean->numel() == M && rstd->numel() == M, "mean/rstd have the wrong size");
  Hold xc(contiguous(x)), gc(contiguous(grad_out)), mc(contiguous(mean)), rc(contiguous(rstd));
  hipStream_t st = current_stream(x->device());
  Hold dx(mask[0] ? new_like(xc.get()) : nullptr);
  std::vector<int64_t> wshape(normalized_shape, normalized_shape + nnorm);
  Hold dw(mask[1] ? new_tensor(wshape, x->dtype, x->device()) : nullptr);
  Hold db(mask[2] ? new_tensor(wshape, x->dtype, x->device()) : nullptr);
  LAMP_DISPATCH_FLOAT(x->dtype, T, {
    using A = acc_t<T>;
    constexpr int W = 16 / sizeof(T);
    const T* xp = static_cast<const Tensor*>(xc.get())->ptr<T>();
    const T* gp = static_cast<const Tensor*>(gc.get())->ptr<T>();
    const T* mp = static_cast<const Tensor*>(mc.get())->ptr<T>();
    const T* rp = static_cast<const Tensor*>(rc.get())->ptr<T>();
    const T* wp = weight ? weight->ptr<T>() : (const T*)nullptr;
    const int64_t npk = D / W;
    const bool al = D % W == 0 && (((uintptr_t)xp | (uintptr_t)gp | (uintptr_t)wp) & 15) == 0;
    if (dx.get() && M > 0) {
      const int64_t blocks = (M * 64 + 255) / 256;
      const bool vec = al && npk <= 64 * 8 && ((uintptr_t)dx->raw() & 15) == 0;
      if (!vec) hipLaunchKernelGGL((ln_bwd_dx_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, gp, xp, mp, rp, wp, dx->ptr<T>(), M, D);
      else if (npk <= 64) hipLaunchKernelGGL((ln_bwd_dx_vec_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, gp, xp, mp, rp, wp, dx->ptr<T>(), M, D);
      else if (npk <= 128) hipLaunchKernelGGL((ln_bwd_dx_vec_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, st, gp, xp, mp, rp, wp, dx->ptr<T>(), M, D);
      else if (npk <= 256) hipLaunchKernelGGL((ln_bwd_dx_vec_kernel<T, 4>), dim3((unsigned)blocks), dim3(256), 0, st, gp, xp, mp, rp, wp, dx->ptr<T>(), M, D);
      else hipLaunchKernelGGL((ln_bwd_dx_vec_kernel<T, 8>), dim3((unsigned)blocks), dim3(256), 0, st, gp, xp, mp, rp, wp, dx->ptr<T>(), M, D);
      LAMP_LAUNCH_CHECK();
    }
    if (dw.get() || db.get()) {
      const int64_t blocks = al ? (npk + 255) / 256 : (D + 255) / 256;
      const int nsplit = pick_split(blocks, std::max<int64_t>(M, 1));
      int64_t ps[1] = {(int64_t)nsplit * D * 2};
      Hold partial(new_tensor(ps, 1, acc_dtype<A>(), x->device()));
      if (al) hipLaunchKernelGGL((ln_bwd_dwdb_vec_kernel<T>), dim3((unsigned)blocks, nsplit), dim3(256), 0, st, gp, xp, mp, rp, partial->ptr<A>(), M, D, nsplit);
      else hipLaunchKernelGGL((ln_bwd_dwdb_kernel<T>), dim3((unsigned)blocks, nsplit), dim3(256), 0, st, gp, xp, mp, rp, partial->ptr<A>(), M, D, nsplit);
      LAMP_LAUNCH_CHECK();
      {
      const int64_t blocks = (D + 255) / 256;
      hipLaunchKernelGGL((ln_bwd_finalize_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, partial->ptr<A>(),
                         dw.get() ? dw->ptr<T>() : (T*)nullptr, db.get() ? db->ptr<T>() : (T*)nullptr, D, nsplit);
      LAMP_LAUNCH_CHECK();
      }
    }
  });
  out3[0] = dx.take(); out3[1] = dw.take(); out3[2] = db.take();
  LAMP_API_END
}

}  // extern "C"
