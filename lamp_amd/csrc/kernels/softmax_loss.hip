// log-softmax / softmax / NLL / MSE kernels.
//
// Replaces ATen.log_softmax, _log_softmax_backward_data, nll_loss_forward, nll_loss_backward,
// mse_loss(+_backward) as lamp calls them (reference:
// lamp-core/src/main/scala/lamp/autograd/ops.scala:955-975 (LogSoftMax), 1176-1206 (MseLoss),
// 1249-1304 (NllLoss); lamp-core/.../nn/LossFunctions.scala:39-55).
// One 64-lane wavefront per softmax row (max and sum by lane shuffles), f32 math for bf16/f32.
// int64 targets are consumed as-is (bit-exact class indices, ignore_index honoured).
#include "device_utils.h"
#include "../core/strided.h"

namespace lamp {

template <class A> __device__ __forceinline__ A t_exp(A x) { return (A)exp((double)x); }
template <> __device__ __forceinline__ float t_exp(float x) { return expf(x); }
template <class A> __device__ __forceinline__ A t_log1p(A x) { return (A)log1p((double)x); }
template <> __device__ __forceinline__ float t_log1p(float x) { return log1pf(x); }
template <class A> __device__ __forceinline__ A t_log(A x) { return (A)log((double)x); }
template <> __device__ __forceinline__ float t_log(float x) { return logf(x); }

// One wave, one row of D elements with stride `inner`: the arithmetic of (log-)softmax and of its backward, shared by the row
// kernels below and by the kernels that fuse a global average pool in front (gap_lsm_*): same order, same values.
template <class T, bool LOG>
__device__ __forceinline__ void softmax_row(const T* xp, T* yp, int64_t D, int64_t inner, int64_t out_inner, int lane) {
  using A = acc_t<T>;
  A m = -INFINITY;
  for (int64_t d = lane; d < D; d += 64) { A v = load_as<A>(xp[d * inner]); m = v > m ? v : m; }
  m = wave_max(m);
  A s = 0;
  for (int64_t d = lane; d < D; d += 64) s += t_exp<A>(load_as<A>(xp[d * inner]) - m);
  s = wave_sum(s);
  if (LOG) {
    const A ls = t_log<A>(s);
    for (int64_t d = lane; d < D; d += 64) yp[d * out_inner] = store_as<T>((A)(load_as<A>(xp[d * inner]) - m - ls));
  } else {
    const A inv = A(1) / s;
    for (int64_t d = lane; d < D; d += 64) yp[d * out_inner] = store_as<T>((A)(t_exp<A>(load_as<A>(xp[d * inner]) - m) * inv));
  }
}
// grad_in = grad - exp(output) * sum(grad)
template <class T>
__device__ __forceinline__ void log_softmax_bwd_row(const T* g, const T* out, T* gi, int64_t D, int64_t inner, int64_t gi_inner, int lane) {
  using A = acc_t<T>;
  A s = 0;
  for (int64_t d = lane; d < D; d += 64) s += load_as<A>(g[d * inner]);
  s = wave_sum(s);
  for (int64_t d = lane; d < D; d += 64) gi[d * gi_inner] = store_as<T>((A)(load_as<A>(g[d * inner]) - t_exp<A>(load_as<A>(out[d * inner])) * s));
}

// x viewed as [outer, D, inner]; one wave per (outer, inner) pair. LOG: log-softmax else softmax.
template <class T, bool LOG>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t outer, int64_t D,
                                                          int64_t inner) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nrows = outer * inner;
  if (wave >= nrows) return;
  const int64_t o = wave / inner, i = wave - o * inner;
  softmax_row<T, LOG>(x + o * D * inner + i, y + o * D * inner + i, D, inner, inner, lane);
}

// ---- global average pool + flatten + log-softmax in one kernel (the tail of Cnn.resnet, cnn.scala:129-136) ---------------------------
// Workgroup = image.  The planes are pooled exactly as avg_pool_global_fwd_vec_kernel pools them (lpp lanes per plane, a sequential sum
// of the packet's elements, a butterfly over the plane's lanes, one rounding to T), the rounded means go to LDS and the first wave runs
// softmax_row over them: the values of the three-operator chain, two launches and one [N, C] round trip less.  The backward does the
// same in reverse: log_softmax_bwd_row into LDS, divided by the plane size and rounded as avg_pool_global_bwd_vec_kernel does, then
// broadcast over the planes in 16-byte packets.
template <class T>
__global__ __launch_bounds__(256) void gap_lsm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int C, int hw, int lpp) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* pl = reinterpret_cast<T*>(smem);
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x, ppp = 256 / lpp, sub = tid % lpp;
  for (int c0 = 0; c0 < C; c0 += ppp) {
    const int c = c0 + tid / lpp;
    A s = 0;
    if (c < C) {
      const Vec<T, W> pk = *reinterpret_cast<const Vec<T, W>*>(x + ((n * C + c) * lpp + sub) * W);
#pragma unroll
      for (int k = 0; k < W; k++) s += load_as<A>(pk.v[k]);
    }
    for (int off = lpp >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (c < C && sub == 0) pl[c] = store_as<T>((A)(s / (A)hw));
  }
  __syncthreads();
  if (tid < 64) softmax_row<T, true>(pl, y + n * C, C, 1, 1, tid);
}
template <class T>
__global__ __launch_bounds__(256) void gap_lsm_bwd_kernel(const T* __restrict__ g, const T* __restrict__ out, T* __restrict__ dx, int C, int hw, int lpp) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* gl = reinterpret_cast<T*>(smem);
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < 64) log_softmax_bwd_row<T>(g + n * C, out + n * C, gl, C, 1, 1, tid);
  __syncthreads();
  const int packets = C * lpp;
  for (int p = tid; p < packets; p += 256) {
    const T v = store_as<T>((A)(load_as<A>(gl[p / lpp]) / (A)hw));
    Vec<T, W> pk;
#pragma unroll
    for (int k = 0; k < W; k++) pk.v[k] = v;
    *reinterpret_cast<Vec<T, W>*>(dx + (n * packets + p) * W) = pk;
  }
}

// ... with the gradient row itself built in LDS from the loss: g[n][c] = nll_loss_backward(grad_loss, target, weight)[n][c] as nll_bwd_kernel
// rounds it (lamp_global_avg_pool_log_softmax_nll_backward: the NllLoss and the pooled LogSoftMax of Cnn.resnet's tail in one backward launch)
template <class T>
__global__ __launch_bounds__(256) void gap_lsm_nll_bwd_kernel(const T* __restrict__ grad, const int64_t* __restrict__ target, const T* __restrict__ w,
                                                              const T* __restrict__ total_weight, int64_t reduction, int64_t ignore,
                                                              const T* __restrict__ out, T* __restrict__ dx, int C, int hw, int lpp) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* gl = reinterpret_cast<T*>(smem);                       // [C]: the pooled gradient
  T* grow = gl + ((C + 7) & ~7);                            // [C]: the loss's gradient row
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x;
  {
    const int64_t t = target[n];
    A v = 0;
    if (t != ignore && t >= 0 && t < C) {
      A g = load_as<A>(reduction == 0 ? grad[n] : grad[0]);
      if (reduction == 1) g = g / load_as<A>(*total_weight);
      v = -(w ? load_as<A>(w[t]) : A(1)) * g;
    }
    const T vt = store_as<T>(v), zero = store_as<T>(A(0));
    for (int c = tid; c < C; c += 256) grow[c] = c == t ? vt : zero;
  }
  __syncthreads();
  if (tid < 64) log_softmax_bwd_row<T>(grow, out + n * C, gl, C, 1, 1, tid);
  __syncthreads();
  const int packets = C * lpp;
  for (int p = tid; p < packets; p += 256) {
    const T v = store_as<T>((A)(load_as<A>(gl[p / lpp]) / (A)hw));
    Vec<T, W> pk;
#pragma unroll
    for (int k = 0; k < W; k++) pk.v[k] = v;
    *reinterpret_cast<Vec<T, W>*>(dx + (n * packets + p) * W) = pk;
  }
}

// contiguous rows (inner == 1), D a multiple of the packet width: 16-byte loads; rows of up to 64 * MAXP packets stay in registers
// (one read of x), longer ones (LM vocabularies) are streamed three times with vector loads (MAXP == 0)
template <class T, bool LOG, int MAXP>
__global__ __launch_bounds__(256) void softmax_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int64_t D) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  constexpr int NP = MAXP > 0 ? MAXP : 1;
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t npk = D / W;
  const Vec<T, W>* xp = reinterpret_cast<const Vec<T, W>*>(x + row * D);
  Vec<T, W>* yp = reinterpret_cast<Vec<T, W>*>(y + row * D);
  Vec<T, W> pk[NP];
  A m = -INFINITY;
  if (MAXP > 0) {
#pragma unroll
    for (int i = 0; i < NP; i++)
      if (lane + 64 * i < npk) {
        pk[i] = xp[lane + 64 * i];
#pragma unroll
        for (int k = 0; k < W; k++) { const A v = load_as<A>(pk[i].v[k]); m = v > m ? v : m; }
      }
  } else {
    for (int64_t p = lane; p < npk; p += 64) {
      const Vec<T, W> t = xp[p];
#pragma unroll
      for (int k = 0; k < W; k++) { const A v = load_as<A>(t.v[k]); m = v > m ? v : m; }
    }
  }
  m = wave_max(m);
  A s = 0;
  if (MAXP > 0) {
#pragma unroll
    for (int i = 0; i < NP; i++)
      if (lane + 64 * i < npk) {
#pragma unroll
        for (int k = 0; k < W; k++) s += t_exp<A>(load_as<A>(pk[i].v[k]) - m);
      }
  } else {
    for (int64_t p = lane; p < npk; p += 64) {
      const Vec<T, W> t = xp[p];
#pragma unroll
      for (int k = 0; k < W; k++) s += t_exp<A>(load_as<A>(t.v[k]) - m);
    }
  }
  s = wave_sum(s);
  const A ls = LOG ? t_log<A>(s) : A(0), inv = LOG ? A(0) : A(1) / s;
  auto emit = [&](const Vec<T, W>& t, int64_t p) {
    Vec<T, W> o;
#pragma unroll
    for (int k = 0; k < W; k++) {
      const A v = load_as<A>(t.v[k]);
      o.v[k] = LOG ? store_as<T>((A)(v - m - ls)) : store_as<T>((A)(t_exp<A>(v - m) * inv));
    }
    yp[p] = o;
  };
  if (MAXP > 0) {
#pragma unroll
    for (int i = 0; i < NP; i++) if (lane + 64 * i < npk) emit(pk[i], lane + 64 * i);
  } else {
    for (int64_t p = lane; p < npk; p += 64) emit(xp[p], p);
  }
}
template <class T>
__global__ __launch_bounds__(256) void log_softmax_bwd_vec_kernel(const T* __restrict__ g, const T* __restrict__ out, T* __restrict__ gi, int64_t rows,
                                                                  int64_t D) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t npk = D / W;
  const Vec<T, W>* gp = reinterpret_cast<const Vec<T, W>*>(g + row * D);
  const Vec<T, W>* op = reinterpret_cast<const Vec<T, W>*>(out + row * D);
  Vec<T, W>* ip = reinterpret_cast<Vec<T, W>*>(gi + row * D);
  A s = 0;
  for (int64_t p = lane; p < npk; p += 64) {
    const Vec<T, W> t = gp[p];
#pragma unroll
    for (int k = 0; k < W; k++) s += load_as<A>(t.v[k]);
  }
  s = wave_sum(s);
  for (int64_t p = lane; p < npk; p += 64) {
    const Vec<T, W> t = gp[p], o = op[p];
    Vec<T, W> r;
#pragma unroll
    for (int k = 0; k < W; k++) r.v[k] = store_as<T>((A)(load_as<A>(t.v[k]) - t_exp<A>(load_as<A>(o.v[k])) * s));
    ip[p] = r;
  }
}

// grad_in = grad - exp(output) * sum(grad)
template <class T>
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const T* __restrict__ g, const T* __restrict__ out, T* __restrict__ gi,
                                                              int64_t outer, int64_t D, int64_t inner) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (wave >= outer * inner) return;
  const int64_t o = wave / inner, i = wave - o * inner;
  const int64_t base = o * D * inner + i;
  log_softmax_bwd_row<T>(g + base, out + base, gi + base, D, inner, inner, lane);
}

// ---- NLL ---------------------------------------------------------------------------------------
// single block: sum_i -w[t_i] x[i, t_i] and sum_i w[t_i] over non-ignored rows
template <class T>
__global__ __launch_bounds__(1024) void nll_fwd_reduce_kernel(const T* __restrict__ x, const int64_t* __restrict__ target,
                                                              const T* __restrict__ w, T* __restrict__ out, T* __restrict__ total_weight,
                                                              int64_t N, int64_t C, int64_t reduction, int64_t ignore, int* __restrict__ assert_word,
                                                              T* acc, double acc_scale) {
  using A = acc_t<T>;
  __shared__ A sm[16];
  A loss = 0, tw = 0;
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
    const int64_t t = target[i];
    if (t == ignore) continue;
    if (t < 0 || t >= C) { *(volatile int*)assert_word = kAssertNllTarget; continue; }  // never read out of bounds; the host raises at its next wait
    const A wt = w ? load_as<A>(w[t]) : A(1);
    loss -= wt * load_as<A>(x[i * C + t]);
    tw += wt;
  }
  loss = block_sum(loss, sm);
  tw = block_sum(tw, sm);
  if (threadIdx.x == 0) {
    *total_weight = store_as<T>(tw);
    const T o = reduction == 1 ? store_as<T>((A)(loss / tw)) : store_as<T>(loss);
    *out = o;
    // the epoch-loss bookkeeping `acc += n * loss` (IOLoops.scala:714) in the same launch: the arithmetic of the add kernel on the ROUNDED loss
    if (acc) *acc = store_as<T>((A)(load_as<A>(*acc) + (A)acc_scale * load_as<A>(o)));
  }
}
// nll_fwd_reduce_kernel and, from the same launch, the INPUT GRADIENT OF THE POOLED LogSoftMax IN FRONT OF IT for an incoming loss gradient of one
// (lamp_nll_loss_forward_pooled_gradient_; Cnn.resnet's tail, cnn.scala:129-136 + SupervisedModel.scala:190-211: the loss is the root of backprop,
// whose derivative autograd.scala:264-282 fills with ones).  Workgroup 0 is nll_fwd_reduce_kernel, statement for statement.  Every other
// workgroup takes 16 rows, a wave each: it sums the total weight itself (the same 1024 partial sums in the same order: the same bits as
// workgroup 0's - it needs nothing from another workgroup), builds the loss's gradient row in LDS as gap_lsm_nll_bwd_kernel does, runs
// log_softmax_bwd_row over it and divides by the plane size with that kernel's roundings.  What leaves is ONE value per (sample, class): every
// element of plane (n, c) of the [N, C, H, W] gradient gap_lsm_nll_bwd_kernel would write equals that value - the consumer reads it through an
// expanded view and the [N, C, H, W] tensor is neither written nor read.  The values are stored CLASS-MAJOR (plane_grad_t[c][n]: the batch norm
// that consumes them walks one channel at a time), sixteen consecutive samples per store segment.
template <class T>
__global__ __launch_bounds__(1024) void nll_fwd_tail_kernel(const T* __restrict__ x, const int64_t* __restrict__ target, const T* __restrict__ w,
                                                            T* __restrict__ out, T* __restrict__ total_weight, int64_t N, int64_t C, int64_t reduction,
                                                            int64_t ignore, int* __restrict__ assert_word, T* acc, double acc_scale,
                                                            T* __restrict__ plane_grad_t, int hw) {
  using A = acc_t<T>;
  __shared__ A sm[16];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool first = blockIdx.x == 0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int CP = ((int)C + 7) & ~7;
  T* gl = reinterpret_cast<T*>(smem) + (size_t)wid * (CP + (int)C);      // [C]: the pooled gradient (this wave's row)
  T* grow = gl + CP;                                                     // [C]: the loss's gradient row; before that, the row's log-probabilities
  const int64_t n0 = ((int64_t)blockIdx.x - 1) * 16, n = n0 + wid;
  const bool live = !first && n < N;
  // this wave's row and target are requested before the scan over all targets (whose round trips they share)
  int64_t trow = 0;
  if (live) {
    trow = target[n];
    for (int c = lane; c < (int)C; c += 64) grow[c] = x[n * C + c];
  }
  A loss = 0, tw = 0;
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
    const int64_t t = target[i];
    if (t == ignore) continue;
    if (t < 0 || t >= C) { if (first) *(volatile int*)assert_word = kAssertNllTarget; continue; }
    const A wt = w ? load_as<A>(w[t]) : A(1);
    if (first) loss -= wt * load_as<A>(x[i * C + t]);
    tw += wt;
  }
  if (first) loss = block_sum(loss, sm);                    // (block-uniform)
  tw = block_sum(tw, sm);
  const T twT = store_as<T>(tw);
  if (first) {
    if (threadIdx.x == 0) {
      *total_weight = twT;
      const T o = reduction == 1 ? store_as<T>((A)(loss / tw)) : store_as<T>(loss);
      *out = o;
      if (acc) *acc = store_as<T>((A)(load_as<A>(*acc) + (A)acc_scale * load_as<A>(o)));
    }
    return;
  }
  if (live) {
    // grad_in = g - exp(output) * sum(g) with g = the one-hot row of nll_bwd_kernel: log_softmax_bwd_row's arithmetic (its sum over the row is
    // the row's only non-zero plus zeros: exact in any order), then the division by the plane size as gap_lsm_nll_bwd_kernel rounds it
    A v = 0;
    if (trow != ignore && trow >= 0 && trow < C) {
      A g = load_as<A>(store_as<T>(A(1)));                  // the root's derivative
      if (reduction == 1) g = g / load_as<A>(twT);
      v = -(w ? load_as<A>(w[trow]) : A(1)) * g;
    }
    const T vt = store_as<T>(v), zero = store_as<T>(A(0));
    A s = 0;
    for (int c = lane; c < (int)C; c += 64) s += load_as<A>(c == trow ? vt : zero);
    s = wave_sum(s);
    for (int c = lane; c < (int)C; c += 64) {
      const T gi = store_as<T>((A)(load_as<A>(c == trow ? vt : zero) - t_exp<A>(load_as<A>(grow[c])) * s));
      gl[c] = store_as<T>((A)(load_as<A>(gi) / (A)hw));
    }
  }
  __syncthreads();
  // class-major store: [c][n0 .. n0 + 16)
  const int rows = (int)((N - n0) < 16 ? (N - n0) : 16);
  for (int idx = threadIdx.x; idx < 16 * (int)C; idx += 1024) {
    const int c = idx >> 4, r = idx & 15;
    if (r < rows) plane_grad_t[(int64_t)c * N + n0 + r] = (reinterpret_cast<const T*>(smem) + (size_t)r * (CP + (int)C))[c];
  }
}
template <class T>
__global__ void nll_fwd_none_kernel(const T* __restrict__ x, const int64_t* __restrict__ target, const T* __restrict__ w,
                                    T* __restrict__ out, int64_t N, int64_t C, int64_t ignore, int* __restrict__ assert_word) {
  using A = acc_t<T>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = target[i];
    A v = 0;
    if (t != ignore && (t < 0 || t >= C)) *(volatile int*)assert_word = kAssertNllTarget;
    if (t != ignore && t >= 0 && t < C) v = -(w ? load_as<A>(w[t]) : A(1)) * load_as<A>(x[i * C + t]);
    out[i] = store_as<T>(v);
  }
}
// grad_input[i, c] = (c == t_i && t_i != ignore) ? -w[t_i] * g : 0, g = grad/total_weight (mean), grad (sum), grad[i] (none)
template <class T>
__global__ void nll_bwd_kernel(const T* __restrict__ grad, const int64_t* __restrict__ target, const T* __restrict__ w,
                               const T* __restrict__ total_weight, T* __restrict__ gi, int64_t N, int64_t C, int64_t reduction,
                               int64_t ignore) {
  using A = acc_t<T>;
  const int64_t n = N * C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / C, c = e - i * C;
    const int64_t t = target[i];
    A v = 0;
    if (c == t && t != ignore) {
      A g = load_as<A>(reduction == 0 ? grad[i] : grad[0]);
      if (reduction == 1) g = g / load_as<A>(*total_weight);
      v = -(w ? load_as<A>(w[t]) : A(1)) * g;
    }
    gi[e] = store_as<T>(v);
  }
}

// smooth_l1 (ATen smooth_l1_loss: |d| < beta ? 0.5 d^2 / beta : |d| - 0.5 beta; beta = 0: l1) and its derivative wrt the input
template <class T>
__global__ void smooth_l1_kernel(const T* __restrict__ x, const T* __restrict__ t, T* __restrict__ out, int64_t n, double beta) {
  using A = acc_t<T>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const A d = load_as<A>(x[i]) - load_as<A>(t[i]);
    const A z = d < A(0) ? -d : d;
    out[i] = store_as<T>((A)(z < (A)beta ? A(0.5) * z * z / (A)beta : z - A(0.5) * (A)beta));
  }
}
template <class T>
__global__ void smooth_l1_bwd_kernel(const T* __restrict__ grad, const T* __restrict__ x, const T* __restrict__ t, T* __restrict__ out, int64_t n,
                                     double beta, double norm, int grad_scalar) {
  using A = acc_t<T>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const A g = load_as<A>(grad_scalar ? grad[0] : grad[i]);
    const A d = load_as<A>(x[i]) - load_as<A>(t[i]);
    A r;
    if (d <= -(A)beta) r = -(A)norm * g;
    else if (d >= (A)beta) r = (A)norm * g;
    else r = (A)norm * d * g / (A)beta;
    out[i] = store_as<T>(r);
  }
}
// binary_cross_entropy_with_logits (ATen): loss = (1 - y) x + (1 + (pw - 1) y) (log1p(exp(-|x|)) + max(-x, 0)); pw broadcasts over x
template <class T>
__global__ void bce_logits_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ pw, T* __restrict__ out, int64_t n, int64_t pw_n) {
  using A = acc_t<T>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const A xv = load_as<A>(x[i]), yv = load_as<A>(y[i]);
    const A ax = xv < A(0) ? -xv : xv, nx = -xv > A(0) ? -xv : A(0);
    const A lse = t_log1p<A>(t_exp<A>(-ax)) + nx;
    const A w = pw ? A(1) + (load_as<A>(pw[i % pw_n]) - A(1)) * yv : A(1);
    out[i] = store_as<T>((A)((A(1) - yv) * xv + w * lse));
  }
}

// mse backward: out = (x - t) * grad * scale  (grad is a scalar for mean/sum, elementwise for none)
template <class T>
__global__ void mse_bwd_kernel(const T* __restrict__ grad, const T* __restrict__ x, const T* __restrict__ t, T* __restrict__ out,
                               int64_t n, double scale, int grad_scalar) {
  using A = acc_t<T>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    A g = load_as<A>(grad_scalar ? grad[0] : grad[i]);
    out[i] = store_as<T>((A)((A)scale * (load_as<A>(x[i]) - load_as<A>(t[i])) * g));
  }
}
template <class T>
__global__ void sqdiff_kernel(const T* __restrict__ x, const T* __restrict__ t, T* __restrict__ out, int64_t n) {
  using A = acc_t<T>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    A d = load_as<A>(x[i]) - load_as<A>(t[i]);
    out[i] = store_as<T>((A)(d * d));
  }
}

static void split_dim(const Tensor* x, int64_t dim, int64_t& outer, int64_t& D, int64_t& inner) {
  int64_t d = wrap_dim(dim, x->ndim);
  outer = 1; inner = 1;
  D = x->ndim ? x->sizes[d] : 1;
  for (int i = 0; i < d; i++) outer *= x->sizes[i];
  for (int i = (int)d + 1; i < x->ndim; i++) inner *= x->sizes[i];
}

template <bool LOG> static Tensor* softmax_impl(const Tensor* x, int64_t dim) {
  check_device_tensor(x, "input");
  Hold xc(contiguous(x));
  Hold y(new_like(xc.get()));
  int64_t outer, D, inner;
  split_dim(x, dim, outer, D, inner);
  const int64_t rows = outer * inner;
  if (rows > 0 && D > 0) {
    const int64_t blocks = (rows * 64 + 255) / 256;
    hipStream_t st = current_stream(x->device());
    LAMP_DISPATCH_FLOAT(x->dtype, T, {
      constexpr int W = 16 / sizeof(T);
      const T* xp = static_cast<const Tensor*>(xc.get())->ptr<T>();
      const int64_t npk = D / W;
      const bool vec = inner == 1 && D % W == 0 && D >= 64 * W / 2 && (((uintptr_t)xp | (uintptr_t)y->raw()) & 15) == 0;
      if (!vec) hipLaunchKernelGGL((softmax_fwd_kernel<T, LOG>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), outer, D, inner);
      else if (npk <= 64) hipLaunchKernelGGL((softmax_fwd_vec_kernel<T, LOG, 1>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), rows, D);
      else if (npk <= 128) hipLaunchKernelGGL((softmax_fwd_vec_kernel<T, LOG, 2>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), rows, D);
      else if (npk <= 256) hipLaunchKernelGGL((softmax_fwd_vec_kernel<T, LOG, 4>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), rows, D);
      else if (npk <= 512) hipLaunchKernelGGL((softmax_fwd_vec_kernel<T, LOG, 8>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), rows, D);
      else hipLaunchKernelGGL((softmax_fwd_vec_kernel<T, LOG, 0>), dim3((unsigned)blocks), dim3(256), 0, st, xp, y->ptr<T>(), rows, D);
    });
    LAMP_LAUNCH_CHECK();
  }
  return y.take();
}

}  // namespace lamp

using namespace lamp;

extern "C" {

int lamp_log_softmax(lamp_tensor** out, const lamp_tensor* x, int64_t dim) { LAMP_API_BEGIN *out = softmax_impl<true>(x, dim); LAMP_API_END }
int lamp_softmax(lamp_tensor** out, const lamp_tensor* x, int64_t dim) { LAMP_API_BEGIN *out = softmax_impl<false>(x, dim); LAMP_API_END }

// lanes per plane for the packet form (as pool.hip's global_pool_lanes), 0: the operators run as the three-call chain
static int gap_lsm_lanes(const lamp_tensor* x) {
  if (!x->is_device() || x->ndim != 4 || !x->is_contiguous() || x->numel() == 0) return 0;
  if (x->dtype != kBF16 && x->dtype != kF16 && x->dtype != kF32 && x->dtype != kF64) return 0;
  const int64_t C = x->sizes[1], hw = x->sizes[2] * x->sizes[3];
  const int64_t D = C, W = 16 / x->itemsize();
  if (D % W == 0 && D >= 64 * W / 2) return 0;            // the chain would take the packet row kernels there (another summation order)
  const int64_t bytes = hw * x->itemsize();
  if (bytes % 16 != 0 || C > 8192) return 0;
  const int64_t lpp = bytes / 16;
  if (lpp < 1 || lpp > 64 || (lpp & (lpp - 1)) != 0) return 0;
  if (((uintptr_t)x->data() & 15) != 0) return 0;
  return (int)lpp;
}
int lamp_global_avg_pool_log_softmax(lamp_tensor** out, const lamp_tensor* x) {
  LAMP_API_BEGIN
  LAMP_CHECK(x != nullptr && x->ndim == 4, "global_avg_pool_log_softmax expects [N, C, H, W]");
  LAMP_CHECK(x->sizes[2] == x->sizes[3], "global_avg_pool_log_softmax: square maps only (avg_pool2d takes one kernel size), got " << x->describe());
  const int lpp = gap_lsm_lanes(x);
  if (!lpp) {                                               // avg_pool2d(k = H) -> flatten -> log_softmax(dim 1)
    lamp_tensor *p = nullptr, *f = nullptr;
    if (lamp_avg_pool2d(&p, x, x->sizes[2], 1, 0, 0, 1) != 0) throw Error(lamp_last_error());
    Hold ph(p);
    int64_t fs[2] = {x->sizes[0], x->sizes[1]};
    if (lamp_reshape(&f, p, fs, 2) != 0) throw Error(lamp_last_error());
    Hold fh(f);
    if (lamp_log_softmax(out, f, 1) != 0) throw Error(lamp_last_error());
    return 0;
  }
  int64_t ys[2] = {x->sizes[0], x->sizes[1]};
  Hold y(new_tensor(ys, 2, x->dtype, x->device()));
  const int C = (int)x->sizes[1], hw = (int)(x->sizes[2] * x->sizes[3]);
  LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((gap_lsm_fwd_kernel<T>), dim3((unsigned)x->sizes[0]), dim3(256), (size_t)C * sizeof(T), current_stream(x->device()),
                                                      x->ptr<T>(), y->ptr<T>(), C, hw, lpp));
  LAMP_LAUNCH_CHECK();
  *out = y.take();
  LAMP_API_END
}
int lamp_global_avg_pool_log_softmax_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* output, const lamp_tensor* x) {
  LAMP_API_BEGIN
  LAMP_CHECK(x != nullptr && x->ndim == 4 && grad && output, "global_avg_pool_log_softmax_backward: bad arguments");
  LAMP_CHECK(grad->ndim == 2 && grad->sizes[0] == x->sizes[0] && grad->sizes[1] == x->sizes[1] && grad->shape() == output->shape() &&
             grad->dtype == x->dtype && output->dtype == x->dtype,
             "global_avg_pool_log_softmax_backward: grad " << grad->describe() << " / output " << output->describe() << " do not match input " << x->describe());
  const int lpp = (grad->is_device() && output->is_device()) ? gap_lsm_lanes(x) : 0;
  if (!lpp) {
    lamp_tensor *gi = nullptr, *g4 = nullptr;
    if (lamp_log_softmax_backward_data(&gi, grad, output, 1) != 0) throw Error(lamp_last_error());
    Hold gih(gi);
    int64_t s4[4] = {x->sizes[0], x->sizes[1], 1, 1};
    if (lamp_reshape(&g4, gi, s4, 4) != 0) throw Error(lamp_last_error());
    Hold g4h(g4);
    if (lamp_avg_pool2d_backward(out, g4, x, x->sizes[2], 1, 0, 0, 1) != 0) throw Error(lamp_last_error());
    return 0;
  }
  Hold gc(contiguous(grad)), oc(contiguous(output));
  Hold dx(new_tensor(x->shape(), x->dtype, x->device()));
  const int C = (int)x->sizes[1], hw = (int)(x->sizes[2] * x->sizes[3]);
  LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((gap_lsm_bwd_kernel<T>), dim3((unsigned)x->sizes[0]), dim3(256), (size_t)C * sizeof(T), current_stream(x->device()),
                                                      static_cast<const Tensor*>(gc.get())->ptr<T>(), static_cast<const Tensor*>(oc.get())->ptr<T>(), dx->ptr<T>(), C, hw, lpp));
  LAMP_LAUNCH_CHECK();
  *out = dx.take();
  LAMP_API_END
}

int lamp_log_softmax_backward_data(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* output, int64_t dim) {
  LAMP_API_BEGIN
  check_device_tensor(grad, "grad_output"); check_device_tensor(output, "output");
  LAMP_CHECK(grad->shape() == output->shape() && grad->dtype == output->dtype, "grad/output mismatch: " << grad->describe() << " vs " << output->describe());
  Hold gc(contiguous(grad)), oc(contiguous(output));
  Hold gi(new_like(oc.get()));
  int64_t outer, D, inner;
  split_dim(output, dim, outer, D, inner);
  const int64_t rows = outer * inner;
  if (rows > 0 && D > 0) {
    const int64_t blocks = (rows * 64 + 255) / 256;
    LAMP_DISPATCH_FLOAT(output->dtype, T, {
      constexpr int W = 16 / sizeof(T);
      const T* gp = static_cast<const Tensor*>(gc.get())->ptr<T>();
      const T* op = static_cast<const Tensor*>(oc.get())->ptr<T>();
      const bool vec = inner == 1 && D % W == 0 && D >= 64 * W / 2 && (((uintptr_t)gp | (uintptr_t)op | (uintptr_t)gi->raw()) & 15) == 0;
      if (vec) hipLaunchKernelGGL((log_softmax_bwd_vec_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, current_stream(output->device()), gp, op, gi->ptr<T>(), rows, D);
      else hipLaunchKernelGGL((log_softmax_bwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, current_stream(output->device()), gp, op, gi->ptr<T>(), outer, D, inner);
    });
    LAMP_LAUNCH_CHECK();
  }
  *out = gi.take();
  LAMP_API_END
}

static void nll_check(const lamp_tensor* x, const lamp_tensor* target, const lamp_tensor* weight) {
  check_device_tensor(x, "input"); check_device_tensor(target, "target");
  LAMP_CHECK(x->ndim == 2, "nll_loss expects a 2-D input (samples x classes), got " << x->describe());
  LAMP_CHECK(target->ndim == 1 && target->dtype == kI64, "nll_loss expects a 1-D int64 target, got " << target->describe());
  LAMP_CHECK(target->sizes[0] == x->sizes[0], "nll_loss: batch size mismatch " << x->describe() << " vs " << target->describe());
  if (weight) {
    check_device_tensor(weight, "weight");
    LAMP_CHECK(weight->numel() == x->sizes[1] && weight->dtype == x->dtype, "nll_loss: weight must have " << x->sizes[1] << " elements of the input dtype");
  }
}

static int nll_forward_impl(lamp_tensor** out, lamp_tensor** total_weight, const lamp_tensor* x, const lamp_tensor* target, const lamp_tensor* weight,
                            int64_t reduction, int64_t ignore_index, lamp_tensor* acc, double acc_scale);
int lamp_nll_loss_forward(lamp_tensor** out, lamp_tensor** total_weight, const lamp_tensor* x, const lamp_tensor* target,
                          const lamp_tensor* weight, int64_t reduction, int64_t ignore_index) {
  return nll_forward_impl(out, total_weight, x, target, weight, reduction, ignore_index, nullptr, 0.0);
}
int lamp_nll_loss_forward_accumulate_(lamp_tensor** out, lamp_tensor** total_weight, const lamp_tensor* x, const lamp_tensor* target,
                                      const lamp_tensor* weight, int64_t reduction, int64_t ignore_index, lamp_tensor* acc, double scale) {
  if (!acc) { set_last_error("nll_loss_forward_accumulate_: null accumulator"); return 1; }
  return nll_forward_impl(out, total_weight, x, target, weight, reduction, ignore_index, acc, scale);
}
static int nll_forward_impl(lamp_tensor** out, lamp_tensor** total_weight, const lamp_tensor* x, const lamp_tensor* target, const lamp_tensor* weight,
                            int64_t reduction, int64_t ignore_index, lamp_tensor* acc, double acc_scale) {
  LAMP_API_BEGIN
  nll_check(x, target, weight);
  if (acc) {
    check_device_tensor(acc, "accumulator");
    LAMP_CHECK(reduction != 0 && acc->numel() == 1 && acc->dtype == x->dtype && acc->device() == x->device(),
               "nll_loss_forward_accumulate_: the accumulator must be a one-element tensor of the input's dtype on its device (and the reduction mean or sum), got "
                   << acc->describe());
  }
  LAMP_CHECK(reduction >= 0 && reduction <= 2, "bad reduction " << reduction);
  Hold xc(contiguous(x)), tc(contiguous(target));
  Hold wc(weight ? contiguous(weight) : nullptr);
  const int64_t N = x->sizes[0], C = x->sizes[1];
  hipStream_t st = current_stream(x->device());
  Hold tw(new_tensor(nullptr, 0, x->dtype, x->device()));
  int* aw = device_assert_word(x->device());
  if (reduction == 0) {
    int64_t sz[1] = {N};
    Hold o(new_tensor(sz, 1, x->dtype, x->device()));
    fill_zero(tw.get());
    if (N > 0) {
      LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((nll_fwd_none_kernel<T>), dim3(grid_for(N, 256)), dim3(256), 0, st,
                                                          xc->ptr<T>(), tc->ptr<int64_t>(), wc.get() ? wc->ptr<T>() : (const T*)nullptr,
                                                          o->ptr<T>(), N, C, ignore_index, aw));
      LAMP_LAUNCH_CHECK();
    }
    *out = o.take();
  } else {
    Hold o(new_tensor(nullptr, 0, x->dtype, x->device()));
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((nll_fwd_reduce_kernel<T>), dim3(1), dim3(1024), 0, st, xc->ptr<T>(),
                                                        tc->ptr<int64_t>(), wc.get() ? wc->ptr<T>() : (const T*)nullptr, o->ptr<T>(),
                                                        tw->ptr<T>(), N, C, reduction, ignore_index, aw, acc ? acc->ptr<T>() : (T*)nullptr, acc_scale));
    LAMP_LAUNCH_CHECK();
    *out = o.take();
  }
  *total_weight = tw.take();
  LAMP_API_END
}

int lamp_nll_loss_forward_pooled_gradient_(lamp_tensor** out, lamp_tensor** total_weight, lamp_tensor** plane_grad, const lamp_tensor* x,
                                           const lamp_tensor* target, const lamp_tensor* weight, int64_t reduction, int64_t ignore_index,
                                           lamp_tensor* acc_or_null, double scale, int64_t plane_elems) {
  LAMP_API_BEGIN
  *plane_grad = nullptr;
  nll_check(x, target, weight);
  LAMP_CHECK(reduction == 1 || reduction == 2, "nll_loss_forward_pooled_gradient_: reduction mean (1) or sum (2), got " << reduction);
  LAMP_CHECK(plane_elems >= 1 && plane_elems <= (1 << 24), "nll_loss_forward_pooled_gradient_: bad plane size " << plane_elems);
  const int64_t N = x->sizes[0], C = x->sizes[1];
  const size_t lds = 16 * (size_t)((((int)C + 7) & ~7) + (int)C) * x->itemsize();
  if (N == 0 || C > 4096 || lds > 48 * 1024) {              // no fused form for this shape: the plain forward, no gradient (the caller runs the backward)
    return nll_forward_impl(out, total_weight, x, target, weight, reduction, ignore_index, acc_or_null, scale);
  }
  if (acc_or_null) {
    check_device_tensor(acc_or_null, "accumulator");
    LAMP_CHECK(acc_or_null->numel() == 1 && acc_or_null->dtype == x->dtype && acc_or_null->device() == x->device(),
               "nll_loss_forward_pooled_gradient_: the accumulator must be a one-element tensor of the input's dtype on its device, got " << acc_or_null->describe());
  }
  Hold xc(contiguous(x)), tc(contiguous(target));
  Hold wc(weight ? contiguous(weight) : nullptr);
  hipStream_t st = current_stream(x->device());
  Hold tw(new_tensor(nullptr, 0, x->dtype, x->device())), o(new_tensor(nullptr, 0, x->dtype, x->device()));
  int64_t ts[2] = {C, N};
  Hold pgt(new_tensor(ts, 2, x->dtype, x->device()));       // class-major values; handed out as their [N, C] transpose (strides [1, N])
  int* aw = device_assert_word(x->device());
  const unsigned blocks = 1u + (unsigned)((N + 15) / 16);
  LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((nll_fwd_tail_kernel<T>), dim3(blocks), dim3(1024), lds, st, static_cast<const Tensor*>(xc.get())->ptr<T>(),
                                                      static_cast<const Tensor*>(tc.get())->ptr<int64_t>(),
                                                      wc.get() ? static_cast<const Tensor*>(wc.get())->ptr<T>() : (const T*)nullptr, o->ptr<T>(), tw->ptr<T>(), N, C,
                                                      reduction, ignore_index, aw, acc_or_null ? acc_or_null->ptr<T>() : (T*)nullptr, scale, pgt->ptr<T>(), (int)plane_elems));
  LAMP_LAUNCH_CHECK();
  *out = o.take();
  *total_weight = tw.take();
  lamp_tensor* tr = nullptr;
  if (lamp_transpose(&tr, pgt.get(), 0, 1) != 0) throw Error(lamp_last_error());
  *plane_grad = tr;
  LAMP_API_END
}

int lamp_nll_loss_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* target,
                           const lamp_tensor* weight, int64_t reduction, int64_t ignore_index, const lamp_tensor* total_weight) {
  LAMP_API_BEGIN
  nll_check(x, target, weight);
  check_device_tensor(grad_out, "grad_output"); check_device_tensor(total_weight, "total_weight");
  LAMP_CHECK(grad_out->dtype == x->dtype && total_weight->dtype == x->dtype, "nll_loss_backward: dtype mismatch");
  const int64_t N = x->sizes[0], C = x->sizes[1];
  if (reduction == 0) LAMP_CHECK(grad_out->numel() == N, "nll_loss_backward: grad_output must have N elements for reduction none");
  else LAMP_CHECK(grad_out->numel() == 1, "nll_loss_backward: grad_output must be a scalar");
  Hold gc(contiguous(grad_out)), tc(contiguous(target));
  Hold wc(weight ? contiguous(weight) : nullptr);
  Hold gi(new_tensor(x->sizes, 2, x->dtype, x->device()));
  if (N * C > 0) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((nll_bwd_kernel<T>), dim3(grid_for(N * C, 256)), dim3(256), 0,
                                                        current_stream(x->device()), gc->ptr<T>(), tc->ptr<int64_t>(),
                                                        wc.get() ? wc->ptr<T>() : (const T*)nullptr, total_weight->ptr<T>(),
                                                        gi->ptr<T>(), N, C, reduction, ignore_index));
    LAMP_LAUNCH_CHECK();
  }
  *out = gi.take();
  LAMP_API_END
}

int lamp_global_avg_pool_log_softmax_nll_backward(lamp_tensor** out, const lamp_tensor* grad_loss, const lamp_tensor* target, const lamp_tensor* weight,
                                                  int64_t reduction, int64_t ignore_index, const lamp_tensor* total_weight, const lamp_tensor* output,
                                                  const lamp_tensor* x) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(output, "output");
  nll_check(output, target, weight);
  check_device_tensor(grad_loss, "grad_output"); check_device_tensor(total_weight, "total_weight");
  LAMP_CHECK(x->ndim == 4 && output->ndim == 2 && output->sizes[0] == x->sizes[0] && output->sizes[1] == x->sizes[1] && output->dtype == x->dtype,
             "global_avg_pool_log_softmax_nll_backward: output " << output->describe() << " does not match input " << x->describe());
  LAMP_CHECK(grad_loss->dtype == x->dtype && total_weight->dtype == x->dtype, "global_avg_pool_log_softmax_nll_backward: dtype mismatch");
  const int64_t N = x->sizes[0];
  if (reduction == 0) LAMP_CHECK(grad_loss->numel() == N, "nll_loss_backward: grad_output must have N elements for reduction none");
  else LAMP_CHECK(grad_loss->numel() == 1, "nll_loss_backward: grad_output must be a scalar");
  const int lpp = (grad_loss->is_device() && output->is_device() && target->is_device()) ? gap_lsm_lanes(x) : 0;
  if (!lpp || N == 0) {
    // the two calls
    lamp_tensor* gy = nullptr;
    if (lamp_nll_loss_backward(&gy, grad_loss, output, target, weight, reduction, ignore_index, total_weight) != 0) throw Error(lamp_last_error());
    Hold gyh(gy);
    if (lamp_global_avg_pool_log_softmax_backward(out, gy, output, x) != 0) throw Error(lamp_last_error());
    return 0;
  }
  Hold gc(contiguous(grad_loss)), tc(contiguous(target)), oc(contiguous(output));
  Hold wc(weight ? contiguous(weight) : nullptr);
  Hold dx(new_tensor(x->shape(), x->dtype, x->device()));
  const int C = (int)x->sizes[1], hw = (int)(x->sizes[2] * x->sizes[3]);
  LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((gap_lsm_nll_bwd_kernel<T>), dim3((unsigned)N), dim3(256), (size_t)(((C + 7) & ~7) + C) * sizeof(T),
                                                      current_stream(x->device()), static_cast<const Tensor*>(gc.get())->ptr<T>(),
                                                      static_cast<const Tensor*>(tc.get())->ptr<int64_t>(),
                                                      wc.get() ? static_cast<const Tensor*>(wc.get())->ptr<T>() : (const T*)nullptr,
                                                      total_weight->ptr<T>(), reduction, ignore_index, static_cast<const Tensor*>(oc.get())->ptr<T>(),
                                                      dx->ptr<T>(), C, hw, lpp));
  LAMP_LAUNCH_CHECK();
  *out = dx.take();
  LAMP_API_END
}

int lamp_mse_loss(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* target, int64_t reduction) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(target, "target");
  LAMP_CHECK(x->shape() == target->shape() && x->dtype == target->dtype, "mse_loss: input/target mismatch " << x->describe() << " vs " << target->describe());
  Hold xc(contiguous(x)), tc(contiguous(target));
  Hold sq(new_like(xc.get()));
  const int64_t n = x->numel();
  if (n > 0) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((sqdiff_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0,
                                                        current_stream(x->device()), xc->ptr<T>(), tc->ptr<T>(), sq->ptr<T>(), n));
    LAMP_LAUNCH_CHECK();
  }
  if (reduction == 0) { *out = sq.take(); return 0; }
  return reduction == 1 ? lamp_mean_all(out, sq.get()) : lamp_sum_all(out, sq.get());
  LAMP_API_END
}

int lamp_smooth_l1_loss(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* target, int64_t reduction, double beta) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(target, "target");
  LAMP_CHECK(x->shape() == target->shape() && x->dtype == target->dtype, "smooth_l1_loss: input/target mismatch " << x->describe() << " vs " << target->describe());
  LAMP_CHECK(beta >= 0, "smooth_l1_loss: negative beta");
  Hold xc(contiguous(x)), tc(contiguous(target));
  Hold el(new_like(xc.get()));
  const int64_t n = x->numel();
  if (n > 0) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((smooth_l1_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, current_stream(x->device()), xc->ptr<T>(),
                                                        tc->ptr<T>(), el->ptr<T>(), n, beta));
    LAMP_LAUNCH_CHECK();
  }
  if (reduction == 0) { *out = el.take(); return 0; }
  return reduction == 1 ? lamp_mean_all(out, el.get()) : lamp_sum_all(out, el.get());
  LAMP_API_END
}
int lamp_smooth_l1_loss_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* target, int64_t reduction, double beta) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(target, "target"); check_device_tensor(grad_out, "grad_output");
  LAMP_CHECK(x->shape() == target->shape() && x->dtype == target->dtype && grad_out->dtype == x->dtype, "smooth_l1_loss_backward: mismatch");
  const int64_t n = x->numel();
  const int grad_scalar = grad_out->numel() == 1;
  if (!grad_scalar) LAMP_CHECK(grad_out->numel() == n, "smooth_l1_loss_backward: grad_output has the wrong size");
  Hold xc(contiguous(x)), tc(contiguous(target)), gc(contiguous(grad_out));
  Hold gi(new_like(xc.get()));
  const double norm = reduction == 1 ? 1.0 / (double)n : 1.0;
  if (n > 0) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((smooth_l1_bwd_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, current_stream(x->device()),
                                                        gc->ptr<T>(), xc->ptr<T>(), tc->ptr<T>(), gi->ptr<T>(), n, beta, norm, grad_scalar));
    LAMP_LAUNCH_CHECK();
  }
  *out = gi.take();
  LAMP_API_END
}
int lamp_binary_cross_entropy_with_logits(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* target, const lamp_tensor* pos_weight, int64_t reduction) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(target, "target");
  LAMP_CHECK(x->shape() == target->shape() && x->dtype == target->dtype, "binary_cross_entropy_with_logits: input/target mismatch");
  Hold xc(contiguous(x)), tc(contiguous(target));
  Hold pc;
  int64_t pw_n = 1;
  if (pos_weight) {
    check_device_tensor(pos_weight, "pos_weight");
    LAMP_CHECK(pos_weight->dtype == x->dtype, "binary_cross_entropy_with_logits: pos_weight dtype");
    // pos_weight broadcasts against the trailing dims of the input: it is dense over the last numel(pos_weight) elements of every row
    auto bs = broadcast_shapes(x->shape(), pos_weight->shape());
    LAMP_CHECK(bs == x->shape(), "binary_cross_entropy_with_logits: pos_weight does not broadcast to the input");
    int lead = 0;
    while (lead < pos_weight->ndim && pos_weight->sizes[lead] == 1) lead++;
    for (int d = lead; d < pos_weight->ndim; d++)
      LAMP_CHECK(pos_weight->sizes[d] == x->sizes[x->ndim - pos_weight->ndim + d], "binary_cross_entropy_with_logits: pos_weight must cover whole trailing dims");
    pc = Hold(contiguous(pos_weight));
    pw_n = pos_weight->numel();
  }
  Hold el(new_like(xc.get()));
  const int64_t n = x->numel();
  if (n > 0) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((bce_logits_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, current_stream(x->device()), xc->ptr<T>(),
                                                        tc->ptr<T>(), pc.get() ? static_cast<const Tensor*>(pc.get())->ptr<T>() : (const T*)nullptr, el->ptr<T>(), n, pw_n));
    LAMP_LAUNCH_CHECK();
  }
  if (reduction == 0) { *out = el.take(); return 0; }
  return reduction == 1 ? lamp_mean_all(out, el.get()) : lamp_sum_all(out, el.get());
  LAMP_API_END
}

int lamp_mse_loss_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, const lamp_tensor* target, int64_t reduction) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(target, "target"); check_device_tensor(grad_out, "grad_output");
  LAMP_CHECK(x->shape() == target->shape() && x->dtype == target->dtype && grad_out->dtype == x->dtype, "mse_loss_backward: mismatch");
  const int64_t n = x->numel();
  int grad_scalar = grad_out->numel() == 1 && n != 1 ? 1 : (grad_out->numel() == 1);
  if (!grad_scalar) LAMP_CHECK(grad_out->numel() == n, "mse_loss_backward: grad_output has the wrong size");
  Hold xc(contiguous(x)), tc(contiguous(target)), gc(contiguous(grad_out));
  Hold gi(new_like(xc.get()));
  const double scale = reduction == 1 ? 2.0 / (double)n : 2.0;
  if (n > 0) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((mse_bwd_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0,
                                                        current_stream(x->device()), gc->ptr<T>(), xc->ptr<T>(), tc->ptr<T>(),
                                                        gi->ptr<T>(), n, scale, grad_scalar));
    LAMP_LAUNCH_CHECK();
  }
  *out = gi.take();
  LAMP_API_END
}

}  // extern "C"
