// Fused multi-tensor optimiser kernels: AdamW, SGDW, global-norm gradient clipping and the
// flat gradient bucket used by data-parallel training.
//
// Replaces the per-tensor ATen call chains of lamp's optimisers with ONE launch over all
// parameter tensors (reference: lamp-core/src/main/scala/lamp/nn/AdamW.scala:101-176 - mul_,
// add_out, mul_, addcmul_out, sqrt, add_(eps), add_out(wd), addcdiv_out per tensor;
// nn/SGD.scala:46-98; nn/package.scala:72-100 gradientClippingInPlace;
// lamp-data/.../distributed/package.scala:690-719 averageGradients).
// The arithmetic and its order follow the reference line by line; the ResNet step has 37
// parameter tensors, i.e. ~300 tiny launches become one.
#include "device_utils.h"
#include "../core/strided.h"

namespace lamp {

void igemm_repack_cached(lamp_tensor* const* params, int n, hipStream_t st);   // conv_igemm.hip
void igemm32_repack_cached(lamp_tensor* const* params, int n, hipStream_t st); // conv_igemm_f32.hip
void small_repack_cached(lamp_tensor* const* params, int n, hipStream_t st);   // conv_small.hip

constexpr int MT_MAX = 40;       // tensors per launch (the descriptor is a by-value kernel argument: 3.4 KB of the 4 KB limit; the ResNet has 37)
constexpr int MT_CHUNK = 1024;   // elements per workgroup (4096: the ResNet's 392 k parameters made ~130 workgroups for 256 CUs - 15 us per AdamW launch)

struct MultiArgs {
  int n;
  void* p[5][MT_MAX];            // up to 5 tensor lists
  int64_t numel[MT_MAX];
  int blk_start[MT_MAX + 1];     // prefix sums of per-tensor chunk counts
  double h[4][MT_MAX];           // per-tensor hyper parameters
  double s[4];                   // scalars
  int flags;
};

static_assert(sizeof(MultiArgs) <= 4096, "MultiArgs is passed by value: HIP kernel arguments are limited to 4 KB");

// the tensor whose chunks contain workgroup chunk `blk`: binary search (the descriptor lives in the kernel-argument segment: every probe
// is a dependent scalar load, and the linear walk over 40 tensors cost a 2-KB chunk more than its data did - 0.8 TB/s in the
// gradient-norm kernels of the language model)
__device__ __forceinline__ int mt_find(const MultiArgs& a, int blk) {
  int lo = 0, hi = a.n - 1;                   // invariant: blk_start[lo] <= blk < blk_start[hi + 1]
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (blk >= a.blk_start[mid]) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// AdamW.scala:124-166.  T = parameter/optimizer-state dtype (f32 or f64; bf16 state allowed),
// G = gradient dtype, M = model parameter dtype when a master (working) copy is used.
// p[0]=param (or master copy), p[1]=grad, p[2]=m, p[3]=v, p[4]=model param to down-cast into (or null)
// h[0]=stepParam (= sched*lr*sqrt(1-b2^t)/(1-b1^t) or sched*lr), h[1]=stepWd, h[2]=beta1, h[3]=beta2, s[0]=eps
template <class T, class G>
__global__ __launch_bounds__(256) void adamw_kernel(MultiArgs a) {
  using A = acc_t<T>;
  const int t = mt_find(a, blockIdx.x);
  const int64_t begin = (int64_t)(blockIdx.x - a.blk_start[t]) * MT_CHUNK;
  const int64_t end = min(begin + MT_CHUNK, a.numel[t]);
  T* p = (T*)a.p[0][t];
  const G* g = (const G*)a.p[1][t];
  T* m = (T*)a.p[2][t];
  T* v = (T*)a.p[3][t];
  G* model = (G*)a.p[4][t];
  const A stepParam = (A)a.h[0][t], stepWd = (A)a.h[1][t], b1 = (A)a.h[2][t], b2 = (A)a.h[3][t];
  const A one_m_b1 = (A)(1.0 - a.h[2][t]), one_m_b2 = (A)(1.0 - a.h[3][t]);
  const A eps = (A)a.s[0];
  for (int64_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
    const A gr = load_as<A>(g[i]);
    A mt = load_as<A>(m[i]) * b1;          // mt *= b1
    mt = mt + one_m_b1 * gr;               // addOut(mt, mt, g, 1-b1)
    A vt = load_as<A>(v[i]) * b2;          // vt *= b2
    vt = vt + (one_m_b2 * gr) * gr;        // addcmulOut(vt, vt, g, g, 1-b2)
    A denom = (A)sqrt((double)vt);
    if (sizeof(A) == 4) denom = sqrtf((float)vt);
    denom = denom + eps;
    A pv = load_as<A>(p[i]);
    if (stepWd != A(0)) pv = pv + (-stepWd) * pv;      // addOut(param, param, param, -stepWd)
    pv = pv + ((-stepParam) * mt) / denom;             // addcdivOut(param, param, mt, denom, -stepParam)
    m[i] = store_as<T>(mt);
    v[i] = store_as<T>(vt);
    p[i] = store_as<T>(pv);
    if (model) model[i] = store_as<G>(load_as<acc_t<G>>(store_as<T>(pv)));
  }
}

// SGD.scala:46-98. p[0]=param, p[1]=grad, p[2]=velocity (or null); h[0]=lr*sched, h[1]=wd*sched, h[2]=momentum
template <class T>
__global__ __launch_bounds__(256) void sgdw_kernel(MultiArgs a) {
  using A = acc_t<T>;
  const int t = mt_find(a, blockIdx.x);
  const int64_t begin = (int64_t)(blockIdx.x - a.blk_start[t]) * MT_CHUNK;
  const int64_t end = min(begin + MT_CHUNK, a.numel[t]);
  T* p = (T*)a.p[0][t];
  const T* g = (const T*)a.p[1][t];
  T* vel = (T*)a.p[2][t];
  const A lr = (A)a.h[0][t], wd = (A)a.h[1][t], mom = (A)a.h[2][t];
  for (int64_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
    A pv = load_as<A>(p[i]);
    const A gr = load_as<A>(g[i]);
    if (!vel) {
      if (wd != A(0)) pv = pv + (-wd) * pv;
      pv = pv + (-lr) * gr;
    } else {
      A vv = load_as<A>(vel[i]) * mom;
      vv = vv + lr * gr;
      if (wd != A(0)) pv = pv + (-wd) * pv;
      pv = pv + A(-1) * vv;
      vel[i] = store_as<T>(vv);
    }
    p[i] = store_as<T>(pv);
  }
}

// sum of squares per workgroup -> partial[blockIdx.x] (deterministic two stage reduction)
template <class T>
__global__ __launch_bounds__(256) void mt_sumsq_kernel(MultiArgs a, acc_t<T>* __restrict__ partial, int per_wg) {
  using A = acc_t<T>;
  __shared__ A sm[4];
  // a workgroup takes `per_wg` consecutive chunks (a 2-KB chunk per workgroup is bound by the dispatch rate: 21 k workgroups in 46 us)
  A acc = 0;
  const int nblk = a.blk_start[a.n];
  int t = blockIdx.x * per_wg < nblk ? mt_find(a, blockIdx.x * per_wg) : 0;
  for (int j = 0; j < per_wg; j++) {
    const int blk = blockIdx.x * per_wg + j;
    if (blk >= nblk) break;
    while (blk >= a.blk_start[t + 1]) t++;
    const int64_t begin = (int64_t)(blk - a.blk_start[t]) * MT_CHUNK;
    const int64_t end = min(begin + MT_CHUNK, a.numel[t]);
    const T* g = (const T*)a.p[0][t];
    // 16-byte packets where the chunk starts on one
    constexpr int W = 16 / sizeof(T);
    int64_t i0 = begin;
    if ((((uintptr_t)(g + begin)) & 15) == 0) {
      const int64_t nv = (end - begin) / W;
      const Vec<T, W>* gv = reinterpret_cast<const Vec<T, W>*>(g + begin);
      for (int64_t v = threadIdx.x; v < nv; v += blockDim.x) {
        const Vec<T, W> pk = gv[v];
#pragma unroll
        for (int k = 0; k < W; k++) { const A x = load_as<A>(pk.v[k]); acc += x * x; }
      }
      i0 = begin + nv * W;
    }
    for (int64_t i = i0 + threadIdx.x; i < end; i += blockDim.x) { const A x = load_as<A>(g[i]); acc += x * x; }
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
// total[0] += sum(partial) ; single workgroup
template <class A>
__global__ __launch_bounds__(256) void mt_accumulate_kernel(const A* __restrict__ partial, int n, A* __restrict__ total, int overwrite) {
  __shared__ A sm[4];
  A acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) total[0] = overwrite ? acc : total[0] + acc;
}
// g *= min(1, theta / sqrt(total))     nn/package.scala:88-97
template <class T>
__global__ __launch_bounds__(256) void mt_clip_scale_kernel(MultiArgs a, const acc_t<T>* __restrict__ total, int per_wg) {
  using A = acc_t<T>;
  // same rounding points as the reference: norm = sqrt(sum) in the gradient dtype, scalar = theta/norm, min(scalar, 1)
  const A norm = (A)load_as<A>(store_as<T>((A)sqrt((double)total[0])));
  A sc = (A)a.s[0] / norm;
  sc = sc < A(1) ? sc : A(1);
  if (sc != sc) sc = A(1) < sc ? A(1) : sc;  // NaN norm: ATen minimum propagates NaN
  const int nblk = a.blk_start[a.n];
  int t = blockIdx.x * per_wg < nblk ? mt_find(a, blockIdx.x * per_wg) : 0;
  for (int j = 0; j < per_wg; j++) {
    const int blk = blockIdx.x * per_wg + j;
    if (blk >= nblk) break;
    while (blk >= a.blk_start[t + 1]) t++;
    const int64_t begin = (int64_t)(blk - a.blk_start[t]) * MT_CHUNK;
    const int64_t end = min(begin + MT_CHUNK, a.numel[t]);
    T* g = (T*)a.p[0][t];
    constexpr int W = 16 / sizeof(T);
    int64_t i0 = begin;
    if ((((uintptr_t)(g + begin)) & 15) == 0) {
      const int64_t nv = (end - begin) / W;
      Vec<T, W>* gv = reinterpret_cast<Vec<T, W>*>(g + begin);
      for (int64_t v = threadIdx.x; v < nv; v += blockDim.x) {
        Vec<T, W> pk = gv[v];
#pragma unroll
        for (int k = 0; k < W; k++) pk.v[k] = store_as<T>((A)(load_as<A>(pk.v[k]) * sc));
        gv[v] = pk;
      }
      i0 = begin + nv * W;
    }
    for (int64_t i = i0 + threadIdx.x; i < end; i += blockDim.x) g[i] = store_as<T>((A)(load_as<A>(g[i]) * sc));
  }
}

// bucket[off_t + i] = scale * t[i] (as f32) ; p[0] = tensors, h[0][t] = element offset inside the bucket
template <class T>
__global__ __launch_bounds__(256) void mt_flatten_kernel(MultiArgs a, float* __restrict__ bucket) {
  const int t = mt_find(a, blockIdx.x);
  const int64_t begin = (int64_t)(blockIdx.x - a.blk_start[t]) * MT_CHUNK;
  const int64_t end = min(begin + MT_CHUNK, a.numel[t]);
  const T* src = (const T*)a.p[0][t];
  float* dst = bucket + (int64_t)a.h[0][t];
  const float scale = (float)a.s[0];
  for (int64_t i = begin + threadIdx.x; i < end; i += blockDim.x) dst[i] = scale * load_as<float>(src[i]);
}
template <class T>
__global__ __launch_bounds__(256) void mt_unflatten_kernel(MultiArgs a, const float* __restrict__ bucket, int64_t last_index) {
  const int t = mt_find(a, blockIdx.x);
  const int64_t begin = (int64_t)(blockIdx.x - a.blk_start[t]) * MT_CHUNK;
  const int64_t end = min(begin + MT_CHUNK, a.numel[t]);
  T* dst = (T*)a.p[0][t];
  const float* src = bucket + (int64_t)a.h[0][t];
  const float div = last_index >= 0 ? bucket[last_index] : 1.0f;
  for (int64_t i = begin + threadIdx.x; i < end; i += blockDim.x) dst[i] = store_as<T>((acc_t<T>)(src[i] / div));
}

// host: iterate over the tensor lists in groups of MT_MAX
template <class Fill, class Launch>
static void for_each_group(int n, const int64_t* numels, Fill fill, Launch launch) {
  for (int base = 0; base < n; base += MT_MAX) {
    MultiArgs a{};
    a.n = std::min(MT_MAX, n - base);
    int blk = 0;
    for (int t = 0; t < a.n; t++) {
      a.numel[t] = numels[base + t];
      a.blk_start[t] = blk;
      blk += (int)((a.numel[t] + MT_CHUNK - 1) / MT_CHUNK);
      fill(a, t, base + t);
    }
    a.blk_start[a.n] = blk;
    if (blk > 0) launch(a, blk);
  }
}

static void check_list(lamp_tensor* const* ts, int n, const char* what, int dtype, int device) {
  for (int i = 0; i < n; i++) {
    check_device_tensor(ts[i], what);
    LAMP_CHECK(ts[i]->is_contiguous(), what << "[" << i << "] must be contiguous");
    if (dtype >= 0) LAMP_CHECK(ts[i]->dtype == dtype, what << "[" << i << "] has dtype " << dtype_name(ts[i]->dtype) << ", expected " << dtype_name(dtype));
    LAMP_CHECK(ts[i]->device() == device, what << "[" << i << "] is on another device");
  }
}

}  // namespace lamp

using namespace lamp;

// The bucket layout on HOST tensors (same offsets, same f32 arithmetic: bucket = scale * g rounded to f32, g = bucket / last as f32):
// a control-plane process can pack / unpack gradients that live in host memory - the exchange protocol of the data-parallel step
// is testable between CPU-only ranks, and host-resident replicas (lamp's CPU device) can take part in an exchange.
template <class T>
static void host_flatten(float* bucket, lamp_tensor* const* ts, int n, const std::vector<int64_t>& offs, double scale) {
  const float sc = (float)scale;
  for (int i = 0; i < n; i++) {
    const T* src = ts[i]->ptr<T>();
    float* dst = bucket + offs[i];
    for (int64_t k = 0, e = ts[i]->numel(); k < e; k++) dst[k] = sc * load_as<float>(src[k]);
  }
}
template <class T>
static void host_unflatten(lamp_tensor* const* ts, int n, const float* bucket, const std::vector<int64_t>& offs, int64_t last) {
  const float div = last >= 0 ? bucket[last] : 1.0f;
  for (int i = 0; i < n; i++) {
    T* dst = ts[i]->ptr<T>();
    const float* src = bucket + offs[i];
    for (int64_t k = 0, e = ts[i]->numel(); k < e; k++) dst[k] = store_as<T>((acc_t<T>)(src[k] / div));
  }
}
static bool all_host(const lamp_tensor* bucket, lamp_tensor* const* ts, int n) {
  if (bucket->is_device()) return false;
  for (int i = 0; i < n; i++) {
    LAMP_CHECK(ts[i] && !ts[i]->is_device(), "host bucket with a device tensor in the list");
    LAMP_CHECK(ts[i]->is_contiguous() && ts[i]->dtype == ts[0]->dtype, "host bucket: tensors must be contiguous and of one dtype");
  }
  return true;
}

extern "C" {

int lamp_gradient_clipping_(lamp_tensor* const* grads, int n, double theta) {
  LAMP_API_BEGIN
  if (n == 0) return 0;
  const int dtype = grads[0]->dtype, dev = grads[0]->device();
  check_list(grads, n, "gradients", dtype, dev);
  hipStream_t st = current_stream(dev);
  std::vector<int64_t> numels(n);
  for (int i = 0; i < n; i++) numels[i] = grads[i]->numel();
  LAMP_DISPATCH_FLOAT(dtype, T, {
    using A = acc_t<T>;
    const int adt = std::is_same<A, double>::value ? kF64 : kF32;
    int64_t one[1] = {1};
    Hold total(new_tensor(one, 1, adt, dev));
    bool first = true;
    for_each_group(n, numels.data(), [&](MultiArgs& a, int t, int gi) { a.p[0][t] = grads[gi]->data(); },
                   [&](MultiArgs& a, int blk) {
                     // up to 16 chunks per workgroup once there are more than 8 chunks per CU (small models keep one: the ResNet has 383 chunks)
                     const int per_wg = std::min(16, std::max(1, blk / (8 * num_cus()))), wgs = (blk + per_wg - 1) / per_wg;
                     int64_t ps[1] = {wgs};
                     Hold partial(new_tensor(ps, 1, adt, dev));
                     hipLaunchKernelGGL((mt_sumsq_kernel<T>), dim3(wgs), dim3(256), 0, st, a, partial->ptr<A>(), per_wg);
                     hipLaunchKernelGGL((mt_accumulate_kernel<A>), dim3(1), dim3(256), 0, st, partial->ptr<A>(), wgs, total->ptr<A>(), first ? 1 : 0);
                     first = false;
                   });
    LAMP_LAUNCH_CHECK();
    for_each_group(n, numels.data(), [&](MultiArgs& a, int t, int gi) { a.p[0][t] = grads[gi]->data(); a.s[0] = theta; },
                   [&](MultiArgs& a, int blk) {
                     const int per_wg = std::min(16, std::max(1, blk / (8 * num_cus()))), wgs = (blk + per_wg - 1) / per_wg;
                     hipLaunchKernelGGL((mt_clip_scale_kernel<T>), dim3(wgs), dim3(256), 0, st, a, total->ptr<A>(), per_wg);
                   });
    LAMP_LAUNCH_CHECK();
  });
  LAMP_API_END
}

int lamp_adamw_step_(lamp_tensor* const* params, lamp_tensor* const* grads, lamp_tensor* const* m, lamp_tensor* const* v,
                     lamp_tensor* const* master, int n, const double* lr, const double* weight_decay, const double* beta1,
                     const double* beta2, double eps, double schedule_factor, int64_t step_count, int debias) {
  LAMP_API_BEGIN
  if (n == 0) return 0;
  check_device_tensor(params[0], "parameter");                 // host parameters: the staging layer runs the step on GPU copies
  const int dev = params[0]->device();
  hipStream_t st = current_stream(dev);
  // group tensors by (state dtype, grad dtype) so that each launch is homogeneous
  for (int pass_state : {kF32, kF64, kBF16, kF16}) {
    for (int pass_grad : {kF32, kF64, kBF16, kF16}) {
      std::vector<int> sel;
      for (int i = 0; i < n; i++) {
        check_device_tensor(params[i], "parameter"); check_device_tensor(grads[i], "gradient");
        const Tensor* state_holder = (master && master[i]) ? master[i] : params[i];
        if (state_holder->dtype == pass_state && grads[i]->dtype == pass_grad) sel.push_back(i);
      }
      if (sel.empty()) continue;
      std::vector<int64_t> numels(sel.size());
      for (size_t k = 0; k < sel.size(); k++) {
        const int i = sel[k];
        const Tensor* holder = (master && master[i]) ? master[i] : params[i];
        numels[k] = holder->numel();
        LAMP_CHECK(grads[i]->numel() == numels[k] && m[i]->numel() == numels[k] && v[i]->numel() == numels[k], "adamw: tensor " << i << " size mismatch");
        LAMP_CHECK(m[i]->dtype == pass_state && v[i]->dtype == pass_state, "adamw: moment buffers of tensor " << i << " must have the dtype of the (working) parameter");
        LAMP_CHECK(holder->is_contiguous() && grads[i]->is_contiguous() && m[i]->is_contiguous() && v[i]->is_contiguous() && params[i]->is_contiguous(), "adamw: tensors must be contiguous");
        if (master && master[i]) LAMP_CHECK(params[i]->dtype == grads[i]->dtype, "adamw mixed precision: model parameter and gradient dtypes differ");
        else LAMP_CHECK(params[i]->dtype == grads[i]->dtype, "adamw: parameter and gradient dtypes differ");
      }
      auto fill = [&](MultiArgs& a, int t, int gi) {
        const int i = sel[gi];
        const bool mixed = master && master[i];
        a.p[0][t] = mixed ? master[i]->data() : params[i]->data();
        a.p[1][t] = grads[i]->data();
        a.p[2][t] = m[i]->data();
        a.p[3][t] = v[i]->data();
        a.p[4][t] = mixed ? params[i]->data() : nullptr;
        const double b1 = beta1[i], b2 = beta2[i];
        const double stepParam = debias ? schedule_factor * lr[i] * std::sqrt(1 - std::pow(b2, (double)step_count)) / (1 - std::pow(b1, (double)step_count))
                                        : schedule_factor * lr[i];
        a.h[0][t] = stepParam;
        a.h[1][t] = stepParam * weight_decay[i];
        a.h[2][t] = b1;
        a.h[3][t] = b2;
        a.s[0] = eps;
      };
#define ADAMW_LAUNCH(T, G) for_each_group((int)sel.size(), numels.data(), fill, [&](MultiArgs& a, int blk) { hipLaunchKernelGGL((adamw_kernel<T, G>), dim3(blk), dim3(256), 0, st, a); })
      if (pass_state == kF32 && pass_grad == kF32) ADAMW_LAUNCH(float, float);
      else if (pass_state == kF64 && pass_grad == kF64) ADAMW_LAUNCH(double, double);
      else if (pass_state == kF32 && pass_grad == kBF16) ADAMW_LAUNCH(float, bf16_t);
      else if (pass_state == kBF16 && pass_grad == kBF16) ADAMW_LAUNCH(bf16_t, bf16_t);
      else if (pass_state == kF32 && pass_grad == kF16) ADAMW_LAUNCH(float, f16_t);      // mixed precision on half parameters (adamw.test.scala:96-127)
      else if (pass_state == kF16 && pass_grad == kF16) ADAMW_LAUNCH(f16_t, f16_t);
      else LAMP_CHECK(false, "adamw: unsupported (state, gradient) dtype pair " << dtype_name(pass_state) << "/" << dtype_name(pass_grad));
#undef ADAMW_LAUNCH
      LAMP_LAUNCH_CHECK();
    }
  }
  igemm_repack_cached(params, n, st);     // the convolution weights' packed images follow the update in one launch (conv_igemm.hip; the narrow ones ride along)
  igemm32_repack_cached(params, n, st);
  small_repack_cached(params, n, st);
  LAMP_API_END
}

int lamp_sgdw_step_(lamp_tensor* const* params, lamp_tensor* const* grads, lamp_tensor* const* velocity, int n, const double* lr,
                    const double* weight_decay, const double* momentum, double schedule_factor) {
  LAMP_API_BEGIN
  if (n == 0) return 0;
  const int dtype = params[0]->dtype, dev = params[0]->device();
  check_list(params, n, "parameters", dtype, dev);
  check_list(grads, n, "gradients", dtype, dev);
  std::vector<int64_t> numels(n);
  for (int i = 0; i < n; i++) {
    numels[i] = params[i]->numel();
    LAMP_CHECK(grads[i]->numel() == numels[i], "sgdw: size mismatch");
    if (velocity && velocity[i]) LAMP_CHECK(velocity[i]->numel() == numels[i] && velocity[i]->dtype == dtype && velocity[i]->is_contiguous(), "sgdw: bad velocity buffer");
  }
  hipStream_t st = current_stream(dev);
  LAMP_DISPATCH_FLOAT(dtype, T, for_each_group(n, numels.data(),
      [&](MultiArgs& a, int t, int gi) {
        a.p[0][t] = params[gi]->data(); a.p[1][t] = grads[gi]->data();
        a.p[2][t] = (velocity && velocity[gi]) ? velocity[gi]->data() : nullptr;
        a.h[0][t] = lr[gi] * schedule_factor; a.h[1][t] = weight_decay[gi] * schedule_factor; a.h[2][t] = momentum ? momentum[gi] : 0.0;
      },
      [&](MultiArgs& a, int blk) { hipLaunchKernelGGL((sgdw_kernel<T>), dim3(blk), dim3(256), 0, st, a); }));
  LAMP_LAUNCH_CHECK();
  igemm_repack_cached(params, n, st);
  igemm32_repack_cached(params, n, st);
  small_repack_cached(params, n, st);
  LAMP_API_END
}

int lamp_flatten_into_(lamp_tensor* bucket, lamp_tensor* const* ts, int n, double scale) {
  LAMP_API_BEGIN
  LAMP_CHECK(bucket, "bucket is null");
  LAMP_CHECK(bucket->dtype == kF32 && bucket->is_contiguous(), "the gradient bucket must be a contiguous f32 tensor");
  if (n == 0) return 0;
  std::vector<int64_t> numels(n), offs(n);
  int64_t off = 0;
  for (int i = 0; i < n; i++) { LAMP_CHECK(ts[i], "null tensor in the list"); numels[i] = ts[i]->numel(); offs[i] = off; off += numels[i]; }
  LAMP_CHECK(off <= bucket->numel(), "bucket too small: " << bucket->numel() << " < " << off);
  if (all_host(bucket, ts, n)) {
    LAMP_DISPATCH_FLOAT(ts[0]->dtype, T, host_flatten<T>(bucket->ptr<float>(), ts, n, offs, scale));
    return 0;
  }
  check_device_tensor(bucket, "bucket");
  const int dtype = ts[0]->dtype, dev = bucket->device();
  check_list(ts, n, "tensors", dtype, dev);
  hipStream_t st = current_stream(dev);
  LAMP_DISPATCH_FLOAT(dtype, T, for_each_group(n, numels.data(),
      [&](MultiArgs& a, int t, int gi) { a.p[0][t] = ts[gi]->data(); a.h[0][t] = (double)offs[gi]; a.s[0] = scale; },
      [&](MultiArgs& a, int blk) { hipLaunchKernelGGL((mt_flatten_kernel<T>), dim3(blk), dim3(256), 0, st, a, bucket->ptr<float>()); }));
  LAMP_LAUNCH_CHECK();
  LAMP_API_END
}

int lamp_unflatten_from_(lamp_tensor* const* ts, int n, const lamp_tensor* bucket, int divide_by_last_element) {
  LAMP_API_BEGIN
  LAMP_CHECK(bucket, "bucket is null");
  LAMP_CHECK(bucket->dtype == kF32 && bucket->is_contiguous(), "the gradient bucket must be a contiguous f32 tensor");
  if (n == 0) return 0;
  std::vector<int64_t> numels(n), offs(n);
  int64_t off = 0;
  for (int i = 0; i < n; i++) { LAMP_CHECK(ts[i], "null tensor in the list"); numels[i] = ts[i]->numel(); offs[i] = off; off += numels[i]; }
  LAMP_CHECK(off <= bucket->numel(), "bucket too small");
  const int64_t last = divide_by_last_element ? bucket->numel() - 1 : -1;
  if (all_host(bucket, ts, n)) {
    LAMP_DISPATCH_FLOAT(ts[0]->dtype, T, host_unflatten<T>(ts, n, bucket->ptr<float>(), offs, last));
    return 0;
  }
  check_device_tensor(bucket, "bucket");
  const int dtype = ts[0]->dtype, dev = bucket->device();
  check_list(ts, n, "tensors", dtype, dev);
  hipStream_t st = current_stream(dev);
  LAMP_DISPATCH_FLOAT(dtype, T, for_each_group(n, numels.data(),
      [&](MultiArgs& a, int t, int gi) { a.p[0][t] = ts[gi]->data(); a.h[0][t] = (double)offs[gi]; },
      [&](MultiArgs& a, int blk) { hipLaunchKernelGGL((mt_unflatten_kernel<T>), dim3(blk), dim3(256), 0, st, a, bucket->ptr<float>(), last); }));
  LAMP_LAUNCH_CHECK();
  LAMP_API_END
}

}  // extern "C"
