// 2-D pooling, forward and backward.
//
// Replaces ATen.avg_pool2d(+_backward) (ceil_mode=false, count_include_pad=true, no divisor
// override - the only configuration lamp passes) and ATen.max_pool2d_with_indices(+_backward)
// (reference: lamp-core/src/main/scala/lamp/autograd/ops.scala:1721-1768 MaxPool2D, 1775-1825 AvgPool2D).
// max-pool indices are int64 flat offsets h*W + w inside each (n, c) plane, first maximum in
// row-major window order wins, NaN propagates - bit-exact with ATen.
#include "device_utils.h"
#include "../core/strided.h"

namespace lamp {

struct PoolGeom {
  int64_t NC, H, W, Ho, Wo;
  int k, s, p, d;
  int count_include_pad;
};

static int64_t pool_out(int64_t in, int k, int s, int p, int d, int ceil_mode) {
  int64_t num = in + 2 * p - d * (k - 1) - 1 + (ceil_mode ? s - 1 : 0);
  int64_t o = num / s + 1;
  if (ceil_mode && (o - 1) * s >= in + p) o--;
  return o;
}

template <class T>
__global__ void avg_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolGeom g) {
  using A = acc_t<T>;
  const int64_t total = g.NC * g.Ho * g.Wo;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int wo = (int)(e % g.Wo), ho = (int)((e / g.Wo) % g.Ho);
    const int64_t nc = e / (g.Wo * g.Ho);
    int hs = ho * g.s - g.p, ws = wo * g.s - g.p;
    int he = min(hs + g.k, (int)g.H + g.p), we = min(ws + g.k, (int)g.W + g.p);
    const int pool = (he - hs) * (we - ws);
    hs = max(hs, 0); ws = max(ws, 0); he = min(he, (int)g.H); we = min(we, (int)g.W);
    A acc = 0;
    const T* xp = x + nc * g.H * g.W;
    for (int h = hs; h < he; h++)
      for (int w = ws; w < we; w++) acc += load_as<A>(xp[h * g.W + w]);
    const int div = g.count_include_pad ? pool : (he - hs) * (we - ws);
    y[e] = store_as<T>((A)(acc / (A)div));
  }
}
template <class T>
__global__ void avg_pool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, PoolGeom g) {
  using A = acc_t<T>;
  const int64_t total = g.NC * g.H * g.W;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(e % g.W), h = (int)((e / g.W) % g.H);
    const int64_t nc = e / (g.W * g.H);
    A acc = 0;
    // windows [ho*s - p, ho*s - p + k) containing h
    int ho_lo = (h + g.p - g.k + 1 + g.s - 1) / g.s; if (h + g.p - g.k + 1 < 0) ho_lo = 0;
    int wo_lo = (w + g.p - g.k + 1 + g.s - 1) / g.s; if (w + g.p - g.k + 1 < 0) wo_lo = 0;
    const int ho_hi = min((h + g.p) / g.s, (int)g.Ho - 1), wo_hi = min((w + g.p) / g.s, (int)g.Wo - 1);
    for (int ho = ho_lo; ho <= ho_hi; ho++)
      for (int wo = wo_lo; wo <= wo_hi; wo++) {
        int hs = ho * g.s - g.p, ws = wo * g.s - g.p;
        int he = min(hs + g.k, (int)g.H + g.p), we = min(ws + g.k, (int)g.W + g.p);
        int div = (he - hs) * (we - ws);
        if (!g.count_include_pad) { div = (min(he, (int)g.H) - max(hs, 0)) * (min(we, (int)g.W) - max(ws, 0)); }
        acc += load_as<A>(dy[(nc * g.Ho + ho) * g.Wo + wo]) / (A)div;
      }
    dx[e] = store_as<T>(acc);
  }
}
template <class T>
__global__ void max_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t* __restrict__ idx, PoolGeom g) {
  using A = acc_t<T>;
  const int64_t total = g.NC * g.Ho * g.Wo;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int wo = (int)(e % g.Wo), ho = (int)((e / g.Wo) % g.Ho);
    const int64_t nc = e / (g.Wo * g.Ho);
    int hs = ho * g.s - g.p, ws = wo * g.s - g.p;
    const int he = min(hs + (g.k - 1) * g.d + 1, (int)g.H), we = min(ws + (g.k - 1) * g.d + 1, (int)g.W);
    while (hs < 0) hs += g.d;
    while (ws < 0) ws += g.d;
    const T* xp = x + nc * g.H * g.W;
    A best = -INFINITY;
    int64_t bi = (int64_t)hs * g.W + ws;
    for (int h = hs; h < he; h += g.d)
      for (int w = ws; w < we; w += g.d) {
        const A v = load_as<A>(xp[h * g.W + w]);
        if (v > best || v != v) { best = v; bi = (int64_t)h * g.W + w; }
      }
    y[e] = store_as<T>(best);
    idx[e] = bi;
  }
}
// gather form (deterministic, no atomics): input element (h, w) collects every window whose argmax it is
template <class T>
__global__ void max_pool_bwd_kernel(const T* __restrict__ dy, const int64_t* __restrict__ idx, T* __restrict__ dx, PoolGeom g) {
  using A = acc_t<T>;
  const int64_t total = g.NC * g.H * g.W;
  const int span = (g.k - 1) * g.d;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(e % g.W), h = (int)((e / g.W) % g.H);
    const int64_t nc = e / (g.W * g.H);
    const int64_t me = (int64_t)h * g.W + w;
    int ho_lo = (h + g.p - span + g.s - 1) / g.s; if (h + g.p - span < 0) ho_lo = 0;
    int wo_lo = (w + g.p - span + g.s - 1) / g.s; if (w + g.p - span < 0) wo_lo = 0;
    const int ho_hi = min((h + g.p) / g.s, (int)g.Ho - 1), wo_hi = min((w + g.p) / g.s, (int)g.Wo - 1);
    A acc = 0;
    for (int ho = ho_lo; ho <= ho_hi; ho++)
      for (int wo = wo_lo; wo <= wo_hi; wo++) {
        const int64_t o = (nc * g.Ho + ho) * g.Wo + wo;
        if (idx[o] == me) acc += load_as<A>(dy[o]);
      }
    dx[e] = store_as<T>(acc);
  }
}

// global average pooling (window == whole plane, the tail of Cnn.resnet: AvgPool2D(k = 8) on 8x8 maps):
// one wavefront per (n, c) plane, coalesced reads, shuffle reduction; backward is a broadcast
template <class T>
__global__ __launch_bounds__(256) void avg_pool_global_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int hw) {
  using A = acc_t<T>;
  const int lane = threadIdx.x & 63;
  const int64_t pl = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (pl >= planes) return;
  A s = 0;
  for (int i = lane; i < hw; i += 64) s += load_as<A>(x[pl * hw + i]);
  s = wave_sum(s);
  if (lane == 0) y[pl] = store_as<T>((A)(s / (A)hw));
}
template <class T>
__global__ void avg_pool_global_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int64_t total, int hw) {
  using A = acc_t<T>;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    dx[e] = store_as<T>((A)(load_as<A>(dy[e / hw]) / (A)hw));
}

// 16-byte packet variants for planes whose byte size is a power-of-two multiple of 16 (8x8 bf16 maps: 8 lanes per plane):
// the scalar kernels above move 2 bytes per lane and run at ~1 TB/s
template <class T>
__global__ __launch_bounds__(256) void avg_pool_global_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int hw, int lpp) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t pl = t / lpp;
  A s = 0;
  if (pl < planes) {
    const Vec<T, W> pk = *reinterpret_cast<const Vec<T, W>*>(x + t * W);
#pragma unroll
    for (int k = 0; k < W; k++) s += load_as<A>(pk.v[k]);
  }
  for (int off = lpp >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (pl < planes && (t % lpp) == 0) y[pl] = store_as<T>((A)(s / (A)hw));
}
template <class T>
__global__ __launch_bounds__(256) void avg_pool_global_bwd_vec_kernel(const T* __restrict__ dy, T* __restrict__ dx, int64_t packets, int hw, int lpp) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < packets; t += (int64_t)gridDim.x * blockDim.x) {
    const T v = store_as<T>((A)(load_as<A>(dy[t / lpp]) / (A)hw));
    Vec<T, W> pk;
#pragma unroll
    for (int k = 0; k < W; k++) pk.v[k] = v;
    *reinterpret_cast<Vec<T, W>*>(dx + t * W) = pk;
  }
}
static int global_pool_lanes(const Tensor* x, int hw) {   // lanes per plane for the packet kernels, 0 = use the scalar kernels
  const int64_t bytes = (int64_t)hw * x->itemsize();
  if (bytes % 16 != 0) return 0;
  const int64_t lpp = bytes / 16;
  if (lpp < 1 || lpp > 64 || (lpp & (lpp - 1)) != 0) return 0;
  return (int)lpp;
}

static PoolGeom pool_geom(const Tensor* x, int64_t k, int64_t s, int64_t p, int64_t d, int ceil_mode, int cip) {
  LAMP_CHECK(x->ndim == 4 || x->ndim == 3, "pooling expects a 3-D or 4-D input, got " << x->describe());
  LAMP_CHECK(k > 0 && s > 0 && p >= 0 && d > 0 && p <= k / 2 + (k == 1 ? 0 : 0) + k, "bad pooling geometry");
  PoolGeom g{};
  const int nd = x->ndim;
  g.H = x->sizes[nd - 2]; g.W = x->sizes[nd - 1];
  g.NC = x->numel() / std::max<int64_t>(g.H * g.W, 1);
  g.k = (int)k; g.s = (int)s; g.p = (int)p; g.d = (int)d;
  g.count_include_pad = cip;
  g.Ho = pool_out(g.H, g.k, g.s, g.p, g.d, ceil_mode);
  g.Wo = pool_out(g.W, g.k, g.s, g.p, g.d, ceil_mode);
  LAMP_CHECK(g.Ho > 0 && g.Wo > 0, "pooling output would be empty");
  return g;
}
static std::vector<int64_t> pooled_shape(const Tensor* x, const PoolGeom& g) {
  std::vector<int64_t> s = x->shape();
  s[s.size() - 2] = g.Ho; s[s.size() - 1] = g.Wo;
  return s;
}

// ---- 1-D max pooling over [N, C, L] (ATen max_pool1d_with_indices; MaxPool1D op, ops.scala:1658-1715) ------------------------------
// indices are positions along L (what the reference's backward index_adds with); ties and NaN as ATen: the first maximum, NaN wins
struct Pool1Geom { int64_t NC, L, Lo; int k, s, p, d; };
template <class T>
__global__ void max_pool1d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t* __restrict__ idx, Pool1Geom g) {
  using A = acc_t<T>;
  const int64_t total = g.NC * g.Lo;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lo = e % g.Lo, nc = e / g.Lo;
    const T* xp = x + nc * g.L;
    int64_t best = -1;
    A bv = 0;
    for (int j = 0; j < g.k; j++) {
      const int64_t l = lo * g.s - g.p + (int64_t)j * g.d;
      if (l < 0 || l >= g.L) continue;
      const A v = load_as<A>(xp[l]);
      if (best < 0 || v > bv || v != v) { if (!(best >= 0 && bv != bv)) { bv = v; best = l; } }
    }
    y[e] = store_as<T>(bv);
    idx[e] = best < 0 ? 0 : best;
  }
}
// gather form (deterministic): input position l collects the gradients of every window whose maximum it was
template <class T>
__global__ void max_pool1d_bwd_kernel(const T* __restrict__ dy, const int64_t* __restrict__ idx, T* __restrict__ dx, Pool1Geom g) {
  using A = acc_t<T>;
  const int64_t total = g.NC * g.L;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = e % g.L, nc = e / g.L;
    A acc = 0;
    // windows lo with lo * s - p <= l <= lo * s - p + (k - 1) d
    int64_t lo_min = (l + g.p - (int64_t)(g.k - 1) * g.d + g.s - 1) / g.s;
    if (l + g.p - (int64_t)(g.k - 1) * g.d < 0) lo_min = 0;
    const int64_t lo_max = min((l + g.p) / g.s, g.Lo - 1);
    for (int64_t lo = lo_min; lo <= lo_max; lo++)
      if (idx[nc * g.Lo + lo] == l) acc += load_as<A>(dy[nc * g.Lo + lo]);
    dx[e] = store_as<T>(acc);
  }
}
static Pool1Geom pool1_geom(const Tensor* x, int64_t kernel, int64_t stride, int64_t padding, int64_t dilation, int ceil_mode) {
  LAMP_CHECK(x->ndim == 3, "assertion failed: Input dimensions must be 3 (MaxPool1D), got " << x->describe());
  LAMP_CHECK(kernel >= 1 && stride >= 1 && padding >= 0 && dilation >= 1 && padding * 2 <= kernel, "max_pool1d: bad kernel / stride / padding / dilation");
  Pool1Geom g;
  g.NC = x->sizes[0] * x->sizes[1]; g.L = x->sizes[2];
  g.k = (int)kernel; g.s = (int)stride; g.p = (int)padding; g.d = (int)dilation;
  g.Lo = pool_out(g.L, g.k, g.s, g.p, g.d, ceil_mode);
  LAMP_CHECK(g.Lo >= 1, "max_pool1d: output would be empty");
  return g;
}

}  // namespace lamp

using namespace lamp;

extern "C" {

int lamp_avg_pool2d(lamp_tensor** out, const lamp_tensor* x, int64_t kernel, int64_t stride, int64_t padding, int ceil_mode,
                    int count_include_pad) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input");
  PoolGeom g = pool_geom(x, kernel, stride, padding, 1, ceil_mode, count_include_pad);
  Hold xc(contiguous(x));
  Hold y(new_tensor(pooled_shape(x, g), x->dtype, x->device()));
  const int64_t total = y->numel();
  const bool global = (g.k == g.H && g.k == g.W && g.p == 0 && g.Ho == 1 && g.Wo == 1);
  if (total && global) {
    const int lpp = (((uintptr_t)xc->data() & 15) == 0) ? global_pool_lanes(xc.get(), (int)(g.H * g.W)) : 0;
    if (lpp) {
      LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((avg_pool_global_fwd_vec_kernel<T>), dim3((unsigned)((g.NC * lpp + 255) / 256)), dim3(256), 0,
                                                          current_stream(x->device()), xc->ptr<T>(), y->ptr<T>(), g.NC, (int)(g.H * g.W), lpp));
    } else {
      LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((avg_pool_global_fwd_kernel<T>), dim3((unsigned)((g.NC * 64 + 255) / 256)), dim3(256), 0,
                                                          current_stream(x->device()), xc->ptr<T>(), y->ptr<T>(), g.NC, (int)(g.H * g.W)));
    }
    LAMP_LAUNCH_CHECK();
  } else if (total) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((avg_pool_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                        current_stream(x->device()), xc->ptr<T>(), y->ptr<T>(), g));
    LAMP_LAUNCH_CHECK();
  }
  *out = y.take();
  LAMP_API_END
}
int lamp_avg_pool2d_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, int64_t kernel, int64_t stride,
                             int64_t padding, int ceil_mode, int count_include_pad) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(grad_out, "grad_out");
  PoolGeom g = pool_geom(x, kernel, stride, padding, 1, ceil_mode, count_include_pad);
  LAMP_CHECK(grad_out->shape() == pooled_shape(x, g) && grad_out->dtype == x->dtype, "avg_pool2d_backward: grad_out " << grad_out->describe() << " has the wrong shape");
  Hold gc(contiguous(grad_out));
  Hold dx(new_tensor(x->shape(), x->dtype, x->device()));
  const int64_t total = dx->numel();
  const bool global = (g.k == g.H && g.k == g.W && g.p == 0 && g.Ho == 1 && g.Wo == 1);
  if (total && global) {
    const int lpp = (((uintptr_t)dx->data() & 15) == 0) ? global_pool_lanes(dx.get(), (int)(g.H * g.W)) : 0;
    if (lpp) {
      const int64_t packets = g.NC * lpp;
      LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((avg_pool_global_bwd_vec_kernel<T>), dim3(grid_for(packets, 256)), dim3(256), 0,
                                                          current_stream(x->device()), gc->ptr<T>(), dx->ptr<T>(), packets, (int)(g.H * g.W), lpp));
    } else {
      LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((avg_pool_global_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                          current_stream(x->device()), gc->ptr<T>(), dx->ptr<T>(), total, (int)(g.H * g.W)));
    }
    LAMP_LAUNCH_CHECK();
  } else if (total) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((avg_pool_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                        current_stream(x->device()), gc->ptr<T>(), dx->ptr<T>(), g));
    LAMP_LAUNCH_CHECK();
  }
  *out = dx.take();
  LAMP_API_END
}
int lamp_max_pool2d_with_indices(lamp_tensor** out, lamp_tensor** indices, const lamp_tensor* x, int64_t kernel, int64_t stride,
                                 int64_t padding, int64_t dilation, int ceil_mode) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input");
  PoolGeom g = pool_geom(x, kernel, stride, padding, dilation, ceil_mode, 1);
  Hold xc(contiguous(x));
  auto os = pooled_shape(x, g);
  Hold y(new_tensor(os, x->dtype, x->device())), idx(new_tensor(os, kI64, x->device()));
  const int64_t total = y->numel();
  if (total) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((max_pool_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                        current_stream(x->device()), xc->ptr<T>(), y->ptr<T>(), idx->ptr<int64_t>(), g));
    LAMP_LAUNCH_CHECK();
  }
  *out = y.take();
  *indices = idx.take();
  LAMP_API_END
}
int lamp_max_pool2d_with_indices_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, int64_t kernel,
                                          int64_t stride, int64_t padding, int64_t dilation, int ceil_mode, const lamp_tensor* indices) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(grad_out, "grad_out"); check_device_tensor(indices, "indices");
  PoolGeom g = pool_geom(x, kernel, stride, padding, dilation, ceil_mode, 1);
  LAMP_CHECK(grad_out->shape() == pooled_shape(x, g) && indices->shape() == grad_out->shape() && indices->dtype == kI64 &&
             grad_out->dtype == x->dtype, "max_pool2d_with_indices_backward: shape/dtype mismatch");
  Hold gc(contiguous(grad_out)), ic(contiguous(indices));
  Hold dx(new_tensor(x->shape(), x->dtype, x->device()));
  const int64_t total = dx->numel();
  if (total) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((max_pool_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                        current_stream(x->device()), gc->ptr<T>(), ic->ptr<int64_t>(), dx->ptr<T>(), g));
    LAMP_LAUNCH_CHECK();
  }
  *out = dx.take();
  LAMP_API_END
}

int lamp_max_pool1d_with_indices(lamp_tensor** out, lamp_tensor** indices, const lamp_tensor* x, int64_t kernel, int64_t stride, int64_t padding,
                                 int64_t dilation, int ceil_mode) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input");
  const Pool1Geom g = pool1_geom(x, kernel, stride, padding, dilation, ceil_mode);
  Hold xc(contiguous(x));
  int64_t os[3] = {x->sizes[0], x->sizes[1], g.Lo};
  Hold y(new_tensor(os, 3, x->dtype, x->device())), idx(new_tensor(os, 3, kI64, x->device()));
  const int64_t total = y->numel();
  if (total) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((max_pool1d_fwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, current_stream(x->device()),
                                                        xc->ptr<T>(), y->ptr<T>(), idx->ptr<int64_t>(), g));
    LAMP_LAUNCH_CHECK();
  }
  *out = y.take();
  *indices = idx.take();
  LAMP_API_END
}
int lamp_max_pool1d_with_indices_backward(lamp_tensor** out, const lamp_tensor* grad_out, const lamp_tensor* x, int64_t kernel, int64_t stride,
                                          int64_t padding, int64_t dilation, int ceil_mode, const lamp_tensor* indices) {
  LAMP_API_BEGIN
  check_device_tensor(x, "input"); check_device_tensor(grad_out, "grad_out"); check_device_tensor(indices, "indices");
  const Pool1Geom g = pool1_geom(x, kernel, stride, padding, dilation, ceil_mode);
  LAMP_CHECK(grad_out->ndim == 3 && grad_out->sizes[2] == g.Lo && indices->shape() == grad_out->shape() && indices->dtype == kI64 && grad_out->dtype == x->dtype,
             "max_pool1d_with_indices_backward: shape/dtype mismatch");
  Hold gc(contiguous(grad_out)), ic(contiguous(indices));
  Hold dx(new_tensor(x->shape(), x->dtype, x->device()));
  const int64_t total = dx->numel();
  if (total) {
    LAMP_DISPATCH_FLOAT(x->dtype, T, hipLaunchKernelGGL((max_pool1d_bwd_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, current_stream(x->device()),
                                                        gc->ptr<T>(), ic->ptr<int64_t>(), dx->ptr<T>(), g));
    LAMP_LAUNCH_CHECK();
  }
  *out = dx.take();
  LAMP_API_END
}

}  // extern "C"
