// Reductions: sum / mean / L2 norm / var+mean / max / min / argmax over arbitrary dim sets, and
// lamp's `unbroadcast` (TensorHelpers.scala:7-41).  HBM-bound: one read of the input, f32
// accumulation for bf16/f32 (f64 for f64), 64-lane shuffle reductions, two-stage (partials +
// finalize) when the output is too small to fill 256 CUs.
//
// Reference call sites: lamp-sten/src/main/scala/lamp/STen.scala:1336-1352 (sum), 1493-1501
// (norm2), 1524-1540 (mean/variance), 1565-1585 (max/min), 987 (argmax);
// lamp-core/src/main/scala/lamp/autograd/ops.scala:623-645, 1034-1078 (Sum/Norm2/Mean/Variance ops).
#include "device_utils.h"
#include "../core/strided.h"

namespace lamp {

enum ReduceOp { kSum = 0, kSumSq = 1, kMax = 2, kMin = 3 };

template <int OP, class A> __device__ __forceinline__ A r_init() {
  if (OP == kMax) return -INFINITY;
  if (OP == kMin) return INFINITY;
  return A(0);
}
template <int OP, class A> __device__ __forceinline__ A r_elem(A x) { return OP == kSumSq ? x * x : x; }
template <int OP, class A> __device__ __forceinline__ A r_comb(A a, A b) {
  if (OP == kMax) return (b > a || b != b) ? b : a;
  if (OP == kMin) return (b < a || b != b) ? b : a;
  return a + b;
}
template <int OP, class A> __device__ __forceinline__ A wave_red(A v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = r_comb<OP, A>(v, __shfl_xor(v, off, 64));
  return v;
}

// Geometry: input viewed as [K0, R1, K1, R2] (contiguous), output [K0, K1]; R1 and R2 are reduced.
struct RGeom {
  int64_t K0, R1, K1, R2;
  int64_t nsplit;      // split of the flattened (R1*R2) range over blockIdx.y
};

// block-per-output kernel: good when R2 (the contiguous reduced run) is large or R1*R2 is large
template <class T, int OP>
__global__ __launch_bounds__(256) void reduce_block_kernel(const T* __restrict__ in, acc_t<T>* __restrict__ partial, RGeom g) {
  using A = acc_t<T>;
  __shared__ A smem[4];
  const int64_t o = blockIdx.x;  // output index in [0, K0*K1)
  const int64_t k0 = o / g.K1, k1 = o - k0 * g.K1;
  const int64_t total = g.R1 * g.R2;
  const int64_t chunk = (total + g.nsplit - 1) / g.nsplit;
  const int64_t begin = blockIdx.y * chunk;
  const int64_t end = begin + chunk < total ? begin + chunk : total;
  A acc = r_init<OP, A>();
  const T* base = in + (k0 * g.R1 * g.K1 + k1) * g.R2;
  if (g.R1 == 1 || g.K1 == 1) {
    // one contiguous run of `total` elements
    for (int64_t i = begin + threadIdx.x; i < end; i += blockDim.x) acc = r_comb<OP, A>(acc, r_elem<OP, A>(load_as<A>(base[i])));
  } else {
    const int64_t rowstride = g.K1 * g.R2;
    for (int64_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
      int64_t r1 = i / g.R2, r2 = i - r1 * g.R2;
      acc = r_comb<OP, A>(acc, r_elem<OP, A>(load_as<A>(base[r1 * rowstride + r2])));
    }
  }
  acc = wave_red<OP, A>(acc);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    A r = smem[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) r = r_comb<OP, A>(r, smem[w]);
    partial[blockIdx.y * (g.K0 * g.K1) + o] = r;
  }
}

// thread-per-output kernel: R2 == 1, threads run along K1 (coalesced), loop over R1
template <class T, int OP>
__global__ __launch_bounds__(256) void reduce_column_kernel(const T* __restrict__ in, acc_t<T>* __restrict__ partial, RGeom g) {
  using A = acc_t<T>;
  const int64_t nout = g.K0 * g.K1;
  const int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (o >= nout) return;
  const int64_t k0 = o / g.K1, k1 = o - k0 * g.K1;
  const int64_t chunk = (g.R1 + g.nsplit - 1) / g.nsplit;
  const int64_t begin = blockIdx.y * chunk;
  const int64_t end = begin + chunk < g.R1 ? begin + chunk : g.R1;
  const T* base = in + k0 * g.R1 * g.K1 + k1;
  A acc = r_init<OP, A>();
  for (int64_t r = begin; r < end; r++) acc = r_comb<OP, A>(acc, r_elem<OP, A>(load_as<A>(base[r * g.K1])));
  partial[blockIdx.y * nout + o] = acc;
}

// column reduction of a dense [K0, R1, K1] input in 16-byte packets: a workgroup covers 32 packets of columns x 8 row lanes, every
// row lane walks its share of the row chunk, the lanes are combined through LDS in a fixed order (run-to-run identical)
template <class T, int OP>
__global__ __launch_bounds__(256) void reduce_column_vec_kernel(const T* __restrict__ in, acc_t<T>* __restrict__ partial, RGeom g) {
  using A = acc_t<T>;
  constexpr int W = 16 / sizeof(T);
  __shared__ A sm[8][32 * W + 1];
  const int cp = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t ppr = g.K1 / W;                                  // packets per row
  const int64_t pk = (int64_t)blockIdx.x * 32 + cp;
  const int64_t k0 = blockIdx.z;
  const int64_t chunk = (g.R1 + g.nsplit - 1) / g.nsplit;
  const int64_t begin = blockIdx.y * chunk;
  const int64_t end = begin + chunk < g.R1 ? begin + chunk : g.R1;
  A acc[W];
#pragma unroll
  for (int k = 0; k < W; k++) acc[k] = r_init<OP, A>();
  if (pk < ppr) {
    const T* base = in + k0 * g.R1 * g.K1 + pk * W;
    int64_t r = begin + rl;
    for (; r + 24 < end; r += 32) {                               // four independent loads in flight
      Vec<T, W> v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const Vec<T, W>*>(base + (r + 8 * u) * g.K1);
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int k = 0; k < W; k++) acc[k] = r_comb<OP, A>(acc[k], r_elem<OP, A>(load_as<A>(v[u].v[k])));
    }
    for (; r < end; r += 8) {
      const Vec<T, W> v = *reinterpret_cast<const Vec<T, W>*>(base + r * g.K1);
#pragma unroll
      for (int k = 0; k < W; k++) acc[k] = r_comb<OP, A>(acc[k], r_elem<OP, A>(load_as<A>(v.v[k])));
    }
  }
#pragma unroll
  for (int k = 0; k < W; k++) sm[rl][cp * W + k] = acc[k];
  __syncthreads();
  const int64_t nout = g.K0 * g.K1;
  for (int c = threadIdx.x; c < 32 * W; c += 256) {
    const int64_t col = (int64_t)blockIdx.x * 32 * W + c;
    if (col < g.K1) {
      A a = sm[0][c];
#pragma unroll
      for (int l = 1; l < 8; l++) a = r_comb<OP, A>(a, sm[l][c]);
      partial[blockIdx.y * nout + k0 * g.K1 + col] = a;
    }
  }
}

// finalize of many partials: 16 outputs x 16 split lanes per workgroup, lanes combined through LDS in a fixed order
template <class T, int OP>
__global__ __launch_bounds__(256) void reduce_finalize_wide_kernel(const acc_t<T>* __restrict__ partial, T* __restrict__ out, int64_t nout, int64_t nsplit,
                                                                   double scale, int do_sqrt) {
  using A = acc_t<T>;
  __shared__ A sm[16][17];
  const int oc = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t o = (int64_t)blockIdx.x * 16 + oc;
  A r = r_init<OP, A>();
  if (o < nout) {
#pragma unroll 4
    for (int64_t s = sl; s < nsplit; s += 16) r = r_comb<OP, A>(r, partial[s * nout + o]);
  }
  sm[sl][oc] = r;
  __syncthreads();
  if (sl == 0 && o < nout) {
    r = sm[0][oc];
#pragma unroll
    for (int l = 1; l < 16; l++) r = r_comb<OP, A>(r, sm[l][oc]);
    if (OP == kSum || OP == kSumSq) r = (A)(r * (A)scale);
    if (do_sqrt) r = (A)sqrt((double)r);
    out[o] = store_as<T>(r);
  }
}

// finalize: combine nsplit partials, apply scale / sqrt, cast
template <class T, int OP>
__global__ void reduce_finalize_kernel(const acc_t<T>* __restrict__ partial, T* __restrict__ out, int64_t nout, int64_t nsplit,
                                       double scale, int do_sqrt) {
  using A = acc_t<T>;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < nout; o += (int64_t)gridDim.x * blockDim.x) {
    A r = partial[o];
    for (int64_t s = 1; s < nsplit; s++) r = r_comb<OP, A>(r, partial[s * nout + o]);
    if (OP == kSum || OP == kSumSq) r = (A)(r * (A)scale);
    if (do_sqrt) r = (A)sqrt((double)r);
    out[o] = store_as<T>(r);
  }
}

// generic fallback: thread per output, strided walk of the reduced index space
template <class T, int OP>
__global__ void reduce_generic_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t nout, int64_t nred, IterArgs keep,
                                      IterArgs red, double scale, int do_sqrt) {
  using A = acc_t<T>;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < nout; o += (int64_t)gridDim.x * blockDim.x) {
    int64_t ko[1];
    iter_offsets<1>(keep, o, ko);
    A acc = r_init<OP, A>();
    for (int64_t r = 0; r < nred; r++) {
      int64_t ro[1];
      iter_offsets<1>(red, r, ro);
      acc = r_comb<OP, A>(acc, r_elem<OP, A>(load_as<A>(in[ko[0] + ro[0]])));
    }
    if (OP == kSum || OP == kSumSq) acc = (A)(acc * (A)scale);
    if (do_sqrt) acc = (A)sqrt((double)acc);
    out[o] = store_as<T>(acc);
  }
}

struct DimPlan {
  std::vector<int64_t> out_keep;   // output shape with keepdim
  std::vector<int64_t> out_nokeep;
  std::vector<bool> reduced;
  int64_t nred = 1, nout = 1;
};
static DimPlan plan_dims(const Tensor* a, const int64_t* dims, int ndims) {
  DimPlan p;
  p.reduced.assign(a->ndim, false);
  if (ndims == 0) for (int i = 0; i < a->ndim; i++) p.reduced[i] = true;
  for (int i = 0; i < ndims; i++) {
    int64_t d = wrap_dim(dims[i], a->ndim);
    if (a->ndim > 0) p.reduced[d] = true;
  }
  for (int i = 0; i < a->ndim; i++) {
    if (p.reduced[i]) { p.out_keep.push_back(1); p.nred *= a->sizes[i]; }
    else { p.out_keep.push_back(a->sizes[i]); p.out_nokeep.push_back(a->sizes[i]); p.nout *= a->sizes[i]; }
  }
  return p;
}

template <class T, int OP>
static void reduce_typed(const Tensor* ac, Tensor* out, const DimPlan& p, double scale, int do_sqrt) {
  hipStream_t st = current_stream(ac->device());
  // collapse to alternating groups
  std::vector<std::pair<bool, int64_t>> groups;  // (reduced, size)
  for (int i = 0; i < ac->ndim; i++) {
    if (ac->sizes[i] == 1) continue;
    if (!groups.empty() && groups.back().first == p.reduced[i]) groups.back().second *= ac->sizes[i];
    else groups.push_back({(bool)p.reduced[i], ac->sizes[i]});
  }
  RGeom g{1, 1, 1, 1, 1};
  bool fits = true;
  {
    // map onto [K0, R1, K1, R2]
    size_t i = 0;
    if (i < groups.size() && !groups[i].first) g.K0 = groups[i++].second;
    if (i < groups.size() && groups[i].first) g.R1 = groups[i++].second;
    if (i < groups.size() && !groups[i].first) g.K1 = groups[i++].second;
    if (i < groups.size() && groups[i].first) g.R2 = groups[i++].second;
    if (i != groups.size()) fits = false;
    // [K0, R1] alone (suffix reduce) is better expressed as K1=K0', R2=R1
    if (fits && g.K1 == 1 && g.R2 == 1) { g.R2 = g.R1; g.R1 = 1; g.K1 = g.K0; g.K0 = 1; }
  }
  const int64_t nout = p.nout;
  if (p.nred == 0 || nout == 0) { if (nout) fill_zero(out); return; }
  if (!fits) {
    // keep / reduced stride tables over the contiguous input
    std::vector<int64_t> ksz, kst, rsz, rst;
    for (int i = 0; i < ac->ndim; i++) (p.reduced[i] ? rsz : ksz).push_back(ac->sizes[i]), (p.reduced[i] ? rst : kst).push_back(ac->strides[i]);
    auto mk = [](const std::vector<int64_t>& sz, const std::vector<int64_t>& stv) {
      IterArgs a;
      a.ndim = std::max<int>((int)sz.size(), 1);
      for (int d = 0; d < kMaxDims; d++) { a.sizes[d] = d < (int)sz.size() ? sz[d] : 1; a.strides[0][d] = d < (int)stv.size() ? stv[d] : 0; }
      return a;
    };
    hipLaunchKernelGGL((reduce_generic_kernel<T, OP>), dim3(grid_for(nout, 256)), dim3(256), 0, st, ac->ptr<T>(), out->ptr<T>(),
                       nout, p.nred, mk(ksz, kst), mk(rsz, rst), scale, do_sqrt);
    LAMP_LAUNCH_CHECK();
    return;
  }
  using A = acc_t<T>;
  const int64_t target_blocks = (int64_t)num_cus() * 4;
  bool column = (g.R2 == 1 && g.K1 >= 64);
  int64_t nsplit = 1;
  if (column) {
    int64_t blocks = (nout + 255) / 256;
    if (sizeof(T) <= 8 && g.K1 % (16 / sizeof(T)) == 0 && g.R1 >= 64) blocks = ((g.K1 / (16 / sizeof(T)) + 31) / 32) * g.K0;
    if (blocks < target_blocks && g.R1 >= 256) nsplit = std::min<int64_t>(std::min<int64_t>(target_blocks / blocks, g.R1 / 64), 256);
  } else {
    int64_t total = g.R1 * g.R2;
    if (nout < target_blocks && total >= 8192) nsplit = std::min<int64_t>(std::min<int64_t>(target_blocks / nout, total / 2048), 1024);
  }
  if (nsplit < 1) nsplit = 1;
  g.nsplit = nsplit;
  int64_t psz[1] = {nsplit * nout};
  Hold partial(new_tensor(psz, 1, std::is_same<A, double>::value ? kF64 : (std::is_same<A, float>::value ? kF32 : kI64), ac->device()));
  constexpr int WV = 16 / sizeof(T);
  const bool column_vec = column && sizeof(T) <= 8 && g.K1 % WV == 0 && ((uintptr_t)ac->data() & 15) == 0 && g.K0 <= 65535 && g.R1 >= 64;
  if (column_vec) {
    const int64_t bx = (g.K1 / WV + 31) / 32;
    dim3 grid((unsigned)bx, (unsigned)nsplit, (unsigned)g.K0);
    hipLaunchKernelGGL((reduce_column_vec_kernel<T, OP>), grid, dim3(256), 0, st, ac->ptr<T>(), partial->ptr<A>(), g);
  } else if (column) {
    dim3 grid((unsigned)((nout + 255) / 256), (unsigned)nsplit);
    hipLaunchKernelGGL((reduce_column_kernel<T, OP>), grid, dim3(256), 0, st, ac->ptr<T>(), partial->ptr<A>(), g);
  } else {
    LAMP_CHECK(nout < (1ll << 31), "too many outputs for the block reduction");
    dim3 grid((unsigned)nout, (unsigned)nsplit);
    int64_t per = (g.R1 * g.R2 + nsplit - 1) / nsplit;
    int block = per >= 256 ? 256 : (per > 64 ? 128 : 64);
    hipLaunchKernelGGL((reduce_block_kernel<T, OP>), grid, dim3(block), 0, st, ac->ptr<T>(), partial->ptr<A>(), g);
  }
  LAMP_LAUNCH_CHECK();
  if (nsplit >= 16 && nout < 65536)
    hipLaunchKernelGGL((reduce_finalize_wide_kernel<T, OP>), dim3((unsigned)((nout + 15) / 16)), dim3(256), 0, st, partial->ptr<A>(), out->ptr<T>(),
                       nout, nsplit, scale, do_sqrt);
  else
    hipLaunchKernelGGL((reduce_finalize_kernel<T, OP>), dim3(grid_for(nout, 256)), dim3(256), 0, st, partial->ptr<A>(), out->ptr<T>(),
                       nout, nsplit, scale, do_sqrt);
  LAMP_LAUNCH_CHECK();
}

// lamp's CPU device: the reductions of tensors that live in host memory, as one scalar loop (sums in f64 / i64; row-major order)
template <class T>
static void host_reduce(const Tensor* ac, Tensor* out, const DimPlan& p, int op, double scale) {
  using A = acc_t<T>;
  const int nd = ac->ndim;
  const int64_t n = ac->numel();
  std::vector<int64_t> ostride(nd, 0);                 // stride of every input dim in the (keepdim) output
  { int64_t run = 1; for (int d = nd - 1; d >= 0; d--) if (!p.reduced[d]) { ostride[d] = run; run *= ac->sizes[d]; } }
  const int64_t nout = std::max<int64_t>(p.nout, 1);
  std::vector<double> accd(std::is_integral<A>::value ? 0 : nout, (op == 3) ? -INFINITY : (op == 4 ? INFINITY : 0.0));
  std::vector<int64_t> acci(std::is_integral<A>::value ? nout : 0, 0);
  const T* src = ac->ptr<T>();
  std::vector<int64_t> idx(nd, 0);
  for (int64_t e = 0; e < n; e++) {
    int64_t o = 0;
    for (int d = 0; d < nd; d++) o += idx[d] * ostride[d];
    if (std::is_integral<A>::value) acci[o] += (int64_t)load_as<A>(src[e]);
    else {
      const double v = (double)load_as<A>(src[e]);
      if (op == 0 || op == 1) accd[o] += v;
      else if (op == 2 || op == 5) accd[o] += v * v;
      else if (op == 3) { if (v > accd[o] || v != v) accd[o] = v; }
      else if (op == 4) { if (v < accd[o] || v != v) accd[o] = v; }
    }
    for (int d = nd - 1; d >= 0; d--) { if (++idx[d] < ac->sizes[d]) break; idx[d] = 0; }
  }
  T* dst = out->ptr<T>();
  for (int64_t o = 0; o < nout; o++) {
    if (std::is_integral<A>::value) dst[o] = store_as<T>((A)(op == 1 ? (double)acci[o] * scale : (double)acci[o]));
    else { double r = accd[o]; if (op == 1) r *= scale; if (op == 2) r = std::sqrt(r); dst[o] = store_as<T>((A)r); }
  }
}

// op: 0 sum, 1 mean, 2 norm2, 3 max, 4 min, 5 sumsq
Tensor* reduce_dims(const Tensor* a, const int64_t* dims, int ndims, bool keepdim, int op) {
  LAMP_CHECK(a != nullptr, "input is null");
  if (!a->is_device()) {
    DimPlan hp = plan_dims(a, dims, ndims);
    Hold hac(contiguous(a));
    Hold hout(new_tensor(keepdim ? hp.out_keep : hp.out_nokeep, a->dtype, -1));
    const double hscale = op == 1 ? (hp.nred ? 1.0 / (double)hp.nred : NAN) : 1.0;
    LAMP_DISPATCH_ALL(a->dtype, T, (host_reduce<T>(hac.get(), hout.get(), hp, op, hscale)));
    return hout.take();
  }
  check_device_tensor(a, "input");
  DimPlan p = plan_dims(a, dims, ndims);
  Hold ac(contiguous(a));
  const auto& oshape = keepdim ? p.out_keep : p.out_nokeep;
  Hold out(new_tensor(oshape, a->dtype, a->device()));
  double scale = 1.0;
  int do_sqrt = 0;
  if (op == 1) { scale = p.nred ? 1.0 / (double)p.nred : NAN; }
  if (op == 2) do_sqrt = 1;
  if (op == 0 || op == 1) { LAMP_DISPATCH_ALL(a->dtype, T, (reduce_typed<T, kSum>(ac.get(), out.get(), p, scale, do_sqrt))); }
  else if (op == 2 || op == 5) { LAMP_DISPATCH_FLOAT(a->dtype, T, (reduce_typed<T, kSumSq>(ac.get(), out.get(), p, scale, do_sqrt))); }
  else if (op == 3) { LAMP_DISPATCH_FLOAT(a->dtype, T, (reduce_typed<T, kMax>(ac.get(), out.get(), p, scale, 0))); }
  else if (op == 4) { LAMP_DISPATCH_FLOAT(a->dtype, T, (reduce_typed<T, kMin>(ac.get(), out.get(), p, scale, 0))); }
  else LAMP_CHECK(false, "bad reduce op");
  return out.take();
}

// ---- var_mean: two pass (mean, then centred sum of squares) through the same machinery --------
template <class T>
__global__ void center_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ mean, int64_t n, IterArgs it) {
  using A = acc_t<T>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t off[3];
    iter_offsets<3>(it, i, off);
    out[off[0]] = store_as<T>((A)(load_as<A>(x[off[1]]) - load_as<A>(mean[off[2]])));
  }
}

// ---- argmax along one dim: [outer, D, inner] -> [outer, inner] i64 -----------------------------
template <class T>
__global__ void argmax_kernel(const T* __restrict__ in, int64_t* __restrict__ out, int64_t outer, int64_t D, int64_t inner) {
  using A = acc_t<T>;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < outer * inner; o += (int64_t)gridDim.x * blockDim.x) {
    int64_t a = o / inner, b = o - a * inner;
    const T* p = in + a * D * inner + b;
    A best = load_as<A>(p[0]);
    int64_t bi = 0;
    for (int64_t d = 1; d < D; d++) {
      A v = load_as<A>(p[d * inner]);
      if (v > best || (v != v && best == best)) { best = v; bi = d; }  // first max wins; NaN propagates like ATen
    }
    out[o] = bi;
  }
}

}  // namespace lamp

using namespace lamp;

extern "C" {

int lamp_sum_all(lamp_tensor** out, const lamp_tensor* a) { LAMP_API_BEGIN *out = reduce_dims(a, nullptr, 0, false, 0); LAMP_API_END }
int lamp_sum_dims(lamp_tensor** out, const lamp_tensor* a, const int64_t* dims, int ndims, int keepdim) {
  LAMP_API_BEGIN
  if (ndims == 0) { *out = reduce_dims(a, nullptr, 0, keepdim, 0); }  // ATen: empty dim list = all
  else *out = reduce_dims(a, dims, ndims, keepdim, 0);
  LAMP_API_END
}
int lamp_mean_all(lamp_tensor** out, const lamp_tensor* a) { LAMP_API_BEGIN *out = reduce_dims(a, nullptr, 0, false, 1); LAMP_API_END }
int lamp_mean_dims(lamp_tensor** out, const lamp_tensor* a, const int64_t* dims, int ndims, int keepdim) {
  LAMP_API_BEGIN *out = reduce_dims(a, dims, ndims, keepdim, 1); LAMP_API_END
}
int lamp_norm2_dims(lamp_tensor** out, const lamp_tensor* a, const int64_t* dims, int ndims, int keepdim) {
  LAMP_API_BEGIN *out = reduce_dims(a, dims, ndims, keepdim, 2); LAMP_API_END
}
int lamp_max_all(lamp_tensor** out, const lamp_tensor* a) { LAMP_API_BEGIN *out = reduce_dims(a, nullptr, 0, false, 3); LAMP_API_END }
int lamp_min_all(lamp_tensor** out, const lamp_tensor* a) { LAMP_API_BEGIN *out = reduce_dims(a, nullptr, 0, false, 4); LAMP_API_END }

int lamp_var_mean_dims(lamp_tensor** var_out, lamp_tensor** mean_out, const lamp_tensor* a, const int64_t* dims, int ndims,
                       int unbiased, int keepdim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "input");
  Hold mean_keep(reduce_dims(a, dims, ndims, true, 1));
  Hold centred(new_like(a));
  {
    const Tensor* ops[3] = {centred.get(), a, mean_keep.get()};
    IterSpace it = make_iter(a->shape(), ops, 3);
    if (it.numel) {
      LAMP_DISPATCH_FLOAT(a->dtype, T, hipLaunchKernelGGL((center_kernel<T>), dim3(grid_for(it.numel, 256)), dim3(256), 0,
                                                          current_stream(a->device()), centred->ptr<T>(), a->ptr<T>(),
                                                          mean_keep->ptr<T>(), it.numel, to_args(it)));
      LAMP_LAUNCH_CHECK();
    }
  }
  Hold ss(reduce_dims(centred.get(), dims, ndims, keepdim, 5));
  DimPlan p = plan_dims(a, dims, ndims);
  double denom = (double)p.nred - (unbiased ? 1.0 : 0.0);
  LAMP_CHECK(lamp_mul_scalar_(ss.get(), denom > 0 ? 1.0 / denom : NAN) == 0, lamp_last_error());
  if (keepdim) { *mean_out = mean_keep.take(); }
  else {
    lamp_tensor* m = nullptr;
    LAMP_CHECK(lamp_view(&m, mean_keep.get(), p.out_nokeep.data(), (int)p.out_nokeep.size()) == 0, lamp_last_error());
    *mean_out = m;
  }
  *var_out = ss.take();
  LAMP_API_END
}

int lamp_argmax(lamp_tensor** out, const lamp_tensor* a, int64_t dim, int keepdim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "input");
  LAMP_CHECK(a->ndim > 0 && a->numel() > 0, "argmax of an empty tensor");
  int64_t d = wrap_dim(dim, a->ndim);
  Hold ac(contiguous(a));
  int64_t outer = 1, inner = 1;
  for (int i = 0; i < d; i++) outer *= a->sizes[i];
  for (int i = (int)d + 1; i < a->ndim; i++) inner *= a->sizes[i];
  std::vector<int64_t> oshape;
  for (int i = 0; i < a->ndim; i++) { if (i == d) { if (keepdim) oshape.push_back(1); } else oshape.push_back(a->sizes[i]); }
  Hold r(new_tensor(oshape, kI64, a->device()));
  LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((argmax_kernel<T>), dim3(grid_for(outer * inner, 256)), dim3(256), 0,
                                                    current_stream(a->device()), ac->ptr<T>(), r->ptr<int64_t>(), outer,
                                                    a->sizes[d], inner));
  LAMP_LAUNCH_CHECK();
  *out = r.take();
  LAMP_API_END
}

// TensorHelpers.unbroadcast (TensorHelpers.scala:7-41): sum the broadcast dims of p back to `target`
int lamp_unbroadcast(lamp_tensor** out, const lamp_tensor* p, const int64_t* target, int ndim) {
  LAMP_API_BEGIN
  check_device_tensor(p, "p");
  LAMP_CHECK(ndim <= p->ndim, "unbroadcast: target has more dims than the gradient");
  bool same = (ndim == p->ndim);
  for (int i = 0; same && i < ndim; i++) same = (target[i] == p->sizes[i]);
  if (same) { *out = retain(p); return 0; }
  int lead = p->ndim - ndim;
  std::vector<int64_t> dims;
  for (int i = 0; i < p->ndim; i++) {
    if (i < lead) dims.push_back(i);
    else if (target[i - lead] == 1 && p->sizes[i] != 1) dims.push_back(i);
    else LAMP_CHECK(target[i - lead] == p->sizes[i], "unbroadcast: size mismatch at dim " << i);
  }
  Hold s(reduce_dims(p, dims.data(), (int)dims.size(), true, 0));
  lamp_tensor* v = nullptr;
  LAMP_CHECK(lamp_view(&v, s.get(), target, ndim) == 0, lamp_last_error());
  *out = v;
  LAMP_API_END
}

}  // extern "C"
