// Direct convolution for the NARROW layers of the CIFAR ResNet (3..16 channels on 32x32 / 16x16 maps:
// stem 3->6 5x5, res1 6->6, res2 6->16 incl. the strided 3x3 and 1x1 projections; reference workload
// example-cifar100/.../cnn.scala:89-131, op ops.scala:1547-1651).  MFMA tiles would be > 90 % padding here, the
// layers are bound by activation traffic, so: one workgroup per image, the image staged ONCE in LDS (fp32, zero
// halo, so no bounds checks), one thread per output pixel accumulating ALL output channels in registers, weights
// read through the scalar cache (wave-uniform indices -> s_load, used as SGPR operands of v_fmac).
//   fprop : y[n, :, ho, wo]  = b + sum_{ci,r,s} x[n, ci, ho*sh - p + r, wo*sw - p + s] * w[:, ci, r, s]
//   dgrad : dx[n, :, h, w]   = sum_{co,r,s}   dy[n, co, (h + p - r)/sh, (w + p - s)/sw] * w[co, :, r, s]
#include <map>
#include <vector>
#include <mutex>
#include <tuple>
#include "device_utils.h"
#include "conv_geom.h"
#include "wgrad_reduce.h"

namespace lamp {

// weights as fp32 in "k-major" order so that the innermost (unrolled) loop over the register-blocked
// channel reads consecutive scalars:  fprop wf[(ci, r, s)][co],  dgrad wf[(co, r, s)][ci]
template <class T>
__global__ void cs_pack_kernel(const T* __restrict__ w, float* __restrict__ wf, int Cout, int Cin, int kh, int kw, int CB, int dgrad) {
  const int total = (dgrad ? Cout : Cin) * kh * kw * CB;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int c = e % CB, krs = e / CB;
    const int s = krs % kw, r = (krs / kw) % kh, k = krs / (kw * kh);
    float v = 0.f;
    if (!dgrad) { if (c < Cout) v = load_as<float>(w[((c * Cin + k) * kh + r) * kw + s]); }
    else { if (c < Cin) v = load_as<float>(w[((k * Cin + c) * kh + r) * kw + s]); }
    wf[e] = v;
  }
}

// CB = register block over output channels (8 or 16)
template <class T, int CB>
__global__ __launch_bounds__(256) void cs_fwd_kernel(const T* __restrict__ x, const float* __restrict__ wf, const T* __restrict__ bias,
                                                     T* __restrict__ y, ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* xs = reinterpret_cast<float*>(smem_raw);            // [Cin][Hp][Wp], zero halo
  const int Hp = (int)g.H + 2 * g.ph, Wp = (int)g.W + 2 * g.pw;
  const int Cin = (int)g.Cin, Cout = (int)g.Cout, H = (int)g.H, W = (int)g.W, Ho = (int)g.Ho, Wo = (int)g.Wo;
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  for (int i = tid; i < Cin * Hp * Wp; i += blockDim.x) xs[i] = 0.f;
  __syncthreads();
  const T* xp = x + n * Cin * H * W;
  for (int i = tid; i < Cin * H * W; i += blockDim.x) {
    const int w_ = i % W, h_ = (i / W) % H, c_ = i / (W * H);
    xs[(c_ * Hp + h_ + g.ph) * Wp + w_ + g.pw] = load_as<float>(xp[i]);
  }
  __syncthreads();
  T* yp = y + n * Cout * Ho * Wo;
  for (int p = tid; p < Ho * Wo; p += blockDim.x) {
    const int ho = p / Wo, wo = p - ho * Wo;
    float acc[CB];
#pragma unroll
    for (int c = 0; c < CB; c++) acc[c] = 0.f;
    const float* xb = xs + (ho * g.sh) * Wp + wo * g.sw;
    const float* wk = wf;
    for (int ci = 0; ci < Cin; ci++) {
      for (int r = 0; r < g.kh; r++) {
        const float* xr = xb + (ci * Hp + r) * Wp;
        for (int s = 0; s < g.kw; s++) {
          const float xv = xr[s];
#pragma unroll
          for (int c = 0; c < CB; c++) acc[c] = fmaf(xv, wk[c], acc[c]);
          wk += CB;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CB; c++)
      if (c < Cout) yp[c * Ho * Wo + p] = store_as<T>(acc[c] + (bias ? load_as<float>(bias[c]) : 0.f));
  }
}

// CB = register block over INPUT channels (the outputs of dgrad)
template <class T, int CB>
__global__ __launch_bounds__(256) void cs_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ wf, T* __restrict__ dx, ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* ds = reinterpret_cast<float*>(smem_raw);            // [Cout][Ho][Wo]
  const int Cin = (int)g.Cin, Cout = (int)g.Cout, H = (int)g.H, W = (int)g.W, Ho = (int)g.Ho, Wo = (int)g.Wo;
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  const T* dp = dy + n * Cout * Ho * Wo;
  for (int i = tid; i < Cout * Ho * Wo; i += blockDim.x) ds[i] = load_as<float>(dp[i]);
  __syncthreads();
  T* xp = dx + n * Cin * H * W;
  for (int p = tid; p < H * W; p += blockDim.x) {
    const int h = p / W, w = p - h * W;
    float acc[CB];
#pragma unroll
    for (int c = 0; c < CB; c++) acc[c] = 0.f;
    const float* wk = wf;
    for (int co = 0; co < Cout; co++) {
      for (int r = 0; r < g.kh; r++) {
        const int hn = h + g.ph - r;
        const int ho = hn / g.sh;
        const bool hv = hn >= 0 && (hn - ho * g.sh) == 0 && ho < Ho;
        for (int s = 0; s < g.kw; s++) {
          const int wn = w + g.pw - s;
          const int wo = wn / g.sw;
          const bool v = hv && wn >= 0 && (wn - wo * g.sw) == 0 && wo < Wo;
          const float gv = v ? ds[(co * Ho + ho) * Wo + wo] : 0.f;
#pragma unroll
          for (int c = 0; c < CB; c++) acc[c] = fmaf(gv, wk[c], acc[c]);
          wk += CB;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CB; c++)
      if (c < Cin) xp[c * H * W + p] = store_as<T>(acc[c]);
  }
}

// ---- round 4: the same layers in f32 and f64 (the precisions the reference's example trains in), specialised ----------------------------
// The generic kernels above spend more instructions on addressing than on arithmetic (runtime tap loops, an integer division per tap in the
// strided dgrad, a division per staged element) and issue one v_fmac per channel.  For square power-of-two maps, kernel 1 / 3 / 5 with
// "same" padding and stride 1 / 2 - every narrow layer of Cnn.resnet - the forms below unroll the taps, stage with 16-byte loads and shifts,
// keep f32 accumulators in register PAIRS (v_pk_fma_f32: two channels per instruction, the weight pair an SGPR operand, the pixel broadcast by
// op_sel - the VALU's full f32 rate, which is also the rate of the f32 matrix pipe) and f64 accumulators for f64 tensors (v_fma_f64 runs at
// the rate of the unpacked f32 FMA).  The strided dgrad maps a thread to an ST x ST block of input pixels: which taps reach which pixel of
// the block is then a compile-time fact (9 of 36 tap-pixel pairs for 3x3 stride 2), not a per-lane test.
typedef float cs_float2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cs_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double cs_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <class T> struct CsAccOf { using type = float; };
template <> struct CsAccOf<double> { using type = double; };

template <class A, int CB> struct CsAcc {
  A v[CB];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int c = 0; c < CB; c++) v[c] = A(0);
  }
  __device__ __forceinline__ void fma(A x, const A* __restrict__ w) {
#pragma unroll
    for (int c = 0; c < CB; c++) v[c] = cs_fma(x, w[c], v[c]);
  }
  __device__ __forceinline__ A get(int c) const { return v[c]; }
};
template <int CB> struct CsAcc<float, CB> {
  cs_float2 v[CB / 2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int c = 0; c < CB / 2; c++) v[c] = cs_float2{0.f, 0.f};
  }
  __device__ __forceinline__ void fma(float x, const float* __restrict__ w) {
    const cs_float2 x2 = {x, x};
#pragma unroll
    for (int c = 0; c < CB / 2; c++) v[c] = __builtin_elementwise_fma(x2, *reinterpret_cast<const cs_float2*>(w + 2 * c), v[c]);
  }
  __device__ __forceinline__ float get(int c) const { return (c & 1) ? v[c >> 1].y : v[c >> 1].x; }
};

// image (or gradient) of one sample -> LDS [C][S + 2 HALO][S + 2 HALO] with a zero halo; S = 1 << shift, 16 bytes per load
template <class T, class A>
__device__ __forceinline__ void cs2_stage(const T* __restrict__ src, A* __restrict__ dst, int C, int shift, int HALO, int tid, int nthreads) {
  const int S = 1 << shift, Sp = S + 2 * HALO;
  constexpr int V = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  if (HALO > 0) {
    for (int i = tid; i < C * Sp * Sp; i += nthreads) dst[i] = A(0);
    __syncthreads();
  }
  const int total = (C << (2 * shift)) / V;
  for (int i = tid; i < total; i += nthreads) {
    const vec_t v = *reinterpret_cast<const vec_t*>(src + (int64_t)i * V);
    const int e = i * V;
    const int w_ = e & (S - 1), h_ = (e >> shift) & (S - 1), c_ = e >> (2 * shift);
    A* d = dst + (c_ * Sp + h_ + HALO) * Sp + w_ + HALO;
#pragma unroll
    for (int k = 0; k < V; k++) d[k] = (A)v[k];
  }
  __syncthreads();
}

// fprop: one workgroup per image, one thread per output pixel (pixels beyond the workgroup's size in further rounds)
template <class T, int CB, int KS, int ST>
__global__ __launch_bounds__(256) void cs2_fwd_kernel(const T* __restrict__ x, const typename CsAccOf<T>::type* __restrict__ wf, const T* __restrict__ bias,
                                                      T* __restrict__ y, int Cin, int Cout, int shift) {
  using A = typename CsAccOf<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  A* xs = reinterpret_cast<A*>(smem_raw);
  constexpr int PAD = KS / 2;
  const int S = 1 << shift, Sp = S + 2 * PAD;
  const int oshift = ST == 2 ? shift - 1 : shift, So = 1 << oshift;
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  cs2_stage<T, A>(x + (n * Cin << (2 * shift)), xs, Cin, shift, PAD, tid, blockDim.x);
  T* yp = y + (n * Cout << (2 * oshift));
  for (int p = tid; p < So * So; p += blockDim.x) {
    const int ho = p >> oshift, wo = p & (So - 1);
    CsAcc<A, CB> acc;
    acc.zero();
    const A* xb = xs + (ho * ST) * Sp + wo * ST;
    const A* wk = wf;
    for (int ci = 0; ci < Cin; ci++) {
#pragma unroll
      for (int r = 0; r < KS; r++)
#pragma unroll
        for (int s = 0; s < KS; s++) acc.fma(xb[(ci * Sp + r) * Sp + s], wk + (r * KS + s) * CB);
      wk += KS * KS * CB;
    }
#pragma unroll
    for (int c = 0; c < CB; c++)
      if (c < Cout) yp[(c << (2 * oshift)) + p] = (T)(acc.get(c) + (bias ? (A)bias[c] : A(0)));
  }
}

// dgrad: one workgroup per image, one thread per ST x ST block of dx; dy staged with a halo of PAD zeros
template <class T, int CB, int KS, int ST>
__global__ __launch_bounds__(256) void cs2_dgrad_kernel(const T* __restrict__ dy, const typename CsAccOf<T>::type* __restrict__ wf, T* __restrict__ dx,
                                                        int Cin, int Cout, int shift, const T* __restrict__ addend) {
  using A = typename CsAccOf<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  A* ds = reinterpret_cast<A*>(smem_raw);
  constexpr int PAD = KS / 2;
  const int oshift = ST == 2 ? shift - 1 : shift, So = 1 << oshift, Sq = So + 2 * PAD;    // dy is So x So; a block of dx = one pixel of dy
  const int S = 1 << shift;
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  cs2_stage<T, A>(dy + (n * Cout << (2 * oshift)), ds, Cout, oshift, PAD, tid, blockDim.x);
  T* xp = dx + (n * Cin << (2 * shift));
  // addend (round 5): dx = round(round(dgrad) + addend) - the second contribution to a residual block's input gradient (autograd.scala:66-84)
  // added here instead of by an elementwise pass over the map (the f32 step had two such passes, 32 us, behind its strided narrow dgrads)
  const T* ap = addend ? addend + (n * Cin << (2 * shift)) : nullptr;
  for (int b = tid; b < So * So; b += blockDim.x) {
    const int hb = b >> oshift, wb = b & (So - 1);
    CsAcc<A, CB> acc[ST][ST];
#pragma unroll
    for (int i = 0; i < ST; i++)
#pragma unroll
      for (int j = 0; j < ST; j++) acc[i][j].zero();
    const A* db = ds + (hb + PAD) * Sq + wb + PAD;
    const A* wk = wf;
    for (int co = 0; co < Cout; co++) {
#pragma unroll
      for (int r = 0; r < KS; r++)
#pragma unroll
        for (int s = 0; s < KS; s++)
#pragma unroll
          for (int i = 0; i < ST; i++)
#pragma unroll
            for (int j = 0; j < ST; j++) {
              // dx[h, w] takes dy[(h + PAD - r) / ST, (w + PAD - s) / ST] where both divisions are exact
              const int hn = i + PAD - r, wn = j + PAD - s;
              if (hn % ST != 0 || wn % ST != 0) continue;
              acc[i][j].fma(db[(co * Sq + hn / ST) * Sq + wn / ST], wk + (r * KS + s) * CB);
            }
      wk += KS * KS * CB;
    }
#pragma unroll
    for (int c = 0; c < CB; c++)
      if (c < Cin) {
#pragma unroll
        for (int i = 0; i < ST; i++) {
          const int64_t off = (c << (2 * shift)) + (hb * ST + i) * S + wb * ST;
          T* o = xp + off;
          if (ST == 2) {
            typedef T pair_t __attribute__((ext_vector_type(2)));
            pair_t v = pair_t{(T)acc[i][0].get(c), (T)acc[i][ST - 1].get(c)};
            if (ap) { const pair_t a = *reinterpret_cast<const pair_t*>(ap + off); v = pair_t{(T)(v[0] + a[0]), (T)(v[1] + a[1])}; }
            *reinterpret_cast<pair_t*>(o) = v;
          } else {
            T v = (T)acc[i][0].get(c);
            if (ap) v = (T)(v + ap[off]);
            o[0] = v;
          }
        }
      }
  }
}

// both weight images of a layer - [fprop: (ci, r, s) x CBf output channels | dgrad: (co, r, s) x CBd input channels] - for up to 16 layers
// in one launch (blockIdx.y = layer)
constexpr int CS2_PACK_MAX = 16;
struct Cs2PackMany { const void* w[CS2_PACK_MAX]; void* wf[CS2_PACK_MAX]; int Cout[CS2_PACK_MAX], Cin[CS2_PACK_MAX], KS[CS2_PACK_MAX], CBf[CS2_PACK_MAX], CBd[CS2_PACK_MAX]; };
template <class T, class A>
__global__ void cs2_pack_kernel(Cs2PackMany a) {
  const int t = blockIdx.y;
  const T* __restrict__ w = static_cast<const T*>(a.w[t]);
  A* __restrict__ wf = static_cast<A*>(a.wf[t]);
  const int Cout = a.Cout[t], Cin = a.Cin[t], KS = a.KS[t], CBf = a.CBf[t], CBd = a.CBd[t];
  const int nf = Cin * KS * KS * CBf, total = nf + Cout * KS * KS * CBd;
  for (int e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += gridDim.x * blockDim.x) {
    const bool dgrad = e0 >= nf;
    const int e = dgrad ? e0 - nf : e0, CB = dgrad ? CBd : CBf;
    const int c = e % CB, krs = e / CB;
    const int s = krs % KS, r = (krs / KS) % KS, k = krs / (KS * KS);
    A v = A(0);
    if (!dgrad) { if (c < Cout) v = (A)w[((c * Cin + k) * KS + r) * KS + s]; }
    else { if (c < Cin) v = (A)w[((k * Cin + c) * KS + r) * KS + s]; }
    wf[e0] = v;
  }
}
static int cs2_cb_of(int64_t ch) { return ch == 6 ? 6 : (ch <= 8 ? 8 : 16); }

// the packed images are kept per (weight view, stream) while the weight's storage is unchanged: the backward pass of a step finds what its
// forward pass packed (13 -> 6 pack launches per ResNet step)
namespace {
struct Cs2PackKey {
  uint64_t uid; int64_t offset; int KS, Cout, Cin, dtype; hipStream_t st;
  bool operator<(const Cs2PackKey& o) const { return std::tie(uid, offset, KS, Cout, Cin, dtype, st) < std::tie(o.uid, o.offset, o.KS, o.Cout, o.Cin, o.dtype, o.st); }
};
struct Cs2PackVal { uint64_t version; Tensor* packed; uint64_t tick;  bool pinned = false; };
std::mutex g_cs2_mu;
std::map<Cs2PackKey, Cs2PackVal> g_cs2_cache;
uint64_t g_cs2_tick = 0;
}  // namespace
template <class T, class A> static Tensor* cs2_packed(const Tensor* w, const ConvGeom& g, hipStream_t st, int64_t* dgrad_offset) {
  const int KS = g.kh, CBf = cs2_cb_of(g.Cout), CBd = cs2_cb_of(g.Cin);
  const int64_t nf = g.Cin * KS * KS * CBf, nd = g.Cout * KS * KS * CBd;
  *dgrad_offset = nf;
  const bool cacheable = w->st->owned && !w->st->scratch;
  const Cs2PackKey key{w->st->uid, w->offset, KS, (int)g.Cout, (int)g.Cin, w->dtype, st};
  const uint64_t ver = w->st->version.load(std::memory_order_relaxed);
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_cs2_mu);
    auto it = g_cs2_cache.find(key);
    if (it != g_cs2_cache.end() && it->second.version == ver) {
      it->second.tick = ++g_cs2_tick;
      if (allocator_capturing()) it->second.pinned = true;
      return retain(it->second.packed);
    }
  }
  int64_t ps[1] = {nf + nd};
  Hold wf(new_tensor(ps, 1, std::is_same<A, float>::value ? kF32 : kF64, w->device()));
  Cs2PackMany pm;
  pm.w[0] = w->raw(); pm.wf[0] = wf->raw(); pm.Cout[0] = (int)g.Cout; pm.Cin[0] = (int)g.Cin; pm.KS[0] = KS; pm.CBf[0] = CBf; pm.CBd[0] = CBd;
  hipLaunchKernelGGL((cs2_pack_kernel<T, A>), dim3((unsigned)std::min<int64_t>(64, (nf + nd + 255) / 256), 1u), dim3(256), 0, st, pm);
  LAMP_LAUNCH_CHECK();
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_cs2_mu);
    auto it = g_cs2_cache.find(key);
    if (it != g_cs2_cache.end()) { release(it->second.packed); g_cs2_cache.erase(it); }
    if (g_cs2_cache.size() >= 64) {             // least recently used entry that no captured graph reads
      auto victim = g_cs2_cache.end();
      for (auto i = g_cs2_cache.begin(); i != g_cs2_cache.end(); ++i)
        if (!i->second.pinned && (victim == g_cs2_cache.end() || i->second.tick < victim->second.tick)) victim = i;
      if (victim != g_cs2_cache.end()) { release(victim->second.packed); g_cs2_cache.erase(victim); }
    }
    g_cs2_cache[key] = Cs2PackVal{ver, retain(wf.get()), ++g_cs2_tick, allocator_capturing()};
  }
  return wf.take();
}

// the optimisers' hook (optim.hip): every cached pair of images whose weight was just updated is re-packed in place, one launch per element
// type - a replayed graph (whose capture found the images in the cache and recorded no pack launch) keeps reading current weights
template <class T, class A> static void cs2_repack_t(lamp_tensor* const* params, int n, hipStream_t st, int dtype) {
  Cs2PackMany pm;
  int cnt = 0;
  std::vector<std::pair<Cs2PackKey, uint64_t>> done;
  auto flush = [&] {                                    // one launch per CS2_PACK_MAX images; the loop goes on (it used to stop)
    if (cnt == 0) return;
    hipLaunchKernelGGL((cs2_pack_kernel<T, A>), dim3(16u, (unsigned)cnt), dim3(256), 0, st, pm);
    LAMP_LAUNCH_CHECK();
    cnt = 0;
  };
  for (int i = 0; i < n; i++) {
    if (cnt == CS2_PACK_MAX) flush();
    const Tensor* w = params[i];
    if (!w || !w->is_device() || w->dtype != dtype || w->ndim != 4 || !w->st->owned || !w->is_contiguous()) continue;
    for (auto& kv : g_cs2_cache) {
      const Cs2PackKey& k = kv.first;
      if (k.uid != w->st->uid || k.offset != w->offset || k.st != st || k.dtype != dtype) continue;
      if (k.Cout != (int)w->sizes[0] || k.Cin != (int)w->sizes[1] || k.KS != (int)w->sizes[2]) continue;
      pm.w[cnt] = w->raw(); pm.wf[cnt] = kv.second.packed->raw(); pm.Cout[cnt] = k.Cout; pm.Cin[cnt] = k.Cin; pm.KS[cnt] = k.KS;
      pm.CBf[cnt] = cs2_cb_of(k.Cout); pm.CBd[cnt] = cs2_cb_of(k.Cin);
      done.push_back({k, w->st->version.load(std::memory_order_relaxed)});
      cnt++;
      break;
    }
  }
  flush();
  for (auto& d : done) {
    auto it = g_cs2_cache.find(d.first);
    if (it != g_cs2_cache.end()) { it->second.version = d.second; it->second.tick = ++g_cs2_tick; }
  }
}
void small_repack_cached(lamp_tensor* const* params, int n, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_cs2_mu);
  if (g_cs2_cache.empty()) return;
  cs2_repack_t<float, float>(params, n, st, kF32);
  cs2_repack_t<double, double>(params, n, st, kF64);
}

static int cs2_shift_of(int64_t v) { int sh = 0; while ((1LL << sh) < v) sh++; return (1LL << sh) == v ? sh : -1; }
static bool cs2_qualifies(const ConvGeom& g) {
  static const bool on = [] { const char* e = getenv("LAMP_CONV_SMALL2"); return !(e && e[0] == '0'); }();
  if (!on || g.groups != 1 || g.transposed || g.dh != 1 || g.dw != 1) return false;
  if (g.Cin > 16 || g.Cout > 16 || g.N < 1) return false;
  if (g.kh != g.kw || !(g.kh == 1 || g.kh == 3 || g.kh == 5) || g.ph != g.kh / 2 || g.pw != g.kh / 2) return false;
  if (g.sh != g.sw || !(g.sh == 1 || g.sh == 2)) return false;
  if (g.H != g.W || cs2_shift_of(g.H) < 2) return false;                 // square power-of-two maps, at least 4 x 4 (16-byte rows)
  if (g.Ho != g.H / g.sh || g.Wo != g.W / g.sw || g.Ho < 1) return false;
  return true;
}

template <class T> static bool cs2_run(const Tensor* in, const Tensor* w, const Tensor* bias, Tensor* out, const ConvGeom& g, bool dgrad, hipStream_t st,
                                       const Tensor* addend = nullptr) {
  using A = typename CsAccOf<T>::type;
  if (!cs2_qualifies(g)) return false;
  const int KS = g.kh, ST = g.sh, PAD = KS / 2;
  const int shift = cs2_shift_of(g.H);
  const int CB = cs2_cb_of(dgrad ? g.Cin : g.Cout);
  const int64_t side = (dgrad ? g.Ho : g.H) + 2 * PAD;
  const size_t lds = (size_t)((dgrad ? g.Cout : g.Cin) * side * side) * sizeof(A);
  if (lds > 150 * 1024) return false;
  int64_t dgrad_off = 0;
  Hold wf(cs2_packed<T, A>(w, g, st, &dgrad_off));
  const A* wfp = static_cast<const Tensor*>(wf.get())->ptr<A>() + (dgrad ? dgrad_off : 0);
  const int64_t px = g.Ho * g.Wo;                        // fprop: output pixels; dgrad: ST x ST blocks of dx
  const int block = px >= 256 ? 256 : (px >= 128 ? 128 : 64);
  KernelTimer kt(dgrad ? "conv_dgrad_small" : "conv_fwd_small", conv_flops(g), conv_bytes(g, sizeof(T)), st);
  const T* bp = bias ? bias->ptr<T>() : (const T*)nullptr;
  const T* addp = (dgrad && addend) ? addend->ptr<T>() : (const T*)nullptr;
#define CS2_GO(CBv, KSv, STv)                                                                                                                       \
  do {                                                                                                                                              \
    if (!dgrad) {                                                                                                                                   \
      if (lds > 64 * 1024) allow_big_lds((const void*)cs2_fwd_kernel<T, CBv, KSv, STv>);                                                            \
      hipLaunchKernelGGL((cs2_fwd_kernel<T, CBv, KSv, STv>), dim3((unsigned)g.N), dim3(block), lds, st, in->ptr<T>(), wfp, bp, out->ptr<T>(), \
                         (int)g.Cin, (int)g.Cout, shift);                                                                                           \
    } else {                                                                                                                                        \
      if (lds > 64 * 1024) allow_big_lds((const void*)cs2_dgrad_kernel<T, CBv, KSv, STv>);                                                          \
      hipLaunchKernelGGL((cs2_dgrad_kernel<T, CBv, KSv, STv>), dim3((unsigned)g.N), dim3(block), lds, st, in->ptr<T>(), wfp, out->ptr<T>(), \
                         (int)g.Cin, (int)g.Cout, shift, addp);                                                                                     \
    }                                                                                                                                               \
  } while (0)
#define CS2_BY_ST(CBv, KSv) do { if (ST == 1) CS2_GO(CBv, KSv, 1); else CS2_GO(CBv, KSv, 2); } while (0)
#define CS2_BY_KS(CBv) do { if (KS == 1) CS2_BY_ST(CBv, 1); else if (KS == 3) CS2_BY_ST(CBv, 3); else CS2_BY_ST(CBv, 5); } while (0)
  if (CB == 6) CS2_BY_KS(6); else if (CB == 8) CS2_BY_KS(8); else CS2_BY_KS(16);
#undef CS2_BY_KS
#undef CS2_BY_ST
#undef CS2_GO
  LAMP_LAUNCH_CHECK();
  return true;
}

static bool cs_qualifies(const ConvGeom& g, bool dgrad) {
  if (g.groups != 1 || g.transposed || g.dh != 1 || g.dw != 1) return false;
  if (g.Cin > 16 || g.Cout > 16) return false;
  const int64_t lds = dgrad ? g.Cout * g.Ho * g.Wo * 4 : g.Cin * (g.H + 2 * g.ph) * (g.W + 2 * g.pw) * 4;
  if (lds > 64 * 1024) return false;
  if (g.N < 1) return false;
  return true;
}

template <class T> static bool cs_run(const Tensor* in, const Tensor* w, const Tensor* bias, Tensor* out, const ConvGeom& g, bool dgrad, hipStream_t st) {
  if (!cs_qualifies(g, dgrad)) return false;
  const int cb_ch = (int)(dgrad ? g.Cin : g.Cout);
  const int CB = cb_ch <= 8 ? 8 : 16;
  const int64_t nk = (dgrad ? g.Cout : g.Cin) * g.kh * g.kw * CB;
  int64_t ps[1] = {nk};
  Hold wf(new_tensor(ps, 1, kF32, in->device()));
  hipLaunchKernelGGL((cs_pack_kernel<T>), dim3(grid_for(nk, 256)), dim3(256), 0, st, w->ptr<T>(), wf->ptr<float>(), (int)g.Cout, (int)g.Cin, g.kh, g.kw,
                     CB, dgrad ? 1 : 0);
  LAMP_LAUNCH_CHECK();
  const size_t lds = dgrad ? (size_t)(g.Cout * g.Ho * g.Wo * 4) : (size_t)(g.Cin * (g.H + 2 * g.ph) * (g.W + 2 * g.pw) * 4);
  const int64_t px = dgrad ? g.H * g.W : g.Ho * g.Wo;
  const int block = px >= 256 ? 256 : (px >= 128 ? 128 : 64);
  KernelTimer kt(dgrad ? "conv_dgrad_small" : "conv_fwd_small", conv_flops(g), conv_bytes(g, sizeof(T)), st);
  if (!dgrad) {
    if (CB == 8) hipLaunchKernelGGL((cs_fwd_kernel<T, 8>), dim3((unsigned)g.N), dim3(block), lds, st, in->ptr<T>(), wf->ptr<float>(), bias ? bias->ptr<T>() : (const T*)nullptr, out->ptr<T>(), g);
    else hipLaunchKernelGGL((cs_fwd_kernel<T, 16>), dim3((unsigned)g.N), dim3(block), lds, st, in->ptr<T>(), wf->ptr<float>(), bias ? bias->ptr<T>() : (const T*)nullptr, out->ptr<T>(), g);
  } else {
    if (CB == 8) hipLaunchKernelGGL((cs_dgrad_kernel<T, 8>), dim3((unsigned)g.N), dim3(block), lds, st, in->ptr<T>(), wf->ptr<float>(), out->ptr<T>(), g);
    else hipLaunchKernelGGL((cs_dgrad_kernel<T, 16>), dim3((unsigned)g.N), dim3(block), lds, st, in->ptr<T>(), wf->ptr<float>(), out->ptr<T>(), g);
  }
  LAMP_LAUNCH_CHECK();
  return true;
}

// wgrad: dw[co, ci, r, s] = sum_{n, ho, wo} dy[n, co, ho, wo] * x[n, ci, ho*sh - p + r, wo*sw - p + s].
// A workgroup walks a range of images (x with zero halo and dy staged in LDS as fp32). A thread owns one
// (ci, r) pair and a slice of the output rows and keeps a CB x KW register block of partial filter taps
// (all output channels x all horizontal taps): per output pixel CB + KW LDS reads feed CB*KW FMAs.
// Slices are combined with LDS atomics, workgroups through a partial buffer + deterministic reduce.
template <class T, int CB, int KW, class A = float>
__global__ __launch_bounds__(256) void cs_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, A* __restrict__ partial, ConvGeom g,
                                                       int PS, int images_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int Hp = (int)g.H + 2 * g.ph, Wp = (int)g.W + 2 * g.pw + KW;   // + KW: slack so the last taps never read past the row
  const int Cin = (int)g.Cin, Cout = (int)g.Cout, H = (int)g.H, W = (int)g.W, Ho = (int)g.Ho, Wo = (int)g.Wo;
  A* xs = reinterpret_cast<A*>(smem_raw);                     // [Cin][Hp][Wp]
  A* ds = xs + Cin * Hp * Wp;                                  // [CB][Ho][Wo] (rows >= Cout zero)
  A* accs = ds + CB * Ho * Wo;                                 // [PS][Cout][Cin][kh][KW] per-slice partials
  const int O = Cout * Cin * g.kh * KW;
  const int tid = threadIdx.x;
  for (int i = tid; i < Cin * Hp * Wp; i += blockDim.x) xs[i] = A(0);
  for (int i = tid; i < CB * Ho * Wo; i += blockDim.x) ds[i] = A(0);
  const int ntask = Cin * g.kh;
  const int task = tid % ntask, slice = tid / ntask;
  const bool active = slice < PS;
  const int ci = task / g.kh, r = task - ci * g.kh;
  A acc[CB][KW];
#pragma unroll
  for (int c = 0; c < CB; c++)
#pragma unroll
    for (int s = 0; s < KW; s++) acc[c][s] = A(0);
  const int64_t n0 = (int64_t)blockIdx.x * images_per_block, n1 = min(n0 + images_per_block, g.N);
  for (int64_t n = n0; n < n1; n++) {
    __syncthreads();
    const T* xp = x + n * Cin * H * W;
    const T* dp = dy + n * Cout * Ho * Wo;
    for (int i = tid; i < Cin * H * W; i += blockDim.x) {
      const int w_ = i % W, h_ = (i / W) % H, c_ = i / (W * H);
      xs[(c_ * Hp + h_ + g.ph) * Wp + w_ + g.pw] = load_as<A>(xp[i]);
    }
    for (int i = tid; i < Cout * Ho * Wo; i += blockDim.x) ds[i] = load_as<A>(dp[i]);
    __syncthreads();
    if (active) {
      for (int ho = slice; ho < Ho; ho += PS) {
        const A* xr = xs + (ci * Hp + ho * g.sh + r) * Wp;
        const A* dr = ds + ho * Wo;
        for (int wo = 0; wo < Wo; wo++) {
          A xv[KW];
#pragma unroll
          for (int s = 0; s < KW; s++) xv[s] = xr[wo * g.sw + s];
#pragma unroll
          for (int c = 0; c < CB; c++) {
            const A gv = dr[c * Ho * Wo + wo];
#pragma unroll
            for (int s = 0; s < KW; s++) acc[c][s] = cs_fma(gv, xv[s], acc[c][s]);
          }
        }
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int c = 0; c < CB; c++)
      if (c < Cout)
#pragma unroll
        for (int s = 0; s < KW; s++) accs[slice * O + ((c * Cin + ci) * g.kh + r) * KW + s] = acc[c][s];
  }
  __syncthreads();
  // fixed-order sum over the row slices: bitwise reproducible
  for (int i = tid; i < O; i += blockDim.x) {
    A a = A(0);
    for (int sl = 0; sl < PS; sl++) a += accs[sl * O + i];
    partial[(int64_t)blockIdx.x * O + i] = a;
  }
}
template <class T, class A = float>
__global__ __launch_bounds__(256) void cs_wgrad_reduce_kernel(const A* __restrict__ partial, T* __restrict__ dw, int O, int nblocks) {
  const int lane = threadIdx.x & 63;
  const int o = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
  if (o >= O) return;
  A a = A(0);
  for (int b = lane; b < nblocks; b += 64) a += partial[(int64_t)b * O + o];
  a = wave_sum(a);
  if (lane == 0) dw[o] = store_as<T>(a);
}

template <class T> static bool cs_wgrad_run(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st) {
  using A = typename CsAccOf<T>::type;
  if (g.groups != 1 || g.transposed || g.dh != 1 || g.dw != 1) return false;
  if (g.Cin > 16 || g.Cout > 16 || g.kh != g.kw || !(g.kw == 1 || g.kw == 3 || g.kw == 5)) return false;
  const int ntask = (int)g.Cin * g.kh;
  if (ntask > 256 || g.N < 1) return false;
  const int CB = g.Cout == 6 ? 6 : (g.Cout <= 8 ? 8 : 16), KW = g.kw;
  const int Hp = (int)g.H + 2 * g.ph, Wp = (int)g.W + 2 * g.pw + KW;
  const int O = (int)(g.Cout * g.Cin * g.kh * KW);
  const int PS = std::min<int>(256 / ntask, (int)g.Ho);
  const size_t lds = (size_t)(g.Cin * Hp * Wp + CB * g.Ho * g.Wo + (int64_t)PS * O) * sizeof(A);
  if (lds > 150 * 1024) return false;
  const int nb = (int)std::min<int64_t>(g.N, (int64_t)num_cus() * 2);
  const int ipb = (int)((g.N + nb - 1) / nb);
  const int nblocks = (int)((g.N + ipb - 1) / ipb);
  int64_t ps[1] = {(int64_t)nblocks * O};
  Hold partial(new_tensor(ps, 1, std::is_same<A, float>::value ? kF32 : kF64, dy->device()));
  {
    KernelTimer kt("conv_wgrad_small", conv_flops(g), conv_bytes(g, sizeof(T)), st);
#define CS_WG(CBv, KWv)                                                                                                              \
  do {                                                                                                                               \
    static bool attr = false;                                                                                                        \
    allow_big_lds((const void*)cs_wgrad_kernel<T, CBv, KWv, A>); \
    hipLaunchKernelGGL((cs_wgrad_kernel<T, CBv, KWv, A>), dim3(nblocks), dim3(256), lds, st, dy->ptr<T>(), x->ptr<T>(), partial->ptr<A>(), g, PS, ipb); \
  } while (0)
    if (CB == 6 && KW == 1) CS_WG(6, 1); else if (CB == 6 && KW == 3) CS_WG(6, 3); else if (CB == 6 && KW == 5) CS_WG(6, 5);
    else if (CB == 8 && KW == 1) CS_WG(8, 1); else if (CB == 8 && KW == 3) CS_WG(8, 3); else if (CB == 8 && KW == 5) CS_WG(8, 5);
    else if (CB == 16 && KW == 1) CS_WG(16, 1); else if (CB == 16 && KW == 3) CS_WG(16, 3); else CS_WG(16, 5);
#undef CS_WG
    LAMP_LAUNCH_CHECK();
  }
  // the per-block sums join the backward pass's batched reduction (wgrad_reduce.hip: one launch for every layer's partial sums; the same
  // lanes-over-blocks + butterfly order as cs_wgrad_reduce_kernel, so the same bits) - seven ~5 us launches of the f32 ResNet step
  if (std::is_same<T, float>::value || std::is_same<T, bf16_t>::value) {
    WgradReduceArgs ra{};
    ra.kind = 1; ra.O = O; ra.nsplit = nblocks; ra.blocks = (int)(((int64_t)O * 64 + 255) / 256);
    ra.dw_f32 = std::is_same<T, float>::value ? 1 : 0;
    wgrad_reduce_enqueue(ra, partial.get(), dw, st);
    return true;
  }
  hipLaunchKernelGGL((cs_wgrad_reduce_kernel<T, A>), dim3((unsigned)(((int64_t)O * 64 + 255) / 256)), dim3(256), 0, st, partial->ptr<A>(), dw->ptr<T>(), O, nblocks);
  LAMP_LAUNCH_CHECK();
  return true;
}
// ---- round 4: weight gradient of the same layers, accumulators in registers ------------------------------------------------------------------
// cs_wgrad_kernel above reads CB + KW values from LDS for CB x KW multiply-adds: LDS-bound.  Here a thread owns output PIXELS (lane = pixel,
// 64 per round) and a wave owns a share of the filter taps: the units (input channel, filter row, half of the output channels when there
// are 16) are dealt round-robin to the eight waves, at most MAXU per wave; for each of its units a thread keeps COH x KS accumulators in
// registers (f32: as pairs of output channels, v_pk_fma_f32) and per pixel reads COH gradients + KS image values for COH x KS
// multiply-adds.  The accumulators live across all the images of the workgroup and are summed over the wave's lanes once at the end.
template <class T, int CO, int KS, int ST, int MAXU>
__global__ __launch_bounds__(512) void cs2_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, typename CsAccOf<T>::type* __restrict__ partial,
                                                        int64_t N, int Cin, int shift, int images_per_block) {
  using A = typename CsAccOf<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int PAD = KS / 2;
  constexpr int COH = CO <= 8 ? CO : CO / 2, NH = CO / COH;
  const int S = 1 << shift, Sp = S + 2 * PAD;
  const int oshift = ST == 2 ? shift - 1 : shift, So = 1 << oshift;
  A* xs = reinterpret_cast<A*>(smem_raw);                      // [Cin][Sp][Sp], zero halo
  A* ds = xs + Cin * Sp * Sp;                                  // [CO][So][So]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nunits = Cin * KS * NH;
  int xoff[MAXU], doff[MAXU];                                   // wave-uniform LDS offsets of this wave's units
#pragma unroll
  for (int k = 0; k < MAXU; k++) {
    const int u = min(wid + 8 * k, nunits - 1);
    const int h = u % NH, cr = u / NH, r = cr % KS, ci = cr / KS;
    xoff[k] = (ci * Sp + r) * Sp;
    doff[k] = (h * COH) << (2 * oshift);
  }
  A acc[MAXU][KS][COH];
#pragma unroll
  for (int k = 0; k < MAXU; k++)
#pragma unroll
    for (int s_ = 0; s_ < KS; s_++)
#pragma unroll
      for (int c = 0; c < COH; c++) acc[k][s_][c] = A(0);

  const int64_t n0 = (int64_t)blockIdx.x * images_per_block, n1 = min(n0 + images_per_block, N);
  for (int64_t n = n0; n < n1; n++) {
    __syncthreads();                                           // the previous image has been consumed
    cs2_stage<T, A>(x + (n * Cin << (2 * shift)), xs, Cin, shift, PAD, tid, 512);
    cs2_stage<T, A>(dy + (n * CO << (2 * oshift)), ds, CO, oshift, 0, tid, 512);
    for (int p = lane; p < So * So; p += 64) {
      const int ho = p >> oshift, wo = p & (So - 1);
      const A* xb = xs + (ho * ST) * Sp + wo * ST;
#pragma unroll
      for (int k = 0; k < MAXU; k++) {
        if (wid + 8 * k < nunits) {
          A gv[COH];
#pragma unroll
          for (int c = 0; c < COH; c++) gv[c] = ds[doff[k] + (c << (2 * oshift)) + p];
#pragma unroll
          for (int s_ = 0; s_ < KS; s_++) {
            const A xv = xb[xoff[k] + s_];
            if constexpr (std::is_same<A, float>::value) {
              const cs_float2 x2 = {xv, xv};
#pragma unroll
              for (int c = 0; c < COH; c += 2) {
                cs_float2 a = {acc[k][s_][c], acc[k][s_][c + 1]};
                a = __builtin_elementwise_fma(cs_float2{gv[c], gv[c + 1]}, x2, a);
                acc[k][s_][c] = a.x; acc[k][s_][c + 1] = a.y;
              }
            } else {
#pragma unroll
              for (int c = 0; c < COH; c++) acc[k][s_][c] = cs_fma(gv[c], xv, acc[k][s_][c]);
            }
          }
        }
      }
    }
  }
  // sum over the wave's lanes (fixed butterfly order), one writer per accumulator: partial[block][((co Cin + ci) KS + r) KS + s]
  const int O = CO * Cin * KS * KS;
#pragma unroll
  for (int k = 0; k < MAXU; k++) {
    const int u = wid + 8 * k;
    if (u < nunits) {
      const int h = u % NH, cr = u / NH, r = cr % KS, ci = cr / KS;
#pragma unroll
      for (int s_ = 0; s_ < KS; s_++)
#pragma unroll
        for (int c = 0; c < COH; c++) {
          const A v = wave_sum(acc[k][s_][c]);
          if (lane == 0) partial[(int64_t)blockIdx.x * O + (((h * COH + c) * Cin + ci) * KS + r) * KS + s_] = v;
        }
    }
  }
}

template <class T> static bool cs2_wgrad_run(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st) {
  using A = typename CsAccOf<T>::type;
  static const bool on = [] { const char* e = getenv("LAMP_CONV_SMALL2_WGRAD"); return !(e && e[0] == '0'); }();
  if (!on || !cs2_qualifies(g)) return false;
  if (!(g.Cout == 6 || g.Cout == 16) || g.Ho < 8) return false;
  const int KS = g.kh, ST = g.sh, PAD = KS / 2;
  const int NH = g.Cout <= 8 ? 1 : 2;
  const int nunits = (int)g.Cin * KS * NH;
  const int maxu = (nunits + 7) / 8;
  if (maxu > (g.Cout == 6 ? 3 : 5)) return false;             // register budget: MAXU x COH x KS accumulators per lane
  const int shift = cs2_shift_of(g.H);
  const int64_t side = g.H + 2 * PAD;
  const size_t lds = (size_t)(g.Cin * side * side + g.Cout * g.Ho * g.Wo) * sizeof(A);
  if (lds > 150 * 1024) return false;
  const int O = (int)(g.Cout * g.Cin * KS * KS);
  const int nb = (int)std::min<int64_t>(g.N, (int64_t)num_cus() * 2);
  const int ipb = (int)((g.N + nb - 1) / nb);
  const int nblocks = (int)((g.N + ipb - 1) / ipb);
  int64_t ps[1] = {(int64_t)nblocks * O};
  Hold partial(new_tensor(ps, 1, std::is_same<A, float>::value ? kF32 : kF64, dy->device()));
  {
    KernelTimer kt("conv_wgrad_small", conv_flops(g), conv_bytes(g, sizeof(T)), st);
#define CS2_WG(COv, KSv, STv, MUv)                                                                                                                  \
  do {                                                                                                                                              \
    if (lds > 64 * 1024) allow_big_lds((const void*)cs2_wgrad_kernel<T, COv, KSv, STv, MUv>);                                                       \
    hipLaunchKernelGGL((cs2_wgrad_kernel<T, COv, KSv, STv, MUv>), dim3(nblocks), dim3(512), lds, st, dy->ptr<T>(), x->ptr<T>(), partial->ptr<A>(),  \
                       g.N, (int)g.Cin, shift, ipb);                                                                                                \
  } while (0)
#define CS2_WG_ST(COv, KSv, MUv) do { if (ST == 1) CS2_WG(COv, KSv, 1, MUv); else CS2_WG(COv, KSv, 2, MUv); } while (0)
#define CS2_WG_MU(COv, KSv)                                                                                                                         \
  do {                                                                                                                                              \
    if (maxu <= 1) CS2_WG_ST(COv, KSv, 1); else if (maxu == 2) CS2_WG_ST(COv, KSv, 2); else if (maxu == 3) CS2_WG_ST(COv, KSv, 3);                  \
    else if (COv == 16 && maxu == 4) CS2_WG_ST(COv, KSv, 4); else CS2_WG_ST(COv, KSv, 5);                                                           \
  } while (0)
    if (g.Cout == 6) { if (KS == 1) CS2_WG_MU(6, 1); else if (KS == 3) CS2_WG_MU(6, 3); else CS2_WG_MU(6, 5); }
    else { if (KS == 1) CS2_WG_MU(16, 1); else if (KS == 3) CS2_WG_MU(16, 3); else CS2_WG_MU(16, 5); }
#undef CS2_WG_MU
#undef CS2_WG_ST
#undef CS2_WG
    LAMP_LAUNCH_CHECK();
  }
  if (std::is_same<T, float>::value) {
    WgradReduceArgs ra{};
    ra.kind = 1; ra.O = O; ra.nsplit = nblocks; ra.blocks = (int)(((int64_t)O * 64 + 255) / 256);
    ra.dw_f32 = 1;
    wgrad_reduce_enqueue(ra, partial.get(), dw, st);
    return true;
  }
  hipLaunchKernelGGL((cs_wgrad_reduce_kernel<T, A>), dim3((unsigned)(((int64_t)O * 64 + 255) / 256)), dim3(256), 0, st, partial->ptr<A>(), dw->ptr<T>(), O, nblocks);
  LAMP_LAUNCH_CHECK();
  return true;
}

bool small_conv_wgrad(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st) {
  if (x->dtype == kBF16) return cs_wgrad_run<bf16_t>(dy, x, dw, g, st);
  if (x->dtype == kF32) return cs2_wgrad_run<float>(dy, x, dw, g, st) || cs_wgrad_run<float>(dy, x, dw, g, st);
  if (x->dtype == kF64) return cs2_wgrad_run<double>(dy, x, dw, g, st) || cs_wgrad_run<double>(dy, x, dw, g, st);
  return false;
}

// f32 accumulation is only parity-safe for bf16/f32 inputs; f64 runs the specialised forms with f64 accumulators or the generic direct kernels
bool small_conv_fwd(const Tensor* x, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, hipStream_t st) {
  if (x->dtype == kBF16) return cs_run<bf16_t>(x, w, bias, y, g, false, st);
  if (x->dtype == kF32) return cs2_run<float>(x, w, bias, y, g, false, st) || cs_run<float>(x, w, bias, y, g, false, st);
  if (x->dtype == kF64) return cs2_run<double>(x, w, bias, y, g, false, st);
  return false;
}
// addend (optional): added in the store of the f32 / f64 kernel where that kernel takes the geometry - *addend_fused says whether it did
bool small_conv_dgrad(const Tensor* dy, const Tensor* w, Tensor* dx, const ConvGeom& g, hipStream_t st, const Tensor* addend, bool* addend_fused) {
  if (addend_fused) *addend_fused = false;
  if (dy->dtype == kBF16) return cs_run<bf16_t>(dy, w, nullptr, dx, g, true, st);
  if (dy->dtype == kF32) {
    if (cs2_run<float>(dy, w, nullptr, dx, g, true, st, addend)) { if (addend_fused) *addend_fused = addend != nullptr; return true; }
    return cs_run<float>(dy, w, nullptr, dx, g, true, st);
  }
  if (dy->dtype == kF64) {
    if (cs2_run<double>(dy, w, nullptr, dx, g, true, st, addend)) { if (addend_fused) *addend_fused = addend != nullptr; return true; }
    return false;
  }
  return false;
}

}  // namespace lamp
