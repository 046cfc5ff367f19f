// Indexing, gathering, scattering, selection and random sampling.
//
// Replaces ATen.index_select / index_add / masked_select / repeat_interleave / topk / one_hot /
// embedding(+_backward) / rand / randn / normal / randint / dropout_ as lamp calls them
// (reference: lamp-core/src/main/scala/lamp/autograd/ops.scala:179-197 (IndexSelect), 1079-1100
// (Dropout), 2141-2170 (Embedding); lamp-data/.../BatchStream.scala:548-549;
// lamp-umap/.../umap.scala:211-227; lamp-knn/.../package.scala:55).
// Index tensors are int64 and every index-valued result is bit-exact.
// RNG is Philox4x32-10 keyed by (seed, per-call offset); bit compatibility with libtorch's
// streams is not required by any reference test.
#include "device_utils.h"
#include "philox.h"
#include <cstring>
#include "../core/strided.h"

namespace lamp {

// ---- atomics -------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ void atomic_add_t(T* p, T v) { atomicAdd(p, v); }
template <> __device__ __forceinline__ void atomic_add_t<int64_t>(int64_t* p, int64_t v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
template <> __device__ __forceinline__ void atomic_add_t<uint8_t>(uint8_t* p, uint8_t v) {
  unsigned int* w = (unsigned int*)((uintptr_t)p & ~(uintptr_t)3);
  const int sh = ((uintptr_t)p & 3) * 8;
  unsigned int old = *w, assumed;
  do {
    assumed = old;
    unsigned int b = ((assumed >> sh) + v) & 0xff;
    old = atomicCAS(w, assumed, (assumed & ~(0xffu << sh)) | (b << sh));
  } while (old != assumed);
}
template <> __device__ __forceinline__ void atomic_add_t<bf16_t>(bf16_t* p, bf16_t v) {
  unsigned int* w = (unsigned int*)((uintptr_t)p & ~(uintptr_t)3);
  const int sh = ((uintptr_t)p & 2) * 8;
  unsigned int old = *w, assumed;
  do {
    assumed = old;
    bf16_t cur; cur.bits = (uint16_t)(assumed >> sh);
    bf16_t nw((float)cur + (float)v);
    old = atomicCAS(w, assumed, (assumed & ~(0xffffu << sh)) | ((unsigned int)nw.bits << sh));
  } while (old != assumed);
}

template <> __device__ __forceinline__ void atomic_add_t<f16_t>(f16_t* p, f16_t v) {
  unsigned int* w = (unsigned int*)((uintptr_t)p & ~(uintptr_t)3);
  const int sh = ((uintptr_t)p & 2) * 8;
  unsigned int old = *w, assumed;
  do {
    assumed = old;
    f16_t cur; cur.bits = (uint16_t)(assumed >> sh);
    f16_t nw((float)cur + (float)v);
    old = atomicCAS(w, assumed, (assumed & ~(0xffffu << sh)) | ((unsigned int)nw.bits << sh));
  } while (old != assumed);
}

// a viewed as [outer, D, inner]
template <class T>
__global__ void index_select_kernel(const T* __restrict__ a, const int64_t* __restrict__ index, T* __restrict__ out, int64_t outer,
                                    int64_t D, int64_t inner, int64_t J) {
  const int64_t total = outer * J * inner;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % inner, j = (e / inner) % J, o = e / (inner * J);
    int64_t k = index[j];
    if (k < 0) k += D;
    out[e] = (k >= 0 && k < D) ? a[(o * D + k) * inner + i] : T{};
  }
}
// row gather (dim 0 of a contiguous tensor, rows a multiple of 16 bytes): 16-byte packets, no 64-bit divisions per element -
// embeddings, the minibatch gather of the device-resident data set, UMAP's index selects of [n, d] tables
__global__ __launch_bounds__(256) void index_select_rows_vec_kernel(const uint4* __restrict__ a, const int64_t* __restrict__ index, uint4* __restrict__ out,
                                                                    int64_t D, int ppr /* packets per row */, int64_t J) {
  const int64_t total = J * ppr;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / ppr;
    const int p = (int)(e - j * ppr);
    int64_t k = index[j];
    if (k < 0) k += D;
    out[e] = (k >= 0 && k < D) ? a[k * ppr + p] : make_uint4(0, 0, 0, 0);
  }
}
// embedding_dense_backward for tables of moderate height: one workgroup per (weight row, 256 columns) walks the index list in
// chunks of 256 (one ballot per wave), and adds the matching gradient rows in ascending token order into an f32 / f64 accumulator -
// no atomics, no zero fill, run-to-run identical, rounded once.  Rows equal to padding_idx receive no gradient (ATen).
template <class T>
__global__ __launch_bounds__(256) void embedding_backward_scan_kernel(const T* __restrict__ grad, const int64_t* __restrict__ idx, T* __restrict__ out,
                                                                      int64_t N, int64_t E, int64_t padding_idx) {
  using A = acc_t<T>;
  const int64_t k = blockIdx.x;
  const int64_t col = (int64_t)blockIdx.y * 256 + threadIdx.x;
  __shared__ unsigned long long masks[4];
  A acc = 0;
  if (k != padding_idx) {
    for (int64_t base = 0; base < N; base += 256) {
      const int64_t n = base + threadIdx.x;
      const unsigned long long b = __ballot(n < N && idx[n] == k);
      if ((threadIdx.x & 63) == 0) masks[threadIdx.x >> 6] = b;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < 4; w++) {
        unsigned long long m = masks[w];
        while (m) {
          const int bit = __ffsll((long long)m) - 1;
          m &= m - 1;
          if (col < E) acc += load_as<A>(grad[(base + w * 64 + bit) * E + col]);
        }
      }
      __syncthreads();
    }
  }
  if (col < E) out[k * E + col] = store_as<T>(acc);
}

template <class T>
__global__ void index_add_kernel(T* __restrict__ self, const int64_t* __restrict__ index, const T* __restrict__ src, int64_t outer,
                                 int64_t D, int64_t inner, int64_t J) {
  const int64_t total = outer * J * inner;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % inner, j = (e / inner) % J, o = e / (inner * J);
    int64_t k = index[j];
    if (k < 0) k += D;
    if (k >= 0 && k < D) atomic_add_t<T>(self + (o * D + k) * inner + i, src[e]);
  }
}
template <class T>
__global__ void repeat_interleave_kernel(const T* __restrict__ a, T* __restrict__ out, int64_t outer, int64_t D, int64_t inner, int64_t rep) {
  const int64_t total = outer * D * rep * inner;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % inner, jr = (e / inner) % (D * rep), o = e / (inner * D * rep);
    out[e] = a[(o * D + jr / rep) * inner + i];
  }
}
template <class T>
__global__ void one_hot_kernel(const int64_t* __restrict__ a, T* __restrict__ out, int64_t n, int64_t C) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * C; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = (T)((e % C) == a[e / C]);
}

// ---- masked_select: order-preserving stream compaction -------------------------------------------
constexpr int MS_ITEMS = 16, MS_BLOCK = 256, MS_TILE = MS_ITEMS * MS_BLOCK;
__global__ __launch_bounds__(MS_BLOCK) void ms_count_kernel(const uint8_t* __restrict__ mask, int64_t n, int64_t* __restrict__ counts) {
  __shared__ int64_t sm[4];
  const int64_t base = (int64_t)blockIdx.x * MS_TILE + threadIdx.x * MS_ITEMS;
  int64_t c = 0;
  for (int k = 0; k < MS_ITEMS; k++) if (base + k < n && mask[base + k]) c++;
  c = block_sum(c, sm);
  if (threadIdx.x == 0) counts[blockIdx.x] = c;
}
__global__ void ms_scan_kernel(int64_t* counts, int64_t nblocks, int64_t* total) {
  // single thread exclusive scan - nblocks is n / 4096, at most a few thousand
  int64_t run = 0;
  for (int64_t i = 0; i < nblocks; i++) { int64_t c = counts[i]; counts[i] = run; run += c; }
  *total = run;
}
template <class T>
__global__ __launch_bounds__(MS_BLOCK) void ms_scatter_kernel(const T* __restrict__ a, const uint8_t* __restrict__ mask, int64_t n,
                                                              const int64_t* __restrict__ offsets, T* __restrict__ out) {
  __shared__ int sm[MS_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * MS_TILE + threadIdx.x * MS_ITEMS;
  int c = 0;
  for (int k = 0; k < MS_ITEMS; k++) if (base + k < n && mask[base + k]) c++;
  sm[threadIdx.x] = c;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 256 per-thread counts
  for (int off = 1; off < MS_BLOCK; off <<= 1) {
    int v = threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  int64_t pos = offsets[blockIdx.x] + sm[threadIdx.x] - c;
  for (int k = 0; k < MS_ITEMS; k++) if (base + k < n && mask[base + k]) out[pos++] = a[base + k];
}

// masked_scatter: out = self with the positions where mask is true replaced by CONSECUTIVE elements of source (row-major order of the
// true positions) - ATen's masked_scatter, which lamp's MaskSelect / ElementWiseMinimum / ElementWiseMaximum backward closures use
// (ops.scala:133-146, 2287-2340).  Same two-level scan as masked_select.
template <class T>
__global__ __launch_bounds__(MS_BLOCK) void ms_masked_scatter_kernel(T* __restrict__ out, const uint8_t* __restrict__ mask, int64_t n,
                                                                     const int64_t* __restrict__ offsets, const T* __restrict__ src, int64_t nsrc) {
  __shared__ int sm[MS_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * MS_TILE + threadIdx.x * MS_ITEMS;
  int c = 0;
  for (int k = 0; k < MS_ITEMS; k++) if (base + k < n && mask[base + k]) c++;
  sm[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < MS_BLOCK; off <<= 1) {
    int v = threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  int64_t pos = offsets[blockIdx.x] + sm[threadIdx.x] - c;
  for (int k = 0; k < MS_ITEMS; k++) if (base + k < n && mask[base + k]) { if (pos < nsrc) out[base + k] = src[pos]; pos++; }
}

// gather / scatter_add along `dim` for tensors of one rank: coordinates of an element of `index` with the coordinate along dim
// replaced by the index value address the other tensor (ATen gather / scatter_add; lamp's ScatterAdd op, ops.scala:410-434)
struct GsGeom { int ndim; int dim; int64_t isz[kMaxDims]; int64_t istr[kMaxDims]; int64_t astr[kMaxDims]; int64_t bstr[kMaxDims]; int64_t dlim; };
template <class T, bool SCATTER>
__global__ void gather_scatter_kernel(T* __restrict__ a /* gather: out ; scatter: self */, const int64_t* __restrict__ index, const T* __restrict__ b /* gather: input ; scatter: src */,
                                      int64_t n, GsGeom g, int* __restrict__ assert_word) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e, io = 0, ao = 0, bo = 0;
    for (int d = g.ndim - 1; d >= 0; d--) {
      const int64_t c = r % g.isz[d]; r /= g.isz[d];
      io += c * g.istr[d];
      if (d != g.dim) { ao += c * g.astr[d]; bo += c * g.bstr[d]; }
      else { if (SCATTER) bo += c * g.bstr[d]; else ao += c * g.astr[d]; }
    }
    const int64_t t = index[io];
    if (t < 0 || t >= g.dlim) { *(volatile int*)assert_word = kAssertIndexRange; continue; }
    if (SCATTER) atomic_add_t<T>(a + ao + t * g.astr[g.dim], b[bo]);
    else a[ao] = b[bo + t * g.bstr[g.dim]];
  }
}
// scatter WITHOUT accumulation (ATen.scatter, STen.scala:1412-1423): self[.., index[e], ..] = src[e] or a scalar.  Duplicate targets: one of
// the writers wins (ATen leaves it unspecified too)
template <class T, bool VALUE>
__global__ void scatter_set_kernel(T* __restrict__ a, const int64_t* __restrict__ index, const T* __restrict__ b, T value, int64_t n, GsGeom g,
                                   int* __restrict__ assert_word) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e, io = 0, ao = 0, bo = 0;
    for (int d = g.ndim - 1; d >= 0; d--) {
      const int64_t c = r % g.isz[d]; r /= g.isz[d];
      io += c * g.istr[d];
      if (d != g.dim) ao += c * g.astr[d];
      bo += c * g.bstr[d];
    }
    const int64_t t = index[io];
    if (t < 0 || t >= g.dlim) { *(volatile int*)assert_word = kAssertIndexRange; continue; }
    a[ao + t * g.astr[g.dim]] = VALUE ? value : b[bo];
  }
}
// index_put / put (STen.scala:1715-1722): K index tensors (each with NI elements) address the first K dimensions of `a`; every index
// position carries `inner` trailing elements.  Negative indices count from the end.
struct IpGeom { int K; const int64_t* idx[kMaxDims]; int64_t size[kMaxDims]; int64_t stride[kMaxDims]; };
template <class T, bool ACC>
__global__ void index_put_kernel(T* __restrict__ a, const T* __restrict__ v, IpGeom g, int64_t NI, int64_t inner, int* __restrict__ assert_word) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < NI * inner; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = e / inner, r = e - p * inner;
    int64_t off = r;
    bool ok = true;
    for (int k = 0; k < g.K; k++) {
      int64_t t = g.idx[k][p];
      if (t < 0) t += g.size[k];
      if (t < 0 || t >= g.size[k]) { ok = false; break; }
      off += t * g.stride[k];
    }
    if (!ok) { *(volatile int*)assert_word = kAssertIndexRange; continue; }
    if (ACC) atomic_add_t<T>(a + off, v[e]); else a[off] = v[e];
  }
}
// tril / triu of the last two dimensions (STen.scala:1883-1884)
template <class T>
__global__ void tri_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t total, int64_t R, int64_t Cn, int64_t diagonal, int lower) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = e % Cn, r = (e / Cn) % R;
    const bool keep = lower ? (c - r <= diagonal) : (c - r >= diagonal);
    T z; memset(&z, 0, sizeof(T));
    out[e] = keep ? x[e] : z;
  }
}
// rows of a PINNED HOST matrix gathered by a device index list, straight over PCIe into a device batch (one pass: no host gather, no staging
// buffer), converted on the fly when the stored type differs (u8 pixels -> the model's type).  A wave copies one row, 16 bytes per lane.
template <class S, class D>
__global__ __launch_bounds__(256) void gather_pinned_rows_kernel(const S* __restrict__ src, const int64_t* __restrict__ index, D* __restrict__ dst, int64_t rows,
                                                                 int64_t width, int64_t src_rows, int* __restrict__ assert_word) {
  const int lane = threadIdx.x & 63;
  constexpr int V = 16 / sizeof(S);                     // elements per 16-byte request over the bus
  for (int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const int64_t t = index[r];
    if (t < 0 || t >= src_rows) { if (lane == 0) *(volatile int*)assert_word = kAssertIndexRange; continue; }
    const S* s = src + t * width;
    D* d = dst + r * width;
    if (width % V == 0 && ((uintptr_t)s & 15) == 0 && ((uintptr_t)d & (sizeof(D) * V - 1) & 15) == 0) {
      // 16 bytes per lane and request, four requests in flight per lane: the reads cross PCIe (microseconds each)
      const int64_t nv = width / V;
      for (int64_t c0 = lane; c0 < nv; c0 += 256) {
        Vec<S, V> v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) if (c0 + 64 * u < nv) v[u] = *reinterpret_cast<const Vec<S, V>*>(s + (c0 + 64 * u) * V);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (c0 + 64 * u < nv) {
            Vec<D, V> o;
#pragma unroll
            for (int k = 0; k < V; k++) o.v[k] = store_as<D>((acc_t<D>)load_as<acc_t<S>>(v[u].v[k]));
            *reinterpret_cast<Vec<D, V>*>(d + (c0 + 64 * u) * V) = o;
          }
      }
    } else {
      for (int64_t c = lane; c < width; c += 64) d[c] = store_as<D>((acc_t<D>)load_as<acc_t<S>>(s[c]));
    }
  }
}
// index_fill along dim (IndexFill op, ops.scala:160-177)
// diag(v, k): out[m, m] zero except out[i + max(-k, 0)][i + max(k, 0)] = v[i]   (ATen diag of a vector)
template <class T>
__global__ void diag_embed_kernel(const T* __restrict__ v, T* __restrict__ out, int64_t n, int64_t m, int64_t off_r, int64_t off_c) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[(i + off_r) * m + i + off_c] = v[i];
}
// cross product along a dimension of size 3 of two contiguous tensors of the same shape [outer, 3, inner]
template <class T>
__global__ void cross_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t outer, int64_t inner) {
  using A = acc_t<T>;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= outer * inner) return;
  const int64_t o = i / inner, r = i - o * inner, base = o * 3 * inner + r;
  const A a0 = load_as<A>(a[base]), a1 = load_as<A>(a[base + inner]), a2 = load_as<A>(a[base + 2 * inner]);
  const A b0 = load_as<A>(b[base]), b1 = load_as<A>(b[base + inner]), b2 = load_as<A>(b[base + 2 * inner]);
  out[base] = store_as<T>(a1 * b2 - a2 * b1);
  out[base + inner] = store_as<T>(a2 * b0 - a0 * b2);
  out[base + 2 * inner] = store_as<T>(a0 * b1 - a1 * b0);
}
template <class T>
__global__ void index_fill_kernel(T* __restrict__ out, const int64_t* __restrict__ index, int64_t nidx, int64_t outer, int64_t D, int64_t inner, T value) {
  const int64_t total = outer * nidx * inner;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in = e % inner, k = (e / inner) % nidx, o = e / (inner * nidx);
    const int64_t t = index[k];
    if (t >= 0 && t < D) out[(o * D + t) * inner + in] = value;
  }
}
// repeat_interleave with one count per slice (ATen repeat_interleave.self_Tensor; RepeatInterleave op, ops.scala:484-509): `starts` is
// the exclusive prefix sum of the counts
template <class T>
__global__ void repeat_interleave_tensor_kernel(const T* __restrict__ a, T* __restrict__ out, const int64_t* __restrict__ starts, int64_t outer, int64_t D,
                                                int64_t inner, int64_t Dout) {
  const int64_t total = outer * Dout * inner;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in = e % inner, j = (e / inner) % Dout, o = e / (inner * Dout);
    int64_t lo = 0, hi = D - 1;                      // largest d with starts[d] <= j
    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (starts[mid] <= j) lo = mid; else hi = mid - 1; }
    out[e] = a[(o * D + lo) * inner + in];
  }
}

// ---- top-k along the last dim, k <= 64: one wavefront per row, threshold + compaction ------------------------------------
// (the kNN graph spends its time here: one call per distance block, rows of ~2048 candidates, k = 10)
//   pass 1: every lane takes the minimum of its strided share of the row; a 64-lane bitonic sort of those minima gives
//           tau = the k-th smallest of them, an upper bound of the row's k-th smallest element;
//   pass 2: the (few) elements <= tau are compacted into an LDS candidate list with ballot / popcount;
//   final : one candidate per lane, bitonic sort by (value, index), lanes 0..k-1 write the result.
// Ordering is lexicographic (value, index) as in the k-pass kernel below, so ties resolve identically.  A row with more than 64
// candidates (massive ties) is handled by the k-pass kernel's rounds inside the same wavefront.
template <class A> __device__ __forceinline__ bool tk_before(A v, int i, A w, int j) { return j < 0 || (i >= 0 && (v < w || (v == w && i < j))); }
template <class A> __device__ __forceinline__ void tk_sort64(A& v, int& i, int lane) {
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const A ov = __shfl_xor(v, j, 64);
      const int oi = __shfl_xor(i, j, 64);
      const bool up = (lane & k2) == 0, lower = (lane & j) == 0;
      const bool other_first = tk_before(ov, oi, v, i);
      if ((lower == up) ? other_first : !other_first) { v = ov; i = oi; }
    }
  }
}
template <class T>
__global__ __launch_bounds__(256) void topk_wave_kernel(const T* __restrict__ a, T* __restrict__ vals, int64_t* __restrict__ idxs, int64_t rows,
                                                        int64_t D, int k, int largest) {
  using A = acc_t<T>;
  __shared__ A cv[4][64];
  __shared__ int ci[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * 4 + wid;
  if (r0 >= rows) return;
  const T* row = a + r0 * D;
  // pass 1: lane minima (index -1 = nothing seen).  Four consecutive elements per lane and load (16-byte loads for f32),
  // several loads in flight: with one scalar load per iteration the scan is latency bound (0.3 TB/s measured)
  A mv = A(0);
  int mi = -1;
  const bool vec4 = (D % 4 == 0) && ((uintptr_t)row % (4 * sizeof(T)) == 0);
  if (vec4) {
#pragma unroll 4
    for (int64_t d = (int64_t)lane * 4; d < D; d += 256) {
      const Vec<T, 4> pk = *reinterpret_cast<const Vec<T, 4>*>(row + d);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        A v = load_as<A>(pk.v[e]);
        if (largest) v = -v;
        if (tk_before(v, (int)d + e, mv, mi)) { mv = v; mi = (int)d + e; }
      }
    }
  } else {
    for (int64_t d = lane; d < D; d += 64) {
      A v = load_as<A>(row[d]);
      if (largest) v = -v;
      if (tk_before(v, (int)d, mv, mi)) { mv = v; mi = (int)d; }
    }
  }
  A sv = mv; int si = mi;
  tk_sort64(sv, si, lane);
  const A tau = __shfl(sv, k - 1, 64);
  const int tau_i = __shfl(si, k - 1, 64);
  int count = 0;
  bool overflow = false;
  if (tau_i >= 0) {   // at least k lanes saw an element: compact everything <= tau
    if (vec4) {
      for (int64_t d0 = 0; d0 < D && !overflow; d0 += 256) {
        const int64_t d = d0 + (int64_t)lane * 4;
        Vec<T, 4> pk;
        if (d < D) pk = *reinterpret_cast<const Vec<T, 4>*>(row + d);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          A v = A(0);
          bool hit = false;
          if (d < D) { v = load_as<A>(pk.v[e]); if (largest) v = -v; hit = v <= tau; }
          const unsigned long long m = __ballot(hit);
          if (m == 0) continue;
          const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
          if (hit && pos < 64) { cv[wid][pos] = v; ci[wid][pos] = (int)d + e; }
          count += __popcll(m);
          if (count > 64) overflow = true;
        }
      }
    } else {
      for (int64_t d0 = 0; d0 < D; d0 += 64) {
        const int64_t d = d0 + lane;
        A v = A(0);
        bool hit = false;
        if (d < D) { v = load_as<A>(row[d]); if (largest) v = -v; hit = v <= tau; }
        const unsigned long long m = __ballot(hit);
        if (m == 0) continue;
        const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
        if (hit && pos < 64) { cv[wid][pos] = v; ci[wid][pos] = (int)d; }
        count += __popcll(m);
        if (count > 64) { overflow = true; break; }
      }
    }
  } else {
    overflow = true;   // fewer than k lanes have elements (D < 64 * ... small rows): take the exact slow path
  }
  if (!overflow) {
    A v = lane < count ? cv[wid][lane] : A(0);
    int i = lane < count ? ci[wid][lane] : -1;
    tk_sort64(v, i, lane);
    if (lane < k) {
      vals[r0 * k + lane] = store_as<T>(largest ? -v : v);
      idxs[r0 * k + lane] = i;
    }
    return;
  }
  // slow path: k rounds of "best element strictly after the previous one"
  A lv = A(0);
  int li = -1;
  for (int r = 0; r < k; r++) {
    A bv = A(0);
    int bi = -1;
    for (int64_t d = lane; d < D; d += 64) {
      A v = load_as<A>(row[d]);
      if (largest) v = -v;
      const bool after = (li < 0) || (v > lv) || (v == lv && (int)d > li);
      if (after && tk_before(v, (int)d, bv, bi)) { bv = v; bi = (int)d; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const A ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (tk_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    lv = bv; li = bi;
    if (lane == 0) {
      vals[r0 * k + r] = store_as<T>(largest ? -bv : bv);
      idxs[r0 * k + r] = bi;
    }
  }
}

// ---- top-k along the last dim: k rounds of lexicographic (value, index) arg-min/max ---------------
template <class T>
__global__ __launch_bounds__(256) void topk_kernel(const T* __restrict__ a, T* __restrict__ vals, int64_t* __restrict__ idxs, int64_t D,
                                                   int64_t k, int largest) {
  using A = acc_t<T>;
  __shared__ A sv[4];
  __shared__ int64_t si[4];
  __shared__ A last_v;
  __shared__ int64_t last_i;
  const T* row = a + (int64_t)blockIdx.x * D;
  if (threadIdx.x == 0) { last_v = 0; last_i = -1; }
  __syncthreads();
  for (int64_t r = 0; r < k; r++) {
    const A lv = last_v;
    const int64_t li = last_i;
    // best candidate strictly after (lv, li) in the ordering
    bool has = false;
    A bv = 0;
    int64_t bi = -1;
    for (int64_t d = threadIdx.x; d < D; d += blockDim.x) {
      A v = load_as<A>(row[d]);
      if (largest) v = -v;
      const bool after = (li < 0) || (v > lv) || (v == lv && d > li);
      if (!after) continue;
      if (!has || v < bv || (v == bv && d < bi)) { has = true; bv = v; bi = d; }
    }
    // wave + block reduction on (bv, bi)
    for (int off = 32; off > 0; off >>= 1) {
      const A ov = __shfl_xor(bv, off, 64);
      const int64_t oi = __shfl_xor(bi, off, 64);
      const int oh = __shfl_xor((int)has, off, 64);
      if (oh && (!has || ov < bv || (ov == bv && oi < bi))) { has = true; bv = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { sv[wid] = bv; si[wid] = has ? bi : -1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      A fv = 0; int64_t fi = -1;
      for (int w = 0; w < (int)(blockDim.x >> 6); w++)
        if (si[w] >= 0 && (fi < 0 || sv[w] < fv || (sv[w] == fv && si[w] < fi))) { fv = sv[w]; fi = si[w]; }
      last_v = fv; last_i = fi;
      vals[(int64_t)blockIdx.x * k + r] = store_as<T>(largest ? -fv : fv);
      idxs[(int64_t)blockIdx.x * k + r] = fi;
    }
    __syncthreads();
  }
}

// ---- Philox4x32-10: philox.h ---------------------------------------------------------------------
// mode 0 uniform [0,1), 1 normal(mean,std), 2 randint [low, high), 3 bernoulli keep-mask scaled by 1/(1-p) multiplied into out
template <class T, int MODE>
__global__ void rng_kernel(T* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset, double p0, double p1) {
  using A = acc_t<T>;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  Philox ph(seed, (uint64_t)tid, offset);
  for (int64_t i = tid * 2; i < n; i += (int64_t)gridDim.x * blockDim.x * 2) {
    const uint4 r = ph.next();
    double a = u01(r.x, r.y), b = u01(r.z, r.w);
    double v0, v1;
    if (MODE == 1) {
      const double rad = sqrt(-2.0 * log(1.0 - a)), ang = 6.283185307179586476925 * b;
      v0 = p0 + p1 * rad * cos(ang); v1 = p0 + p1 * rad * sin(ang);
    } else if (MODE == 2) {
      v0 = floor(p0 + a * (p1 - p0)); v1 = floor(p0 + b * (p1 - p0));
    } else if (MODE == 3) {
      v0 = (a >= p0) ? 1.0 / (1.0 - p0) : 0.0; v1 = (b >= p0) ? 1.0 / (1.0 - p0) : 0.0;
    } else { v0 = a; v1 = b; }
    if (MODE == 3) {
      out[i] = store_as<T>((A)(load_as<A>(out[i]) * (A)v0));
      if (i + 1 < n) out[i + 1] = store_as<T>((A)(load_as<A>(out[i + 1]) * (A)v1));
    } else {
      out[i] = store_as<T>((A)v0);
      if (i + 1 < n) out[i + 1] = store_as<T>((A)v1);
    }
  }
}

static void split3(const Tensor* a, int64_t dim, int64_t& outer, int64_t& D, int64_t& inner) {
  const int64_t d = wrap_dim(dim, a->ndim);
  outer = inner = 1;
  D = a->ndim ? a->sizes[d] : 1;
  for (int i = 0; i < d; i++) outer *= a->sizes[i];
  for (int i = (int)d + 1; i < a->ndim; i++) inner *= a->sizes[i];
}
static void check_index(const Tensor* index) {
  check_device_tensor(index, "index");
  LAMP_CHECK(index->dtype == kI64, "index must be int64, got " << index->describe());
  LAMP_CHECK(index->ndim <= 1, "index must be a vector");
}

template <int MODE> static Tensor* rng_new(const int64_t* sizes, int ndim, int dtype, int device, double p0, double p1) {
  if (device < 0) {
    // lamp's CPU device (STen.rand / randint with CPU options, e.g. Umap.umap's default device): drawn by the same Philox kernel on
    // the calling thread's current GPU and copied to host memory - one generator, one stream of numbers, whichever device is named
    Hold d(rng_new<MODE>(sizes, ndim, dtype, current_device(), p0, p1));
    Hold h(new_tensor(sizes, ndim, dtype, -1));
    if (d->numel() > 0) copy_into(h.get(), d.get());
    return h.take();
  }
  Hold t(new_tensor(sizes, ndim, dtype, device));
  const int64_t n = t->numel();
  if (n) {
    const int grid = grid_for((n + 1) / 2, 256);
    const uint64_t off = next_philox_offset((uint64_t)((n + 1) / 2 / ((int64_t)grid * 256) + 2));
    LAMP_DISPATCH_ALL(dtype, T, hipLaunchKernelGGL((rng_kernel<T, MODE>), dim3(grid), dim3(256), 0, current_stream(device), t->ptr<T>(), n,
                                                   philox_seed(), off, p0, p1));
    LAMP_LAUNCH_CHECK();
  }
  return t.take();
}

}  // namespace lamp

using namespace lamp;

extern "C" {

int lamp_index_select(lamp_tensor** out, const lamp_tensor* a, int64_t dim, const lamp_tensor* index) {
  LAMP_API_BEGIN
  if (a && !a->is_device()) {
    // lamp's CPU device: BatchStream gathers a minibatch of the host-resident data set into its pinned buffer
    // (BatchStream.scala:540-573): a row copy per index on the host
    LAMP_CHECK(index && !index->is_device() && index->dtype == kI64 && index->ndim <= 1, "index_select on a host tensor needs a host int64 index vector");
    Hold ac(contiguous(a)), ic(contiguous(index));
    int64_t outer, D, inner;
    split3(a, dim, outer, D, inner);
    const int64_t J = index->numel();
    std::vector<int64_t> oshape = a->shape();
    if (a->ndim) oshape[wrap_dim(dim, a->ndim)] = J;
    Hold r(new_tensor(oshape, a->dtype, -1));
    const size_t row = (size_t)inner * a->itemsize();
    const char* src = static_cast<const char*>(static_cast<const Tensor*>(ac.get())->raw());
    char* dst = static_cast<char*>(r->data());
    const int64_t* ix = static_cast<const Tensor*>(ic.get())->ptr<int64_t>();
    for (int64_t o = 0; o < outer; o++)
      for (int64_t j = 0; j < J; j++) {
        LAMP_CHECK(ix[j] >= 0 && ix[j] < D, "index_select: index " << ix[j] << " out of range for a dimension of size " << D);
        memcpy(dst + ((size_t)o * J + j) * row, src + ((size_t)o * D + ix[j]) * row, row);
      }
    *out = r.take();
    return 0;
  }
  check_device_tensor(a, "self"); check_index(index);
  Hold ac(contiguous(a)), ic(contiguous(index));
  int64_t outer, D, inner;
  split3(a, dim, outer, D, inner);
  const int64_t J = index->numel();
  std::vector<int64_t> oshape = a->shape();
  if (a->ndim) oshape[wrap_dim(dim, a->ndim)] = J;
  Hold r(new_tensor(oshape, a->dtype, a->device()));
  const int64_t total = outer * J * inner;
  if (total) {
    const int64_t row_bytes = inner * (int64_t)a->itemsize();
    const void* ap = static_cast<const Tensor*>(ac.get())->raw();
    if (outer == 1 && row_bytes % 16 == 0 && row_bytes / 16 < (1 << 30) && (((uintptr_t)ap | (uintptr_t)r->raw()) & 15) == 0) {
      const int ppr = (int)(row_bytes / 16);
      hipLaunchKernelGGL(index_select_rows_vec_kernel, dim3(grid_for(J * ppr, 256)), dim3(256), 0, current_stream(a->device()), (const uint4*)ap,
                         static_cast<const Tensor*>(ic.get())->ptr<int64_t>(), (uint4*)r->data(), D, ppr, J);
    } else {
      LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((index_select_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                        current_stream(a->device()), ac->ptr<T>(), ic->ptr<int64_t>(), r->ptr<T>(), outer, D, inner, J));
    }
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_index_add_(lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* source) {
  LAMP_API_BEGIN
  check_device_tensor(self, "self"); check_index(index); check_device_tensor(source, "source");
  LAMP_CHECK(self->is_contiguous(), "index_add_ needs a contiguous destination");
  LAMP_CHECK(self->dtype == source->dtype, "index_add: dtype mismatch");
  int64_t outer, D, inner;
  split3(self, dim, outer, D, inner);
  const int64_t J = index->numel();
  std::vector<int64_t> sshape = self->shape();
  if (self->ndim) sshape[wrap_dim(dim, self->ndim)] = J;
  LAMP_CHECK(source->shape() == sshape, "index_add: source " << source->describe() << " has the wrong shape");
  Hold sc(contiguous(source)), ic(contiguous(index));
  const int64_t total = outer * J * inner;
  if (total) {
    LAMP_DISPATCH_ALL(self->dtype, T, hipLaunchKernelGGL((index_add_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                         current_stream(self->device()), self->ptr<T>(), ic->ptr<int64_t>(), sc->ptr<T>(), outer, D, inner, J));
    LAMP_LAUNCH_CHECK();
  }
  LAMP_API_END
}
int lamp_index_add(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* source) {
  LAMP_API_BEGIN
  check_device_tensor(self, "self");
  Hold r(new_like(self));
  copy_into(r.get(), self);
  LAMP_CHECK(lamp_index_add_(r.get(), dim, index, source) == 0, lamp_last_error());
  *out = r.take();
  LAMP_API_END
}
int lamp_masked_select(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* mask) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_device_tensor(mask, "mask");
  LAMP_CHECK(mask->dtype == kBool || mask->dtype == kU8, "masked_select expects a bool mask");
  auto shape = broadcast_shapes(a->shape(), mask->shape());
  Hold ae(new_tensor(shape, a->dtype, a->device())), me(new_tensor(shape, mask->dtype, a->device()));
  copy_into(ae.get(), a);
  copy_into(me.get(), mask);
  const int64_t n = ae->numel();
  const int64_t nblocks = (n + MS_TILE - 1) / MS_TILE;
  hipStream_t st = current_stream(a->device());
  int64_t cs[1] = {nblocks + 1};
  Hold counts(new_tensor(cs, 1, kI64, a->device()));
  int64_t total = 0;
  if (n) {
    hipLaunchKernelGGL(ms_count_kernel, dim3((unsigned)nblocks), dim3(MS_BLOCK), 0, st, me->ptr<uint8_t>(), n, counts->ptr<int64_t>());
    hipLaunchKernelGGL(ms_scan_kernel, dim3(1), dim3(1), 0, st, counts->ptr<int64_t>(), nblocks, counts->ptr<int64_t>() + nblocks);
    LAMP_LAUNCH_CHECK();
    HIP_CHECK(hipMemcpyAsync(&total, counts->ptr<int64_t>() + nblocks, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
  int64_t os[1] = {total};
  Hold r(new_tensor(os, 1, a->dtype, a->device()));
  if (total) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((ms_scatter_kernel<T>), dim3((unsigned)nblocks), dim3(MS_BLOCK), 0, st, ae->ptr<T>(),
                                                      me->ptr<uint8_t>(), n, counts->ptr<int64_t>(), r->ptr<T>()));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_repeat_interleave(lamp_tensor** out, const lamp_tensor* a, int64_t repeats, int64_t dim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  LAMP_CHECK(repeats >= 0, "repeats must be non-negative");
  Hold ac(contiguous(a));
  int64_t outer, D, inner;
  split3(a, dim, outer, D, inner);
  std::vector<int64_t> oshape = a->shape();
  if (a->ndim) oshape[wrap_dim(dim, a->ndim)] = D * repeats; else oshape = {repeats};
  Hold r(new_tensor(oshape, a->dtype, a->device()));
  const int64_t total = r->numel();
  if (total) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((repeat_interleave_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                      current_stream(a->device()), ac->ptr<T>(), r->ptr<T>(), outer, D, inner, repeats));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_masked_scatter(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* mask, const lamp_tensor* source) {
  LAMP_API_BEGIN
  check_device_tensor(self, "self"); check_device_tensor(mask, "mask"); check_device_tensor(source, "source");
  LAMP_CHECK(mask->dtype == kBool || mask->dtype == kU8, "masked_scatter expects a bool mask");
  LAMP_CHECK(source->dtype == self->dtype, "masked_scatter: source dtype " << source->describe() << " differs from self " << self->describe());
  Hold r(new_tensor(self->shape(), self->dtype, self->device())), me(new_tensor(self->shape(), mask->dtype, self->device()));
  copy_into(r.get(), self);
  copy_into(me.get(), mask);                      // mask broadcasts to self
  Hold sc(contiguous(source));
  const int64_t n = r->numel();
  if (n) {
    const int64_t nblocks = (n + MS_TILE - 1) / MS_TILE;
    hipStream_t st = current_stream(self->device());
    int64_t cs[1] = {nblocks + 1};
    Hold counts(new_tensor(cs, 1, kI64, self->device()));
    hipLaunchKernelGGL(ms_count_kernel, dim3((unsigned)nblocks), dim3(MS_BLOCK), 0, st, me->ptr<uint8_t>(), n, counts->ptr<int64_t>());
    hipLaunchKernelGGL(ms_scan_kernel, dim3(1), dim3(1), 0, st, counts->ptr<int64_t>(), nblocks, counts->ptr<int64_t>() + nblocks);
    LAMP_DISPATCH_ALL(self->dtype, T, hipLaunchKernelGGL((ms_masked_scatter_kernel<T>), dim3((unsigned)nblocks), dim3(MS_BLOCK), 0, st, r->ptr<T>(),
                                                         me->ptr<uint8_t>(), n, counts->ptr<int64_t>(), sc->ptr<T>(), sc->numel()));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
static GsGeom gs_geom(const Tensor* a, const Tensor* index, const Tensor* b, int64_t dim, bool scatter) {
  LAMP_CHECK(index->dtype == kI64, "index must be int64, got " << index->describe());
  LAMP_CHECK(a->ndim == index->ndim && b->ndim == index->ndim && index->ndim >= 1, "gather / scatter_add: self, index and source must have one rank");
  GsGeom g;
  g.ndim = index->ndim;
  g.dim = (int)wrap_dim(dim, index->ndim);
  for (int d = 0; d < g.ndim; d++) {
    g.isz[d] = index->sizes[d]; g.istr[d] = index->strides[d]; g.astr[d] = a->strides[d]; g.bstr[d] = b->strides[d];
    if (d != g.dim) LAMP_CHECK(index->sizes[d] <= a->sizes[d] && index->sizes[d] <= b->sizes[d], "gather / scatter_add: index is larger than the tensors along dim " << d);
  }
  g.dlim = scatter ? a->sizes[g.dim] : b->sizes[g.dim];
  if (scatter) LAMP_CHECK(index->sizes[g.dim] <= b->sizes[g.dim], "scatter_add: index is longer than src along dim");
  return g;
}
int lamp_gather(lamp_tensor** out, const lamp_tensor* a, int64_t dim, const lamp_tensor* index) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_device_tensor(index, "index");
  Hold r(new_tensor(index->shape(), a->dtype, a->device()));
  const GsGeom g = gs_geom(r.get(), index, a, dim, false);
  const int64_t n = index->numel();
  if (n) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((gather_scatter_kernel<T, false>), dim3(grid_for(n, 256)), dim3(256), 0, current_stream(a->device()),
                                                      r->ptr<T>(), index->ptr<int64_t>(), a->ptr<T>(), n, g, device_assert_word(a->device())));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_scatter_add(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* src) {
  LAMP_API_BEGIN
  check_device_tensor(self, "self"); check_device_tensor(index, "index"); check_device_tensor(src, "src");
  LAMP_CHECK(self->dtype == src->dtype, "scatter_add: dtype mismatch");
  Hold r(new_tensor(self->shape(), self->dtype, self->device()));
  copy_into(r.get(), self);
  const GsGeom g = gs_geom(r.get(), index, src, dim, true);
  const int64_t n = index->numel();
  if (n) {
    LAMP_DISPATCH_ALL(self->dtype, T, hipLaunchKernelGGL((gather_scatter_kernel<T, true>), dim3(grid_for(n, 256)), dim3(256), 0, current_stream(self->device()),
                                                         r->ptr<T>(), index->ptr<int64_t>(), src->ptr<T>(), n, g, device_assert_word(self->device())));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
static void scatter_set(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* src, double value) {
  check_device_tensor(self, "self"); check_device_tensor(index, "index");
  if (src) { check_device_tensor(src, "src"); LAMP_CHECK(self->dtype == src->dtype, "scatter: dtype mismatch"); }
  Hold r(new_tensor(self->shape(), self->dtype, self->device()));
  copy_into(r.get(), self);
  const GsGeom g = gs_geom(r.get(), index, src ? src : index, dim, true);
  const int64_t n = index->numel();
  if (n) {
    hipStream_t st = current_stream(self->device());
    if (src) {
      LAMP_DISPATCH_ALL(self->dtype, T, hipLaunchKernelGGL((scatter_set_kernel<T, false>), dim3(grid_for(n, 256)), dim3(256), 0, st, r->ptr<T>(), index->ptr<int64_t>(),
                                                           src->ptr<T>(), T{}, n, g, device_assert_word(self->device())));
    } else {
      LAMP_DISPATCH_ALL(self->dtype, T, hipLaunchKernelGGL((scatter_set_kernel<T, true>), dim3(grid_for(n, 256)), dim3(256), 0, st, r->ptr<T>(), index->ptr<int64_t>(),
                                                           (const T*)nullptr, store_as<T>((acc_t<T>)value), n, g, device_assert_word(self->device())));
    }
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
}
int lamp_scatter(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* src) {
  LAMP_API_BEGIN scatter_set(out, self, dim, index, src, 0.0); LAMP_API_END
}
int lamp_scatter_value(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, double value) {
  LAMP_API_BEGIN scatter_set(out, self, dim, index, nullptr, value); LAMP_API_END
}
// out = self with out[indices[0][p], .., indices[K-1][p], ...] (+)= values[p, ...]: the index tensors are broadcast to one shape I, values to
// I x (the remaining dimensions of self)
static void index_put_impl(lamp_tensor** out, const lamp_tensor* self, lamp_tensor* const* indices, int n, const lamp_tensor* values, int accumulate) {
  check_device_tensor(self, "self"); check_device_tensor(values, "values");
  LAMP_CHECK(n >= 1 && n <= self->ndim, "index_put: " << n << " index tensors for " << self->describe());
  LAMP_CHECK(values->dtype == self->dtype, "index_put: values " << values->describe() << " differ in dtype from self " << self->describe());
  std::vector<int64_t> ishape;
  for (int k = 0; k < n; k++) {
    LAMP_CHECK(indices[k] != nullptr, "index_put: undefined index tensors (None) are not supported");
    check_device_tensor(indices[k], "index");
    LAMP_CHECK(indices[k]->dtype == kI64, "index_put: indices must be int64, got " << indices[k]->describe());
    ishape = k == 0 ? indices[k]->shape() : broadcast_shapes(ishape, indices[k]->shape());
  }
  Hold r(new_tensor(self->shape(), self->dtype, self->device()));
  copy_into(r.get(), self);
  IpGeom g;
  g.K = n;
  std::vector<Hold> idx;
  int64_t NI = 1;
  for (int64_t v : ishape) NI *= v;
  for (int k = 0; k < n; k++) {
    Hold e(new_tensor(ishape, kI64, self->device()));
    copy_into(e.get(), indices[k]);                       // broadcast
    g.idx[k] = e->ptr<int64_t>(); g.size[k] = r->sizes[k]; g.stride[k] = r->strides[k];
    idx.push_back(std::move(e));
  }
  std::vector<int64_t> vshape = ishape;
  int64_t inner = 1;
  for (int d = n; d < self->ndim; d++) { vshape.push_back(self->sizes[d]); inner *= self->sizes[d]; }
  Hold ve(new_tensor(vshape, self->dtype, self->device()));
  copy_into(ve.get(), values);                            // broadcast
  if (NI * inner) {
    hipStream_t st = current_stream(self->device());
    if (accumulate) {
      LAMP_DISPATCH_ALL(self->dtype, T, hipLaunchKernelGGL((index_put_kernel<T, true>), dim3(grid_for(NI * inner, 256)), dim3(256), 0, st, r->ptr<T>(), ve->ptr<T>(), g, NI,
                                                           inner, device_assert_word(self->device())));
    } else {
      LAMP_DISPATCH_ALL(self->dtype, T, hipLaunchKernelGGL((index_put_kernel<T, false>), dim3(grid_for(NI * inner, 256)), dim3(256), 0, st, r->ptr<T>(), ve->ptr<T>(), g, NI,
                                                           inner, device_assert_word(self->device())));
    }
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
}
int lamp_index_put(lamp_tensor** out, const lamp_tensor* self, lamp_tensor* const* indices, int n, const lamp_tensor* values, int accumulate) {
  LAMP_API_BEGIN index_put_impl(out, self, indices, n, values, accumulate); LAMP_API_END
}
// ATen.put: self viewed as one long vector
int lamp_put(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* index, const lamp_tensor* values, int accumulate) {
  LAMP_API_BEGIN
  check_device_tensor(self, "self"); check_device_tensor(index, "index"); check_device_tensor(values, "values");
  LAMP_CHECK(index->numel() == values->numel(), "put: index " << index->describe() << " and values " << values->describe() << " differ in length");
  Hold sc(contiguous(self));
  int64_t flat[1] = {sc->numel()}, one[1] = {1}, nn[1] = {index->numel()};
  Hold sf(new_view(sc.get(), flat, one, 1, sc->offset));
  Hold ic(contiguous(index)), vc(contiguous(values));
  Hold i1(new_view(ic.get(), nn, one, 1, ic->offset)), v1(new_view(vc.get(), nn, one, 1, vc->offset));
  lamp_tensor* ids[1] = {i1.get()};
  lamp_tensor* o = nullptr;
  index_put_impl(&o, sf.get(), ids, 1, v1.get(), accumulate);
  Hold oh(o);
  int64_t cst[kMaxDims], run = 1;
  for (int i = self->ndim - 1; i >= 0; i--) { cst[i] = run; run *= self->sizes[i]; }
  *out = new_view(oh.get(), self->sizes, cst, self->ndim, oh->offset);
  LAMP_API_END
}
// ATen.index_copy: out = self with out.select(dim, index[i]) = source.select(dim, i)
int lamp_index_copy(lamp_tensor** out, const lamp_tensor* self, int64_t dim, const lamp_tensor* index, const lamp_tensor* source) {
  LAMP_API_BEGIN
  check_device_tensor(self, "self"); check_index(index); check_device_tensor(source, "source");
  LAMP_CHECK(self->ndim >= 1 && source->ndim == self->ndim && source->dtype == self->dtype, "index_copy: source " << source->describe() << " does not match self " << self->describe());
  const int64_t d = wrap_dim(dim, self->ndim);
  LAMP_CHECK(source->sizes[d] == index->numel(), "index_copy: source has " << source->sizes[d] << " slices along dim, index " << index->numel());
  // move dim first on both, then it is an index_put with one index tensor
  lamp_tensor *st = nullptr, *srt = nullptr;
  LAMP_CHECK(lamp_transpose(&st, self, 0, d) == 0, lamp_last_error());
  Hold sth(st);
  LAMP_CHECK(lamp_transpose(&srt, source, 0, d) == 0, lamp_last_error());
  Hold srth(srt);
  Hold sc(contiguous(sth.get()));
  Hold i1(contiguous(index));
  lamp_tensor* ids[1] = {i1.get()};
  lamp_tensor* o = nullptr;
  index_put_impl(&o, sc.get(), ids, 1, srth.get(), 0);
  Hold oh(o);
  lamp_tensor* back = nullptr;
  LAMP_CHECK(lamp_transpose(&back, oh.get(), 0, d) == 0, lamp_last_error());
  Hold bh(back);
  *out = contiguous(bh.get());
  LAMP_API_END
}
static void tri_impl(lamp_tensor* dst_or_null, lamp_tensor** out, const lamp_tensor* a, int64_t diagonal, int lower) {
  check_device_tensor(a, "self");
  LAMP_CHECK(a->ndim >= 2, "tril / triu expects at least a matrix, got " << a->describe());
  Hold ac(contiguous(a));
  Hold r(new_tensor(a->shape(), a->dtype, a->device()));
  const int64_t total = ac->numel();
  if (total) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((tri_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, current_stream(a->device()), ac->ptr<T>(), r->ptr<T>(), total,
                                                      a->sizes[a->ndim - 2], a->sizes[a->ndim - 1], diagonal, lower));
    LAMP_LAUNCH_CHECK();
  }
  if (dst_or_null) copy_into(dst_or_null, r.get()); else *out = r.take();
}
int lamp_tril(lamp_tensor** out, const lamp_tensor* a, int64_t diagonal) { LAMP_API_BEGIN tri_impl(nullptr, out, a, diagonal, 1); LAMP_API_END }
int lamp_triu(lamp_tensor** out, const lamp_tensor* a, int64_t diagonal) { LAMP_API_BEGIN tri_impl(nullptr, out, a, diagonal, 0); LAMP_API_END }
int lamp_tril_out(lamp_tensor* out, const lamp_tensor* a, int64_t diagonal) {
  LAMP_API_BEGIN
  check_device_tensor(out, "out");
  LAMP_CHECK(out->shape() == a->shape() && out->dtype == a->dtype, "tril_out: out " << out->describe() << " does not match self " << a->describe());
  tri_impl(out, nullptr, a, diagonal, 1);                 // through a temporary: out may be self (STen.tril_)
  LAMP_API_END
}
// ATen.diagonal: a VIEW of the diagonal of dimensions (dim1, dim2), appended as the last dimension (STen.scala:1885-1886)
int lamp_diagonal(lamp_tensor** out, const lamp_tensor* a, int64_t offset, int64_t dim1, int64_t dim2) {
  LAMP_API_BEGIN
  LAMP_CHECK(a != nullptr, "self is null");
  LAMP_CHECK(a->ndim >= 2, "diagonal expects at least a matrix, got " << a->describe());
  const int64_t d1 = wrap_dim(dim1, a->ndim), d2 = wrap_dim(dim2, a->ndim);
  LAMP_CHECK(d1 != d2, "diagonal: the two dimensions must differ");
  const int64_t r0 = offset < 0 ? -offset : 0, c0 = offset > 0 ? offset : 0;
  const int64_t n = std::max<int64_t>(0, std::min(a->sizes[d1] - r0, a->sizes[d2] - c0));
  int64_t sz[kMaxDims], st[kMaxDims];
  int nd = 0;
  for (int i = 0; i < a->ndim; i++) if (i != d1 && i != d2) { sz[nd] = a->sizes[i]; st[nd] = a->strides[i]; nd++; }
  sz[nd] = n; st[nd] = a->strides[d1] + a->strides[d2]; nd++;
  *out = new_view(a, sz, st, nd, a->offset + (n > 0 ? r0 * a->strides[d1] + c0 * a->strides[d2] : 0));
  LAMP_API_END
}
// ATen.trace: the sum of the main diagonal of a matrix (STen.scala:1322)
int lamp_trace(lamp_tensor** out, const lamp_tensor* a) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  LAMP_CHECK(a->ndim == 2, "trace expects a matrix, got " << a->describe());
  lamp_tensor* dg = nullptr;
  LAMP_CHECK(lamp_diagonal(&dg, a, 0, 0, 1) == 0, lamp_last_error());
  Hold dh(dg);
  Hold dc(contiguous(dh.get()));
  lamp_tensor* s_ = nullptr;
  LAMP_CHECK(lamp_sum_all(&s_, dc.get()) == 0, lamp_last_error());
  *out = s_;
  LAMP_API_END
}
// out[i, ...] = (out_dtype) pinned[index[i], ...]: `pinned` lives in page-locked host memory, `index` (int64 vector) and the result on `index`'s
// device; runs on that device's current stream.  The minibatch gather of a host-resident data set (BatchStream.scala:539-556 gathers on the
// host, stages in a pinned buffer and copies: here the GPU reads the rows it wants over PCIe).
int lamp_index_select_pinned(lamp_tensor** out, const lamp_tensor* pinned, const lamp_tensor* index, int out_dtype) {
  LAMP_API_BEGIN
  LAMP_CHECK(pinned != nullptr && !pinned->is_device() && pinned->st && pinned->st->pinned, "index_select_pinned: the source must be a pinned host tensor (lamp_pin_memory)");
  check_index(index);
  LAMP_CHECK(pinned->ndim >= 1 && pinned->is_contiguous(), "index_select_pinned: the source must be contiguous");
  const int odt = out_dtype < 0 ? pinned->dtype : out_dtype;
  std::vector<int64_t> oshape = pinned->shape();
  oshape[0] = index->numel();
  Hold r(new_tensor(oshape, odt, index->device()));
  Hold ic(contiguous(index));
  const int64_t rows = index->numel(), width = pinned->sizes[0] ? pinned->numel() / pinned->sizes[0] : 0;
  if (rows * width) {
    hipStream_t st = current_stream(index->device());
    // The reads cross PCIe: ~50 GB/s whatever the grid, as long as ~100 KB of requests are in flight - a few dozen workgroups do that.
    // A grid that covers the chip (tried first: two workgroups on every CU for the 0.5 ms a 25 MB batch takes) starved the training step it is
    // meant to run beside: the step's first kernel waited 300 us for a slot and then ran 10 x slower (rocprofv3 timeline, scripts/epoch_overlap_probe.py)
    static const int64_t gp_wgs = [] { const char* e = getenv("LAMP_PINNED_GATHER_WGS"); return (int64_t)(e ? std::max(1, atoi(e)) : 48); }();
    const dim3 grid((unsigned)std::min<int64_t>((rows + 3) / 4, gp_wgs));
    int* aw = device_assert_word(index->device());
#define GP_LAUNCH(S_, D_) hipLaunchKernelGGL((gather_pinned_rows_kernel<S_, D_>), grid, dim3(256), 0, st, pinned->ptr<S_>(), ic->ptr<int64_t>(), r->ptr<D_>(), rows, width, pinned->sizes[0], aw)
    if (pinned->dtype == odt) { LAMP_DISPATCH_ALL(odt, T, GP_LAUNCH(T, T)); }
    else if (pinned->dtype == kF32 && odt == kBF16) GP_LAUNCH(float, bf16_t);
    else if (pinned->dtype == kU8 && odt == kBF16) GP_LAUNCH(uint8_t, bf16_t);
    else if (pinned->dtype == kU8 && odt == kF32) GP_LAUNCH(uint8_t, float);
    else if (pinned->dtype == kF64 && odt == kF32) GP_LAUNCH(double, float);
    else LAMP_CHECK(false, "index_select_pinned: no conversion from " << dtype_name(pinned->dtype) << " to " << dtype_name(odt));
#undef GP_LAUNCH
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_index_fill(lamp_tensor** out, const lamp_tensor* a, int64_t dim, const lamp_tensor* index, double value) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_index(index);
  Hold r(new_tensor(a->shape(), a->dtype, a->device()));
  copy_into(r.get(), a);
  Hold ic(contiguous(index));
  int64_t outer, D, inner;
  split3(a, dim, outer, D, inner);
  const int64_t total = outer * ic->numel() * inner;
  if (total) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((index_fill_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, current_stream(a->device()), r->ptr<T>(),
                                                      ic->ptr<int64_t>(), ic->numel(), outer, D, inner, store_as<T>((acc_t<T>)value)));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_diag(lamp_tensor** out, const lamp_tensor* a, int64_t diagonal) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  LAMP_CHECK(a->ndim == 1 || a->ndim == 2, "diag expects a vector or a matrix, got " << a->describe());
  if (a->ndim == 1) {
    const int64_t n = a->sizes[0], m = n + (diagonal < 0 ? -diagonal : diagonal);
    Hold ac(contiguous(a));
    int64_t sh[2] = {m, m};
    Hold r(new_tensor(sh, 2, a->dtype, a->device()));
    fill_zero(r.get());
    if (n) {
      LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((diag_embed_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, current_stream(a->device()),
                                                        ac->ptr<T>(), r->ptr<T>(), n, m, diagonal < 0 ? -diagonal : (int64_t)0, diagonal > 0 ? diagonal : (int64_t)0));
      LAMP_LAUNCH_CHECK();
    }
    *out = r.take();
  } else {
    // the k-th diagonal of a matrix: a strided view (stride0 + stride1), copied out
    const int64_t R = a->sizes[0], Cc = a->sizes[1];
    const int64_t r0 = diagonal < 0 ? -diagonal : 0, c0 = diagonal > 0 ? diagonal : 0;
    const int64_t n = std::max<int64_t>(0, std::min(R - r0, Cc - c0));
    int64_t sz[1] = {n}, st[1] = {a->strides[0] + a->strides[1]};
    Hold v(new_view(a, sz, st, 1, a->offset + r0 * a->strides[0] + c0 * a->strides[1]));
    Hold r(new_tensor(sz, 1, a->dtype, a->device()));
    copy_into(r.get(), v.get());
    *out = r.take();
  }
  LAMP_API_END
}
int lamp_cross(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b, int64_t dim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_device_tensor(b, "other"); check_same_device(a, b);
  LAMP_CHECK(a->dtype == b->dtype && a->shape() == b->shape(), "cross: operands differ: " << a->describe() << " vs " << b->describe());
  int64_t outer, D, inner;
  split3(a, dim, outer, D, inner);
  LAMP_CHECK(D == 3, "cross: dimension " << dim << " has size " << D << ", expected 3");
  Hold ac(contiguous(a)), bc(contiguous(b));
  Hold r(new_tensor(a->shape(), a->dtype, a->device()));
  if (outer * inner) {
    LAMP_DISPATCH_FLOAT(a->dtype, T, hipLaunchKernelGGL((cross_kernel<T>), dim3(grid_for(outer * inner, 256)), dim3(256), 0, current_stream(a->device()),
                                                        ac->ptr<T>(), bc->ptr<T>(), r->ptr<T>(), outer, inner));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_repeat_interleave_tensor(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* repeats, int64_t dim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_index(repeats);
  int64_t outer, D, inner;
  split3(a, dim, outer, D, inner);
  LAMP_CHECK(repeats->numel() == D, "repeat_interleave: " << repeats->numel() << " counts for " << D << " slices");
  // the output size is the sum of the counts: one small device -> host copy (as ATen)
  std::vector<int64_t> rep((size_t)D), starts((size_t)D);
  Hold rc(contiguous(repeats));
  if (D) LAMP_CHECK(lamp_copy_to_host(rc.get(), rep.data(), (size_t)D * 8) == 0, lamp_last_error());
  int64_t Dout = 0;
  for (int64_t d = 0; d < D; d++) { LAMP_CHECK(rep[d] >= 0, "repeats must be non-negative"); starts[d] = Dout; Dout += rep[d]; }
  int64_t dd[1] = {D};
  Hold st_t(new_tensor(dd, 1, kI64, a->device()));
  if (D) LAMP_CHECK(lamp_copy_from_host(st_t.get(), starts.data(), (size_t)D * 8) == 0, lamp_last_error());
  Hold ac(contiguous(a));
  std::vector<int64_t> oshape = a->shape();
  if (a->ndim) oshape[wrap_dim(dim, a->ndim)] = Dout; else oshape = {Dout};
  Hold r(new_tensor(oshape, a->dtype, a->device()));
  const int64_t total = r->numel();
  if (total) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((repeat_interleave_tensor_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                                      current_stream(a->device()), ac->ptr<T>(), r->ptr<T>(), st_t->ptr<int64_t>(), outer, D, inner, Dout));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_topk(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t k, int64_t dim, int largest, int sorted) {
  LAMP_API_BEGIN
  (void)sorted;  // results always come back sorted (a legal answer for sorted=false)
  check_device_tensor(a, "self");
  LAMP_CHECK(a->ndim >= 1, "topk of a 0-dim tensor");
  const int64_t d = wrap_dim(dim, a->ndim);
  LAMP_CHECK(k >= 0 && k <= a->sizes[d], "topk: k = " << k << " out of range for dimension of size " << a->sizes[d]);
  // move `dim` last
  lamp_tensor* tr = nullptr;
  LAMP_CHECK(lamp_transpose(&tr, a, d, a->ndim - 1) == 0, lamp_last_error());
  Hold trh(tr);
  Hold ac(contiguous(tr));
  const int64_t D = ac->sizes[ac->ndim - 1];
  const int64_t rows = D ? ac->numel() / D : 0;
  std::vector<int64_t> oshape = ac->shape();
  oshape.back() = k;
  Hold v(new_tensor(oshape, a->dtype, a->device())), ix(new_tensor(oshape, kI64, a->device()));
  if (rows && k) {
    if (k <= 64 && D < (int64_t)1 << 31) {
      LAMP_DISPATCH_FLOAT(a->dtype, T, hipLaunchKernelGGL((topk_wave_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, current_stream(a->device()),
                                                          ac->ptr<T>(), v->ptr<T>(), ix->ptr<int64_t>(), rows, D, (int)k, largest));
    } else {
      LAMP_DISPATCH_FLOAT(a->dtype, T, hipLaunchKernelGGL((topk_kernel<T>), dim3((unsigned)rows), dim3(256), 0, current_stream(a->device()),
                                                          ac->ptr<T>(), v->ptr<T>(), ix->ptr<int64_t>(), D, k, largest));
    }
    LAMP_LAUNCH_CHECK();
  }
  lamp_tensor *vo = nullptr, *io = nullptr;
  LAMP_CHECK(lamp_transpose(&vo, v.get(), d, a->ndim - 1) == 0, lamp_last_error());
  Hold vh(vo);
  LAMP_CHECK(lamp_transpose(&io, ix.get(), d, a->ndim - 1) == 0, lamp_last_error());
  *values = vh.take();
  *indices = io;
  LAMP_API_END
}
int lamp_one_hot(lamp_tensor** out, const lamp_tensor* a, int64_t num_classes) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  LAMP_CHECK(a->dtype == kI64 && num_classes > 0, "one_hot expects int64 input and a positive class count");
  Hold ac(contiguous(a));
  std::vector<int64_t> oshape = a->shape();
  oshape.push_back(num_classes);
  Hold r(new_tensor(oshape, kI64, a->device()));
  const int64_t n = a->numel();
  if (n) {
    hipLaunchKernelGGL((one_hot_kernel<int64_t>), dim3(grid_for(n * num_classes, 256)), dim3(256), 0, current_stream(a->device()),
                       ac->ptr<int64_t>(), r->ptr<int64_t>(), n, num_classes);
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
int lamp_embedding(lamp_tensor** out, const lamp_tensor* weight, const lamp_tensor* indices) {
  LAMP_API_BEGIN
  check_device_tensor(weight, "weight"); check_device_tensor(indices, "indices");
  LAMP_CHECK(weight->ndim == 2 && indices->dtype == kI64, "embedding expects a 2-D weight and int64 indices");
  Hold ic(contiguous(indices));
  int64_t flat[1] = {ic->numel()};
  Hold iv(new_view(ic.get(), flat, (int64_t[]){1}, 1, ic->offset));
  lamp_tensor* sel = nullptr;
  LAMP_CHECK(lamp_index_select(&sel, weight, 0, iv.get()) == 0, lamp_last_error());
  Hold sh(sel);
  std::vector<int64_t> oshape = indices->shape();
  oshape.push_back(weight->sizes[1]);
  return lamp_view(out, sel, oshape.data(), (int)oshape.size());
  LAMP_API_END
}
int lamp_embedding_backward(lamp_tensor** out, const lamp_tensor* grad, const lamp_tensor* indices, int64_t num_weights, int64_t padding_idx) {
  LAMP_API_BEGIN
  check_device_tensor(grad, "grad"); check_device_tensor(indices, "indices");
  LAMP_CHECK(indices->dtype == kI64, "embedding_backward: indices must be long");
  const int64_t E = grad->sizes[grad->ndim - 1];
  int64_t ws[2] = {num_weights, E};
  Hold r(new_tensor(ws, 2, grad->dtype, grad->device()));
  Hold gc(contiguous(grad)), ic(contiguous(indices));
  const int64_t N = ic->numel();
  LAMP_CHECK(gc->numel() == N * E, "embedding_backward: grad " << grad->describe() << " does not match indices " << indices->describe());
  const bool floating = grad->dtype == kF32 || grad->dtype == kF64 || grad->dtype == kBF16;
  if (floating && num_weights > 0 && E > 0 && num_weights <= 65535 * 16 && (double)num_weights * (double)N <= (double)(1ll << 28)) {
    LAMP_DISPATCH_FLOAT(grad->dtype, T, hipLaunchKernelGGL((embedding_backward_scan_kernel<T>), dim3((unsigned)num_weights, (unsigned)((E + 255) / 256)),
                                                           dim3(256), 0, current_stream(grad->device()), gc->ptr<T>(), ic->ptr<int64_t>(), r->ptr<T>(), N, E,
                                                           padding_idx));
    LAMP_LAUNCH_CHECK();
  } else {
    fill_zero(r.get());
    int64_t gs[2] = {N, E}, gst[2] = {E, 1}, is_[1] = {N}, ist[1] = {1};
    Hold g2(new_view(gc.get(), gs, gst, 2, gc->offset)), i1(new_view(ic.get(), is_, ist, 1, ic->offset));
    LAMP_CHECK(lamp_index_add_(r.get(), 0, i1.get(), g2.get()) == 0, lamp_last_error());
    if (padding_idx >= 0 && padding_idx < num_weights) {   // the row only ever received the gradients of the padding tokens
      int64_t rs[1] = {E}, rst[1] = {1};
      Hold row(new_view(r.get(), rs, rst, 1, r->offset + padding_idx * E));
      fill_zero(row.get());
    }
  }
  *out = r.take();
  LAMP_API_END
}

int lamp_rand(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device) {
  LAMP_API_BEGIN *out = rng_new<0>(sizes, ndim, dtype, device, 0, 1); LAMP_API_END
}
int lamp_randn(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device) {
  LAMP_API_BEGIN *out = rng_new<1>(sizes, ndim, dtype, device, 0, 1); LAMP_API_END
}
int lamp_normal(lamp_tensor** out, double mean, double std, const int64_t* sizes, int ndim, int dtype, int device) {
  LAMP_API_BEGIN *out = rng_new<1>(sizes, ndim, dtype, device, mean, std); LAMP_API_END
}
int lamp_randint(lamp_tensor** out, int64_t low, int64_t high, const int64_t* sizes, int ndim, int dtype, int device) {
  LAMP_API_BEGIN
  LAMP_CHECK(high > low, "randint: high must be greater than low");
  *out = rng_new<2>(sizes, ndim, dtype, device, (double)low, (double)high);
  LAMP_API_END
}
int lamp_dropout_(lamp_tensor* self, double p, int training) {
  LAMP_API_BEGIN
  check_device_tensor(self, "self");
  LAMP_CHECK(p >= 0 && p <= 1, "dropout probability must be in [0, 1]");
  if (!training || p == 0.0) return 0;
  LAMP_CHECK(self->is_contiguous(), "dropout_ needs a contiguous tensor");
  const int64_t n = self->numel();
  if (p == 1.0) { fill_zero(self); return 0; }
  if (n) {
    const int grid = grid_for((n + 1) / 2, 256);
    const uint64_t off = next_philox_offset((uint64_t)((n + 1) / 2 / ((int64_t)grid * 256) + 2));
    LAMP_DISPATCH_FLOAT(self->dtype, T, hipLaunchKernelGGL((rng_kernel<T, 3>), dim3(grid), dim3(256), 0, current_stream(self->device()),
                                                           self->ptr<T>(), n, philox_seed(), off, p, 0.0));
    LAMP_LAUNCH_CHECK();
  }
  LAMP_API_END
}

}  // extern "C"
