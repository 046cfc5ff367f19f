// Sorting family of the STen surface: sort / argsort (STen.scala:1592, 1761), randperm (:274), multinomial (:259-264, the
// language model's sampler: lamp-data/.../languagemodel/package.scala:100), unique (:1037-1055), bincount (:1034), median along a dimension
// (:1553-1557).  Off the training hot path: one generic mechanism, no tuning.
//
// Every element becomes a pair (key, position): key = an order preserving 64-bit image of the value (NaN above everything, as ATen sorts;
// complemented for a descending sort), position = its index along the sorted dimension.  Pairs are sorted ascending by (key, position) with
// a bitonic network - chunks of 2048 pairs in LDS, the strides beyond a chunk in global memory - so equal values keep their order in
// both directions: the result is ATen's STABLE sort, which is also a legal answer of the unstable one.  Rows are padded to a power of
// two with (max key, position >= length) pairs, which sort behind every real element.
#include "device_utils.h"
#include "../core/strided.h"
#include <vector>

namespace lamp {

constexpr int SORT_CH = 2048;       // pairs a workgroup sorts in LDS

template <class T> __device__ __forceinline__ uint64_t sort_key_of(T v);
template <> __device__ __forceinline__ uint64_t sort_key_of<double>(double v) {
  if (v != v) return ~0ull;
  if (v == 0.0) v = 0.0;                      // -0.0 == +0.0 for ATen's comparisons: one key (a stable sort keeps their order, unique counts them once)
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
template <> __device__ __forceinline__ uint64_t sort_key_of<float>(float v) { return sort_key_of<double>((double)v); }
template <> __device__ __forceinline__ uint64_t sort_key_of<bf16_t>(bf16_t v) { return sort_key_of<double>((double)(float)v); }
template <> __device__ __forceinline__ uint64_t sort_key_of<f16_t>(f16_t v) { return sort_key_of<double>((double)(float)v); }
template <> __device__ __forceinline__ uint64_t sort_key_of<int64_t>(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
template <> __device__ __forceinline__ uint64_t sort_key_of<int32_t>(int32_t v) { return sort_key_of<int64_t>((int64_t)v); }
template <> __device__ __forceinline__ uint64_t sort_key_of<uint8_t>(uint8_t v) { return sort_key_of<int64_t>((int64_t)v); }

template <class T>
__global__ void sort_init_kernel(const T* __restrict__ x, uint64_t* __restrict__ keys, int* __restrict__ idx, int64_t rows, int64_t L, int64_t P, int descending) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < rows * P; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / P, p = e - r * P;
    uint64_t k = ~0ull;
    if (p < L) { k = sort_key_of<T>(x[r * L + p]); if (descending) k = ~k; }
    keys[e] = k; idx[e] = (int)p;
  }
}
__device__ __forceinline__ bool pair_less(uint64_t ka, int ia, uint64_t kb, int ib) { return ka < kb || (ka == kb && ia < ib); }

// all strides j < chunk of the merges k = k_first .. k_last (k_first = 2: a full sort of every chunk; k_first = k_last = k: the tail of merge k)
__global__ __launch_bounds__(256) void sort_local_kernel(uint64_t* __restrict__ keys, int* __restrict__ idx, int64_t P, int chunk, int64_t k_first, int64_t k_last) {
  __shared__ uint64_t sk[SORT_CH];
  __shared__ int si[SORT_CH];
  const int64_t base = (int64_t)blockIdx.x * chunk;
  for (int t = threadIdx.x; t < chunk; t += 256) { sk[t] = keys[base + t]; si[t] = idx[base + t]; }
  __syncthreads();
  for (int64_t k = k_first; k <= k_last; k <<= 1) {
    for (int64_t j = min(k >> 1, (int64_t)chunk >> 1); j >= 1; j >>= 1) {
      for (int t = threadIdx.x; t < chunk / 2; t += 256) {
        const int lo = (int)(((t / j) * 2 * j) + (t % j)), hi = lo + (int)j;
        const int64_t g = (base + lo) & (P - 1);                    // position inside the row: the direction of merge k
        const bool up = (g & k) == 0;
        const bool sw = pair_less(sk[hi], si[hi], sk[lo], si[lo]);
        if (sw == up) { const uint64_t a = sk[lo]; sk[lo] = sk[hi]; sk[hi] = a; const int b = si[lo]; si[lo] = si[hi]; si[hi] = b; }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < chunk; t += 256) { keys[base + t] = sk[t]; idx[base + t] = si[t]; }
}
// one stride j >= chunk of merge k, in global memory
__global__ void sort_global_kernel(uint64_t* __restrict__ keys, int* __restrict__ idx, int64_t total_pairs, int64_t P, int64_t k, int64_t j) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total_pairs; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lo = (t / j) * 2 * j + (t % j), hi = lo + j;
    const bool up = ((lo & (P - 1)) & k) == 0;
    const uint64_t ka = keys[lo], kb = keys[hi];
    const int ia = idx[lo], ib = idx[hi];
    if (pair_less(kb, ib, ka, ia) == up) { keys[lo] = kb; keys[hi] = ka; idx[lo] = ib; idx[hi] = ia; }
  }
}
template <class T>
__global__ void sort_finish_kernel(const T* __restrict__ x, const int* __restrict__ idx, T* __restrict__ vals, int64_t* __restrict__ oi, int64_t rows, int64_t L,
                                   int64_t P, int64_t take) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < rows * take; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / take, p = e - r * take;
    const int i = idx[r * P + p];
    if (vals) vals[e] = x[r * L + i];
    if (oi) oi[e] = i;
  }
}

static int64_t pow2_at_least(int64_t n) { int64_t p = 1; while (p < n) p <<= 1; return p; }

// sorts the (key, position) pairs of `rows` rows of P pairs each, in place
static void sort_pairs(uint64_t* keys, int* idx, int64_t rows, int64_t P, hipStream_t st) {
  const int64_t total = rows * P;
  if (total <= 1 || P <= 1) return;
  // chunks must tile the array: P is a power of two; short rows share a chunk as long as whole chunks divide the array
  int ch = (int)std::min<int64_t>(P, SORT_CH);
  if (P < SORT_CH) while ((int64_t)ch * 2 <= SORT_CH && total % ((int64_t)ch * 2) == 0) ch *= 2;
  const unsigned blocks = (unsigned)(total / ch);
  hipLaunchKernelGGL(sort_local_kernel, dim3(blocks), dim3(256), 0, st, keys, idx, P, ch, (int64_t)2, std::min<int64_t>(P, (int64_t)ch));
  for (int64_t k = (int64_t)ch * 2; k <= P; k <<= 1) {
    for (int64_t j = k >> 1; j >= ch; j >>= 1)
      hipLaunchKernelGGL(sort_global_kernel, dim3(grid_for(total / 2, 256)), dim3(256), 0, st, keys, idx, total / 2, P, k, j);
    hipLaunchKernelGGL(sort_local_kernel, dim3(blocks), dim3(256), 0, st, keys, idx, P, ch, k, k);
  }
  LAMP_LAUNCH_CHECK();
}

// argsort of the last dimension of a contiguous tensor: idx workspace [rows][P] (positions in sorted order)
struct Sorted { Hold keys, idx; int64_t rows, L, P; };
static Sorted sort_last_dim(const Tensor* ac, bool descending, hipStream_t st) {
  Sorted s;
  s.L = ac->ndim ? ac->sizes[ac->ndim - 1] : 1;
  s.rows = s.L ? ac->numel() / s.L : 0;
  s.P = pow2_at_least(std::max<int64_t>(s.L, 1));
  LAMP_CHECK(s.L < ((int64_t)1 << 31), "sort: dimension of " << s.L << " elements is too long");
  int64_t n[1] = {std::max<int64_t>(s.rows * s.P, 1)};
  s.keys = Hold(new_tensor(n, 1, kI64, ac->device()));
  s.idx = Hold(new_tensor(n, 1, kI32, ac->device()));
  if (s.rows * s.P > 0) {
    LAMP_DISPATCH_ALL(ac->dtype, T, hipLaunchKernelGGL((sort_init_kernel<T>), dim3(grid_for(s.rows * s.P, 256)), dim3(256), 0, st, ac->ptr<T>(),
                                                       (uint64_t*)s.keys->ptr<int64_t>(), s.idx->ptr<int32_t>(), s.rows, s.L, s.P, descending ? 1 : 0));
    LAMP_LAUNCH_CHECK();
    sort_pairs((uint64_t*)s.keys->ptr<int64_t>(), s.idx->ptr<int32_t>(), s.rows, s.P, st);
  }
  return s;
}

// values / indices of the first `take` sorted elements of every row along `dim` (take = the whole dimension: sort)
static void sort_dim(const Tensor* a, int64_t dim, bool descending, int64_t take, Tensor** values, Tensor** indices) {
  check_device_tensor(a, "self");
  const int nd = a->ndim;
  const int64_t d = nd ? wrap_dim(dim, nd) : 0;
  hipStream_t st = current_stream(a->device());
  Hold tr;
  if (nd) { lamp_tensor* t = nullptr; LAMP_CHECK(lamp_transpose(&t, a, d, nd - 1) == 0, lamp_last_error()); tr = Hold(t); } else tr = Hold(retain(a));
  Hold ac(contiguous(tr.get()));
  Sorted s = sort_last_dim(ac.get(), descending, st);
  if (take < 0) take = s.L;
  std::vector<int64_t> oshape = ac->shape();
  if (nd) oshape.back() = take;
  Hold v(values ? new_tensor(oshape, a->dtype, a->device()) : nullptr), ix(indices ? new_tensor(oshape, kI64, a->device()) : nullptr);
  if (s.rows * take > 0) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((sort_finish_kernel<T>), dim3(grid_for(s.rows * take, 256)), dim3(256), 0, st, ac->ptr<T>(), s.idx->ptr<int32_t>(),
                                                      v.get() ? v->ptr<T>() : (T*)nullptr, ix.get() ? ix->ptr<int64_t>() : (int64_t*)nullptr, s.rows, s.L, s.P, take));
    LAMP_LAUNCH_CHECK();
  }
  auto back = [&](Hold& h, Tensor** out) {
    if (!out) return;
    if (!nd) { *out = h.take(); return; }
    lamp_tensor* t = nullptr;
    LAMP_CHECK(lamp_transpose(&t, h.get(), d, nd - 1) == 0, lamp_last_error());
    *out = t;
  };
  back(v, values); back(ix, indices);
}

// ---- random keys (randperm, sampling without replacement) ---------------------------------------------------------------------------
struct PhiloxS {      // Philox4x32-10, as kernels/index.hip
  uint32_t key[2], ctr[4];
  __device__ PhiloxS(uint64_t seed, uint64_t subsequence, uint64_t offset) {
    key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    ctr[0] = (uint32_t)offset; ctr[1] = (uint32_t)(offset >> 32); ctr[2] = (uint32_t)subsequence; ctr[3] = (uint32_t)(subsequence >> 32);
  }
  __device__ uint4 next() {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
      c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    if (++ctr[0] == 0) ++ctr[1];
    return make_uint4(c0, c1, c2, c3);
  }
};
__device__ __forceinline__ double u01s(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }

// keys of a random permutation: 63 random bits per element (the top bit stays clear: padding pairs carry the maximum key)
__global__ void randperm_keys_kernel(uint64_t* __restrict__ keys, int* __restrict__ idx, int64_t n, int64_t P, uint64_t seed, uint64_t offset) {
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (tid >= P) return;
  PhiloxS ph(seed, (uint64_t)tid, offset);
  const uint4 r = ph.next();
  keys[tid] = tid < n ? ((((uint64_t)r.x << 32) | r.y) >> 1) : ~0ull;
  idx[tid] = (int)tid;
}
__global__ void idx_to_i64_kernel(const int* __restrict__ idx, int64_t* __restrict__ out, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = idx[i];
}
// sampling WITHOUT replacement = the num_samples smallest of the exponential clocks E_i / p_i (p_i = 0 never fires)
template <class T>
__global__ void multinomial_keys_kernel(const T* __restrict__ p, uint64_t* __restrict__ keys, int* __restrict__ idx, int64_t rows, int64_t L, int64_t P, uint64_t seed,
                                        uint64_t offset) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= rows * P) return;
  const int64_t r = e / P, c = e - r * P;
  uint64_t k = ~0ull;
  if (c < L) {
    PhiloxS ph(seed, (uint64_t)e, offset);
    const uint4 q = ph.next();
    const double pr = (double)load_as<acc_t<T>>(p[r * L + c]);
    const double t = pr > 0.0 ? -log(1.0 - u01s(q.x, q.y)) / pr : INFINITY;
    k = sort_key_of<double>(t);
  }
  keys[e] = k; idx[e] = (int)c;
}
// sampling WITH replacement: one workgroup per row builds the running sums in LDS-sized pieces; every sample is a binary search
template <class T>
__global__ __launch_bounds__(256) void multinomial_cdf_kernel(const T* __restrict__ p, double* __restrict__ cdf, int64_t L) {
  __shared__ double part[256];
  const T* pr = p + (int64_t)blockIdx.x * L;
  double* cr = cdf + (int64_t)blockIdx.x * L;
  const int64_t per = (L + 255) / 256, b = threadIdx.x * per, e = min(b + per, L);
  double s = 0.0;
  for (int64_t i = b; i < e; i++) s += (double)load_as<acc_t<T>>(pr[i]);
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double run = 0.0; for (int t = 0; t < 256; t++) { const double v = part[t]; part[t] = run; run += v; } }
  __syncthreads();
  double run = part[threadIdx.x];
  for (int64_t i = b; i < e; i++) { run += (double)load_as<acc_t<T>>(pr[i]); cr[i] = run; }
}
__global__ void multinomial_sample_kernel(const double* __restrict__ cdf, int64_t* __restrict__ out, int64_t rows, int64_t L, int64_t ns, uint64_t seed, uint64_t offset) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= rows * ns) return;
  const int64_t r = e / ns;
  const double* c = cdf + r * L;
  PhiloxS ph(seed, (uint64_t)e, offset);
  const uint4 q = ph.next();
  const double u = u01s(q.x, q.y) * c[L - 1];
  int64_t lo = 0, hi = L - 1;                       // first index with cdf > u
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (c[mid] > u) hi = mid; else lo = mid + 1; }
  out[e] = lo;
}

// ---- unique / bincount ----------------------------------------------------------------------------------------------------------------
// flags[i] = 1 where sorted element i starts a new run; serial scan by blocks (the counts are small next to the sort)
// nan_key (floating tensors): the key all NaNs share; NaN != NaN, so each NaN is a value of its own, as in ATen
__global__ void unique_flag_kernel(const uint64_t* __restrict__ keys, int64_t* __restrict__ flag, int64_t n, int nan_key) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1] || (nan_key && keys[i] == ~0ull)) ? 1 : 0;
}
// ATen.median propagates NaN: a slice that holds one returns NaN and the position of its FIRST NaN.  x viewed as [outer][L][inner].
template <class T>
__global__ void median_nan_kernel(const T* __restrict__ x, T* __restrict__ vals, int64_t* __restrict__ idx, int64_t outer, int64_t L, int64_t inner) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= outer * inner) return;
  const int64_t o = e / inner, i = e - o * inner;
  for (int64_t l = 0; l < L; l++) {
    const T v = x[(o * L + l) * inner + i];
    const float f = load_as<float>(v);
    if (f != f) { vals[e] = v; idx[e] = l; return; }
  }
}
// one workgroup per row: every weight finite and >= 0, the row's sum > 0 (ATen raises "invalid multinomial distribution")
template <class T>
__global__ __launch_bounds__(256) void multinomial_check_kernel(const T* __restrict__ p, int64_t L, int* __restrict__ assert_word) {
  __shared__ double part[256];
  const T* pr = p + (int64_t)blockIdx.x * L;
  double s = 0.0;
  bool bad = false;
  for (int64_t i = threadIdx.x; i < L; i += 256) {
    const double v = (double)load_as<acc_t<T>>(pr[i]);
    if (!(v >= 0.0) || v > 1.7e308) bad = true;
    s += v;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < 256; k++) t += part[k]; if (!(t > 0.0)) bad = true; }
  if (bad) *(volatile int*)assert_word = kAssertMultinomial;
}
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int64_t* __restrict__ v, int64_t* __restrict__ sums, int64_t n) {
  __shared__ int64_t red[256];
  const int64_t b = (int64_t)blockIdx.x * 4096;
  int64_t s = 0;
  for (int64_t i = b + threadIdx.x; i < min(b + 4096, n); i += 256) s += v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}
__global__ void scan_serial_kernel(int64_t* __restrict__ sums, int64_t nb) {       // exclusive, in place; sums[nb] = total
  int64_t run = 0;
  for (int64_t i = 0; i < nb; i++) { const int64_t v = sums[i]; sums[i] = run; run += v; }
  sums[nb] = run;
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const int64_t* __restrict__ v, const int64_t* __restrict__ sums, int64_t* __restrict__ incl, int64_t n) {
  // inclusive scan of one 4096-element block: thread t owns 16 consecutive elements
  __shared__ int64_t part[256];
  const int64_t b = (int64_t)blockIdx.x * 4096 + threadIdx.x * 16;
  int64_t s = 0;
  for (int k = 0; k < 16; k++) if (b + k < n) s += v[b + k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int64_t run = sums[blockIdx.x]; for (int t = 0; t < 256; t++) { const int64_t x = part[t]; part[t] = run; run += x; } }
  __syncthreads();
  int64_t run = part[threadIdx.x];
  for (int k = 0; k < 16; k++) if (b + k < n) { run += v[b + k]; incl[b + k] = run; }
}
template <class T>
__global__ void unique_write_kernel(const T* __restrict__ x, const int* __restrict__ idx, const int64_t* __restrict__ flag, const int64_t* __restrict__ incl,
                                    T* __restrict__ vals, int64_t* __restrict__ inverse, int64_t* __restrict__ counts, int64_t n, int64_t nu) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t run = incl[i] - 1;
  if (inverse) inverse[idx[i]] = run;
  if (flag[i]) {
    vals[run] = x[idx[i]];
    if (counts) {
      // the run ends where the next one starts: binary search for the first j > i with incl[j] > incl[i]
      int64_t lo = i + 1, hi = n;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (incl[mid] > incl[i]) hi = mid; else lo = mid + 1; }
      counts[run] = lo - i;
    }
  }
  (void)nu;
}
template <class W>
__global__ void bincount_kernel(const int64_t* __restrict__ x, const W* __restrict__ w, W* __restrict__ out, int64_t n, int64_t size, int* __restrict__ assert_word) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = x[i];
    if (b < 0 || b >= size) { *(volatile int*)assert_word = kAssertIndexRange; continue; }
    if constexpr (std::is_same<W, int64_t>::value) atomicAdd((unsigned long long*)(out + b), 1ull);
    else atomicAdd(out + b, w[i]);
  }
}
__global__ __launch_bounds__(256) void max_i64_kernel(const int64_t* __restrict__ x, int64_t n, int64_t* __restrict__ out /* [2]: max, min */) {
  __shared__ int64_t mx[256], mn[256];
  int64_t a = INT64_MIN, b = INT64_MAX;
  for (int64_t i = threadIdx.x; i < n; i += 256) { a = max(a, x[i]); b = min(b, x[i]); }
  mx[threadIdx.x] = a; mn[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { mx[threadIdx.x] = max(mx[threadIdx.x], mx[threadIdx.x + o]); mn[threadIdx.x] = min(mn[threadIdx.x], mn[threadIdx.x + o]); } __syncthreads(); }
  if (threadIdx.x == 0) { out[0] = mx[0]; out[1] = mn[0]; }
}

}  // namespace lamp

using namespace lamp;

extern "C" {

int lamp_sort(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t dim, int descending) {
  LAMP_API_BEGIN
  sort_dim(a, dim, descending != 0, -1, values, indices);
  LAMP_API_END
}
int lamp_argsort(lamp_tensor** out, const lamp_tensor* a, int stable, int64_t dim, int descending) {
  LAMP_API_BEGIN
  (void)stable;                                   // the sort is always stable
  sort_dim(a, dim, descending != 0, -1, nullptr, out);
  LAMP_API_END
}
// the LOWER median of every slice along dim and its position (ATen.median with a dimension)
int lamp_median_dim(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t dim, int keepdim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  const int nd = a->ndim;
  const int64_t d = nd ? wrap_dim(dim, nd) : 0;
  const int64_t L = nd ? a->sizes[d] : 1;
  LAMP_CHECK(L > 0, "median of an empty dimension");
  lamp_tensor *v = nullptr, *ix = nullptr;
  sort_dim(a, dim, false, -1, &v, &ix);
  Hold vh(v), ih(ix);
  lamp_tensor *vs = nullptr, *is = nullptr;
  if (nd) {
    LAMP_CHECK(lamp_select(&vs, vh.get(), d, (L - 1) / 2) == 0, lamp_last_error());
    Hold t1(vs);
    LAMP_CHECK(lamp_select(&is, ih.get(), d, (L - 1) / 2) == 0, lamp_last_error());
    Hold t2(is);
    Hold vc(contiguous(t1.get())), ic(contiguous(t2.get()));
    if (a->dtype == kF32 || a->dtype == kF64 || a->dtype == kBF16 || a->dtype == kF16) {
      Hold ac(contiguous(a));
      int64_t outer = 1, inner = 1;
      for (int i = 0; i < d; i++) outer *= a->sizes[i];
      for (int i = (int)d + 1; i < nd; i++) inner *= a->sizes[i];
      if (outer * inner > 0) {
        hipStream_t st = current_stream(a->device());
        LAMP_DISPATCH_FLOAT(a->dtype, T, hipLaunchKernelGGL((median_nan_kernel<T>), dim3(grid_for(outer * inner, 256)), dim3(256), 0, st, ac->ptr<T>(), vc->ptr<T>(),
                                                             ic->ptr<int64_t>(), outer, L, inner));
        LAMP_LAUNCH_CHECK();
      }
    }
    if (keepdim) {
      lamp_tensor *vu = nullptr, *iu = nullptr;
      LAMP_CHECK(lamp_unsqueeze(&vu, vc.get(), d) == 0, lamp_last_error());
      Hold t3(vu);
      LAMP_CHECK(lamp_unsqueeze(&iu, ic.get(), d) == 0, lamp_last_error());
      *values = t3.take(); *indices = iu;
    } else { *values = vc.take(); *indices = ic.take(); }
  } else { *values = vh.take(); *indices = ih.take(); }
  LAMP_API_END
}
int lamp_randperm(lamp_tensor** out, int64_t n, int dtype, int device) {
  LAMP_API_BEGIN
  LAMP_CHECK(n >= 0 && n < ((int64_t)1 << 31), "randperm: n = " << n << " out of range");
  LAMP_CHECK(dtype == kI64, "randperm returns int64 (STen.randperm's default options), got dtype " << dtype);
  const int dev = device < 0 ? current_device() : device;
  hipStream_t st = current_stream(dev);
  const int64_t P = pow2_at_least(std::max<int64_t>(n, 1));
  int64_t ws[1] = {P}, os[1] = {n};
  Hold keys(new_tensor(ws, 1, kI64, dev)), idx(new_tensor(ws, 1, kI32, dev)), r(new_tensor(os, 1, kI64, dev));
  if (n) {
    const uint64_t off = next_philox_offset(2);
    hipLaunchKernelGGL(randperm_keys_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, (uint64_t*)keys->ptr<int64_t>(), idx->ptr<int32_t>(), n, P, philox_seed(), off);
    LAMP_LAUNCH_CHECK();
    sort_pairs((uint64_t*)keys->ptr<int64_t>(), idx->ptr<int32_t>(), 1, P, st);
    hipLaunchKernelGGL(idx_to_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx->ptr<int32_t>(), r->ptr<int64_t>(), n);
    LAMP_LAUNCH_CHECK();
  }
  if (device < 0) { Hold h(new_tensor(os, 1, kI64, -1)); if (n) copy_into(h.get(), r.get()); *out = h.take(); }
  else *out = r.take();
  LAMP_API_END
}
int lamp_multinomial(lamp_tensor** out, const lamp_tensor* probs, int64_t num_samples, int replacement) {
  LAMP_API_BEGIN
  check_device_tensor(probs, "probs");
  LAMP_CHECK(probs->ndim == 1 || probs->ndim == 2, "multinomial expects a vector or a matrix of weights, got " << probs->describe());
  LAMP_CHECK(probs->dtype == kF32 || probs->dtype == kF64 || probs->dtype == kBF16 || probs->dtype == kF16, "multinomial expects floating weights");
  Hold pc(contiguous(probs));
  const int64_t L = pc->sizes[pc->ndim - 1], rows = pc->ndim == 2 ? pc->sizes[0] : 1;
  LAMP_CHECK(num_samples > 0 && L > 0, "multinomial: nothing to sample");
  LAMP_CHECK(replacement || num_samples <= L, "multinomial: cannot draw " << num_samples << " samples without replacement from " << L << " categories");
  hipStream_t st = current_stream(probs->device());
  std::vector<int64_t> oshape = pc->ndim == 2 ? std::vector<int64_t>{rows, num_samples} : std::vector<int64_t>{num_samples};
  Hold r(new_tensor(oshape, kI64, probs->device()));
  LAMP_DISPATCH_FLOAT(pc->dtype, T, hipLaunchKernelGGL((multinomial_check_kernel<T>), dim3((unsigned)rows), dim3(256), 0, st, pc->ptr<T>(), L, device_assert_word(probs->device())));
  LAMP_LAUNCH_CHECK();
  if (replacement) {
    int64_t cs[1] = {rows * L};
    Hold cdf(new_tensor(cs, 1, kF64, probs->device()));
    LAMP_DISPATCH_FLOAT(pc->dtype, T, hipLaunchKernelGGL((multinomial_cdf_kernel<T>), dim3((unsigned)rows), dim3(256), 0, st, pc->ptr<T>(), cdf->ptr<double>(), L));
    const uint64_t off = next_philox_offset(2);
    hipLaunchKernelGGL(multinomial_sample_kernel, dim3((unsigned)((rows * num_samples + 255) / 256)), dim3(256), 0, st, cdf->ptr<double>(), r->ptr<int64_t>(), rows, L,
                       num_samples, philox_seed(), off);
    LAMP_LAUNCH_CHECK();
  } else {
    const int64_t P = pow2_at_least(L);
    int64_t ws[1] = {rows * P};
    Hold keys(new_tensor(ws, 1, kI64, probs->device())), idx(new_tensor(ws, 1, kI32, probs->device()));
    const uint64_t off = next_philox_offset(2);
    LAMP_DISPATCH_FLOAT(pc->dtype, T, hipLaunchKernelGGL((multinomial_keys_kernel<T>), dim3((unsigned)((rows * P + 255) / 256)), dim3(256), 0, st, pc->ptr<T>(),
                                                         (uint64_t*)keys->ptr<int64_t>(), idx->ptr<int32_t>(), rows, L, P, philox_seed(), off));
    LAMP_LAUNCH_CHECK();
    sort_pairs((uint64_t*)keys->ptr<int64_t>(), idx->ptr<int32_t>(), rows, P, st);
    hipLaunchKernelGGL((sort_finish_kernel<int64_t>), dim3(grid_for(rows * num_samples, 256)), dim3(256), 0, st, (const int64_t*)nullptr, idx->ptr<int32_t>(), (int64_t*)nullptr,
                       r->ptr<int64_t>(), rows, L, P, num_samples);
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
// ATen._unique2 of the flattened tensor: sorted unique values, the index of every element's value in them, the multiplicities.
// The output length is data dependent: one host synchronisation.  inverse / counts may be NULL.
int lamp_unique(lamp_tensor** values, lamp_tensor** inverse_or_null, lamp_tensor** counts_or_null, const lamp_tensor* a) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  hipStream_t st = current_stream(a->device());
  Hold ac(contiguous(a));
  const int64_t n = ac->numel();
  int64_t flat[1] = {n};
  Hold af(new_view(ac.get(), flat, std::vector<int64_t>{1}.data(), 1, ac->offset));
  Sorted s = sort_last_dim(af.get(), false, st);
  int64_t nu = 0;
  const int64_t nb = (n + 4095) / 4096;
  int64_t fs[1] = {std::max<int64_t>(n, 1)}, bs[1] = {nb + 1};
  Hold flag(new_tensor(fs, 1, kI64, a->device())), incl(new_tensor(fs, 1, kI64, a->device())), sums(new_tensor(bs, 1, kI64, a->device()));
  if (n) {
    const int nan_key = (a->dtype == kF32 || a->dtype == kF64 || a->dtype == kBF16 || a->dtype == kF16) ? 1 : 0;
    hipLaunchKernelGGL(unique_flag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint64_t*)s.keys->ptr<int64_t>(), flag->ptr<int64_t>(), n, nan_key);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, flag->ptr<int64_t>(), sums->ptr<int64_t>(), n);
    hipLaunchKernelGGL(scan_serial_kernel, dim3(1), dim3(1), 0, st, sums->ptr<int64_t>(), nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, flag->ptr<int64_t>(), sums->ptr<int64_t>(), incl->ptr<int64_t>(), n);
    LAMP_LAUNCH_CHECK();
    HIP_CHECK(hipMemcpyAsync(&nu, sums->ptr<int64_t>() + nb, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
  int64_t us[1] = {nu};
  Hold v(new_tensor(us, 1, a->dtype, a->device()));
  Hold inv(inverse_or_null ? new_tensor(a->shape(), kI64, a->device()) : nullptr), cnt(counts_or_null ? new_tensor(us, 1, kI64, a->device()) : nullptr);
  if (n) {
    // P == n is not required: the first n sorted pairs are the real elements (padding sorts last)
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((unique_write_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, af->ptr<T>(), s.idx->ptr<int32_t>(),
                                                      flag->ptr<int64_t>(), incl->ptr<int64_t>(), v->ptr<T>(), inv.get() ? inv->ptr<int64_t>() : (int64_t*)nullptr,
                                                      cnt.get() ? cnt->ptr<int64_t>() : (int64_t*)nullptr, n, nu));
    LAMP_LAUNCH_CHECK();
  }
  *values = v.take();
  if (inverse_or_null) *inverse_or_null = inv.take();
  if (counts_or_null) *counts_or_null = cnt.take();
  LAMP_API_END
}
// ATen.bincount: int64 counts (no weights) or sums of f64 / f32 weights per value of a non-negative int64 vector; one host synchronisation
int lamp_bincount(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* weights_or_null, int64_t minlength) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  LAMP_CHECK(a->dtype == kI64 && a->ndim == 1, "bincount expects an int64 vector, got " << a->describe());
  if (weights_or_null) {
    check_device_tensor(weights_or_null, "weights");
    LAMP_CHECK((weights_or_null->dtype == kF64 || weights_or_null->dtype == kF32) && weights_or_null->numel() == a->numel(), "bincount: weights must be f32 / f64 of the input's length");
  }
  hipStream_t st = current_stream(a->device());
  Hold ac(contiguous(a));
  const int64_t n = ac->numel();
  int64_t mm[2] = {-1, 0};
  if (n) {
    int64_t two[1] = {2};
    Hold m(new_tensor(two, 1, kI64, a->device()));
    hipLaunchKernelGGL(max_i64_kernel, dim3(1), dim3(256), 0, st, ac->ptr<int64_t>(), n, m->ptr<int64_t>());
    LAMP_LAUNCH_CHECK();
    HIP_CHECK(hipMemcpyAsync(mm, m->ptr<int64_t>(), 16, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    LAMP_CHECK(mm[1] >= 0, "bincount: negative value " << mm[1]);
  }
  int64_t size[1] = {std::max<int64_t>(mm[0] + 1, minlength)};
  const int odt = weights_or_null ? weights_or_null->dtype : kI64;
  Hold r(new_tensor(size, 1, odt, a->device()));
  fill_zero(r.get());
  if (n) {
    if (!weights_or_null) hipLaunchKernelGGL((bincount_kernel<int64_t>), dim3(grid_for(n, 256)), dim3(256), 0, st, ac->ptr<int64_t>(), (const int64_t*)nullptr, r->ptr<int64_t>(), n, size[0], device_assert_word(a->device()));
    else {
      Hold wc(contiguous(weights_or_null));
      if (odt == kF64) hipLaunchKernelGGL((bincount_kernel<double>), dim3(grid_for(n, 256)), dim3(256), 0, st, ac->ptr<int64_t>(), wc->ptr<double>(), r->ptr<double>(), n, size[0], device_assert_word(a->device()));
      else hipLaunchKernelGGL((bincount_kernel<float>), dim3(grid_for(n, 256)), dim3(256), 0, st, ac->ptr<int64_t>(), wc->ptr<float>(), r->ptr<float>(), n, size[0], device_assert_word(a->device()));
    }
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}


// ---- mode / unique along a dimension / unique_consecutive / cartesian_prod (round 5) ----------------------------------------
// STen.mode (STen.scala:1561), STen.unique(dim, ...) (:1059), STen.uniqueConsecutive (:1068), STen.cartesianProduct (:674).  Off every hot path:
// compositions of the sort above with thread-per-slice scans; data-dependent output lengths cost one host synchronisation, as lamp_unique.
}  // extern "C"

namespace lamp {
// the dimension d moved to the front, the others in their order (ATen's moveaxis(d, 0)), contiguous: [L][M]
static Tensor* slices_first(const Tensor* a, int64_t d) {
  Hold cur(retain(a));
  for (int64_t k = d; k > 0; k--) {
    lamp_tensor* t = nullptr;
    LAMP_CHECK(lamp_transpose(&t, cur.get(), k, k - 1) == 0, lamp_last_error());
    cur = Hold(t);
  }
  return contiguous(cur.get());
}
// values sorted along the last dimension with their original positions: per row the SMALLEST most frequent value and the position of its LAST
// occurrence (ATen.mode): the first longest run of the ascending stable sort, whose last member has the largest original index
template <class T>
__global__ void mode_rows_kernel(const T* __restrict__ v, const int64_t* __restrict__ ix, T* __restrict__ ov, int64_t* __restrict__ oi, int64_t rows, int64_t L) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const T* vr = v + r * L;
  const int64_t* ir = ix + r * L;
  int64_t best_end = 0, best_len = 0, start = 0;
  for (int64_t i = 1; i <= L; i++) {
    if (i == L || !(load_as<acc_t<T>>(vr[i]) == load_as<acc_t<T>>(vr[i - 1]))) {
      if (i - start > best_len) { best_len = i - start; best_end = i - 1; }
      start = i;
    }
  }
  ov[r] = vr[best_end];
  oi[r] = ir[best_end];
}
// flag[r] = 1 where slice order[r] differs from slice order[r - 1] (order == nullptr: the slices as they stand); x is [L][M]
template <class T>
__global__ void slice_flag_kernel(const T* __restrict__ x, const int* __restrict__ order, int64_t* __restrict__ flag, int64_t L, int64_t M) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= L) return;
  if (r == 0) { flag[0] = 1; return; }
  const T* a = x + (order ? (int64_t)order[r] : r) * M;
  const T* b = x + (order ? (int64_t)order[r - 1] : r - 1) * M;
  int64_t f = 0;
  for (int64_t j = 0; j < M; j++)
    if (!(load_as<acc_t<T>>(a[j]) == load_as<acc_t<T>>(b[j]))) { f = 1; break; }
  flag[r] = f;
}
// one pass of the least-significant-column-first lexicographic sort: key of the slice that currently stands at rank r, position = r
template <class T>
__global__ void slice_keys_kernel(const T* __restrict__ x, const int* __restrict__ order, uint64_t* __restrict__ keys, int* __restrict__ idx, int64_t L, int64_t P,
                                  int64_t M, int64_t col) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= P) return;
  keys[r] = r < L ? sort_key_of<T>(x[(int64_t)order[r] * M + col]) : ~0ull;
  idx[r] = (int)r;
}
__global__ void compose_order_kernel(const int* __restrict__ order, const int* __restrict__ idx, int* __restrict__ out, int64_t L) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < L) out[r] = order[idx[r]];
}
__global__ void iota_i32_kernel(int* __restrict__ o, int64_t n) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < n) o[r] = (int)r;
}
// runs of equal slices in rank order -> the slice that starts each run, every slice's run, the run lengths
__global__ void runs_write_kernel(const int* __restrict__ order, const int64_t* __restrict__ flag, const int64_t* __restrict__ incl, int64_t* __restrict__ starts,
                                  int64_t* __restrict__ inverse, int64_t* __restrict__ counts, int64_t L) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= L) return;
  const int64_t run = incl[r] - 1, pos = order ? (int64_t)order[r] : r;
  if (inverse) inverse[pos] = run;
  if (flag[r]) {
    starts[run] = pos;
    if (counts) {
      int64_t lo = r + 1, hi = L;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (incl[mid] > incl[r]) hi = mid; else lo = mid + 1; }
      counts[run] = lo - r;
    }
  }
}
// shared tail of unique_dim / unique_consecutive: flags in rank order -> (values = the run-starting slices along d, inverse, counts)
static void unique_slices(const Tensor* a, int64_t d, const Tensor* xs /* [L][M] */, const int* order, Tensor** values, Tensor** inverse, Tensor** counts, hipStream_t st) {
  const int64_t L = a->sizes[d], M = L ? xs->numel() / L : 0;
  int64_t nu = 0;
  const int64_t nb = (L + 4095) / 4096;
  int64_t fs[1] = {std::max<int64_t>(L, 1)}, bs[1] = {nb + 1};
  Hold flag(new_tensor(fs, 1, kI64, a->device())), incl(new_tensor(fs, 1, kI64, a->device())), sums(new_tensor(bs, 1, kI64, a->device()));
  if (L) {
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((slice_flag_kernel<T>), dim3(grid_for(L, 256)), dim3(256), 0, st, xs->ptr<T>(), order, flag->ptr<int64_t>(), L, M));
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, flag->ptr<int64_t>(), sums->ptr<int64_t>(), L);
    hipLaunchKernelGGL(scan_serial_kernel, dim3(1), dim3(1), 0, st, sums->ptr<int64_t>(), nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, flag->ptr<int64_t>(), sums->ptr<int64_t>(), incl->ptr<int64_t>(), L);
    LAMP_LAUNCH_CHECK();
    HIP_CHECK(hipMemcpyAsync(&nu, sums->ptr<int64_t>() + nb, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
  int64_t us[1] = {nu}, ls[1] = {L};
  Hold starts(new_tensor(us, 1, kI64, a->device()));
  Hold inv(inverse ? new_tensor(ls, 1, kI64, a->device()) : nullptr), cnt(counts ? new_tensor(us, 1, kI64, a->device()) : nullptr);
  if (L) {
    hipLaunchKernelGGL(runs_write_kernel, dim3(grid_for(L, 256)), dim3(256), 0, st, order, flag->ptr<int64_t>(), incl->ptr<int64_t>(), starts->ptr<int64_t>(),
                       inv.get() ? inv->ptr<int64_t>() : (int64_t*)nullptr, cnt.get() ? cnt->ptr<int64_t>() : (int64_t*)nullptr, L);
    LAMP_LAUNCH_CHECK();
  }
  lamp_tensor* v = nullptr;
  LAMP_CHECK(lamp_index_select(&v, a, d, starts.get()) == 0, lamp_last_error());
  *values = v;
  if (inverse) *inverse = inv.take();
  if (counts) *counts = cnt.take();
}
}  // namespace lamp

extern "C" {

int lamp_mode(lamp_tensor** values, lamp_tensor** indices, const lamp_tensor* a, int64_t dim, int keepdim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  const int nd = a->ndim;
  const int64_t d = nd ? wrap_dim(dim, nd) : 0;
  const int64_t L = nd ? a->sizes[d] : 1;
  LAMP_CHECK(L > 0, "mode of an empty dimension");
  lamp_tensor *v = nullptr, *ix = nullptr;
  sort_dim(a, dim, false, -1, &v, &ix);
  Hold vh(v), ih(ix);
  Hold vt, it;
  if (nd) {
    lamp_tensor *t1 = nullptr, *t2 = nullptr;
    LAMP_CHECK(lamp_transpose(&t1, vh.get(), d, nd - 1) == 0, lamp_last_error());
    Hold h1(t1);
    LAMP_CHECK(lamp_transpose(&t2, ih.get(), d, nd - 1) == 0, lamp_last_error());
    Hold h2(t2);
    vt = Hold(contiguous(h1.get())); it = Hold(contiguous(h2.get()));
  } else { vt = Hold(contiguous(vh.get())); it = Hold(contiguous(ih.get())); }
  const int64_t rows = vt->numel() / L;
  // the transposed shape without its last dimension: dimension d holds what was the last one
  std::vector<int64_t> oshape;
  for (int i = 0; i + 1 < nd; i++) oshape.push_back(vt->sizes[i]);
  Hold ov(new_tensor(oshape, a->dtype, a->device())), oi(new_tensor(oshape, kI64, a->device()));
  if (rows) {
    hipStream_t st = current_stream(a->device());
    LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((mode_rows_kernel<T>), dim3(grid_for(rows, 256)), dim3(256), 0, st, vt->ptr<T>(), it->ptr<int64_t>(), ov->ptr<T>(),
                                                      oi->ptr<int64_t>(), rows, L));
    LAMP_LAUNCH_CHECK();
  }
  // back to the input's dimension order: with keepdim the reduced dimension returns (size 1) to position d
  auto finish = [&](Hold& h, lamp_tensor** out) {
    if (!nd) { *out = h.take(); return; }
    lamp_tensor* u = nullptr;
    LAMP_CHECK(lamp_unsqueeze(&u, h.get(), nd - 1) == 0, lamp_last_error());            // [..., last-at-d ..., 1]
    Hold hu(u);
    lamp_tensor* t = nullptr;
    LAMP_CHECK(lamp_transpose(&t, hu.get(), d, nd - 1) == 0, lamp_last_error());        // the 1 at d, the former last dimension back at the end
    Hold ht(t);
    if (keepdim) { *out = contiguous(ht.get()); return; }
    lamp_tensor* q = nullptr;
    LAMP_CHECK(lamp_squeeze(&q, ht.get(), d) == 0, lamp_last_error());
    Hold hq(q);
    *out = contiguous(hq.get());
  };
  finish(ov, values); finish(oi, indices);
  LAMP_API_END
}

int lamp_unique_consecutive(lamp_tensor** values, lamp_tensor** inverse_or_null, lamp_tensor** counts_or_null, const lamp_tensor* a, int64_t dim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  LAMP_CHECK(a->ndim >= 1, "unique_consecutive along a dimension needs at least one dimension");
  const int64_t d = wrap_dim(dim, a->ndim);
  hipStream_t st = current_stream(a->device());
  Hold xs(slices_first(a, d));
  unique_slices(a, d, xs.get(), nullptr, values, inverse_or_null, counts_or_null, st);
  LAMP_API_END
}

int lamp_unique_dim(lamp_tensor** values, lamp_tensor** inverse_or_null, lamp_tensor** counts_or_null, const lamp_tensor* a, int64_t dim) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self");
  LAMP_CHECK(a->ndim >= 1, "unique along a dimension needs at least one dimension");
  const int64_t d = wrap_dim(dim, a->ndim);
  hipStream_t st = current_stream(a->device());
  Hold xs(slices_first(a, d));
  const int64_t L = a->sizes[d], M = L ? xs->numel() / L : 0;
  LAMP_CHECK(L < ((int64_t)1 << 31), "unique: dimension of " << L << " slices is too long");
  const int64_t P = pow2_at_least(std::max<int64_t>(L, 1));
  int64_t ps[1] = {P}, ls[1] = {std::max<int64_t>(L, 1)};
  Hold keys(new_tensor(ps, 1, kI64, a->device())), idx(new_tensor(ps, 1, kI32, a->device()));
  Hold order(new_tensor(ls, 1, kI32, a->device())), next(new_tensor(ls, 1, kI32, a->device()));
  if (L) {
    hipLaunchKernelGGL(iota_i32_kernel, dim3(grid_for(L, 256)), dim3(256), 0, st, order->ptr<int32_t>(), L);
    // slices sorted lexicographically over their flattened elements: stable sorts by column M - 1, M - 2, ..., 0
    for (int64_t col = M - 1; col >= 0; col--) {
      LAMP_DISPATCH_ALL(a->dtype, T, hipLaunchKernelGGL((slice_keys_kernel<T>), dim3(grid_for(P, 256)), dim3(256), 0, st, xs->ptr<T>(), order->ptr<int32_t>(),
                                                        (uint64_t*)keys->ptr<int64_t>(), idx->ptr<int32_t>(), L, P, M, col));
      LAMP_LAUNCH_CHECK();
      sort_pairs((uint64_t*)keys->ptr<int64_t>(), idx->ptr<int32_t>(), 1, P, st);
      hipLaunchKernelGGL(compose_order_kernel, dim3(grid_for(L, 256)), dim3(256), 0, st, order->ptr<int32_t>(), idx->ptr<int32_t>(), next->ptr<int32_t>(), L);
      LAMP_LAUNCH_CHECK();
      std::swap(order, next);
    }
  }
  unique_slices(a, d, xs.get(), order->ptr<int32_t>(), values, inverse_or_null, counts_or_null, st);
  LAMP_API_END
}

int lamp_cartesian_prod(lamp_tensor** out, lamp_tensor* const* tensors, int n) {
  LAMP_API_BEGIN
  LAMP_CHECK(n >= 1, "cartesian_prod of no tensors");
  for (int i = 0; i < n; i++) LAMP_CHECK(tensors[i] && tensors[i]->ndim == 1, "cartesian_prod expects 1-D tensors");
  if (n == 1) { *out = retain(tensors[0]); return 0; }
  std::vector<int64_t> full(n);
  for (int i = 0; i < n; i++) full[i] = tensors[i]->sizes[0];
  std::vector<Hold> cols;
  std::vector<lamp_tensor*> raw;
  for (int i = 0; i < n; i++) {
    std::vector<int64_t> one(n, 1);
    one[i] = full[i];
    lamp_tensor *v = nullptr, *e = nullptr, *f = nullptr;
    LAMP_CHECK(lamp_reshape(&v, tensors[i], one.data(), n) == 0, lamp_last_error());
    Hold hv(v);
    LAMP_CHECK(lamp_expand(&e, hv.get(), full.data(), n) == 0, lamp_last_error());
    Hold he(e);
    const int64_t flat[1] = {-1};
    LAMP_CHECK(lamp_reshape(&f, he.get(), flat, 1) == 0, lamp_last_error());
    cols.emplace_back(f);
    raw.push_back(cols.back().get());
  }
  return lamp_stack(out, raw.data(), n, 1);
  LAMP_API_END
}

}  // extern "C"
