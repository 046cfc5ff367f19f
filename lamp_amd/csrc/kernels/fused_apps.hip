// kNN graph, UMAP layout loss/gradient and scaled-dot-product attention.
//
// kNN   : lamp-knn/src/main/scala/lamp/knn/package.scala:21-80 - D = max(0, |q|^2 + |x|^2^T - 2 q.x^T),
//         k smallest per row; here the data set is walked in column chunks so the q x n distance
//         matrix is never materialised (per chunk: MFMA GEMM -> fused distance epilogue -> per-row
//         top-k; chunk winners are merged by a final top-k).  Index results are exact.
// UMAP  : lamp-umap/src/main/scala/lamp/umap/umap.scala:132-176 (loss) - one fused kernel evaluates
//         the loss and scatter-adds its gradient into the locations' gradient (f64 atomics).
// SDPA  : lamp-sten/.../STen.scala:501-584, ops.scala:2342-2390; CPU semantics are the composed
//         softmax(Q K^T * scale + mask) V of Transformer.scala:784-804 (the fused op has no CPU
//         known-answer test in the reference: parity is pinned to the composed form).
#include "device_utils.h"
#include "philox.h"
#include "../core/strided.h"

namespace lamp {

void knn_distance_block(Tensor* out, const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn);   // gemm.hip

Tensor* reduce_dims(const Tensor* a, const int64_t* dims, int ndims, bool keepdim, int op);

// dist[i, j] = max(0, qn[i] + dn[j] - 2 * outer[i, j])   (same operation order as the reference chain)
template <class T>
__global__ void knn_dist_kernel(T* __restrict__ outer, const T* __restrict__ qn, const T* __restrict__ dn, int64_t Q, int64_t Nc) {
  using A = acc_t<T>;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < Q * Nc; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / Nc, j = e - i * Nc;
    const A o2 = load_as<A>(store_as<T>((A)(load_as<A>(outer[e]) * A(2))));
    const A s = load_as<A>(store_as<T>((A)(load_as<A>(qn[i]) + load_as<A>(dn[j]))));
    const A v = s - o2;
    outer[e] = store_as<T>(v > A(0) ? v : A(0));
  }
}
// Jaccard: dist[i, j] = 1 - outer / ((qn[i] + dn[j]) - outer)   (knn/package.scala:32-44, same operation order)
template <class T>
__global__ void knn_jaccard_kernel(T* __restrict__ outer, const T* __restrict__ qn, const T* __restrict__ dn, int64_t Q, int64_t Nc) {
  using A = acc_t<T>;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < Q * Nc; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / Nc, j = e - i * Nc;
    const A o = load_as<A>(outer[e]);
    const A s = load_as<A>(store_as<T>((A)(load_as<A>(qn[i]) + load_as<A>(dn[j]))));
    const A den = load_as<A>(store_as<T>((A)(s - o)));
    const A sim = load_as<A>(store_as<T>((A)(o / den)));
    outer[e] = store_as<T>((A)(A(1) - sim));
  }
}
// out[i, j] = src[i, idx[i, j]] (+ column offset table for index merging)
template <class T>
__global__ void gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx, T* __restrict__ out, int64_t rows, int64_t k, int64_t srccols) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < rows * k; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / k;
    out[e] = src[i * srccols + idx[e]];
  }
}
__global__ void add_offset_kernel(int64_t* __restrict__ idx, int64_t n, int64_t offset) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) idx[e] += offset;
}

// ---- UMAP ------------------------------------------------------------------------------------------------
// one thread per pair; pairs [0, E1) are attractive (index1, index2, b), pairs [E1, E1 + E2) repulsive (index3, index4)
template <class T>
__global__ __launch_bounds__(256) void umap_pairs_kernel(const T* __restrict__ loc, int64_t D, const int64_t* __restrict__ i1, const int64_t* __restrict__ i2,
                                                         const T* __restrict__ b, int64_t E1, const int64_t* __restrict__ i3, const int64_t* __restrict__ i4,
                                                         int64_t E2, const T* __restrict__ bsum, double min_dist, int balance, double strength,
                                                         double w1, double w2, double w3, double w4, T* __restrict__ grad, double* __restrict__ loss_acc, const int64_t* __restrict__ e2_kept, int skip_self) {
  __shared__ double sm[4];
  double local = 0.0;
  const double attr_scale = balance ? 1.0 / (double)bsum[0] : 1.0;
  const double rep_scale = balance ? strength / (double)(e2_kept ? *e2_kept : E2) : 1.0;   // E2 counts the pairs that are kept
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E1 + E2; e += (int64_t)gridDim.x * blockDim.x) {
    const bool attr = e < E1;
    const int64_t a = attr ? i1[e] : i3[e - E1], c = attr ? i2[e] : i4[e - E1];
    if (skip_self && !attr && a == c) continue;                      // the reference drops these pairs before the loss (umap.scala:221-227)
    double d2 = 0.0;
    for (int64_t k = 0; k < D; k++) { const double df = (double)loc[a * D + k] - (double)loc[c * D + k]; d2 += df * df; }
    const double d = sqrt(d2);
    double dl_dd;   // d loss / d distance for this pair
    if (attr) {
      const double bb = (double)b[e];
      if (min_dist == 0.0) { local += bb * d * attr_scale; dl_dd = bb * attr_scale; }                    // -( -(b d) ) / sum b
      else {
        const double f = d <= min_dist ? 1.0 : exp(min_dist - d);
        local += -bb * log(f) * attr_scale;
        dl_dd = d <= min_dist ? 0.0 : bb * attr_scale;
      }
    } else {
      if (min_dist == 0.0) {
        const double ex = exp(-d);
        local += -rep_scale * log1p(-ex);
        dl_dd = -rep_scale * ex / (1.0 - ex);
      } else {
        const double f = d <= min_dist ? 1.0 : exp(min_dist - d);
        local += -rep_scale * log1p(-f + 1e-6);
        dl_dd = d <= min_dist ? 0.0 : -rep_scale * f / (1.0 + 1e-6 - f);
      }
    }
    const double wa = attr ? w1 : w3, wc = attr ? w2 : w4;
    for (int64_t k = 0; k < D; k++) {
      const double unit = ((double)loc[a * D + k] - (double)loc[c * D + k]) / d;   // diff / norm (NaN at d == 0, as in the reference)
      atomicAdd(&grad[a * D + k], (T)(wa * dl_dd * unit));
      atomicAdd(&grad[c * D + k], (T)(-wc * dl_dd * unit));
    }
  }
  local = block_sum(local, sm);
  if (threadIdx.x == 0) atomicAdd(loss_acc, local);
}
// 2-D layouts (the default numDim): a lane pair per pair of points, one lane per coordinate.
// Float atomics execute at the memory side as 64-byte requests, ~20 G requests/s chip-wide, and that rate - not bytes - bounds
// this kernel: the generic kernel above issues 4 requests per pair.  Here (1) the x and y adds of a point leave in ONE wave
// instruction from adjacent lanes (same 64-byte line: one request), and (2) the contributions to the FIRST point of a pair are
// summed across the wave before the atomic: the edge list is sorted by that index (runs of ~k attractive and ~k * negatives
// repulsive pairs per point), so a segmented scan leaves one atomic per run.  ~1.1 requests per pair instead of 4.
// The distance and the loss terms are computed exactly as above (d^2 = dx^2 + dy^2 in that order on both lanes).
// SAMPLED (round 6): the negative pairs are not read from (i3, i4) but DRAWN here - pair q = e * neg + j of edge e is (i1[e], the j-th negative of e),
// a counter-based draw (umap_negative) that umap_count_kept_kernel reproduces for the normaliser: what Umap.optimize builds per iteration with
// repeatInterleave + randint + ne + two maskedSelects (umap.scala:211-227) - 0.72 GB of indices written and read back at 1M points - is gone.
// The stream is not torch's (nor was randint's); the distribution is randint(0, hi)'s, value for value the one lamp_umap_negatives materialises.
struct UmapSampled { int neg; double hi; uint64_t seed, offset; };
// the j-th negative of edge e: uniform in [0, hi) as rng_kernel<long, 2> draws randint(0, hi) - floor(u * hi), u a 53-bit uniform; Philox counter =
// (offset + j / 2, subsequence e), the two doubles of a block serve j and j + 1
__device__ __forceinline__ int64_t umap_negative(const UmapSampled& sp, int64_t e, int j) {
  Philox ph(sp.seed, (uint64_t)e, sp.offset + (uint64_t)(j >> 1));
  const uint4 r = ph.next();
  const double u = (j & 1) ? u01(r.z, r.w) : u01(r.x, r.y);
  return (int64_t)floor(0.0 + u * (sp.hi - 0.0));
}
template <class T, bool SAMPLED = false>
__global__ __launch_bounds__(256) void umap_pairs2_kernel(const T* __restrict__ loc, const int64_t* __restrict__ i1, const int64_t* __restrict__ i2,
                                                          const T* __restrict__ b, int64_t E1, const int64_t* __restrict__ i3, const int64_t* __restrict__ i4,
                                                          int64_t E2, const T* __restrict__ bsum, double min_dist, int balance, double strength,
                                                          double w1, double w2, double w3, double w4, T* __restrict__ grad, double* __restrict__ loss_acc, const int64_t* __restrict__ e2_kept, int skip_self,
                                                          UmapSampled sp) {
  __shared__ double sm[4];
  double local = 0.0;
  const double attr_scale = balance ? 1.0 / (double)bsum[0] : 1.0;
  const double rep_scale = balance ? strength / (double)(e2_kept ? *e2_kept : E2) : 1.0;   // E2 counts the pairs that are kept
  const int lane = threadIdx.x & 63, dim = threadIdx.x & 1;
  const int64_t E = E1 + E2;
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 1);
  const int64_t rounds = (E + stride - 1) / stride;                // the same trip count for every lane: the loop body shuffles
  int64_t e = blockIdx.x * (int64_t)(blockDim.x >> 1) + (threadIdx.x >> 1);
  for (int64_t it = 0; it < rounds; it++, e += stride) {
    const int64_t ee = e < E ? e : E - 1;
    const bool attr = ee < E1;
    int64_t a, c;
    if constexpr (SAMPLED) {
      if (attr) { a = i1[ee]; c = i2[ee]; }
      else {
        const unsigned q = (unsigned)(ee - E1), eq = q / (unsigned)sp.neg;       // (host: E2 < 2^31)
        a = i1[eq];
        c = umap_negative(sp, (int64_t)eq, (int)(q - eq * (unsigned)sp.neg));
      }
    } else { a = attr ? i1[ee] : i3[ee - E1]; c = attr ? i2[ee] : i4[ee - E1]; }
    const bool valid = e < E && !(skip_self && !attr && a == c);    // the reference drops negative pairs that hit themselves
    const double diff = (double)loc[a * 2 + dim] - (double)loc[c * 2 + dim];
    const double sq = diff * diff, sq_o = __shfl_xor(sq, 1, 64);
    const double d2 = dim == 0 ? (0.0 + sq) + sq_o : (0.0 + sq_o) + sq;
    const double d = sqrt(d2);
    double dl_dd, term;
    if (attr) {
      const double bb = (double)b[ee];
      if (min_dist == 0.0) { term = bb * d * attr_scale; dl_dd = bb * attr_scale; }
      else {
        const double f = d <= min_dist ? 1.0 : exp(min_dist - d);
        term = -bb * log(f) * attr_scale;
        dl_dd = d <= min_dist ? 0.0 : bb * attr_scale;
      }
    } else {
      if (min_dist == 0.0) {
        const double ex = exp(-d);
        term = -rep_scale * log1p(-ex);
        dl_dd = -rep_scale * ex / (1.0 - ex);
      } else {
        const double f = d <= min_dist ? 1.0 : exp(min_dist - d);
        term = -rep_scale * log1p(-f + 1e-6);
        dl_dd = d <= min_dist ? 0.0 : -rep_scale * f / (1.0 + 1e-6 - f);
      }
    }
    if (valid && dim == 0) local += term;
    const double unit = diff / d;                                   // NaN at d == 0, as in the reference
    const double wa = attr ? w1 : w3, wc = attr ? w2 : w4;
    T ga = valid ? (T)(wa * dl_dd * unit) : T(0);
    const T gc = (T)(-wc * dl_dd * unit);
    if (valid) atomicAdd(&grad[c * 2 + dim], gc);
    // segmented inclusive scan over the pairs of the wave that share the first point (lane distance 2 per pair)
    const int64_t akey = valid ? a : -1 - (int64_t)lane;            // invalid lanes never merge
    const int64_t aprev = __shfl_up(akey, 2, 64), anext = __shfl_down(akey, 2, 64);
    int closed = lane < 2 || aprev != akey;                         // the run's first pair is already included in the sum
#pragma unroll
    for (int off = 2; off < 64; off <<= 1) {
      const T up = __shfl_up(ga, off, 64);
      const int cup = __shfl_up(closed, off, 64);
      if (lane >= off && !closed) { ga += up; closed = cup; }
    }
    if (valid && (lane >= 62 || anext != akey)) atomicAdd(&grad[a * 2 + dim], ga);
  }
  local = block_sum(local, sm);
  if (threadIdx.x == 0) atomicAdd(loss_acc, local);
}
template <class T> __global__ void cast_scalar_kernel(const double* in, T* out) { *out = (T)(*in); }
// the number of drawn negatives that do not hit their own point (the normaliser of the repulsion): the draws of umap_pairs2_kernel<., true>
__global__ __launch_bounds__(256) void umap_count_kept_kernel(const int64_t* __restrict__ i1, int64_t E1, UmapSampled sp, int64_t* __restrict__ out) {
  __shared__ double sm[4];
  double local = 0.0;   // exact below 2^53
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E1; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = i1[e];
    for (int j = 0; j < sp.neg; j++) local += umap_negative(sp, e, j) != a ? 1.0 : 0.0;
  }
  local = block_sum(local, sm);
  if (threadIdx.x == 0 && local != 0.0) atomicAdd((unsigned long long*)out, (unsigned long long)local);
}
// ... and the same draws written out: ii = i1 repeated neg times, jj = the negatives (lamp_umap_negatives: tests, sharded edge lists, layouts
// of more than two dimensions)
__global__ __launch_bounds__(256) void umap_negatives_kernel(const int64_t* __restrict__ i1, int64_t E1, UmapSampled sp, int64_t* __restrict__ ii, int64_t* __restrict__ jj) {
  const int64_t n = E1 * sp.neg;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = q / sp.neg;
    ii[q] = i1[e];
    jj[q] = umap_negative(sp, e, (int)(q - e * sp.neg));
  }
}
__global__ __launch_bounds__(256) void count_ne_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n, int64_t* __restrict__ out) {
  __shared__ double sm[4];
  double local = 0.0;   // exact below 2^53
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) local += a[e] != b[e] ? 1.0 : 0.0;
  local = block_sum(local, sm);
  if (threadIdx.x == 0 && local != 0.0) atomicAdd((unsigned long long*)out, (unsigned long long)local);
}

// ---- attention helpers ---------------------------------------------------------------------------------------
// scores[b, i, j] = scale * scores[b, i, j] (+ -inf above the diagonal); lse[b, i] = logsumexp_j; p = exp(s - lse)
template <class T>
__global__ __launch_bounds__(256) void sdpa_softmax_kernel(T* __restrict__ s, T* __restrict__ lse, int64_t rows, int64_t Sq, int64_t Sk, double scale, int causal) {
  using A = acc_t<T>;
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t i = row % Sq;
  T* p = s + row * Sk;
  const int64_t lim = causal ? (i + 1 < Sk ? i + 1 : Sk) : Sk;
  A m = -INFINITY;
  for (int64_t j = lane; j < lim; j += 64) { const A v = load_as<A>(p[j]) * (A)scale; m = v > m ? v : m; }
  m = wave_max(m);
  A sum = 0;
  for (int64_t j = lane; j < lim; j += 64) sum += (A)exp((double)(load_as<A>(p[j]) * (A)scale - m));
  sum = wave_sum(sum);
  const A l = m + (A)log((double)sum);
  if (lane == 0) lse[row] = store_as<T>(l);
  for (int64_t j = lane; j < Sk; j += 64) p[j] = store_as<T>(j < lim ? (A)exp((double)(load_as<A>(p[j]) * (A)scale - l)) : A(0));
}
// ds = p * (dp - rowsum(dp * p)) * scale   (in place on dp)
template <class T>
__global__ __launch_bounds__(256) void sdpa_softmax_bwd_kernel(T* __restrict__ dp, const T* __restrict__ p, int64_t rows, int64_t Sk, double scale) {
  using A = acc_t<T>;
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  A dot = 0;
  for (int64_t j = lane; j < Sk; j += 64) dot += load_as<A>(dp[row * Sk + j]) * load_as<A>(p[row * Sk + j]);
  dot = wave_sum(dot);
  for (int64_t j = lane; j < Sk; j += 64) {
    const int64_t e = row * Sk + j;
    dp[e] = store_as<T>((A)(load_as<A>(p[e]) * (load_as<A>(dp[e]) - dot) * (A)scale));
  }
}

// attention.hip: (B, heads, S, d) bf16 tensors in any row-contiguous layout (contiguous, or views of (B, S, heads, d) storage)
bool flash_attention_fwd(const Tensor* q, const Tensor* k, const Tensor* v, Tensor* out, Tensor* lse, int is_causal, double scale, hipStream_t st);
bool flash_attention_bwd(const Tensor* go, const Tensor* q, const Tensor* k, const Tensor* v, const Tensor* out, const Tensor* lse, Tensor* dq, Tensor* dk,
                         Tensor* dv, Tensor* dsum, int is_causal, double scale, hipStream_t st);
Tensor* at_new_like_layout(const Tensor* like, int64_t B, int64_t H, int64_t S, int64_t D, int dtype);

// scores[B*H, Sq, Sk] = scores * scale + bias (bias broadcast against (B, heads, Sq, Sk)) in place
static void sdpa_add_bias(Tensor* scores, const Tensor* bias, int64_t B, int64_t H, int64_t Sq, int64_t Sk, double scale) {
  int64_t s4[4] = {B, H, Sq, Sk};
  lamp_tensor* v4 = nullptr;
  LAMP_CHECK(lamp_view(&v4, scores, s4, 4) == 0, lamp_last_error());
  Hold h4(v4);
  LAMP_CHECK(lamp_mul_scalar_(v4, scale) == 0, lamp_last_error());
  LAMP_CHECK(lamp_add_(v4, bias, 1.0) == 0, lamp_last_error());
}

bool knn_fused(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t dim,
               int64_t k, hipStream_t st, int kind);   // knn_fused.hip
bool knn_split(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t dim, int64_t k,
               hipStream_t st);             // knn_split.hip: large f32 / f64 searches through the f16 matrix pipe, same result

static Hold call1(int (*fn)(lamp_tensor**, const lamp_tensor*), const Tensor* a) {
  lamp_tensor* o = nullptr;
  LAMP_CHECK(fn(&o, a) == 0, lamp_last_error());
  return Hold(o);
}

// ---- UMAP edge weights (umap.scala:14-113) ------------------------------------------------------------------------------
// rho_i = smallest positive kNN distance of point i; sigma_i = the reference's bisection (binarySearch: start 1, double while
// the upper bound is infinite, stop at |f - log2 k| < 1e-6 or after 1000 steps) of f(s) = sum_d exp(-max(0, d - rho_i) / s).
__global__ void umap_rho_sigma_kernel(const double* __restrict__ dist, double* __restrict__ rho, double* __restrict__ sigma, int64_t n, int k) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {   // grid_for caps the grid
    const double* d = dist + i * k;
    double r = INFINITY;
    for (int j = 0; j < k; j++) if (d[j] > 0.0 && d[j] < r) r = d[j];
    const double target = log((double)k) / log(2.0);
    double lo = 0.0, hi = INFINITY, mid = 1.0;
    for (int it = 0; it <= 1000; it++) {
      double f = 0.0;
      for (int j = 0; j < k; j++) f += exp((-1.0 * fmax(0.0, d[j] - r)) / mid);
      if (fabs(f - target) < 1e-6) break;
      if (f > target) { hi = mid; mid = (lo + mid) * 0.5; }
      else { lo = mid; mid = isinf(hi) ? mid * 2.0 : (hi + mid) * 0.5; }
    }
    rho[i] = r;
    sigma[i] = mid;
  }
}
// one thread per (i, jidx): b = w_ij + w_ji - w_ij * w_ji with w_ji looked up in j's neighbour list (0 if i is not in it)
__global__ void umap_edge_b_kernel(const double* __restrict__ dist, const int64_t* __restrict__ knn, const double* __restrict__ rho,
                                   const double* __restrict__ sigma, double* __restrict__ oi, double* __restrict__ oj, double* __restrict__ ob,
                                   uint8_t* __restrict__ keep, int64_t n, int k) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * k; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / k;
    const int64_t j = knn[e];
    oi[e] = (double)i; oj[e] = (double)j;
    if (j == i || j < 0 || j >= n) { keep[e] = 0; ob[e] = 0.0; continue; }
    const double wij = exp((-1.0 * fmax(0.0, dist[e] - rho[i])) / sigma[i]);
    double wji = 0.0;
    for (int l = 0; l < k; l++)
      if (knn[j * k + l] == i) { wji = exp((-1.0 * fmax(0.0, dist[j * k + l] - rho[j])) / sigma[j]); break; }
    ob[e] = wij + wji - wij * wji;
    keep[e] = 1;
  }
}

// knnDistances of Umap.umap (umap.scala:382-402): the exact f64 Euclidean distance between row i and each of its neighbours,
// summed in the reference's order (left to right over the columns)
template <class T>
__global__ void knn_row_distance_kernel(const T* __restrict__ data, const int64_t* __restrict__ idx, double* __restrict__ out, int64_t n, int64_t k,
                                        int64_t d) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * k; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / k, j = idx[e];
    const T* a = data + i * d;
    const T* b = data + j * d;
    double s = 0.0;
    for (int64_t c = 0; c < d; c++) { const double t = (double)load_as<acc_t<T>>(a[c]) - (double)load_as<acc_t<T>>(b[c]); s += t * t; }
    out[e] = sqrt(s);
  }
}

}  // namespace lamp

using namespace lamp;

extern "C" {

}  // extern "C"
static void knn_impl(lamp_tensor** indices, lamp_tensor** distances, const lamp_tensor* data, const lamp_tensor* query, int64_t k, int kind) {
  check_device_tensor(data, "data"); check_device_tensor(query, "query");
  LAMP_CHECK(data->ndim == 2 && query->ndim == 2 && data->sizes[1] == query->sizes[1] && data->dtype == query->dtype,
             "knn: data " << data->describe() << " and query " << query->describe() << " must be 2-D with equal width and dtype");
  const int64_t N = data->sizes[0], Q = query->sizes[0], dim = data->sizes[1];
  LAMP_CHECK(k >= 1 && k <= N, "knn: k = " << k << " out of range for " << N << " points");
  Hold dc(contiguous(data)), qc(contiguous(query));
  hipStream_t st = current_stream(data->device());
  // Euclidean: squared norms (v * v).rowSum; Jaccard: v.rowSum
  int64_t one = 1;
  Hold dn, qn;
  if (kind == 0) {
    Hold d2(call1(lamp_square, dc.get())), q2(call1(lamp_square, qc.get()));
    dn = Hold(reduce_dims(d2.get(), &one, 1, true, 0)); qn = Hold(reduce_dims(q2.get(), &one, 1, true, 0));
  } else {
    dn = Hold(reduce_dims(dc.get(), &one, 1, true, 0)); qn = Hold(reduce_dims(qc.get(), &one, 1, true, 0));
  }
  // f32 / f64, up to 128 features, k <= 16: top-k fused into the distance GEMM, no distance block at all (knn_fused.hip).  Widths
  // other than 64 / 128 are zero-padded to the next of the two (distances unchanged: the padding adds 0 to norms and dot products)
  if (Q > 0 && (data->dtype == kF32 || data->dtype == kF64) && dim <= 128 && k <= 16) {
    const int64_t pdim = dim <= 64 ? 64 : 128;
    Hold dpad, qpad;
    const Tensor *dsrc = dc.get(), *qsrc = qc.get();
    if (pdim != dim) {
      auto pad = [&](const Tensor* t, int64_t rows) {
        int64_t ps[2] = {rows, pdim};
        Hold z(new_tensor(ps, 2, t->dtype, t->device()));
        fill_zero(z.get());
        lamp_tensor* view = nullptr;
        LAMP_CHECK(lamp_narrow(&view, z.get(), 1, 0, dim) == 0, lamp_last_error());
        Hold hv(view);
        LAMP_CHECK(lamp_copy_(view, t, 1) == 0, lamp_last_error());
        return z;
      };
      dpad = pad(dc.get(), N); qpad = pad(qc.get(), Q);
      dsrc = dpad.get(); qsrc = qpad.get();
    }
    int64_t os[2] = {Q, k};
    Hold fi(new_tensor(os, 2, kI64, data->device())), fv(new_tensor(os, 2, data->dtype, data->device()));
    Hold dnc(contiguous(dn.get())), qnc(contiguous(qn.get()));
    if ((kind == 0 && knn_split(qsrc, dsrc, qnc.get(), dnc.get(), fi.get(), fv.get(), Q, N, pdim, k, st)) ||
        knn_fused(qsrc, dsrc, qnc.get(), dnc.get(), fi.get(), fv.get(), Q, N, pdim, k, st, kind)) {
      *indices = fi.take();
      if (distances) *distances = fv.take();
      return;
    }
  }
  // column chunk so that the Q x chunk block stays around 256 MB
  int64_t chunk = std::max<int64_t>(k, std::min<int64_t>(N, (int64_t)(256ll << 20) / (int64_t)(std::max<int64_t>(Q, 1) * data->itemsize())));
  chunk = std::max<int64_t>(chunk, std::min<int64_t>(N, 1024));
  const int64_t nchunks = (N + chunk - 1) / chunk;
  std::vector<Hold> cand_v, cand_i;
  for (int64_t c = 0; c < nchunks; c++) {
    const int64_t lo = c * chunk, len = std::min(chunk, N - lo);
    lamp_tensor *dsl = nullptr, *dnsl = nullptr;
    LAMP_CHECK(lamp_narrow(&dsl, dc.get(), 0, lo, len) == 0, lamp_last_error()); Hold hd(dsl);
    LAMP_CHECK(lamp_narrow(&dnsl, dn.get(), 0, lo, len) == 0, lamp_last_error()); Hold hdn(dnsl);
    int64_t os[2] = {Q, len};
    Hold outer(new_tensor(os, 2, data->dtype, data->device()));
    if (kind == 0 && (data->dtype == kF32 || data->dtype == kF64)) {
      // distance block in the GEMM epilogue: the q x chunk block is written once and never re-read before the top-k
      Hold hdnc(contiguous(dnsl));
      knn_distance_block(outer.get(), qc.get(), dsl, qn.get(), hdnc.get());
    } else {
      LAMP_CHECK(lamp_addmm_out_transposed2(outer.get(), outer.get(), qc.get(), dsl, 0.0, 1.0) == 0, lamp_last_error());   // q . x^T
      if (Q * len > 0) {
        if (kind == 0) {
          LAMP_DISPATCH_FLOAT(data->dtype, T, hipLaunchKernelGGL((knn_dist_kernel<T>), dim3(grid_for(Q * len, 256)), dim3(256), 0, st,
                                                                 outer->ptr<T>(), qn->ptr<T>(), hdn->ptr<T>(), Q, len));
        } else {
          LAMP_DISPATCH_FLOAT(data->dtype, T, hipLaunchKernelGGL((knn_jaccard_kernel<T>), dim3(grid_for(Q * len, 256)), dim3(256), 0, st,
                                                                 outer->ptr<T>(), qn->ptr<T>(), hdn->ptr<T>(), Q, len));
        }
        LAMP_LAUNCH_CHECK();
      }
    }
    const int64_t kk = std::min(k, len);
    lamp_tensor *tv = nullptr, *ti = nullptr;
    LAMP_CHECK(lamp_topk(&tv, &ti, outer.get(), kk, 1, 0, 1) == 0, lamp_last_error());
    Hold hv(tv), hi(ti);
    Hold ic(contiguous(ti));
    if (lo) { hipLaunchKernelGGL(add_offset_kernel, dim3(grid_for(Q * kk, 256)), dim3(256), 0, st, ic->ptr<int64_t>(), Q * kk, lo); LAMP_LAUNCH_CHECK(); }
    cand_v.emplace_back(contiguous(tv));
    cand_i.emplace_back(ic.take());
  }
  Hold best_v, best_i;
  if (nchunks == 1) { best_v = std::move(cand_v[0]); best_i = std::move(cand_i[0]); }
  else {
    std::vector<lamp_tensor*> vs, is;
    for (auto& h : cand_v) vs.push_back(h.get());
    for (auto& h : cand_i) is.push_back(h.get());
    lamp_tensor *allv = nullptr, *alli = nullptr;
    LAMP_CHECK(lamp_cat(&allv, vs.data(), (int)vs.size(), 1) == 0, lamp_last_error()); Hold hav(allv);
    LAMP_CHECK(lamp_cat(&alli, is.data(), (int)is.size(), 1) == 0, lamp_last_error()); Hold hai(alli);
    lamp_tensor *tv = nullptr, *ti = nullptr;
    LAMP_CHECK(lamp_topk(&tv, &ti, allv, k, 1, 0, 1) == 0, lamp_last_error());
    Hold hv(tv), hi(ti);
    Hold pos(contiguous(ti));
    int64_t os[2] = {Q, k};
    Hold gi(new_tensor(os, 2, kI64, data->device()));
    hipLaunchKernelGGL((gather_rows_kernel<int64_t>), dim3(grid_for(Q * k, 256)), dim3(256), 0, st, alli->ptr<int64_t>(), pos->ptr<int64_t>(),
                       gi->ptr<int64_t>(), Q, k, alli->sizes[1]);
    LAMP_LAUNCH_CHECK();
    best_v = Hold(contiguous(tv));
    best_i = std::move(gi);
  }
  *indices = best_i.take();
  if (distances) *distances = best_v.take();
}
extern "C" {
int lamp_knn_squared_euclidean(lamp_tensor** indices, lamp_tensor** distances, const lamp_tensor* data, const lamp_tensor* query, int64_t k) {
  LAMP_API_BEGIN
  knn_impl(indices, distances, data, query, k, 0);
  LAMP_API_END
}
int lamp_knn_jaccard(lamp_tensor** indices, lamp_tensor** distances, const lamp_tensor* data, const lamp_tensor* query, int64_t k) {
  LAMP_API_BEGIN
  knn_impl(indices, distances, data, query, k, 1);
  LAMP_API_END
}

int lamp_knn_row_distances(lamp_tensor** out, const lamp_tensor* data, const lamp_tensor* indices) {
  LAMP_API_BEGIN
  check_device_tensor(data, "data"); check_device_tensor(indices, "indices");
  LAMP_CHECK(data->ndim == 2 && indices->ndim == 2 && indices->dtype == kI64 && indices->sizes[0] == data->sizes[0],
             "knn row distances: expected data [n, d] and i64 indices [n, k], got " << data->describe() << " and " << indices->describe());
  Hold dc(contiguous(data)), ic(contiguous(indices));
  const int64_t n = indices->sizes[0], k = indices->sizes[1], d = data->sizes[1];
  int64_t os[2] = {n, k};
  Hold r(new_tensor(os, 2, kF64, data->device()));
  if (n * k > 0) {
    LAMP_DISPATCH_FLOAT(data->dtype, T, hipLaunchKernelGGL((knn_row_distance_kernel<T>), dim3(grid_for(n * k, 256)), dim3(256), 0, current_stream(data->device()),
                                                           dc->ptr<T>(), ic->ptr<int64_t>(), r->ptr<double>(), n, k, d));
    LAMP_LAUNCH_CHECK();
  }
  *out = r.take();
  LAMP_API_END
}
/* rows (i, j, b) for every neighbour j != i of every point i, in the reference's emission order (umap.scala:50-113) */
int lamp_umap_edge_weights(lamp_tensor** out, const lamp_tensor* knn_distances, const lamp_tensor* knn) {
  LAMP_API_BEGIN
  check_device_tensor(knn_distances, "knn_distances"); check_device_tensor(knn, "knn");
  LAMP_CHECK(knn_distances->ndim == 2 && knn->ndim == 2 && knn_distances->shape() == knn->shape() && knn_distances->dtype == kF64 && knn->dtype == kI64,
             "umap edge weights: expected [n, k] f64 distances and i64 indices, got " << knn_distances->describe() << " and " << knn->describe());
  const int64_t n = knn->sizes[0];
  const int k = (int)knn->sizes[1];
  LAMP_CHECK(k >= 1, "umap edge weights: k must be positive");
  Hold dc(contiguous(knn_distances)), kc(contiguous(knn));
  hipStream_t st = current_stream(knn->device());
  int64_t ns[1] = {n}, es[1] = {n * k};
  Hold rho(new_tensor(ns, 1, kF64, knn->device())), sigma(new_tensor(ns, 1, kF64, knn->device()));
  Hold oi(new_tensor(es, 1, kF64, knn->device())), oj(new_tensor(es, 1, kF64, knn->device())), ob(new_tensor(es, 1, kF64, knn->device()));
  Hold keep(new_tensor(es, 1, kBool, knn->device()));
  if (n > 0) {
    hipLaunchKernelGGL(umap_rho_sigma_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, dc->ptr<double>(), rho->ptr<double>(), sigma->ptr<double>(), n, k);
    LAMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(umap_edge_b_kernel, dim3(grid_for(n * k, 256)), dim3(256), 0, st, dc->ptr<double>(), kc->ptr<int64_t>(), rho->ptr<double>(),
                       sigma->ptr<double>(), oi->ptr<double>(), oj->ptr<double>(), ob->ptr<double>(), keep->ptr<uint8_t>(), n, k);
    LAMP_LAUNCH_CHECK();
  }
  // order-preserving compaction of the three columns, then [m, 3]
  lamp_tensor* cols[3] = {nullptr, nullptr, nullptr};
  const Tensor* src[3] = {oi.get(), oj.get(), ob.get()};
  std::vector<Hold> held;
  for (int c = 0; c < 3; c++) {
    LAMP_CHECK(lamp_masked_select(&cols[c], src[c], keep.get()) == 0, lamp_last_error());
    held.emplace_back(cols[c]);
  }
  return lamp_stack(out, cols, 3, 1);
  LAMP_API_END
}

}  // extern "C"
static void umap_loss_grad_impl(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations, const lamp_tensor* index1, const lamp_tensor* index2,
                                const lamp_tensor* b, const lamp_tensor* index3, const lamp_tensor* index4, double min_dist, int balance,
                                double repulsion_strength, const double* term_weights, int skip_self,
                                const lamp_tensor* bsum_global = nullptr, const lamp_tensor* kept_global = nullptr, const UmapSampled* sampled = nullptr) {
  // sampled: the negatives are drawn inside the kernels (index3 / index4 are null; 2-D layouts)
  check_device_tensor(locations, "locations"); check_device_tensor(grad_accum, "grad_accum");
  check_device_tensor(index1, "index1"); check_device_tensor(index2, "index2");
  if (!sampled) { check_device_tensor(index3, "index3"); check_device_tensor(index4, "index4"); }
  check_device_tensor(b, "b");
  LAMP_CHECK(locations->ndim == 2 && locations->is_contiguous() && grad_accum->is_contiguous() && grad_accum->shape() == locations->shape() &&
             grad_accum->dtype == locations->dtype, "umap: locations/grad must be contiguous [n, dim] tensors of one dtype");
  LAMP_CHECK(locations->dtype == kF64 || locations->dtype == kF32, "umap layout runs in f64 (reference) or f32");
  for (const lamp_tensor* ix : {index1, index2, index3, index4})
    if (ix) LAMP_CHECK(ix->dtype == kI64 && ix->ndim == 1 && ix->is_contiguous(), "umap: indices must be contiguous int64 vectors");
  const int64_t E1 = index1->numel(), E2 = sampled ? E1 * sampled->neg : index3->numel();
  LAMP_CHECK(index2->numel() == E1 && b->numel() == E1 && (sampled || index4->numel() == E2) && b->dtype == locations->dtype && b->is_contiguous(), "umap: edge list size mismatch");
  LAMP_CHECK(!sampled || (locations->sizes[1] == 2 && E2 < ((int64_t)1 << 31)), "internal: the sampled form is the 2-D kernel's");
  const UmapSampled spv = sampled ? *sampled : UmapSampled{1, 1.0, 0, 0};
  hipStream_t st = current_stream(locations->device());
  // sharded edge lists (one slice per rank): the normalisers are the GLOBAL sum of b and count of kept negatives, handed in
  Hold bsum;
  if (bsum_global) {
    check_device_tensor(bsum_global, "bsum"); LAMP_CHECK(bsum_global->dtype == b->dtype && bsum_global->numel() == 1, "umap: bsum must be one element of b's dtype");
    lamp_tensor* r = nullptr; LAMP_CHECK(lamp_tensor_retain(bsum_global, &r) == 0, lamp_last_error()); bsum = Hold(r);
  } else bsum = Hold(reduce_dims(b, nullptr, 0, false, 0));
  int64_t one[1] = {1};
  Hold acc(new_tensor(one, 1, kF64, locations->device()));
  fill_zero(acc.get());
  Hold out(new_tensor(nullptr, 0, locations->dtype, locations->device()));
  Hold kept;                                   // number of negative pairs that do not hit themselves (device scalar; no host sync)
  if (kept_global) {
    check_device_tensor(kept_global, "kept"); LAMP_CHECK(kept_global->dtype == kI64 && kept_global->numel() == 1, "umap: kept must be one int64");
    lamp_tensor* r = nullptr; LAMP_CHECK(lamp_tensor_retain(kept_global, &r) == 0, lamp_last_error()); kept = Hold(r);
  } else if (skip_self) {
    kept = Hold(new_tensor(one, 1, kI64, locations->device()));
    fill_zero(kept.get());
    if (E2 > 0 && sampled) { hipLaunchKernelGGL(umap_count_kept_kernel, dim3(grid_for(E1, 256)), dim3(256), 0, st, index1->ptr<int64_t>(), E1, spv, kept->ptr<int64_t>()); LAMP_LAUNCH_CHECK(); }
    else if (E2 > 0) { hipLaunchKernelGGL(count_ne_kernel, dim3(grid_for(E2, 256)), dim3(256), 0, st, index3->ptr<int64_t>(), index4->ptr<int64_t>(), E2, kept->ptr<int64_t>()); LAMP_LAUNCH_CHECK(); }
  }
  const int64_t* keptp = kept.get() ? static_cast<const Tensor*>(kept.get())->ptr<int64_t>() : nullptr;
  const double w[4] = {term_weights ? term_weights[0] : 1.0, term_weights ? term_weights[1] : 1.0, term_weights ? term_weights[2] : 1.0,
                       term_weights ? term_weights[3] : 1.0};
  static const bool pairs2 = [] { const char* e = getenv("LAMP_UMAP_PAIRS2"); return !(e && e[0] == '0'); }();
  const int64_t* i3p = index3 ? index3->ptr<int64_t>() : nullptr;
  const int64_t* i4p = index4 ? index4->ptr<int64_t>() : nullptr;
  if ((pairs2 || sampled) && locations->sizes[1] == 2 && E1 + E2 > 0) {
    // algorithmic traffic per pair: two int64 indices, two 2-D points gathered, b for the attractive pairs, and the read-modify-write of
    // two gradient points.  The `flops` slot carries the number of memory-side atomic requests if nothing were merged (two per pair,
    // x and y of a point share a 64-byte line): bench.py divides by the duration for the atomic-request rate.
    const double esz = (double)dtype_size(locations->dtype);
    // (sampled: the negatives cost one int64 per EDGE - the first point - instead of two per pair)
    const double idx_bytes = sampled ? (double)E1 * 16.0 + (double)E1 * 8.0 : (double)(E1 + E2) * 16.0;
    KernelTimer kt("umap_pairs2", 2.0 * (double)(E1 + E2), idx_bytes + (double)(E1 + E2) * (4.0 * esz + 8.0 * esz) + (double)E1 * esz, st);
#define UMAP_P2(T_, S_)                                                                                                                                         \
  hipLaunchKernelGGL((umap_pairs2_kernel<T_, S_>), dim3(grid_for(2 * (E1 + E2), 256)), dim3(256), 0, st, locations->ptr<T_>(), index1->ptr<int64_t>(),           \
                     index2->ptr<int64_t>(), b->ptr<T_>(), E1, i3p, i4p, E2, bsum->ptr<T_>(), min_dist, balance, repulsion_strength, w[0], w[1], w[2], w[3],      \
                     grad_accum->ptr<T_>(), acc->ptr<double>(), keptp, skip_self, spv)
    if (locations->dtype == kF64) {
      if (sampled) UMAP_P2(double, true); else UMAP_P2(double, false);
      hipLaunchKernelGGL((cast_scalar_kernel<double>), dim3(1), dim3(1), 0, st, acc->ptr<double>(), out->ptr<double>());
    } else {
      if (sampled) UMAP_P2(float, true); else UMAP_P2(float, false);
      hipLaunchKernelGGL((cast_scalar_kernel<float>), dim3(1), dim3(1), 0, st, acc->ptr<double>(), out->ptr<float>());
    }
#undef UMAP_P2
  } else if (locations->dtype == kF64) {
    hipLaunchKernelGGL((umap_pairs_kernel<double>), dim3(grid_for(E1 + E2, 256)), dim3(256), 0, st, locations->ptr<double>(), locations->sizes[1],
                       index1->ptr<int64_t>(), index2->ptr<int64_t>(), b->ptr<double>(), E1, i3p, i4p, E2,
                       bsum->ptr<double>(), min_dist, balance, repulsion_strength, w[0], w[1], w[2], w[3], grad_accum->ptr<double>(), acc->ptr<double>(), keptp, skip_self);
    hipLaunchKernelGGL((cast_scalar_kernel<double>), dim3(1), dim3(1), 0, st, acc->ptr<double>(), out->ptr<double>());
  } else {
    hipLaunchKernelGGL((umap_pairs_kernel<float>), dim3(grid_for(E1 + E2, 256)), dim3(256), 0, st, locations->ptr<float>(), locations->sizes[1],
                       index1->ptr<int64_t>(), index2->ptr<int64_t>(), b->ptr<float>(), E1, i3p, i4p, E2,
                       bsum->ptr<float>(), min_dist, balance, repulsion_strength, w[0], w[1], w[2], w[3], grad_accum->ptr<float>(), acc->ptr<double>(), keptp, skip_self);
    hipLaunchKernelGGL((cast_scalar_kernel<float>), dim3(1), dim3(1), 0, st, acc->ptr<double>(), out->ptr<float>());
  }
  LAMP_LAUNCH_CHECK();
  *loss = out.take();
}
extern "C" {
int lamp_umap_loss_grad(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations, const lamp_tensor* index1, const lamp_tensor* index2,
                        const lamp_tensor* b, const lamp_tensor* index3, const lamp_tensor* index4, double min_dist, int balance,
                        double repulsion_strength, const double* term_weights) {
  LAMP_API_BEGIN
  umap_loss_grad_impl(loss, grad_accum, locations, index1, index2, b, index3, index4, min_dist, balance, repulsion_strength, term_weights, 0);
  LAMP_API_END
}
// building blocks of the layout with the EDGE LIST sharded over ranks: number of pairs a != b as a device scalar, and
// the loss / gradient of a slice of the pairs under global normalisers; the caller all-reduces the count, the gradient and the loss
int lamp_count_ne(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN
  check_device_tensor(a, "a"); check_device_tensor(b, "b");
  LAMP_CHECK(a->dtype == kI64 && b->dtype == kI64 && a->is_contiguous() && b->is_contiguous() && a->numel() == b->numel(), "count_ne: two contiguous int64 tensors of one size");
  int64_t one[1] = {1};
  Hold r(new_tensor(one, 1, kI64, a->device()));
  fill_zero(r.get());
  const int64_t n = a->numel();
  if (n > 0) { hipLaunchKernelGGL(count_ne_kernel, dim3(grid_for(n, 256)), dim3(256), 0, current_stream(a->device()), a->ptr<int64_t>(), b->ptr<int64_t>(), n, r->ptr<int64_t>()); LAMP_LAUNCH_CHECK(); }
  *out = r.take();
  LAMP_API_END
}
int lamp_umap_loss_grad_sharded(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations, const lamp_tensor* index1,
                                const lamp_tensor* index2, const lamp_tensor* b, const lamp_tensor* index3, const lamp_tensor* index4, double min_dist,
                                int balance, double repulsion_strength, const double* term_weights, const lamp_tensor* bsum_global,
                                const lamp_tensor* kept_global) {
  LAMP_API_BEGIN
  LAMP_CHECK(bsum_global && kept_global, "umap sharded: the global normalisers are required");
  umap_loss_grad_impl(loss, grad_accum, locations, index1, index2, b, index3, index4, min_dist, balance, repulsion_strength, term_weights, 1, bsum_global,
                      kept_global);
  LAMP_API_END
}
// the negatives of Umap.optimize as the library draws them: ii = index1.repeatInterleave(n), jj = randint(0, high, [E1 * n]) (umap.scala:211-213),
// from ONE counter block of the generator (lamp_manual_seed + the calls since): lamp_umap_loss_grad_sampled draws exactly these when it takes the
// generator at the same point
static UmapSampled umap_sampled_draw(int64_t negatives_per_edge, int64_t high) {
  LAMP_CHECK(negatives_per_edge >= 1 && negatives_per_edge <= 1024, "umap: negatives per edge must be in [1, 1024], got " << negatives_per_edge);
  LAMP_CHECK(high >= 1, "umap: randint(0, high) needs high >= 1, got " << high);
  UmapSampled sp;
  sp.neg = (int)negatives_per_edge; sp.hi = (double)high; sp.seed = philox_seed();
  sp.offset = next_philox_offset((uint64_t)(negatives_per_edge + 1) / 2 + 1);
  return sp;
}
int lamp_umap_negatives(lamp_tensor** ii, lamp_tensor** jj, const lamp_tensor* index1, int64_t negatives_per_edge, int64_t high) {
  LAMP_API_BEGIN
  check_device_tensor(index1, "index1");
  LAMP_CHECK(index1->dtype == kI64 && index1->ndim == 1 && index1->is_contiguous(), "umap: indices must be contiguous int64 vectors");
  const UmapSampled sp = umap_sampled_draw(negatives_per_edge, high);
  const int64_t E1 = index1->numel();
  int64_t n[1] = {E1 * negatives_per_edge};
  Hold a(new_tensor(n, 1, kI64, index1->device())), c(new_tensor(n, 1, kI64, index1->device()));
  if (n[0] > 0) {
    hipLaunchKernelGGL(umap_negatives_kernel, dim3(grid_for(n[0], 256)), dim3(256), 0, current_stream(index1->device()), index1->ptr<int64_t>(), E1, sp,
                       a->ptr<int64_t>(), c->ptr<int64_t>());
    LAMP_LAUNCH_CHECK();
  }
  *ii = a.take(); *jj = c.take();
  LAMP_API_END
}
int lamp_umap_loss_grad_sampled(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations, const lamp_tensor* index1, const lamp_tensor* index2,
                                const lamp_tensor* b, int64_t negatives_per_edge, int64_t high, double min_dist, int balance, double repulsion_strength,
                                const double* term_weights) {
  LAMP_API_BEGIN
  check_device_tensor(locations, "locations"); check_device_tensor(index1, "index1");
  const int64_t E1 = index1->numel();
  if (locations->ndim == 2 && locations->sizes[1] == 2 && E1 * negatives_per_edge < ((int64_t)1 << 31)) {
    const UmapSampled sp = umap_sampled_draw(negatives_per_edge, high);
    umap_loss_grad_impl(loss, grad_accum, locations, index1, index2, b, nullptr, nullptr, min_dist, balance, repulsion_strength, term_weights, 1, nullptr, nullptr, &sp);
  } else {
    // other layouts: the same draws, written out, through the generic kernel
    lamp_tensor *ii = nullptr, *jj = nullptr;
    if (lamp_umap_negatives(&ii, &jj, index1, negatives_per_edge, high) != 0) throw Error(lamp_last_error());
    Hold hi_(ii), hj_(jj);
    umap_loss_grad_impl(loss, grad_accum, locations, index1, index2, b, ii, jj, min_dist, balance, repulsion_strength, term_weights, 1);
  }
  LAMP_API_END
}
int lamp_umap_loss_grad_skip_self(lamp_tensor** loss, lamp_tensor* grad_accum, const lamp_tensor* locations, const lamp_tensor* index1,
                                  const lamp_tensor* index2, const lamp_tensor* b, const lamp_tensor* index3, const lamp_tensor* index4, double min_dist,
                                  int balance, double repulsion_strength, const double* term_weights) {
  LAMP_API_BEGIN
  umap_loss_grad_impl(loss, grad_accum, locations, index1, index2, b, index3, index4, min_dist, balance, repulsion_strength, term_weights, 1);
  LAMP_API_END
}

// q, k, v: (B, heads, S, d).  out: (B, heads, Sq, d), logsumexp: (B, heads, Sq)
int lamp_scaled_dot_product_attention(lamp_tensor** out, lamp_tensor** logsumexp, const lamp_tensor* q, const lamp_tensor* k, const lamp_tensor* v,
                                      int is_causal, double scale) {
  return lamp_scaled_dot_product_attention_bias(out, logsumexp, q, k, v, nullptr, is_causal, scale);
}
// softmax(q k^T * scale + attn_bias (+ causal mask)) v.  attn_bias (ScaledDotProductAttention's `attentionBias: Option[STen]`,
// ops.scala:2342-2390; no gradient flows into it) broadcasts against (B, heads, Sq, Sk); with a bias the composed kernels run (the
// flash kernels take the mask-free and causal forms).
int lamp_scaled_dot_product_attention_bias(lamp_tensor** out, lamp_tensor** logsumexp, const lamp_tensor* q, const lamp_tensor* k, const lamp_tensor* v,
                                           const lamp_tensor* attn_bias, int is_causal, double scale) {
  LAMP_API_BEGIN
  check_device_tensor(q, "query"); check_device_tensor(k, "key"); check_device_tensor(v, "value");
  LAMP_CHECK(q->ndim == 4 && k->ndim == 4 && v->ndim == 4, "attention expects (B, heads, S, d) tensors");
  LAMP_CHECK(q->dtype == k->dtype && q->dtype == v->dtype, "attention: dtype mismatch");
  const int64_t B = q->sizes[0], H = q->sizes[1], Sq = q->sizes[2], D = q->sizes[3], Sk = k->sizes[2], Dv = v->sizes[3];
  LAMP_CHECK(k->sizes[0] == B && k->sizes[1] == H && k->sizes[3] == D && v->sizes[0] == B && v->sizes[1] == H && v->sizes[2] == Sk, "attention: shape mismatch");
  if (scale <= 0) scale = 1.0 / std::sqrt((double)D);
  int64_t ls[3] = {B, H, Sq};
  const int64_t rows = B * H * Sq;
  if (attn_bias) {
    check_device_tensor(attn_bias, "attn_bias");
    LAMP_CHECK(attn_bias->dtype == q->dtype, "attention: attn_bias dtype " << attn_bias->describe() << " differs from the query's");
    LAMP_CHECK(broadcast_shapes({B, H, Sq, Sk}, attn_bias->shape()) == (std::vector<int64_t>{B, H, Sq, Sk}), "attention: attn_bias does not broadcast to (B, heads, Sq, Sk)");
  }
  if (rows && Sk && q->dtype == kBF16 && !attn_bias) {   // fused flash form (bf16, head dim 64 / 128): no S x S intermediate; logsumexp is f32 as in ATen
    // strided operands are read in place; the result takes q's layout, so a (B, heads, S, d) view of (B, S, heads, d) projections
    // comes back as such a view (lamp's transposeOut is then free)
    Hold fo(at_new_like_layout(q, B, H, Sq, Dv, q->dtype)), lse32(new_tensor(ls, 3, kF32, q->device()));
    if (flash_attention_fwd(q, k, v, fo.get(), lse32.get(), is_causal, scale, current_stream(q->device()))) {
      *out = fo.take();
      *logsumexp = lse32.take();
      return 0;
    }
  }
  Hold qc(contiguous(q)), kc(contiguous(k)), vc(contiguous(v));
  int64_t qs[3] = {B * H, Sq, D}, ks[3] = {B * H, Sk, D}, vs[3] = {B * H, Sk, Dv};
  lamp_tensor *q3 = nullptr, *k3 = nullptr, *v3 = nullptr;
  LAMP_CHECK(lamp_view(&q3, qc.get(), qs, 3) == 0, lamp_last_error()); Hold hq(q3);
  LAMP_CHECK(lamp_view(&k3, kc.get(), ks, 3) == 0, lamp_last_error()); Hold hk(k3);
  LAMP_CHECK(lamp_view(&v3, vc.get(), vs, 3) == 0, lamp_last_error()); Hold hv(v3);
  Hold lse(new_tensor(ls, 3, q->dtype, q->device()));
  int64_t ss[3] = {B * H, Sq, Sk};
  Hold scores(new_tensor(ss, 3, q->dtype, q->device()));
  LAMP_CHECK(lamp_baddbmm_out_transposed2(scores.get(), scores.get(), q3, k3, 0.0, 1.0) == 0, lamp_last_error());
  double sm_scale = scale;
  if (attn_bias && rows && Sk) {            // scores = q k^T * scale + bias, then the row softmax with scale 1
    sdpa_add_bias(scores.get(), attn_bias, B, H, Sq, Sk, scale);
    sm_scale = 1.0;
  }
  if (rows) {
    LAMP_DISPATCH_FLOAT(q->dtype, T, hipLaunchKernelGGL((sdpa_softmax_kernel<T>), dim3((unsigned)((rows * 64 + 255) / 256)), dim3(256), 0,
                                                        current_stream(q->device()), scores->ptr<T>(), lse->ptr<T>(), rows, Sq, Sk, sm_scale, is_causal));
    LAMP_LAUNCH_CHECK();
  }
  lamp_tensor* o3 = nullptr;
  LAMP_CHECK(lamp_bmm(&o3, scores.get(), v3) == 0, lamp_last_error()); Hold ho(o3);
  int64_t os[4] = {B, H, Sq, Dv};
  LAMP_CHECK(lamp_view(out, o3, os, 4) == 0, lamp_last_error());
  *logsumexp = lse.take();
  LAMP_API_END
}

int lamp_scaled_dot_product_attention_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* q, const lamp_tensor* k,
                                               const lamp_tensor* v, const lamp_tensor* out, const lamp_tensor* logsumexp, int is_causal, double scale) {
  return lamp_scaled_dot_product_attention_bias_backward(out3, grad_out, q, k, v, out, logsumexp, nullptr, is_causal, scale);
}
int lamp_scaled_dot_product_attention_bias_backward(lamp_tensor* out3[3], const lamp_tensor* grad_out, const lamp_tensor* q, const lamp_tensor* k,
                                                    const lamp_tensor* v, const lamp_tensor* out, const lamp_tensor* logsumexp, const lamp_tensor* attn_bias,
                                                    int is_causal, double scale) {
  LAMP_API_BEGIN
  check_device_tensor(q, "query"); check_device_tensor(k, "key"); check_device_tensor(v, "value"); check_device_tensor(grad_out, "grad_out");
  const int64_t B = q->sizes[0], H = q->sizes[1], Sq = q->sizes[2], D = q->sizes[3], Sk = k->sizes[2], Dv = v->sizes[3];
  if (scale <= 0) scale = 1.0 / std::sqrt((double)D);
  // flash form: needs the forward's output and its f32 logsumexp (what the fused forward returns); operands are read in place in
  // whatever row-contiguous layout they have, each gradient takes the layout of its operand
  if (attn_bias) check_device_tensor(attn_bias, "attn_bias");
  if (!attn_bias && q->dtype == kBF16 && out && logsumexp && logsumexp->dtype == kF32 && out->dtype == kBF16 && B * H * Sq > 0 && Sk > 0 && grad_out->dtype == kBF16 &&
      logsumexp->numel() == B * H * Sq && out->numel() == B * H * Sq * Dv && out->ndim == 4 && grad_out->ndim == 4) {
    Hold lc(contiguous(logsumexp));
    int64_t n1[1] = {B * H * Sq};
    Hold dsum(new_tensor(n1, 1, kF32, q->device()));
    Hold fdq, fdk, fdv;
    // q, k, v are the three column blocks of ONE projection x . [Wq | Wk | Wv] (rows of 3 heads d elements; host: F::packed_self_attention):
    // their gradients are written as the three column blocks of one buffer, so that dX and d[Wq | Wk | Wv] are ONE product each
    const bool packed = q->st == k->st && k->st == v->st && Sq == Sk && D == Dv && q->strides[3] == 1 && q->strides[1] == D && q->strides[2] == 3 * H * D &&
                        q->strides[0] == Sq * 3 * H * D && k->offset == q->offset + H * D && v->offset == k->offset + H * D;
    bool same_strides = packed;
    for (int i = 0; same_strides && i < 4; i++) same_strides = k->strides[i] == q->strides[i] && v->strides[i] == q->strides[i];
    if (same_strides) {
      int64_t bs[3] = {B, Sq, 3 * H * D};
      Hold buf(new_tensor(bs, 3, kBF16, q->device()));
      int64_t vs[4] = {B, H, Sq, D};
      fdq = Hold(new_view(buf.get(), vs, q->strides, 4, buf->offset));
      fdk = Hold(new_view(buf.get(), vs, q->strides, 4, buf->offset + H * D));
      fdv = Hold(new_view(buf.get(), vs, q->strides, 4, buf->offset + 2 * H * D));
    } else {
      fdq = Hold(at_new_like_layout(q, B, H, Sq, D, kBF16)); fdk = Hold(at_new_like_layout(k, B, H, Sk, D, kBF16)); fdv = Hold(at_new_like_layout(v, B, H, Sk, Dv, kBF16));
    }
    if (flash_attention_bwd(grad_out, q, k, v, out, lc.get(), fdq.get(), fdk.get(), fdv.get(), dsum.get(), is_causal, scale, current_stream(q->device()))) {
      out3[0] = fdq.take(); out3[1] = fdk.take(); out3[2] = fdv.take();
      return 0;
    }
    // a gradient with an odd layout (e.g. expanded): retry on a dense copy before giving the fused form up
    Hold gc2(contiguous(grad_out)), oc2(contiguous(out));
    if ((gc2.get()->raw() != grad_out->raw() || oc2.get()->raw() != out->raw()) &&
        flash_attention_bwd(gc2.get(), q, k, v, oc2.get(), lc.get(), fdq.get(), fdk.get(), fdv.get(), dsum.get(), is_causal, scale, current_stream(q->device()))) {
      out3[0] = fdq.take(); out3[1] = fdk.take(); out3[2] = fdv.take();
      return 0;
    }
  }
  Hold qc(contiguous(q)), kc(contiguous(k)), vc(contiguous(v)), gc(contiguous(grad_out));
  // composed: P is recomputed from q and k (same numerics as the composed forward)
  int64_t qs[3] = {B * H, Sq, D}, ks[3] = {B * H, Sk, D}, vs[3] = {B * H, Sk, Dv}, gs[3] = {B * H, Sq, Dv};
  lamp_tensor *q3 = nullptr, *k3 = nullptr, *v3 = nullptr, *g3 = nullptr;
  LAMP_CHECK(lamp_view(&q3, qc.get(), qs, 3) == 0, lamp_last_error()); Hold hq(q3);
  LAMP_CHECK(lamp_view(&k3, kc.get(), ks, 3) == 0, lamp_last_error()); Hold hk(k3);
  LAMP_CHECK(lamp_view(&v3, vc.get(), vs, 3) == 0, lamp_last_error()); Hold hv(v3);
  LAMP_CHECK(lamp_view(&g3, gc.get(), gs, 3) == 0, lamp_last_error()); Hold hg(g3);
  int64_t ss[3] = {B * H, Sq, Sk}, ls[1] = {B * H * Sq};
  Hold p(new_tensor(ss, 3, q->dtype, q->device())), lse(new_tensor(ls, 1, q->dtype, q->device()));
  LAMP_CHECK(lamp_baddbmm_out_transposed2(p.get(), p.get(), q3, k3, 0.0, 1.0) == 0, lamp_last_error());
  const int64_t rows = B * H * Sq;
  hipStream_t st = current_stream(q->device());
  double sm_scale = scale;
  if (attn_bias && rows && Sk) { sdpa_add_bias(p.get(), attn_bias, B, H, Sq, Sk, scale); sm_scale = 1.0; }
  if (rows) {
    LAMP_DISPATCH_FLOAT(q->dtype, T, hipLaunchKernelGGL((sdpa_softmax_kernel<T>), dim3((unsigned)((rows * 64 + 255) / 256)), dim3(256), 0, st,
                                                        p->ptr<T>(), lse->ptr<T>(), rows, Sq, Sk, sm_scale, is_causal));
    LAMP_LAUNCH_CHECK();
  }
  // dV = P^T dO ; dP = dO V^T ; dS = P (dP - rowsum(dP P)) scale ; dQ = dS K ; dK = dS^T Q
  Hold dv(new_tensor(vs, 3, q->dtype, q->device())), dp(new_tensor(ss, 3, q->dtype, q->device()));
  LAMP_CHECK(lamp_baddbmm_out_transposed1(dv.get(), dv.get(), p.get(), g3, 0.0, 1.0) == 0, lamp_last_error());
  LAMP_CHECK(lamp_baddbmm_out_transposed2(dp.get(), dp.get(), g3, v3, 0.0, 1.0) == 0, lamp_last_error());
  if (rows) {
    LAMP_DISPATCH_FLOAT(q->dtype, T, hipLaunchKernelGGL((sdpa_softmax_bwd_kernel<T>), dim3((unsigned)((rows * 64 + 255) / 256)), dim3(256), 0, st,
                                                        dp->ptr<T>(), p->ptr<T>(), rows, Sk, scale));
    LAMP_LAUNCH_CHECK();
  }
  lamp_tensor* dq3 = nullptr;
  LAMP_CHECK(lamp_bmm(&dq3, dp.get(), k3) == 0, lamp_last_error()); Hold hdq(dq3);
  Hold dk(new_tensor(ks, 3, q->dtype, q->device()));
  LAMP_CHECK(lamp_baddbmm_out_transposed1(dk.get(), dk.get(), dp.get(), q3, 0.0, 1.0) == 0, lamp_last_error());
  LAMP_CHECK(lamp_view(&out3[0], dq3, q->sizes, 4) == 0, lamp_last_error());
  LAMP_CHECK(lamp_view(&out3[1], dk.get(), k->sizes, 4) == 0, lamp_last_error());
  LAMP_CHECK(lamp_view(&out3[2], dv.get(), v->sizes, 4) == 0, lamp_last_error());
  LAMP_API_END
}

}  // extern "C"
