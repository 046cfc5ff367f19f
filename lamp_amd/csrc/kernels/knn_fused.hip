// k nearest neighbours by squared Euclidean distance, f32 and f64, with the top-k selection fused into the distance GEMM.
//
// Reference: lamp-knn/src/main/scala/lamp/knn/package.scala:60-121 (knnSearch / knnMinibatched with SquaredEuclideanDistance):
//   d(q, x) = max(0, (|q|^2 + |x|^2) - 2 q.x), then topk(k, largest = false) per query row.  The reference materialises a
//   minibatch x n distance block per query minibatch; at 1M x 1M that is 4 TB through HBM.  Here the n x n block never exists:
//   a workgroup owns 128 query rows, keeps their features in registers, streams the whole data set through LDS in tiles of 64
//   points (LDS-DMA, two buffers), and every wave keeps the running k best of its 32 rows in LDS.
//
// Matrix part: v_mfma_f32_16x16x4_f32 (wave = 32 queries x 64 points) or v_mfma_f64_16x16x4_f64 (32 x 32; lamp's default
// DoublePrecision).  A 16-byte LDS read gives a lane E = 4 (f32) or 2 (f64) consecutive k's of one point, so MFMA i of chunk j
// contracts k in {4 E j + E g + i : g = 0..3}; the query fragments are loaded with the same pattern.
// Selection part: a lane ends a tile with 4 rows x 1 column per accumulator tile.  It compares its distances with the rows'
// current k-th best (registers); only when some lane passes (probability ~ k / points seen) the wave appends the candidates to a
// private LDS buffer (ballot + mbcnt, no atomics) and lane r inserts the entries of row r into that row's sorted (value, index)
// list.  Ties resolve by the lower index, as the unfused path does.
#include "device_utils.h"
#include <type_traits>

namespace lamp {

typedef float kf_f4 __attribute__((ext_vector_type(4)));
typedef double kf_d2 __attribute__((ext_vector_type(2)));
typedef double kf_d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char kf_lds_t;
typedef const __attribute__((address_space(1))) char kf_glb_t;

constexpr int KF_BQ = 128, KF_KMAX = 16, KF_RC = 24;   // KF_RC: a row's buffer is flushed when it holds more than KF_RC - 16 (a tile can add 16)

template <class T> struct KfTraits;
template <> struct KfTraits<float> {
  typedef kf_f4 chunk_t;                       // 16 bytes of one point
  typedef kf_f4 acc_t;
  static constexpr int E = 4, BC = 64;         // elements per chunk, points per tile (32 KiB)
  static __device__ __forceinline__ int row_of(int g, int r) { return 4 * g + r; }   // D register r of lane group g
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
template <> struct KfTraits<double> {
  typedef kf_d2 chunk_t;
  typedef kf_d4 acc_t;
  static constexpr int E = 2, BC = 32;
  static __device__ __forceinline__ int row_of(int g, int r) { return 4 * r + g; }   // the f64 MFMA interleaves the rows
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};

template <int OFF, class V> __device__ __forceinline__ void kf_read128(V& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
template <int I, int N, class F> __device__ __forceinline__ void kf_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); kf_static_for<I + 1, N>(f); }
}

// per-wave selection state in LDS
template <class T> struct KfWaveState {
  T lv[32][KF_KMAX];          // sorted ascending by (value, index)
  int li[32][KF_KMAX];
  T thr[32];                  // lv[row][k - 1]
  T bv[32][KF_RC];            // per-row candidate buffers (round 3, from knn_split.hip): a visit of the selection appends - slot = LDS atomic
  int bi[32][KF_RC];          // increment of the row's counter -, a flush lets every row's lane insert its own entries, all rows at once
  int bn[32];
};

template <class T, int DIM, int KIND>   // KIND 0: squared Euclidean (qn, dn = squared norms), 1: Jaccard (qn, dn = row sums)
__global__ __launch_bounds__(256, 1) void knn_fused_kernel(const T* __restrict__ q, const T* __restrict__ x, const T* __restrict__ qn,
                                                           const T* __restrict__ dn, int64_t* __restrict__ out_idx, T* __restrict__ out_val,
                                                           int Q, int N, int k, int chunk_len) {
  using TR = KfTraits<T>;
  using chunk_t = typename TR::chunk_t;
  using acc_t = typename TR::acc_t;
  constexpr int E = TR::E, KF_BC = TR::BC, NCT = KF_BC / 16;   // column tiles of 16 points
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NJ = DIM / (4 * E);           // chunk steps: 4 lane groups x E elements each
  constexpr int ROWB = DIM * (int)sizeof(T);  // bytes of one point in LDS
  constexpr int NCHK = ROWB / 16;             // 16-byte chunks per point (>= 16)
  constexpr int TILE = KF_BC * ROWB;
  static_assert(NCHK >= 16, "the swizzle needs at least 16 chunks per row");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int q0 = blockIdx.x * KF_BQ + wid * 32;
  // few queries against many points (chunk_len > 0): blockIdx.y takes a slice of the data set, every (query block, slice) its own
  // result rows [slice][Q][k] with global indices; knn_merge_chunks_kernel picks the k best of a query's slices
  int idx_base = 0;
  if (chunk_len > 0) {
    idx_base = blockIdx.y * chunk_len;
    x += (int64_t)idx_base * DIM; dn += idx_base;
    out_idx += (int64_t)blockIdx.y * Q * k; out_val += (int64_t)blockIdx.y * Q * k;
    N = min(chunk_len, N - idx_base);
  }
  KfWaveState<T>* ws = reinterpret_cast<KfWaveState<T>*>(smem + 2 * TILE) + wid;

  // ---- query fragments and norms
  chunk_t qf[2][NJ];
  T qnr[2][4];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    int row = q0 + 16 * t + c16; row = row < Q ? row : Q - 1;
#pragma unroll
    for (int j = 0; j < NJ; j++) qf[t][j] = *reinterpret_cast<const chunk_t*>(q + (int64_t)row * DIM + 4 * E * j + E * g);
#pragma unroll
    for (int r = 0; r < 4; r++) { int rr = q0 + 16 * t + TR::row_of(g, r); rr = rr < Q ? rr : Q - 1; qnr[t][r] = qn[rr]; }
  }
  // ---- selection state
  if (lane < 32) {
    for (int i = 0; i < KF_KMAX; i++) { ws->lv[lane][i] = (T)INFINITY; ws->li[lane][i] = 0x7fffffff; }
    ws->thr[lane] = (T)INFINITY;
    ws->bn[lane] = 0;
  }
  T thr[2][4];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) thr[t][r] = (T)INFINITY;

  auto dma_tile = [&](int it, int buf) {
    const int col0 = it * KF_BC;
    constexpr int PIECES = TILE / 1024;
#pragma unroll
    for (int i = 0; i < PIECES / 4; i++) {
      const int piece = wid * (PIECES / 4) + i;
      const int pos = piece * 64 + lane;
      const int row = pos / NCHK, cs = pos % NCHK;
      const int c = cs ^ (row & 15);
      int col = col0 + row; col = col < N ? col : N - 1;
      __builtin_amdgcn_global_load_lds((kf_glb_t*)(x + (int64_t)col * DIM + c * E), (kf_lds_t*)(smem + buf * TILE + piece * 1024), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned bbase[NJ];                         // byte address of chunk j of this lane's point in col tile 0, buffer 0
#pragma unroll
  for (int j = 0; j < NJ; j++) { bbase[j] = lds0 + c16 * ROWB + ((((4 * j + g) ^ c16)) << 4); asm volatile("" : "+v"(bbase[j])); }

  const int nit = (N + KF_BC - 1) / KF_BC;
  T dnc[NCT], dnn[NCT], dno[NCT];                // column norms of the tile being multiplied / the next one / the one being filtered
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) { int col = 16 * ct + c16; col = col < N ? col : N - 1; dnn[ct] = dn[col]; dno[ct] = T(0); }
  acc_t old[2][NCT];                             // dot products of the previous tile: filtered while this tile multiplies
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) old[t][ct] = acc_t{0, 0, 0, 0};
  int col0_old = N;                            // no valid column: the first pass filters nothing

  // distances of one accumulator tile in the reference's operation order: (|q|^2 + |x|^2) - 2 q.x, then the clamp
  // Jaccard (knn/package.scala:32-44): 1 - q.x / ((sum q + sum x) - q.x), 0 / 0 -> NaN, which no threshold test ever passes
  auto dist4 = [&](const acc_t& a, int t, T dnv, T* v) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if constexpr (KIND == 0) { const T d = (qnr[t][r] + dnv) - T(2) * a[r]; v[r] = d > T(0) ? d : T(0); }
      else { const T den = (qnr[t][r] + dnv) - a[r]; v[r] = T(1) - a[r] / den; }
    }
  };
  // slow path (probability ~ k / points seen per candidate).  Inserting every candidate at once - the lane that owns the row shifting
  // its sorted list through LDS while the other lanes wait - cost ~1500 cycles per candidate (knn_split.hip's counters); so a visit only
  // APPENDS the passing (value, index) to the row's buffer, and when some buffer could overflow with the next tile all 32 row lanes
  // insert their own entries at the same time.  Thresholds are a little stale in between (a few more candidates pass); entries of one
  // tile reach a buffer in any order, so the insertion compares (value, index) - ties go to the lower index as before.
  auto flush_rows = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 32) {
      T* lv = ws->lv[lane];
      int* li = ws->li[lane];
      const int n = ws->bn[lane];
      for (int e = 0; e < n; e++) {
        const T cvv = ws->bv[lane][e];
        const int cii = ws->bi[lane][e];
        if (!(cvv < lv[k - 1] || (cvv == lv[k - 1] && cii < li[k - 1]))) continue;
        int p = k - 1;
        while (p > 0 && (lv[p - 1] > cvv || (lv[p - 1] == cvv && li[p - 1] > cii))) { lv[p] = lv[p - 1]; li[p] = li[p - 1]; p--; }
        lv[p] = cvv; li[p] = cii;
      }
      if (n) { ws->thr[lane] = lv[k - 1]; ws->bn[lane] = 0; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int r = 0; r < 4; r++) thr[tt][r] = ws->thr[16 * tt + TR::row_of(g, r)];
  };
  auto select_tile = [&](const acc_t (&a)[2][NCT], const T* dnv, int c0) {
    bool full = false;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) {
        const int col = c0 + 16 * ct + c16;
        T v[4];
        dist4(a[t][ct], t, dnv[ct], v);
        bool pass[4], any = false;
#pragma unroll
        for (int r = 0; r < 4; r++) { pass[r] = col < N && v[r] < thr[t][r]; any |= pass[r]; }
        if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (pass[r]) {
            const int row = 16 * t + TR::row_of(g, r);
            const int slot = atomicAdd(&ws->bn[row], 1);       // the lanes of a group that hit the same row get distinct slots
            ws->bv[row][slot] = v[r]; ws->bi[row][slot] = col;
            full |= slot >= KF_RC - 17;
          }
        }
        if (__builtin_amdgcn_ballot_w64(full) != 0) { flush_rows(); full = false; }
      }
  };
  // branch-free test "does any candidate of the tile beat its row's k-th best": plain VALU work the scheduler can place
  // between the MFMAs of the next tile
  auto any_pass = [&](const acc_t (&a)[2][NCT], const T* dnv, int c0) {
    bool any = false;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) {
        T v[4];
        dist4(a[t][ct], t, dnv[ct], v);
        const bool in = c0 + 16 * ct + c16 < N;
#pragma unroll
        for (int r = 0; r < 4; r++) any |= in && v[r] < thr[t][r];
      }
    return any;
  };

  dma_tile(0, 0);
  for (int it = 0; it < nit; it++) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) dnc[ct] = dnn[ct];
    if (it + 1 < nit) {
      dma_tile(it + 1, buf ^ 1);
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) { int col = (it + 1) * KF_BC + 16 * ct + c16; col = col < N ? col : N - 1; dnn[ct] = dn[col]; }
    }
    // ---- 32 x 64 dot products of tile `it`, with the filter of tile `it - 1` in the same basic block
    acc_t acc[2][NCT];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) acc[t][ct] = acc_t{0, 0, 0, 0};
    const unsigned boff = buf * TILE;
    chunk_t bf[2][NCT];
    auto b_issue = [&](auto jc, chunk_t* dst) {
      constexpr int j = decltype(jc)::value;
      kf_static_for<0, NCT>([&](auto ctc) { constexpr int ct = decltype(ctc)::value; kf_read128<ct * 16 * ROWB>(dst[ct], bbase[j] + boff); });
    };
    auto b_fence = [&](chunk_t* f, bool last) {
      if constexpr (NCT == 4) { if (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory");
                                else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory"); }
      else { if (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]) : : "memory");
             else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]), "+v"(f[1]) : : "memory"); }
    };
    b_issue(std::integral_constant<int, 0>{}, bf[0]);
    const bool hit = any_pass(old, dno, col0_old);
    kf_static_for<0, NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + 1 < NJ) { b_issue(std::integral_constant<int, j + 1>{}, bf[(j + 1) & 1]); b_fence(bf[j & 1], false); }
      else b_fence(bf[j & 1], true);
#pragma unroll
      for (int i = 0; i < E; i++)
#pragma unroll
        for (int ct = 0; ct < NCT; ct++)
#pragma unroll
          for (int t = 0; t < 2; t++) acc[t][ct] = TR::mfma(qf[t][j][i], bf[j & 1][ct][i], acc[t][ct]);
    });
    if (__builtin_amdgcn_ballot_w64(hit) != 0) select_tile(old, dno, col0_old);
    col0_old = it * KF_BC;
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) dno[ct] = dnc[ct];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) old[t][ct] = acc[t][ct];
  }
  select_tile(old, dno, col0_old);
  flush_rows();
  // ---- results: row lane of the wave
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < 32 && q0 + lane < Q) {
    for (int i = 0; i < k; i++) {
      out_idx[(int64_t)(q0 + lane) * k + i] = ws->li[lane][i] == 0x7fffffff ? (int64_t)0x7fffffff : (int64_t)ws->li[lane][i] + idx_base;
      out_val[(int64_t)(q0 + lane) * k + i] = ws->lv[lane][i];
    }
  }
}

// the k best (value, then lower index) of a query's `chunks` sorted lists: one thread per query (few queries: that is why there are chunks)
template <class T>
__global__ __launch_bounds__(256) void knn_merge_chunks_kernel(const int64_t* __restrict__ ci, const T* __restrict__ cv, int64_t* __restrict__ out_idx,
                                                               T* __restrict__ out_val, int Q, int k, int chunks) {
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= Q) return;
  int head[64];                                 // next unread entry of every chunk's list (chunks <= 64)
  for (int c = 0; c < chunks; c++) head[c] = 0;
  for (int o = 0; o < k; o++) {
    int best = -1; T bv = (T)INFINITY; int64_t bi = 0x7fffffff;
    for (int c = 0; c < chunks; c++) {
      if (head[c] >= k) continue;
      const int64_t e = ((int64_t)c * Q + qi) * k + head[c];
      const T v = cv[e]; const int64_t i = ci[e];
      if (best < 0 || v < bv || (v == bv && i < bi)) { best = c; bv = v; bi = i; }
    }
    head[best]++;
    out_idx[(int64_t)qi * k + o] = bi;
    out_val[(int64_t)qi * k + o] = bv;
  }
}

// indices [Q, k] i64, values [Q, k] of the data's dtype; qn [Q], dn [N] squared norms.  Returns false when the shape is not covered.
template <class T, int D, int KIND>
static void knn_fused_launch(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t k,
                             hipStream_t st) {
  static bool attr = false;
  allow_big_lds((const void*)knn_fused_kernel<T, D, KIND>);
  const int64_t wgs = (Q + KF_BQ - 1) / KF_BQ;
  const size_t lds = (size_t)2 * KfTraits<T>::BC * D * sizeof(T) + 4 * sizeof(KfWaveState<T>);
  // A workgroup streams the whole data set: with fewer workgroups than CUs the search takes as long as a full round whatever Q is
  // (442 queries against 1M points: 77 ms).  Then the data set is cut into slices over blockIdx.y and the per-slice lists are merged.
  const int64_t cus = num_cus();
  int64_t chunks = 1;
  if (wgs * 2 <= cus && N >= 32768) chunks = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(64, cus / wgs), N / 8192));
  if (chunks > 1 && k <= N / chunks) {
    const int64_t chunk_len = (N + chunks - 1) / chunks;
    chunks = (N + chunk_len - 1) / chunk_len;
    Hold ti(new_tensor({chunks, Q, k}, kI64, x->device())), tv(new_tensor({chunks, Q, k}, q->dtype, x->device()));
    hipLaunchKernelGGL((knn_fused_kernel<T, D, KIND>), dim3((unsigned)wgs, (unsigned)chunks), dim3(256), lds, st, q->ptr<T>(), x->ptr<T>(), qn->ptr<T>(), dn->ptr<T>(),
                       ti->ptr<int64_t>(), tv->ptr<T>(), (int)Q, (int)N, (int)k, (int)chunk_len);
    hipLaunchKernelGGL((knn_merge_chunks_kernel<T>), dim3(grid_for(Q, 256)), dim3(256), 0, st, ti->ptr<int64_t>(), tv->ptr<T>(), idx->ptr<int64_t>(), val->ptr<T>(), (int)Q,
                       (int)k, (int)chunks);
    return;
  }
  hipLaunchKernelGGL((knn_fused_kernel<T, D, KIND>), dim3((unsigned)wgs), dim3(256), lds, st, q->ptr<T>(), x->ptr<T>(), qn->ptr<T>(), dn->ptr<T>(), idx->ptr<int64_t>(),
                     val->ptr<T>(), (int)Q, (int)N, (int)k, 0);
}

bool knn_fused(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t dim,
               int64_t k, hipStream_t st, int kind) {
  static const bool enabled = [] { const char* e = getenv("LAMP_KNN_FUSED"); return !(e && e[0] == '0'); }();
  const bool f32 = q->dtype == kF32, f64 = q->dtype == kF64;
  if (!enabled || !(f32 || f64) || !(dim == 64 || dim == 128) || k < 1 || k > KF_KMAX || N > 0x7fffff00 || Q > 0x7fffff00 || N < k || Q < 1) return false;
  if ((((uintptr_t)q->data() | (uintptr_t)x->data()) & 15) != 0) return false;
  KernelTimer kt(f32 ? "knn_fused_f32" : "knn_fused_f64", 2.0 * (double)Q * N * dim, ((double)Q + N) * dim * (f32 ? 4 : 8), st);
#define KF_GO(TT, DD) do { if (kind == 0) knn_fused_launch<TT, DD, 0>(q, x, qn, dn, idx, val, Q, N, k, st); else knn_fused_launch<TT, DD, 1>(q, x, qn, dn, idx, val, Q, N, k, st); } while (0)
  if (f32) { if (dim == 128) KF_GO(float, 128); else KF_GO(float, 64); }
  else     { if (dim == 128) KF_GO(double, 128); else KF_GO(double, 64); }
#undef KF_GO
  LAMP_LAUNCH_CHECK();
  return true;
}

}  // namespace lamp
