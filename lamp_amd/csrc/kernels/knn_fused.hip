// k nearest neighbours by squared Euclidean distance, f32, with the top-k selection fused into the distance GEMM.
//
// Reference: lamp-knn/src/main/scala/lamp/knn/package.scala:60-121 (knnSearch / knnMinibatched with SquaredEuclideanDistance):
//   d(q, x) = max(0, (|q|^2 + |x|^2) - 2 q.x), then topk(k, largest = false) per query row.  The reference materialises a
//   minibatch x n distance block per query minibatch; at 1M x 1M that is 4 TB through HBM.  Here the n x n block never exists:
//   a workgroup owns 128 query rows, keeps their features in registers, streams the whole data set through LDS in tiles of 64
//   points (LDS-DMA, two buffers), and every wave keeps the running k best of its 32 rows in LDS.
//
// Matrix part: v_mfma_f32_16x16x4_f32, wave = 32 queries x 64 points.  A 16-byte LDS read gives a lane four k's of one point, so
// MFMA i of a 16-wide k-chunk j contracts k in {16 j + 4 g + i : g = 0..3}; the query fragments are loaded with the same pattern.
// Selection part: a lane ends a tile with 4 rows x 1 column per accumulator tile.  It compares its distances with the rows'
// current k-th best (registers); only when some lane passes (probability ~ k / points seen) the wave appends the candidates to a
// private LDS buffer (ballot + mbcnt, no atomics) and lane r inserts the entries of row r into that row's sorted (value, index)
// list.  Ties resolve by the lower index, as the unfused path does.
#include "device_utils.h"
#include <type_traits>

namespace lamp {

typedef float kf_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char kf_lds_t;
typedef const __attribute__((address_space(1))) char kf_glb_t;

constexpr int KF_BQ = 128, KF_BC = 64, KF_KMAX = 16, KF_CAND = 256;

template <int OFF> __device__ __forceinline__ void kf_read128(kf_f4& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
#define KF_FENCE4(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : : "memory")
template <int I, int N, class F> __device__ __forceinline__ void kf_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); kf_static_for<I + 1, N>(f); }
}

// per-wave selection state in LDS
struct KfWaveState {
  float lv[32][KF_KMAX];      // sorted ascending by (value, index)
  int li[32][KF_KMAX];
  float thr[32];              // lv[row][k - 1]
  float cv[KF_CAND];          // candidates of the accumulator tile just filtered
  int ci[KF_CAND];
  int cr[KF_CAND];
};

template <int DIM>
__global__ __launch_bounds__(256, 1) void knn_fused_kernel(const float* __restrict__ q, const float* __restrict__ x, const float* __restrict__ qn,
                                                           const float* __restrict__ dn, int64_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                           int Q, int N, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NJ = DIM / 16;                // 16-wide k-chunks
  constexpr int ROWB = DIM * 4;               // bytes of one point in LDS
  constexpr int NCHK = DIM / 4;               // 16-byte chunks per point (>= 16)
  constexpr int TILE = KF_BC * ROWB;
  static_assert(NCHK >= 16, "the swizzle needs at least 16 chunks per row");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int q0 = blockIdx.x * KF_BQ + wid * 32;
  KfWaveState* ws = reinterpret_cast<KfWaveState*>(smem + 2 * TILE) + wid;

  // ---- query fragments and norms
  kf_f4 qf[2][NJ];
  float qnr[2][4];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    int row = q0 + 16 * t + c16; row = row < Q ? row : Q - 1;
#pragma unroll
    for (int j = 0; j < NJ; j++) qf[t][j] = *reinterpret_cast<const kf_f4*>(q + (int64_t)row * DIM + 16 * j + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; r++) { int rr = q0 + 16 * t + 4 * g + r; rr = rr < Q ? rr : Q - 1; qnr[t][r] = qn[rr]; }
  }
  // ---- selection state
  if (lane < 32) {
    for (int i = 0; i < KF_KMAX; i++) { ws->lv[lane][i] = INFINITY; ws->li[lane][i] = 0x7fffffff; }
    ws->thr[lane] = INFINITY;
  }
  float thr[2][4];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) thr[t][r] = INFINITY;

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
      __builtin_amdgcn_global_load_lds((kf_glb_t*)(x + (int64_t)col * DIM + c * 4), (kf_lds_t*)(smem + buf * TILE + piece * 1024), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned bbase[NJ];                         // byte address of chunk j of this lane's point in col tile 0, buffer 0
#pragma unroll
  for (int j = 0; j < NJ; j++) { bbase[j] = lds0 + c16 * ROWB + ((((4 * j + g) ^ c16)) << 4); asm volatile("" : "+v"(bbase[j])); }

  const int nit = (N + KF_BC - 1) / KF_BC;
  float dnc[4], dnn[4], dno[4];                // column norms of the tile being multiplied / the next one / the one being filtered
#pragma unroll
  for (int ct = 0; ct < 4; ct++) { int col = 16 * ct + c16; col = col < N ? col : N - 1; dnn[ct] = dn[col]; dno[ct] = 0.f; }
  kf_f4 old[2][4];                             // dot products of the previous tile: filtered while this tile multiplies
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int ct = 0; ct < 4; ct++) old[t][ct] = kf_f4{0.f, 0.f, 0.f, 0.f};
  int col0_old = N;                            // no valid column: the first pass filters nothing

  // distances of one accumulator tile in the reference's operation order: (|q|^2 + |x|^2) - 2 q.x, then the clamp
  auto dist4 = [&](const kf_f4& a, int t, float dnv, float* v) {
#pragma unroll
    for (int r = 0; r < 4; r++) { const float d = (qnr[t][r] + dnv) - 2.f * a[r]; v[r] = d > 0.f ? d : 0.f; }
  };
  // slow path (probability ~ k / points seen per candidate): append the passing candidates of every accumulator tile to the
  // wave's buffer (ballot + prefix count) and let lane r insert the entries of row r into that row's sorted list
  auto select_tile = [&](const kf_f4 (&a)[2][4], const float* dnv, int c0) {
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        const int col = c0 + 16 * ct + c16;
        float v[4];
        dist4(a[t][ct], t, dnv[ct], v);
        bool pass[4], any = false;
#pragma unroll
        for (int r = 0; r < 4; r++) { pass[r] = col < N && v[r] < thr[t][r]; any |= pass[r]; }
        if (__builtin_amdgcn_ballot_w64(any) == 0) continue;
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned long long m = __builtin_amdgcn_ballot_w64(pass[r]);
          if (pass[r]) {
            const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            ws->cv[pos] = v[r]; ws->ci[pos] = col; ws->cr[pos] = 16 * t + 4 * g + r;
          }
          cnt += __builtin_popcountll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane < 32) {
          float* lv = ws->lv[lane];
          int* li = ws->li[lane];
          for (int e = 0; e < cnt; e++) {
            if (ws->cr[e] != lane) continue;
            const float cvv = ws->cv[e];
            const int cii = ws->ci[e];
            if (!(cvv < lv[k - 1] || (cvv == lv[k - 1] && cii < li[k - 1]))) continue;
            int p = k - 1;
            while (p > 0 && (lv[p - 1] > cvv || (lv[p - 1] == cvv && li[p - 1] > cii))) { lv[p] = lv[p - 1]; li[p] = li[p - 1]; p--; }
            lv[p] = cvv; li[p] = cii;
          }
          ws->thr[lane] = lv[k - 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
          for (int r = 0; r < 4; r++) thr[tt][r] = ws->thr[16 * tt + 4 * g + r];
      }
  };
  // branch-free test "does any candidate of the tile beat its row's k-th best": plain VALU work the scheduler can place
  // between the MFMAs of the next tile
  auto any_pass = [&](const kf_f4 (&a)[2][4], const float* dnv, int c0) {
    bool any = false;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        float v[4];
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
    for (int ct = 0; ct < 4; ct++) dnc[ct] = dnn[ct];
    if (it + 1 < nit) {
      dma_tile(it + 1, buf ^ 1);
#pragma unroll
      for (int ct = 0; ct < 4; ct++) { int col = (it + 1) * KF_BC + 16 * ct + c16; col = col < N ? col : N - 1; dnn[ct] = dn[col]; }
    }
    // ---- 32 x 64 dot products of tile `it`, with the filter of tile `it - 1` in the same basic block
    kf_f4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++) acc[t][ct] = kf_f4{0.f, 0.f, 0.f, 0.f};
    const unsigned boff = buf * TILE;
    kf_f4 bf[2][4];
    auto b_issue = [&](auto jc, kf_f4* dst) {
      constexpr int j = decltype(jc)::value;
      kf_read128<0 * 16 * ROWB>(dst[0], bbase[j] + boff);
      kf_read128<1 * 16 * ROWB>(dst[1], bbase[j] + boff);
      kf_read128<2 * 16 * ROWB>(dst[2], bbase[j] + boff);
      kf_read128<3 * 16 * ROWB>(dst[3], bbase[j] + boff);
    };
    b_issue(std::integral_constant<int, 0>{}, bf[0]);
    const bool hit = any_pass(old, dno, col0_old);
    kf_static_for<0, NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + 1 < NJ) { b_issue(std::integral_constant<int, j + 1>{}, bf[(j + 1) & 1]); KF_FENCE4("s_waitcnt lgkmcnt(4)", bf[j & 1]); }
      else KF_FENCE4("s_waitcnt lgkmcnt(0)", bf[j & 1]);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
          for (int t = 0; t < 2; t++) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[t][j][i], bf[j & 1][ct][i], acc[t][ct], 0, 0, 0);
    });
    if (__builtin_amdgcn_ballot_w64(hit) != 0) select_tile(old, dno, col0_old);
    col0_old = it * KF_BC;
#pragma unroll
    for (int ct = 0; ct < 4; ct++) dno[ct] = dnc[ct];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int ct = 0; ct < 4; ct++) old[t][ct] = acc[t][ct];
  }
  select_tile(old, dno, col0_old);
  // ---- results: row lane of the wave
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < 32 && q0 + lane < Q) {
    for (int i = 0; i < k; i++) {
      out_idx[(int64_t)(q0 + lane) * k + i] = ws->li[lane][i];
      out_val[(int64_t)(q0 + lane) * k + i] = ws->lv[lane][i];
    }
  }
}

// indices [Q, k] i64, values [Q, k] f32; qn [Q], dn [N] squared norms.  Returns false when the shape is not covered.
bool knn_fused(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t dim,
               int64_t k, hipStream_t st) {
  static const bool enabled = [] { const char* e = getenv("LAMP_KNN_FUSED"); return !(e && e[0] == '0'); }();
  if (!enabled || q->dtype != kF32 || !(dim == 64 || dim == 128) || k < 1 || k > KF_KMAX || N > 0x7fffff00 || Q > 0x7fffff00 || N < k || Q < 1) return false;
  if ((((uintptr_t)q->data() | (uintptr_t)x->data()) & 15) != 0) return false;
  const dim3 grid((unsigned)((Q + KF_BQ - 1) / KF_BQ));
  const size_t lds = (size_t)2 * KF_BC * dim * 4 + 4 * sizeof(KfWaveState);
  KernelTimer kt("knn_fused_f32", 2.0 * (double)Q * N * dim, ((double)Q + N) * dim * 4, st);
#define KF_LAUNCH(D)                                                                                                                       \
  do {                                                                                                                                     \
    static bool attr = false;                                                                                                              \
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)knn_fused_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
    hipLaunchKernelGGL((knn_fused_kernel<D>), grid, dim3(256), lds, st, q->ptr<float>(), x->ptr<float>(), qn->ptr<float>(), dn->ptr<float>(),   \
                       idx->ptr<int64_t>(), val->ptr<float>(), (int)Q, (int)N, (int)k);                                                     \
  } while (0)
  if (dim == 128) KF_LAUNCH(128); else KF_LAUNCH(64);
#undef KF_LAUNCH
  LAMP_LAUNCH_CHECK();
  return true;
}

}  // namespace lamp
