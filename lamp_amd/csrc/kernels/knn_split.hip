// k nearest neighbours of f32 points through the 16-bit matrix pipe: a filter pass that cannot lose a neighbour + an exact re-rank.
//
// Reference: lamp-knn/src/main/scala/lamp/knn/package.scala:60-121 (knnSearch with SquaredEuclideanDistance, f32):
//   d(q, x) = max(0, (|q|^2 + |x|^2) - 2 q.x), topk(k, largest = false) per query.
// knn_fused.hip computes q.x on v_mfma_f32_16x16x4_f32: 157 TFLOP/s is all the f32 matrix pipe has, and a 1M x 1M x 128 search is
// 256 TFLOP.  The f16 pipe is 16 x wider, and an f32 value splits exactly into f16 pieces (11 bits each):
//   y = y0 + y1 + r,  y0 = f16(y), y1 = f16(y - y0):  |r| <= 2^-22 |y|   (bf16 pieces, 8 bits each, need three planes and six products
//   for the same 22 - 24 bits: measured first, twice the matrix time).
// Pass 0 (knn_split_planes_kernel): distances do not change when every point moves by the same vector, but the error of a split product
//   scales with |q| |x| - so the data set's mean row is subtracted first; the centred rows are scaled by a power of two that puts their
//   largest coordinate near 2^14 (f16's range; the pieces of small coordinates that fall into f16's denormals are 2^-39 of that, far
//   below the bound) and written as two f16 planes, with the squared norm of the centred row.
// Pass 1 (knn_split_kernel): q.x ~ (q0.x0 + q1.x0 + q0.x1) / scale^2 - three f16 products per feature, 3/16 of the f32 pipe's time -
//   with the top-k selection fused in; the 16 best candidates per query by a = (|q|^2 + |x|^2) - 2 (that sum) are kept.
//   Error of the sum: the dropped products (q1.x1, q0.r, r.x0: <= 3.1 * 2^-22 |q||x| = 7.4e-7) plus the f32 accumulation of 12 MFMA results (each a
//   32-term sum with <= 6 internal roundings, then 12 roundings of the running sum: 18 * 2^-24 = 1.07e-6 relative to sum |q_i x_i|): 1.8e-6 |q||x|,
//   bounded by c = 1.1875 * 2^-19 = 2.27e-6 (the largest error seen on the test sets is 4e-7).  So |a - d| <= eps_q =
//   2 c |q_c| max|x_c| + 2^-21 (|q|^2 + max|x|^2) + 1.5 * 2^-21 (|q_c|^2 + max|x_c|^2): the second term is the f32 formula's own rounding
//   (f64 searches: 2^-50); the third is everything the centred norms carry (_c = after the mean row was subtracted): the rounding of the
//   centred rows to f32 before the split (2 * 2^-24 of a squared norm), their f32 accumulation (one fma per lane + a six-level butterfly:
//   8 roundings), the sum of the two norms and the final subtraction (1 each) - 12 * 2^-24.  Round 3 relied on slack in c (then 2^-18) for
//   the third term; a far outlier opposite the mean makes max|x_c|^2 large and that slack did not cover it (test: an f64 set with
//   such an outlier and near-tied k-th / (k+1)-th neighbours).  With every term explicit, c is its derived value plus 25 %.
// Pass 2 (knn_rerank_kernel) recomputes d for the 16 candidates from the ORIGINAL f32 data (dot product accumulated in f64, rounded once,
//   then the reference's formula in f32), sorts them by (d, index) and returns the first k.  Every point that is NOT a candidate has
//   a >= a_16, hence d >= a_16 - eps_q: when the k-th re-ranked distance is strictly below that, the k neighbours are exactly those an
//   exact search returns.  Queries that fail the test (ties at the k-th neighbour, more than 16 - k points within eps of it) are
//   collected and run through knn_fused.hip.
// Whether to run the filter at all: an exact search of ~1000 queries spread over the query set tells how many it would prove; below 85 %
// (lattices, duplicated points: neighbourhoods that are ties within f32) the exact kernel does the whole search.
// The result is that of an exact f32 search; only the time depends on the data.
#include "device_utils.h"
#include "../core/tensor.h"
#include <type_traits>
#include <vector>

namespace lamp {

typedef _Float16 ks_h8 __attribute__((ext_vector_type(8)));
typedef float ks_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char ks_lds_t;
typedef const __attribute__((address_space(1))) char ks_glb_t;

constexpr int KS_BQ = 256;        // queries per workgroup: 4 waves x 64
constexpr int KS_M = 16;          // candidates kept per query
constexpr int KS_RC = 24;           // capacity of a row's candidate buffer: flushed when a row holds more than KS_RC - 16 (a tile can add 16)

template <int OFF, class V> __device__ __forceinline__ void ks_read128(V& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
template <int I, int N, class F> __device__ __forceinline__ void ks_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ks_static_for<I + 1, N>(f); }
}

__device__ unsigned long long ks_dbg[8];      // diagnostic build (LAMP_KNN_SPLIT_DBG=3): visits, tiles tested, candidates, flushes, cycles in the selection

// points per LDS tile: as many as two tiles + the selection state leave room for (160 KB), in whole 1-KiB DMA pieces per wave
constexpr int ks_tile_points(int dim, int planes) { return planes == 2 || dim == 64 ? 64 : 48; }

struct KsWaveState {
  float lv[64][KS_M];         // the row's 16 best so far, UNSORTED (the re-rank orders its candidates anyway): an insertion replaces the
  int li[64][KS_M];           // largest entry and finds the new largest - no shifting through LDS
  float thr[64];              // the largest value of the row's list
  int mpos[64];               // and where it sits (the highest index among equal values)
  float qn[64];               // the rows' squared norms (the selection reads norms and thresholds from here: indexing the register
                              // copies by a runtime tile number would move those arrays to scratch)
  float bv[64][KS_RC];        // per-row candidate buffers: a visit of the selection appends (LDS atomic counter per row), a flush lets
  int bi[64][KS_RC];          // every row's lane insert its own entries - all rows at the same time
  int bn[64];                 // entries in the row's buffer
};

// rows of PL f16 planes [y0(DIM) | y1(DIM) | ..] of y = (row - mean) * scale (scale: a power of two that brings the largest coordinate near
// 2^14, so every piece is a normal f16 number or a denormal far below the error bound), and the squared norm of (row - mean) in f32:
// one wave per row.  out == nullptr: the norms only.
template <int DIM, int PL, class T>
__global__ __launch_bounds__(256) void knn_split_planes_kernel(const T* __restrict__ x, const T* __restrict__ mean, float scale, _Float16* __restrict__ out,
                                                               float* __restrict__ norm, int64_t rows) {
  constexpr int PER = DIM / 64;                       // features per lane (1 or 2)
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < PER; e++) {
    const int c = lane * PER + e;
    const float cen = (float)(x[r * DIM + c] - mean[c]);      // (f64 rows: the difference in f64, then one rounding to f32 - 2^-24, far inside the bound)
    s = __builtin_fmaf(cen, cen, s);
    float rest = cen * scale;
#pragma unroll
    for (int pl = 0; pl < PL; pl++) {
      const _Float16 piece = (_Float16)rest;          // round to nearest even
      rest -= (float)piece;                           // exact: the piece holds the leading 11 bits of what was left
      if (out) out[r * (PL * DIM) + pl * DIM + c] = piece;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) norm[r] = s;
}

// The structure of knn_fused_kernel (one workgroup = its queries' fragments in registers, the data set streamed through two LDS
// buffers by LDS-DMA, the filter of tile i - 1 in the basic block that multiplies tile i) with
//   * wave = 64 queries (four 16-row tiles) x BC points (64 with two planes; PL = 3 - 33 bits, kept for experiments - 48), a B fragment read feeds
//     4 MFMAs per query plane it meets;
//   * an LDS row per point = its PL planes, 16-byte chunk c of row r at c ^ (r & 15): chunk step j is one 32-deep k-slab of
//     v_mfma_f32_16x16x32_f16 (lane group g = k 8g .. 8g + 7), steps [p NH, (p + 1) NH) are plane p, multiplied by the query planes
//     0 .. PL - 1 - p;
//   * KS_M = 16 candidates per row whatever k is (the re-rank needs the margin).
template <int DIM, int PL, int DBG = 0>
__global__ __launch_bounds__(256, 1) void knn_split_kernel(const _Float16* __restrict__ qs, const _Float16* __restrict__ xs, const float* __restrict__ qn,
                                                           const float* __restrict__ dn, int* __restrict__ out_idx, float* __restrict__ out_val, int Q,
                                                           int N, float m2 /* -2 / scale^2: the dot products of the scaled planes back to the rows' units */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS_BC = ks_tile_points(DIM, PL);   // points per tile
  constexpr int ROWB = 2 * PL * DIM;          // bytes of one point in LDS (PL DIM bf16)
  constexpr int NCHK = ROWB / 16;             // 16-byte chunks per point (>= 16)
  constexpr int NJ = ROWB / 64;               // chunk steps: 4 lane groups x 16 bytes each
  constexpr int NH = NJ / PL;                 // steps of one plane
  constexpr int NCT = KS_BC / 16;
  constexpr int NACC = 4 * NCT;               // accumulator tiles of a wave
  constexpr int TILE = KS_BC * ROWB;
  static_assert(NCHK >= 16, "the swizzle needs at least 16 chunks per row");
  static_assert(TILE % 4096 == 0, "a tile is whole 1-KiB pieces per wave");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int q0 = blockIdx.x * KS_BQ + wid * 64;
  KsWaveState* ws = reinterpret_cast<KsWaveState*>(smem + 2 * TILE) + wid;

  // ---- query fragments (A operand: lane = row c16 of the tile, k = 8 g .. 8 g + 7 of the slab) and norms
  ks_h8 qf[PL][4][NH];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    int row = q0 + 16 * t + c16; row = row < Q ? row : Q - 1;
    const _Float16* qr = qs + (int64_t)row * (PL * DIM);
#pragma unroll
    for (int pl = 0; pl < PL; pl++)
#pragma unroll
      for (int j = 0; j < NH; j++) qf[pl][t][j] = *reinterpret_cast<const ks_h8*>(qr + pl * DIM + 32 * j + 8 * g);
  }
  // ---- selection state
  for (int i = 0; i < KS_M; i++) { ws->lv[lane][i] = INFINITY; ws->li[lane][i] = 0x7fffffff - i; }
  ws->thr[lane] = INFINITY;
  ws->mpos[lane] = 0;
  ws->bn[lane] = 0;
  { int rr = q0 + lane; rr = rr < Q ? rr : Q - 1; ws->qn[lane] = qn[rr]; }

  auto dma_tile = [&](int it, int buf) {
    const int col0 = it * KS_BC;
    constexpr int PIECES = TILE / 1024;
#pragma unroll
    for (int i = 0; i < PIECES / 4; i++) {
      const int piece = wid * (PIECES / 4) + i;
      const int pos = piece * 64 + lane;
      const int row = pos / NCHK, cs = pos % NCHK;
      const int c = cs ^ (row & 15);
      int col = col0 + row; col = col < N ? col : N - 1;
      __builtin_amdgcn_global_load_lds((ks_glb_t*)(xs + (int64_t)col * (PL * DIM) + c * 8), (ks_lds_t*)(smem + buf * TILE + piece * 1024), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned bbase[NJ];                         // byte address of chunk step j of this lane's point in col tile 0, buffer 0
#pragma unroll
  for (int j = 0; j < NJ; j++) { bbase[j] = lds0 + c16 * ROWB + ((((4 * j + g) ^ c16)) << 4); asm volatile("" : "+v"(bbase[j])); }

  const int nit = (N + KS_BC - 1) / KS_BC;
  float dnc[NCT], dnn[NCT], dno[NCT];         // column norms of the tile being multiplied / the next one / the one being filtered
#pragma unroll
  for (int ct = 0; ct < NCT; ct++) { int col = 16 * ct + c16; col = col < N ? col : N - 1; dnn[ct] = dn[col]; dno[ct] = 0.f; }
  ks_f4 old[4][NCT];                          // dot products of the previous tile: filtered while this tile multiplies
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) old[t][ct] = ks_f4{0, 0, 0, 0};
  int col0_old = N;                           // no valid column: the first pass filters nothing

  // The selection proper.  What it costs is LDS round trips and idle lanes, not arithmetic (measured with in-kernel counters, 1M points:
  // a wave sees ~3700 visits and ~17000 candidates; inserted one at a time by the lane that owns the row while 63 lanes wait, a
  // candidate cost ~1500 cycles and the selection as much as all the MFMAs), so it is split:
  //   visit (a wave whose filter fired; only the accumulator tiles that fired): exact test; a passing (value, index) goes to ITS ROW's
  //     buffer - slot = LDS atomic increment of the row's counter.  Thresholds stay as they are (a few more candidates pass later);
  //   flush (some row's buffer could overflow with the next tile, or the end): lane r = row r inserts the entries of its own buffer into
  //     its list - 64 rows in parallel, the trip count is the fullest row's.  An insertion replaces the list's largest entry and finds the
  //     new largest (the list is unsorted: the re-rank orders its candidates anyway); entries of one tile reach the buffer in any
  //     order, so "better" is decided on (value, index).
  // ONE copy of the visit's code for the sixteen accumulator tiles (a runtime loop; the switch moves a tile's values out of the
  // register array): unrolled, it was 150 KB of instructions.
  unsigned long long dbg_visits = 0, dbg_tiles = 0, dbg_cands = 0, dbg_flushes = 0, dbg_cycles = 0, dbg_flush_cycles = 0;
  auto flush_candidates = [&]() {
    unsigned long long t0f = 0;
    if constexpr (DBG == 3) { dbg_flushes++; t0f = __builtin_amdgcn_s_memtime(); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {
      float* lv = ws->lv[lane];
      int* li = ws->li[lane];
      float worst = ws->thr[lane];
      int wpos = ws->mpos[lane];
      int wi = li[wpos];
      const int n = ws->bn[lane];
      if constexpr (DBG == 3) dbg_cands += n;      // (lane 0's rows only: scaled by 64 in the report)
      for (int e = 0; e < n; e++) {
        const float cvv = ws->bv[lane][e];
        const int cii = ws->bi[lane][e];
        if (!(cvv < worst || (cvv == worst && cii < wi))) continue;
        lv[wpos] = cvv; li[wpos] = cii;
        float4 a4[4]; int4 i4[4];                           // the new worst entry: largest value, of equal values the highest index
#pragma unroll
        for (int c = 0; c < 4; c++) { a4[c] = *reinterpret_cast<const float4*>(lv + 4 * c); i4[c] = *reinterpret_cast<const int4*>(li + 4 * c); }
        float mv = a4[0].x; int mi = i4[0].x, mp = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float vv[4] = {a4[c].x, a4[c].y, a4[c].z, a4[c].w};
          const int ii[4] = {i4[c].x, i4[c].y, i4[c].z, i4[c].w};
#pragma unroll
          for (int w2 = 0; w2 < 4; w2++) {
            const bool gt = vv[w2] > mv || (vv[w2] == mv && ii[w2] > mi);
            mv = gt ? vv[w2] : mv; mi = gt ? ii[w2] : mi; mp = gt ? 4 * c + w2 : mp;
          }
        }
        worst = mv; wi = mi; wpos = mp;
      }
      if (n) { ws->thr[lane] = worst; ws->mpos[lane] = wpos; ws->bn[lane] = 0; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if constexpr (DBG == 3) dbg_flush_cycles += __builtin_amdgcn_s_memtime() - t0f;
  };
  auto select_tile = [&](const ks_f4 (&a)[4][NCT], const float* dnv, int c0, unsigned tmask) {
    unsigned long long t0v = 0;
    if constexpr (DBG == 3) { dbg_visits++; t0v = __builtin_amdgcn_s_memtime(); }
    bool full = false;                            // some row of this lane's group is within one tile of its buffer's capacity
    while (tmask) {
      if constexpr (DBG == 3) dbg_tiles++;
      const int idx = __builtin_ctz(tmask);
      tmask &= tmask - 1;
      const int t = idx / NCT, ct = idx % NCT;
      // the row group's norms and thresholds: requested first, so that the round trip runs under the switch below
      const float4 qq = *reinterpret_cast<const float4*>(ws->qn + 16 * t + 4 * g), tt4 = *reinterpret_cast<const float4*>(ws->thr + 16 * t + 4 * g);
      ks_f4 av = ks_f4{0, 0, 0, 0}; float dn1 = 0.f;
      switch (idx) {
#define KS_CASE(I) case I: if constexpr ((I) < NACC) { av = a[(I) / NCT][(I) % NCT]; dn1 = dnv[(I) % NCT]; } break;
        KS_CASE(0) KS_CASE(1) KS_CASE(2) KS_CASE(3) KS_CASE(4) KS_CASE(5) KS_CASE(6) KS_CASE(7)
        KS_CASE(8) KS_CASE(9) KS_CASE(10) KS_CASE(11) KS_CASE(12) KS_CASE(13) KS_CASE(14) KS_CASE(15)
        default: break;
#undef KS_CASE
      }
      const float q4[4] = {qq.x, qq.y, qq.z, qq.w}, t4[4] = {tt4.x, tt4.y, tt4.z, tt4.w};
      const int col = c0 + 16 * ct + c16;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float d = __builtin_fmaf(m2, av[r], q4[r] + dn1);
        const float v = d > 0.f ? d : 0.f;
        // (<=: an equal value with a lower index than the list's worst entry must reach the flush, which decides on (value, index))
        if (col < N && v <= t4[r]) {
          const int row = 16 * t + 4 * g + r;
          const int slot = atomicAdd(&ws->bn[row], 1);     // ds_add_rtn_u32: the sixteen lanes of a group that hit the same row get distinct slots
          ws->bv[row][slot] = v; ws->bi[row][slot] = col;
          full |= slot >= KS_RC - 17;
        }
      }
      if (__builtin_amdgcn_ballot_w64(full) != 0) { flush_candidates(); full = false; }
    }
    if constexpr (DBG == 3) dbg_cycles += __builtin_amdgcn_s_memtime() - t0v;
  };
  // The filter "does any distance of the previous tile reach its row's 16th best" as integer arithmetic between the MFMAs of this tile:
  // for non-negative floats a <= b <=> bits(a) <= bits(b).  One fma, one max, one subtract, one min per value; one bit per accumulator
  // tile decides.  Columns beyond N repeat the last point: they can only cause a needless visit of select_tile, which tests the column.
  auto filter_piece = [&](auto idxc, const ks_f4& a, float dnv, unsigned& mask) {
    constexpr int idx = decltype(idxc)::value, t = idx / NCT;
    // the row group's norms and thresholds come from LDS (two 16-byte reads, the same address for the sixteen lanes of a group): held
    // in registers next to the 128 registers of query fragments they were spilled to scratch
    const float4 qq = *reinterpret_cast<const float4*>(ws->qn + 16 * t + 4 * g);
    const int4 th = *reinterpret_cast<const int4*>(ws->thr + 16 * t + 4 * g);
    // (the clamp is part of the test: the bit pattern of a negative distance minus a threshold's would wrap around)
    int m = (int)__float_as_uint(fmaxf(__builtin_fmaf(m2, a[0], qq.x + dnv), 0.f)) - th.x;
    m = min(m, (int)__float_as_uint(fmaxf(__builtin_fmaf(m2, a[1], qq.y + dnv), 0.f)) - th.y);
    m = min(m, (int)__float_as_uint(fmaxf(__builtin_fmaf(m2, a[2], qq.z + dnv), 0.f)) - th.z);
    m = min(m, (int)__float_as_uint(fmaxf(__builtin_fmaf(m2, a[3], qq.w + dnv), 0.f)) - th.w);
    mask |= (__builtin_amdgcn_ballot_w64(m <= 0) != 0 ? 1u : 0u) << idx;       // one bit per accumulator tile, kept in a scalar register
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
      for (int ct = 0; ct < NCT; ct++) { int col = (it + 1) * KS_BC + 16 * ct + c16; col = col < N ? col : N - 1; dnn[ct] = dn[col]; }
    }
    ks_f4 acc[4][NCT];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) acc[t][ct] = ks_f4{0, 0, 0, 0};
    const unsigned boff = buf * TILE;
    ks_h8 bf[2][NCT];
    auto b_issue = [&](auto jc, ks_h8* dst) {
      constexpr int j = decltype(jc)::value;
      ks_static_for<0, NCT>([&](auto ctc) { constexpr int ct = decltype(ctc)::value; ks_read128<ct * 16 * ROWB>(dst[ct], bbase[j] + boff); });
    };
    auto b_fence = [&](ks_h8* f, bool last) {
      if constexpr (NCT == 4) {
        if (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : : "memory");
      } else if constexpr (NCT == 3) {
        if (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
      } else {
        if (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]), "+v"(f[1]) : : "memory");
      }
    };
    b_issue(std::integral_constant<int, 0>{}, bf[0]);
    unsigned fmask = 0;                           // accumulator tiles of the previous tile in which some distance reaches its row's threshold
    ks_static_for<0, NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j + 1 < NJ) { b_issue(std::integral_constant<int, j + 1>{}, bf[(j + 1) & 1]); b_fence(bf[j & 1], false); }
      else b_fence(bf[j & 1], true);
      // data plane p = j / NH against the query planes 0 .. PL - 1 - p; NACC independent accumulators between two uses of the same one
      // (a dependent MFMA waits for its predecessor's result)
      constexpr int p = j / NH, jj = j % NH;
      ks_static_for<0, PL - p>([&](auto qc) {
        constexpr int qp = decltype(qc)::value;
#pragma unroll
        for (int ct = 0; ct < NCT; ct++)
#pragma unroll
          for (int t = 0; t < 4; t++) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[qp][t][jj], bf[j & 1][ct], acc[t][ct], 0, 0, 0);
      });
      // the filter of the PREVIOUS tile's accumulator tiles, spread over the steps: vector work that issues while the matrix pipe runs
      constexpr int PER = (NACC + NJ - 1) / NJ;
      if constexpr (DBG != 2)
        ks_static_for<0, PER>([&](auto uc) {
          constexpr int idx = j * PER + decltype(uc)::value;
          if constexpr (idx < NACC) filter_piece(std::integral_constant<int, idx>{}, old[idx / NCT][idx % NCT], dno[idx % NCT], fmask);
        });
    });
    unsigned tmask = col0_old < N ? fmask : 0u;
    if (DBG == 1 || DBG == 2) tmask = 0;
    if (tmask) select_tile(old, dno, col0_old, tmask);
    col0_old = it * KS_BC;
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) dno[ct] = dnc[ct];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int ct = 0; ct < NCT; ct++) old[t][ct] = acc[t][ct];
  }
  select_tile(old, dno, col0_old, (1u << NACC) - 1u);
  flush_candidates();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if constexpr (DBG == 3) {
    if (lane == 0) {
      atomicAdd(&ks_dbg[0], dbg_visits); atomicAdd(&ks_dbg[1], dbg_tiles); atomicAdd(&ks_dbg[2], dbg_cands); atomicAdd(&ks_dbg[3], dbg_flushes);
      atomicAdd(&ks_dbg[4], dbg_cycles); atomicAdd(&ks_dbg[5], dbg_flush_cycles); atomicAdd(&ks_dbg[6], 1ull);
    }
  }
  if (q0 + lane < Q) {
    for (int i = 0; i < KS_M; i++) out_idx[(int64_t)(q0 + lane) * KS_M + i] = ws->li[lane][i];
    out_val[q0 + lane] = ws->thr[lane];         // a_16: no point outside the list has a smaller approximate distance
  }
}

// One wave per query: the exact distances of its KS_M candidates (f32 data, the dot product summed in f64 and rounded once, then
// the reference's (|q|^2 + |x|^2) - 2 q.x in f32 with the clamp), ranked by (distance, index).  The first k go to the result; the
// query is appended to `failed` unless the k-th distance is strictly below every distance a non-candidate can have.
template <int DIM, class T>
__global__ __launch_bounds__(256) void knn_rerank_kernel(const T* __restrict__ q, const T* __restrict__ x, const T* __restrict__ qn, const T* __restrict__ dn,
                                                         const float* __restrict__ qn_c, const float* __restrict__ dn_c_max, const T* __restrict__ dn_max, float c_dot,
                                                         const int* __restrict__ cand_idx, const float* __restrict__ cand_val, int64_t* __restrict__ out_idx,
                                                         T* __restrict__ out_val, int* __restrict__ failed, int* __restrict__ nfailed, int Q, int N, int k) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= Q) return;
  const int c = lane >> 2, part = lane & 3;             // candidate, quarter of the features
  const int ci = cand_idx[(int64_t)qi * KS_M + c];
  const bool valid = ci >= 0 && ci < N;                   // fewer than KS_M points: the padding entries
  const T* qr = q + (int64_t)qi * DIM + part * (DIM / 4);
  const T* xr = x + (int64_t)(valid ? ci : 0) * DIM + part * (DIM / 4);
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < DIM / 4; i++) s += (double)qr[i] * (double)xr[i];
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  // the reference's formula in the data's precision (f32: the dot product rounded once to f32 first)
  const T qnv = qn[qi];
  T d = (qnv + dn[valid ? ci : 0]) - T(2) * (T)s;
  d = d > T(0) ? d : T(0);
  if (!valid) d = (T)INFINITY;
  // rank among the KS_M candidates by (d, index): every lane of a candidate computes the same rank
  int rank = 0;
#pragma unroll
  for (int o = 0; o < KS_M; o++) {
    const T od = __shfl(d, o * 4);
    const int oi = __shfl(ci, o * 4);
    rank += (od < d || (od == d && oi < ci)) ? 1 : 0;
  }
  if (part == 0 && rank < k) {
    out_idx[(int64_t)qi * k + rank] = ci;
    out_val[(int64_t)qi * k + rank] = d;
  }
  // the proof: a point outside the candidate list has an approximate distance >= a_last, and |approximate - exact| <= eps
  // (the filter worked on the centred rows: its error scales with their norms; the formula's own rounding - f32 only - with the original ones)
  const double a_last = (double)cand_val[qi];
  const double noise = std::is_same<T, float>::value ? 0x1p-21 * ((double)qnv + (double)dn_max[0]) : 0x1p-50 * ((double)qnv + (double)dn_max[0]);
  // + the f32 arithmetic of the quantities the FILTER compared: the centred norms qn_c / dn_c are accumulated in f32 and the
  // centred rows are rounded to f32 before they are split - at most 2^-21 + 2^-22 of (|q_c|^2 + |x_c|^2), bounded by 2^-20 of it; on the f64
  // path nothing else covers these (its `noise` is 2^-50)
  const double filter_noise = 0x1.8p-21 * ((double)qn_c[qi] + (double)dn_c_max[0]);
  const double eps = 2.0 * (double)c_dot * sqrt((double)qn_c[qi]) * sqrt((double)dn_c_max[0]) + noise + filter_noise;
  const bool all_points_are_candidates = N <= KS_M;
  if (part == 0 && rank == k - 1 && !all_points_are_candidates && !((double)d < a_last - eps)) {
    const int slot = atomicAdd(nfailed, 1);
    failed[slot] = qi;
  }
}

template <class T>
__global__ __launch_bounds__(256) void knn_gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ rows, T* __restrict__ dst, int64_t n,
                                                              int64_t width) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * width; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / width, c = e - r * width;
    dst[e] = src[(int64_t)rows[r] * width + c];
  }
}
template <class T>
__global__ __launch_bounds__(256) void knn_scatter_results_kernel(const int64_t* __restrict__ si, const T* __restrict__ sv, const int* __restrict__ rows,
                                                                  int64_t* __restrict__ di, T* __restrict__ dv, int64_t n, int64_t k) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * k; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / k, c = e - r * k;
    di[(int64_t)rows[r] * k + c] = si[e];
    dv[(int64_t)rows[r] * k + c] = sv[e];
  }
}

__global__ void knn_strided_ids_kernel(int* __restrict__ ids, int64_t n, int64_t stride) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) ids[i] = (int)(i * stride);
}
// would the filter prove this query?  d16 - dk of an EXACT search against the error bound (both ends of the gap move by at most eps)
template <class T>
__global__ void knn_predict_kernel(const T* __restrict__ val16, const T* __restrict__ qn, const float* __restrict__ qn_c, const float* __restrict__ dn_c_max,
                                   const T* __restrict__ dn_max, float c_dot, int* __restrict__ unproven, int S, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const double noise = (std::is_same<T, float>::value ? 0x1p-21 : 0x1p-50) * ((double)qn[i] + (double)dn_max[0]);
  const double filter_noise = 0x1.8p-21 * ((double)qn_c[i] + (double)dn_c_max[0]);     // as in knn_rerank_kernel
  const double eps = 2.0 * (double)c_dot * sqrt((double)qn_c[i]) * sqrt((double)dn_c_max[0]) + noise + filter_noise;
  const double gap = (double)val16[(int64_t)i * KS_M + KS_M - 1] - (double)val16[(int64_t)i * KS_M + k - 1];
  if (!(gap > 2.0 * eps)) atomicAdd(unproven, 1);
}

bool knn_fused(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t dim,
               int64_t k, hipStream_t st, int kind);   // knn_fused.hip
Tensor* reduce_dims(const Tensor* a, const int64_t* dims, int ndims, bool keepdim, int op);

namespace {
// (process-wide switches and diagnostics, not synchronised: set the mode before searching from several threads)
int g_knn_split_mode = 1;      // 0 never, 1 where it pays (large searches), 2 whenever the shape is covered (tests)
int64_t g_knn_split_failed = 0;
int g_knn_split_planes = 0;    // planes of the last search (0: the split path did not run)
}
void knn_split_set_mode(int mode) { g_knn_split_mode = mode; }
int64_t knn_split_last_failed() { return g_knn_split_failed; }
int knn_split_last_planes() { return g_knn_split_planes; }

namespace {
constexpr float KS_C_DOT = 0x1.3p-19f;   // 2.27e-6: |filter's dot product - exact| <= KS_C_DOT |q_c| |x_c| (header comment)

// the rows of `src`, centred and scaled, as PL f16 planes (planes == nullptr: none) + the squared norms of the centred rows
template <int DIM, int PL, class T>
void make_planes(const Tensor* src, int64_t row0, const Tensor* mean, float scale, Tensor* planes, Tensor* norm, int64_t rows, hipStream_t st) {
  hipLaunchKernelGGL((knn_split_planes_kernel<DIM, PL, T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, src->ptr<T>() + row0 * DIM, mean->ptr<T>(), scale,
                     planes ? reinterpret_cast<_Float16*>(planes->raw()) : (_Float16*)nullptr, norm->ptr<float>(), rows);
  LAMP_LAUNCH_CHECK();
}
// filter + re-rank of all queries with PL planes; returns the number of queries without proof, their ids in failed[1 ..]
template <int DIM, int PL, class T>
int split_pass(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, const Tensor* dn_max, const Tensor* mean, float scale, const Tensor* dnc,
               const Tensor* dnc_max, const Tensor* qnc, Tensor* idx, Tensor* val, Tensor* failed, int64_t Q, int64_t N, int64_t k, hipStream_t st) {
  const int dev = x->device();
  Hold xs(new_tensor({N, (int64_t)PL * DIM}, kF16, dev)), qs(new_tensor({Q, (int64_t)PL * DIM}, kF16, dev)), scratch_n(new_tensor({std::max(N, Q)}, kF32, dev));
  make_planes<DIM, PL, T>(x, 0, mean, scale, xs.get(), scratch_n.get(), N, st);
  make_planes<DIM, PL, T>(q, 0, mean, scale, qs.get(), scratch_n.get(), Q, st);
  Hold ci(new_tensor({Q, (int64_t)KS_M}, kI32, dev)), cv(new_tensor({Q}, kF32, dev));        // candidates (unsorted) and a_16 per query
  {
    // declared: the algorithmic work of the search, as knn_fused declares it (the f16 pipe executes PL (PL + 1) / 2 times the products)
    KernelTimer kt(PL == 2 ? "knn_split_f16x3" : "knn_split_f16x6", 2.0 * (double)Q * N * DIM, ((double)Q + N) * DIM * sizeof(T), st);
    constexpr int BC = ks_tile_points(DIM, PL);
    const size_t lds = (size_t)2 * BC * 2 * PL * DIM + 4 * sizeof(KsWaveState);
    const float m2 = -2.f / (scale * scale);
    const char* dbg = getenv("LAMP_KNN_SPLIT_DBG");
    const int dm = dbg ? atoi(dbg) : 0;
#define KS_LAUNCH(DB) do { allow_big_lds((const void*)knn_split_kernel<DIM, PL, DB>); hipLaunchKernelGGL((knn_split_kernel<DIM, PL, DB>), dim3((unsigned)((Q + KS_BQ - 1) / KS_BQ)), dim3(256), lds, st, reinterpret_cast<const _Float16*>(qs->raw()), reinterpret_cast<const _Float16*>(xs->raw()), qnc->ptr<float>(), dnc->ptr<float>(), ci->ptr<int>(), cv->ptr<float>(), (int)Q, (int)N, m2); } while (0)
    if (dm == 3) KS_LAUNCH(3); else KS_LAUNCH(0);
#undef KS_LAUNCH
    LAMP_LAUNCH_CHECK();
    if (dm == 3) {
      HIP_CHECK(hipStreamSynchronize(st));
      unsigned long long h[8] = {};
      HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ks_dbg), sizeof(h)));
      fprintf(stderr, "knn_split dbg (%d planes): waves %llu  visits/wave %.0f  tiles tested/wave %.0f  candidates/wave %.0f  flushes/wave %.1f  selection cycles/wave %.3e (flush part %.3e)\n",
              PL, h[6], (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] * 64 / h[6], (double)h[3] / h[6], (double)h[4] / h[6], (double)h[5] / h[6]);
      unsigned long long z[8] = {};
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(ks_dbg), z, sizeof(z)));
    }
  }
  HIP_CHECK(hipMemsetAsync(failed->raw(), 0, sizeof(int), st));
  hipLaunchKernelGGL((knn_rerank_kernel<DIM, T>), dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, st, q->ptr<T>(), x->ptr<T>(), qn->ptr<T>(), dn->ptr<T>(),
                     qnc->ptr<float>(), dnc_max->ptr<float>(), dn_max->ptr<T>(), KS_C_DOT, ci->ptr<int>(), cv->ptr<float>(), idx->ptr<int64_t>(), val->ptr<T>(),
                     failed->ptr<int>() + 1, failed->ptr<int>(), (int)Q, (int)N, (int)k);
  LAMP_LAUNCH_CHECK();
  int nfail = 0;
  HIP_CHECK(hipMemcpyAsync(&nfail, failed->raw(), sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return nfail;
}
// the queries listed in failed[1 .. nfail] through the exact kernel
template <int DIM, class T>
void exact_for_failed(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, const Tensor* failed, int nfail, int64_t N,
                      int64_t k, hipStream_t st) {
  const int dev = x->device();
  Hold fq(new_tensor({(int64_t)nfail, (int64_t)DIM}, q->dtype, dev)), fqn(new_tensor({(int64_t)nfail}, q->dtype, dev));
  const int* rows = failed->ptr<int>() + 1;
  hipLaunchKernelGGL((knn_gather_rows_kernel<T>), dim3(grid_for((int64_t)nfail * DIM, 256)), dim3(256), 0, st, q->ptr<T>(), rows, fq->ptr<T>(), (int64_t)nfail, (int64_t)DIM);
  hipLaunchKernelGGL((knn_gather_rows_kernel<T>), dim3(grid_for((int64_t)nfail, 256)), dim3(256), 0, st, qn->ptr<T>(), rows, fqn->ptr<T>(), (int64_t)nfail, (int64_t)1);
  LAMP_LAUNCH_CHECK();
  Hold fi(new_tensor({(int64_t)nfail, k}, kI64, dev)), fv(new_tensor({(int64_t)nfail, k}, q->dtype, dev));
  LAMP_CHECK(knn_fused(fq.get(), x, fqn.get(), dn, fi.get(), fv.get(), nfail, N, DIM, k, st, 0), "internal: the exact kNN kernel refused the fallback queries");
  hipLaunchKernelGGL((knn_scatter_results_kernel<T>), dim3(grid_for((int64_t)nfail * k, 256)), dim3(256), 0, st, fi->ptr<int64_t>(), fv->ptr<T>(), rows, idx->ptr<int64_t>(),
                     val->ptr<T>(), (int64_t)nfail, k);
  LAMP_LAUNCH_CHECK();
}
}  // namespace

template <int DIM, class T>
static bool knn_split_run(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t k,
                          hipStream_t st, bool forced) {
  const int dev = x->device();
  const int64_t zero = 0;
  if (N < KS_M) return false;
  Hold mean(reduce_dims(x, &zero, 1, false, 1));                  // the data set's mean row: the filter works on (row - mean)
  Hold dn_max(reduce_dims(dn, nullptr, 0, false, 3));
  Hold failed(new_tensor({Q + 1}, kI32, dev));                    // [0] = count, then the queries
  // squared norms of the centred rows (their largest sets the error bound and the scale of the f16 planes)
  Hold dnc(new_tensor({N}, kF32, dev)), qnc(new_tensor({Q}, kF32, dev));
  make_planes<DIM, 2, T>(x, 0, mean.get(), 1.f, nullptr, dnc.get(), N, st);
  make_planes<DIM, 2, T>(q, 0, mean.get(), 1.f, nullptr, qnc.get(), Q, st);
  Hold dnc_max(reduce_dims(dnc.get(), nullptr, 0, false, 3)), qnc_max(reduce_dims(qnc.get(), nullptr, 0, false, 3));
  // Is the filter worth running?  An exact search of ~1000 queries spread over the query set (16 neighbours each; the data set in slices
  // over the CUs: ~3 ms at 1M points) gives their d_k and d_16; the filter proves a query when that gap exceeds twice its error bound.
  // If it would leave more than 15 % of the sample unproven, the neighbourhoods of this data are ties within f32 (lattices, duplicates)
  // and the filter would only add its time to the exact kernel's: not used.
  const int64_t S = std::min<int64_t>(Q, 1024), stride = Q / S;
  Hold ids(new_tensor({S}, kI32, dev)), sq(new_tensor({S, (int64_t)DIM}, q->dtype, dev)), sqn(new_tensor({S}, q->dtype, dev)), sqc(new_tensor({S}, kF32, dev));
  hipLaunchKernelGGL(knn_strided_ids_kernel, dim3(grid_for(S, 256)), dim3(256), 0, st, ids->ptr<int>(), S, stride);
  hipLaunchKernelGGL((knn_gather_rows_kernel<T>), dim3(grid_for(S * DIM, 256)), dim3(256), 0, st, q->ptr<T>(), ids->ptr<int>(), sq->ptr<T>(), S, (int64_t)DIM);
  hipLaunchKernelGGL((knn_gather_rows_kernel<T>), dim3(grid_for(S, 256)), dim3(256), 0, st, qn->ptr<T>(), ids->ptr<int>(), sqn->ptr<T>(), S, (int64_t)1);
  hipLaunchKernelGGL((knn_gather_rows_kernel<float>), dim3(grid_for(S, 256)), dim3(256), 0, st, qnc->ptr<float>(), ids->ptr<int>(), sqc->ptr<float>(), S, (int64_t)1);
  LAMP_LAUNCH_CHECK();
  Hold si(new_tensor({S, (int64_t)KS_M}, kI64, dev)), sv(new_tensor({S, (int64_t)KS_M}, q->dtype, dev));
  if (!knn_fused(sq.get(), x, sqn.get(), dn, si.get(), sv.get(), S, N, DIM, KS_M, st, 0)) return false;
  HIP_CHECK(hipMemsetAsync(failed->raw(), 0, sizeof(int), st));
  hipLaunchKernelGGL((knn_predict_kernel<T>), dim3(grid_for(S, 256)), dim3(256), 0, st, sv->ptr<T>(), sqn->ptr<T>(), sqc->ptr<float>(), dnc_max->ptr<float>(),
                     dn_max->ptr<T>(), KS_C_DOT, failed->ptr<int>(), (int)S, (int)k);
  LAMP_LAUNCH_CHECK();
  int unproven = 0;
  float mx[2] = {0.f, 0.f};
  HIP_CHECK(hipMemcpyAsync(&unproven, failed->raw(), sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&mx[0], dnc_max->raw(), sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipMemcpyAsync(&mx[1], qnc_max->raw(), sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  if (!forced && (int64_t)unproven * 100 > 15 * S) return false;
  // a power of two that puts the largest coordinate of any centred row at or below 2^14 (f16 holds 65504)
  const float big = std::sqrt(std::max(mx[0], mx[1]));
  if (!std::isfinite(big)) return false;                          // NaN / Inf in the data or the queries: the exact kernel's behaviour, not an error of this path
  const float scale = big > 0.f ? std::exp2(std::floor(14.f - std::log2(big))) : 1.f;
  static const int env_planes = [] { const char* e = getenv("LAMP_KNN_SPLIT_PLANES"); return e ? atoi(e) : 0; }();
  const int planes = env_planes == 3 ? 3 : 2;
  const int nf = planes == 2 ? split_pass<DIM, 2, T>(q, x, qn, dn, dn_max.get(), mean.get(), scale, dnc.get(), dnc_max.get(), qnc.get(), idx, val, failed.get(), Q, N, k, st)
                             : split_pass<DIM, 3, T>(q, x, qn, dn, dn_max.get(), mean.get(), scale, dnc.get(), dnc_max.get(), qnc.get(), idx, val, failed.get(), Q, N, k, st);
  if (nf) exact_for_failed<DIM, T>(q, x, qn, dn, idx, val, failed.get(), nf, N, k, st);
  g_knn_split_failed = nf;
  g_knn_split_planes = planes;
  return true;
}

// f32 / f64 squared-Euclidean search of 64 / 128 features, k <= 12 (16 candidates leave a margin of at least 4).  false: not covered / not worth it.
// (f64 - lamp's default DoublePrecision: the same f16 filter on the rows rounded to f32 after centring; the re-rank and the proof in f64.)
bool knn_split(const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn, Tensor* idx, Tensor* val, int64_t Q, int64_t N, int64_t dim, int64_t k,
               hipStream_t st) {
  static const int env_mode = [] { const char* e = getenv("LAMP_KNN_SPLIT"); return e ? atoi(e) : -1; }();
  const int mode = env_mode >= 0 ? env_mode : g_knn_split_mode;
  g_knn_split_planes = 0;
  const bool f32 = q->dtype == kF32, f64 = q->dtype == kF64;
  if (mode == 0 || !(f32 || f64) || !(dim == 64 || dim == 128) || k < 1 || k > 12 || N > 0x7fffff00 || Q > 0x7fffff00 || N < k || Q < 1) return false;
  if ((((uintptr_t)q->data() | (uintptr_t)x->data()) & 15) != 0) return false;
  // the extra passes (mean, norms, planes, sample, re-rank) and the host round trips for the verdicts cost ~0.5 ms: below ~4e9 distance evaluations the exact kernel is as fast
  if (mode == 1 && ((double)Q * (double)N < 4.0e9 || N < 16384)) return false;
  const bool forced = mode == 2;
  if (f32) return dim == 128 ? knn_split_run<128, float>(q, x, qn, dn, idx, val, Q, N, k, st, forced) : knn_split_run<64, float>(q, x, qn, dn, idx, val, Q, N, k, st, forced);
  return dim == 128 ? knn_split_run<128, double>(q, x, qn, dn, idx, val, Q, N, k, st, forced) : knn_split_run<64, double>(q, x, qn, dn, idx, val, Q, N, k, st, forced);
}

}  // namespace lamp

extern "C" {
/* 0: never use the split-bf16 filter, 1: where it pays (default), 2: whenever the shape is covered.  LAMP_KNN_SPLIT overrides. */
int lamp_knn_split_mode(int mode) {
  LAMP_API_BEGIN
  LAMP_CHECK(mode >= 0 && mode <= 2, "knn split mode must be 0, 1 or 2");
  lamp::knn_split_set_mode(mode);
  LAMP_API_END
}
/* queries of the last split search on this process that the filter could not prove and the exact kernel re-ran */
int lamp_knn_split_last_failed(int64_t* out) {
  LAMP_API_BEGIN
  *out = lamp::knn_split_last_failed();
  LAMP_API_END
}
/* bf16 planes per value the last search used: 2 or 3; 0 = the split path did not run */
int lamp_knn_split_last_planes(int* out) {
  LAMP_API_BEGIN
  *out = lamp::knn_split_last_planes();
  LAMP_API_END
}
}
