// GEMM family on the gfx950 matrix cores.
//
// Replaces ATen.mm / bmm / addmm / baddbmm / matmul and aten-scala's custom natives
// Tensor.addmm_out_transposed1/2, Tensor.baddbmm_out_transposed1/2 (reference call sites:
// lamp-sten/src/main/scala/lamp/STen.scala:391-449,1146,1220-1240;
// lamp-core/src/main/scala/lamp/autograd/ops.scala:665-724 - MatMul/BatchedMatMul forward
// `mm`, backward dA += p.B^T (transposed2) and dB += A^T.p (transposed1), beta = alpha = 1).
//
//   C[m,n] = beta * S[m,n] + alpha * sum_k A(m,k) * B(k,n)
//
// All four storage combinations are handled WITHOUT materialising a transpose: an operand
// whose K index is contiguous in memory is staged as [rows][K] and its MFMA fragments are
// read with ds_read_b128; an operand whose K index is strided (B of a row-major mm, both
// operands of A^T.B) is staged in its natural [K][cols] layout (coalesced 16-byte global
// loads) and its fragments come from the CDNA4 transposing LDS read ds_read_b64_tr_b16.
//
// bf16: v_mfma_f32_16x16x32_bf16, fp32 accumulate, one rounding to bf16 in the epilogue.
// f32 : v_mfma_f32_16x16x4_f32 - exact f32 FMA chain (no tf32 on gfx950), parity <= 1e-5.
// f64 : v_mfma_f64_16x16x4_f64 (the reference's own tests run in double).
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include "device_utils.h"
#include "../core/strided.h"

namespace lamp {

typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef double d4_t __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const void* S;          // beta operand (may alias C, may be null)
  int64_t M, N, K;
  int64_t a_rs, a_cs;     // A(m,k) = A[m*a_rs + k*a_cs]; exactly one of them is 1 for the fast path
  int64_t b_rs, b_cs;     // B(k,n) = B[k*b_rs + n*b_cs]
  int64_t ldc;            // C(m,n) = C[m*ldc + n]
  int64_t s_rs, s_cs;     // S(m,n) = S[m*s_rs + n*s_cs] (0 strides broadcast)
  int64_t a_bs, b_bs, c_bs, s_bs;  // batch strides
  int batch;
  int a_vec, b_vec;       // 16-byte vector loads legal for this operand
  double alpha, beta;
  int tiles_m, tiles_n;
  // optional kNN epilogue of the f32/f64 kernel: C = max(0, (knn_q[m] + knn_d[n]) - 2 * C)   (knn/package.scala:21-30)
  const void* knn_q;
  const void* knn_d;
  // split-K (bf16 256 x 256 kernel): blockIdx.z walks K chunks (a_bs / b_bs = one chunk along K) and the raw f32 accumulators go to
  // C = float[split][M][N]; gemm_splitk_reduce_kernel sums the slices and applies alpha / beta
  int split_f32;
  // bf16: round alpha * acc to bf16 before the beta operand is added - bitwise the chain `mm` then `add` that lamp's Linear issues
  int round_first;
  // split-K of the f32 / f64 kernel: blockIdx.z = K chunk of g.K (the last one shorter: k_split_total - z g.K), raw sums to C[z][M][N]
  int64_t k_split_total;
};

// ================================================================================================
// bf16 kernel: 128 x 128 x 64 tile, 256 threads = 4 waves (2 x 2), 64 x 64 per wave
// ================================================================================================
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

// LDS image of a K-contiguous operand tile: [128 rows][64 k] bf16, 128-byte rows of eight
// 16-byte chunks, chunk' = chunk ^ (row & 7)  (conflict-free ds_read_b128 for the
// 16-row x 16-byte fragment read pattern).
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// LDS image of a K-strided operand tile: [64 k][128 cols] bf16, 256-byte rows of eight
// 32-byte chunks, chunk' = chunk ^ f(k) with f(k) = (k & 3) | ((k >> 3) & 1) << 2 (the rows a
// 32-lane half touches in one ds_read_b64_tr_b16 land in 8 different chunks).
__device__ __forceinline__ int ks_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int ks_off(int k, int col8 /* 16-byte chunk index 0..15 */) {
  return k * 256 + ((((col8 >> 1) ^ ks_swz(k))) << 5) + ((col8 & 1) << 4);
}

// global -> registers for one 128x64 (or 64x128) operand tile: 4 x 16 bytes per thread
template <bool KC>
__device__ __forceinline__ void stage_load(uint4 (&r)[4], const bf16_t* __restrict__ base, int64_t ld, int64_t row0,
                                           int64_t k0, int64_t rows, int64_t K, int vec_ok, int tid) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = tid + i * 256;
    int64_t gr, gk;     // "row" = M or N index, "k" = K index
    int64_t addr;
    bool full;
    if (KC) {           // [rows][K], K contiguous: 8 chunks per row
      gr = row0 + (c >> 3);
      gk = k0 + ((c & 7) << 3);
      addr = gr * ld + gk;
      full = (gr < rows) && (gk + 8 <= K);
    } else {            // [K][rows], row index contiguous: 16 chunks per k
      gk = k0 + (c >> 4);
      gr = row0 + ((c & 15) << 3);
      addr = gk * ld + gr;
      full = (gk < K) && (gr + 8 <= rows);
    }
    if (full && vec_ok) {
      r[i] = *reinterpret_cast<const uint4*>(base + addr);
    } else {
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        bool ok = KC ? (gr < rows && gk + j < K) : (gk < K && gr + j < rows);
        e[j] = ok ? base[addr + j].bits : (unsigned short)0;
      }
      r[i] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
  }
}
template <bool KC> __device__ __forceinline__ void stage_store(const uint4 (&r)[4], char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = tid + i * 256;
    const int off = KC ? kc_off(c >> 3, c & 7) : ks_off(c >> 4, c & 15);
    *reinterpret_cast<uint4*>(lds + off) = r[i];
  }
}

// fragment for MFMA 16x16x32: lane l needs 8 consecutive k (k = 32*s + 8*(l>>4) + j) of row/col (l & 15)
template <bool KC> __device__ __forceinline__ bf8_t load_frag(const char* lds, int tile_row0, int s, int lane) {
  if (KC) {
    const int row = tile_row0 + (lane & 15);
    const int chunk = s * 4 + (lane >> 4);
    s8_t v = *reinterpret_cast<const s8_t*>(lds + kc_off(row, chunk));
    return __builtin_bit_cast(bf8_t, v);
  } else {
    // transposing read: within a 16-lane group lane 4q+p supplies the address of row q, columns 4p..4p+3
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int k = s * 32 + g * 8 + q;
    const int col8 = (tile_row0 >> 3) + (p >> 1);           // 16-byte chunk along the contiguous index
    const int off0 = ks_off(k, col8) + ((p & 1) << 3);
    const int off1 = ks_off(k + 4, col8) + ((p & 1) << 3);
    typedef __attribute__((address_space(3))) s4_t* lds_ptr_t;
    s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(lds + off0));
    s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(lds + off1));
    s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8_t, v);
  }
}

// E = bf16_t or f16_t: the two 16-bit floating types share the staging (bit patterns) and the fragment layout; only the MFMA opcode
// (v_mfma_f32_16x16x32_bf16 / _f16) and the epilogue's rounding differ
template <class E> __device__ __forceinline__ f4_t mfma16(bf8_t a, bf8_t b, f4_t c);
template <> __device__ __forceinline__ f4_t mfma16<bf16_t>(bf8_t a, bf8_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
template <> __device__ __forceinline__ f4_t mfma16<f16_t>(bf8_t a, bf8_t b, f4_t c) {
  typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}
template <bool AKC, bool BKC, class E = bf16_t>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware tile order: consecutive tiles of one XCD share A row-panels in that XCD's L2
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  // group tiles in column strips of 8 tile rows for L2 reuse of the B panel
  const int GROUP = 8;
  const int per_group = GROUP * g.tiles_n;
  const int group = bid / per_group;
  const int first_m = group * GROUP;
  const int gsize = min(g.tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int bz = blockIdx.z;
  const bf16_t* A = (const bf16_t*)g.A + bz * g.a_bs;
  const bf16_t* B = (const bf16_t*)g.B + bz * g.b_bs;
  const int64_t lda = AKC ? g.a_rs : g.a_cs;
  const int64_t ldb = BKC ? g.b_cs : g.b_rs;

  // stage layout: [A0][B0][A1][B1]
#define As_(buf) (smem + (buf) * 2 * TILE_BYTES)
#define Bs_(buf) (smem + (buf) * 2 * TILE_BYTES + TILE_BYTES)

  f4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (int)((g.K + BK - 1) / BK);
  uint4 ra[4], rb[4];
  stage_load<AKC>(ra, A, lda, m0, 0, g.M, g.K, g.a_vec, tid);
  stage_load<BKC>(rb, B, ldb, n0, 0, g.N, g.K, g.b_vec, tid);
  stage_store<AKC>(ra, As_(0), tid);
  stage_store<BKC>(rb, Bs_(0), tid);
  __syncthreads();

  for (int t = 0; t < nk; t++) {
    const int cur = t & 1;
    if (t + 1 < nk) {
      stage_load<AKC>(ra, A, lda, m0, (int64_t)(t + 1) * BK, g.M, g.K, g.a_vec, tid);
      stage_load<BKC>(rb, B, ldb, n0, (int64_t)(t + 1) * BK, g.N, g.K, g.b_vec, tid);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
      bf8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = load_frag<AKC>(As_(cur), wr * 64 + i * 16, s, lane);
#pragma unroll
      for (int j = 0; j < 4; j++) fb[j] = load_frag<BKC>(Bs_(cur), wc * 64 + j * 16, s, lane);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16<E>(fa[i], fb[j], acc[i][j]);
    }
    if (t + 1 < nk) {
      stage_store<AKC>(ra, As_(cur ^ 1), tid);
      stage_store<BKC>(rb, Bs_(cur ^ 1), tid);
    }
    __syncthreads();
  }

  // epilogue: C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  E* C = (E*)g.C + bz * g.c_bs;
  const E* S = g.S ? (const E*)g.S + bz * g.s_bs : nullptr;
  const float alpha = (float)g.alpha, beta = (float)g.beta;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t col = n0 + wc * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t row = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row < g.M && col < g.N) {
          float v = alpha * acc[i][j][r];
          if (g.round_first) v = (float)E(v);
          if (S) v += beta * (float)S[row * g.s_rs + col * g.s_cs];
          C[row * g.ldc + col] = E(v);
        }
      }
    }
  }
}

// ================================================================================================
// bf16 large-tile kernel for aligned shapes (M % 256 == 0, N % 128 == 0, K % 64 == 0):
// 256 x 128 x 64 tile, 512 threads = 8 waves (4 x 2), 64 x 64 per wave.
//   * operand tiles arrive by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a ring of THREE
//     stages, two K-steps ahead of their use; the LDS images are the same swizzled images as above, with
//     the swizzle applied to the per-lane SOURCE address (the DMA destination is lane-linear);
//   * a K-step is a READ phase (all 16 fragments of the step + the DMA of step t+2) and an MFMA phase
//     (32 MFMAs), each closed by a raw s_barrier; waves 4-7 (the second wave of every SIMD) run one phase
//     behind waves 0-3, so the matrix core and the LDS pipe of a SIMD are busy at the same time;
//   * DMA ordering: each wave retires its own DMA(t+1) with a counted vmcnt before the barrier closing
//     READ(t) (DMA(t+2) stays in flight); DMA(t+2) reuses the slot last read in READ(t-1), whose reads
//     were retired (lgkmcnt(0)) before the barrier closing that phase;
//   * the B fragment is the MFMA's first operand: D rows = n, D cols = m, so a lane's four accumulator
//     registers are four CONSECUTIVE columns of C and the epilogue stores 8 bytes per lane.
// ================================================================================================
constexpr int PM = 256, PN = 128, PK = 64;
constexpr int P_A_BYTES = PM * PK * 2, P_B_BYTES = PN * PK * 2, P_SLOT = P_A_BYTES + P_B_BYTES;   // 32 + 16 KiB

__device__ __forceinline__ int ks_off_p(int k, int col8, int pitch) { return k * pitch + ((((col8 >> 1) ^ ks_swz(k))) << 5) + ((col8 & 1) << 4); }

typedef __attribute__((address_space(3))) char pp_lds_t;
typedef const __attribute__((address_space(1))) char pp_glb_t;

// one operand tile of one K-step: ROWS x 64, global -> LDS by DMA; 1 KiB pieces, ROWS / 64 pieces per wave
template <bool KC, int ROWS>
__device__ __forceinline__ void pp_stage_dma(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t k0, char* img, int wid, int lane) {
  constexpr int PPW = ROWS * PK * 2 / 1024 / 8;
#pragma unroll
  for (int i = 0; i < PPW; i++) {
    const int piece = wid * PPW + i;
    const int p = piece * 64 + lane;                       // 16-byte position inside the image
    const bf16_t* src;
    if (KC) {                                              // image [ROWS][8 chunks], chunk' = chunk ^ (row & 7)
      const int row = p >> 3, chunk = (p & 7) ^ (row & 7);
      src = base + (row0 + row) * ld + k0 + chunk * 8;
    } else {                                               // image [64 k][ROWS / 8 chunks], 32-byte pairs XOR ks_swz(k)
      constexpr int CPR = ROWS / 8;
      const int k = p / CPR, c = p % CPR;
      const int col8 = ((((c >> 1) ^ ks_swz(k))) << 1) | (c & 1);
      src = base + (k0 + k) * ld + row0 + col8 * 8;
    }
    __builtin_amdgcn_global_load_lds((pp_glb_t*)src, (pp_lds_t*)(img + piece * 1024), 16, 0, 0);
  }
}
template <bool KC, int ROWS> __device__ __forceinline__ bf8_t pp_frag(const char* img, int tile_row0, int s, int lane) {
  if (KC) {
    s8_t v = *reinterpret_cast<const s8_t*>(img + kc_off(tile_row0 + (lane & 15), s * 4 + (lane >> 4)));
    return __builtin_bit_cast(bf8_t, v);
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int k = s * 32 + g * 8 + q;
    const int col8 = (tile_row0 >> 3) + (p >> 1);
    // Inline asm, not __builtin_amdgcn_ds_read_tr16_b64: hipcc cannot tell that the builtin's read does not alias the LDS-DMA
    // in flight and drains it with s_waitcnt vmcnt(0) in front of the first transposing read of every phase, which serialises the
    // weight stream with the compute (measured: 1.07 vs 1.36 PFLOP/s against the all-ds_read_b128 layout).  The results are
    // made visible to the compiler's users by pp_frag_fence() below.
    s4_t lo, hi;
    const unsigned a0 = (unsigned)(uintptr_t)(img + ks_off_p(k, col8, ROWS * 2) + ((p & 1) << 3));
    const unsigned a1 = (unsigned)(uintptr_t)(img + ks_off_p(k + 4, col8, ROWS * 2) + ((p & 1) << 3));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8_t, v);
  }
}

// s_waitcnt lgkmcnt(0) (optionally with a vmcnt) that "produces" the fragments: consumers of asm-issued LDS reads cannot be
// scheduled above it
#define PP_FENCE4(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : : "memory")
#define PP_FENCE8(WAIT, F) \
  asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]) : : "memory")

template <bool AKC, bool BKC>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int GROUP = 4;                                      // column strips of 4 tile rows (1024 rows of A) per L2
  const int per_group = GROUP * g.tiles_n;
  const int group = bid / per_group;
  const int first_m = group * GROUP;
  const int gsize = min(g.tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: guards raw s_barriers
  const int wr = wid >> 1, wc = wid & 1;                      // 4 x 2 waves, 64 x 64 each
  const int64_t m0 = (int64_t)tm * PM, n0 = (int64_t)tn * PN;
  const int bz = blockIdx.z;
  const bf16_t* A = (const bf16_t*)g.A + bz * g.a_bs;
  const bf16_t* B = (const bf16_t*)g.B + bz * g.b_bs;
  const int64_t lda = AKC ? g.a_rs : g.a_cs;
  const int64_t ldb = BKC ? g.b_cs : g.b_rs;

  f4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (int)(g.K / PK);
  auto dma = [&](int t, int slot) {
    char* sb = smem + slot * P_SLOT;
    pp_stage_dma<AKC, PM>(A, lda, m0, (int64_t)t * PK, sb, wid, lane);
    pp_stage_dma<BKC, PN>(B, ldb, n0, (int64_t)t * PK, sb + P_A_BYTES, wid, lane);
  };
  dma(0, 0);
  if (nk > 1) dma(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int grp = wid >> 2;                                  // waves w and w + 4 share a SIMD
  if (grp == 1) __builtin_amdgcn_s_barrier();
  bf8_t fa0[4], fa1[4], fb0[4], fb1[4];
  int slot = 0;
  for (int t = 0; t < nk; t++) {
    const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
    const char* as = smem + slot * P_SLOT;
    const char* bs = as + P_A_BYTES;
    // ---- READ(t)
    if (t + 2 < nk) dma(t + 2, slot2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      fa0[i] = pp_frag<AKC, PM>(as, wr * 64 + i * 16, 0, lane);
      fa1[i] = pp_frag<AKC, PM>(as, wr * 64 + i * 16, 1, lane);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      fb0[j] = pp_frag<BKC, PN>(bs, wc * 64 + j * 16, 0, lane);
      fb1[j] = pp_frag<BKC, PN>(bs, wc * 64 + j * 16, 1, lane);
    }
    if (t + 2 < nk) PP_FENCE4("s_waitcnt vmcnt(6) lgkmcnt(0)", fa0);
    else PP_FENCE4("s_waitcnt vmcnt(0) lgkmcnt(0)", fa0);
    PP_FENCE4("", fa1); PP_FENCE4("", fb0); PP_FENCE4("", fb1);
    __builtin_amdgcn_s_barrier();
    // ---- MFMA(t)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    slot = slot1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();

  // epilogue: D rows = n (four consecutive per lane), D cols = m (lane & 15)
  bf16_t* C = (bf16_t*)g.C + bz * g.c_bs;
  const bf16_t* S = g.S ? (const bf16_t*)g.S + bz * g.s_bs : nullptr;
  const float alpha = (float)g.alpha, beta = (float)g.beta;
  const bool s_vec = S && g.s_cs == 1 && (g.s_rs % 4 == 0) && (((uintptr_t)S & 7) == 0);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t row = m0 + wr * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t col = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) { v[r] = alpha * acc[i][j][r]; if (g.round_first) v[r] = (float)bf16_t(v[r]); }
      if (S) {
        if (s_vec) {                                       // beta operand contiguous along n: one 8-byte load
          const uint2 sv = *reinterpret_cast<const uint2*>(S + row * g.s_rs + col);
          bf16_t e0, e1, e2, e3;
          e0.bits = (unsigned short)(sv.x & 0xffffu); e1.bits = (unsigned short)(sv.x >> 16);
          e2.bits = (unsigned short)(sv.y & 0xffffu); e3.bits = (unsigned short)(sv.y >> 16);
          v[0] += beta * (float)e0; v[1] += beta * (float)e1; v[2] += beta * (float)e2; v[3] += beta * (float)e3;
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] += beta * (float)S[row * g.s_rs + (col + r) * g.s_cs];
        }
      }
      const bf16_t o0(v[0]), o1(v[1]), o2(v[2]), o3(v[3]);
      uint2 pk;
      pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
      pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
      *reinterpret_cast<uint2*>(C + row * g.ldc + col) = pk;
    }
  }
}

// ================================================================================================
// bf16 256 x 256 x 64 kernel (M % 256 == 0, N % 256 == 0, K % 64 == 0, >= 200 tiles): 8 waves (2 x 4), 128 x 64 per wave.
// At 4096^3 the 256 x 128 kernel above moves ~9 TB/s from L2 into the CUs and stops there; the square 256 tile halves the
// bytes per FLOP (128 FLOP/B).  LDS holds TWO 64 KiB stages, so the DMA of stage t+1 is issued one stage ahead, by every
// wave in its first READ phase of stage t, and retired (vmcnt(0)) before the barrier that closes the LAST phase of stage t.
// A stage is four phases, READ(k-step 0) / MFMA / READ(k-step 1) / MFMA, each closed by a raw s_barrier, with waves 4-7
// running one phase behind waves 0-3 (ping-pong on every SIMD).  In units of phases, stage t occupies 4t..4t+3 for group 0
// and 4t+1..4t+4 for group 1:
//   * slot (t+1)&1 held stage t-1, last read by group 1 in phase 4t-1; group 0 issues DMA(t+1) in phase 4t, group 1 in 4t+1;
//   * both groups retire their DMA(t+1) before the barrier closing phase 4t+3 (group 0: end of its second MFMA phase,
//     group 1: end of its second READ phase); the first read of stage t+1 is in phase 4t+4.
// ================================================================================================
constexpr int QM = 256, QN = 256;
constexpr int Q_IMG = QM * PK * 2, Q_SLOT = 2 * Q_IMG;     // 32 KiB per operand image, 64 KiB per stage

// -DGEMM_STAMP (diagnostic builds only, `make EXTRA=-DGEMM_STAMP`; scripts/gemm_clock_probe.py): thread 0 of every workgroup stamps the
// shader clock (s_memtime) AND the constant 100 MHz clock (s_memrealtime) at the kernel's start, around the main loop and at its end,
// into a buffer nothing else reads: in-kernel clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
#ifdef GEMM_STAMP
__device__ unsigned long long gemm_stamps[512 * 8];
#define GQ_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 512 && blockIdx.z == 0) {                                  \
    gemm_stamps[blockIdx.x * 8 + 2 * (k)] = __builtin_amdgcn_s_memtime();                                                \
    gemm_stamps[blockIdx.x * 8 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int lamp_debug_gemm_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm_stamps), sizeof(gemm_stamps)) == hipSuccess ? 0 : 1; }
#else
#define GQ_STAMP(k) do { } while (0)
#endif

template <bool AKC, bool BKC>
__global__ __launch_bounds__(512) void gemm_bf16_pp2_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GQ_STAMP(0);
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int GROUP = 4;
  const int per_group = GROUP * g.tiles_n;
  const int group = bid / per_group;
  const int first_m = group * GROUP;
  const int gsize = min(g.tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;                      // 2 x 4 waves, 128 x 64 each; wr is also the ping-pong group
  const int64_t m0 = (int64_t)tm * QM, n0 = (int64_t)tn * QN;
  const int bz = blockIdx.z;
  const bf16_t* A = (const bf16_t*)g.A + bz * g.a_bs;
  const bf16_t* B = (const bf16_t*)g.B + bz * g.b_bs;
  const int64_t lda = AKC ? g.a_rs : g.a_cs;
  const int64_t ldb = BKC ? g.b_cs : g.b_rs;

  f4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (int)(g.K / PK);
  auto dma = [&](int t, int slot) {
    char* sb = smem + slot * Q_SLOT;
    pp_stage_dma<AKC, QM>(A, lda, m0, (int64_t)t * PK, sb, wid, lane);
    pp_stage_dma<BKC, QN>(B, ldb, n0, (int64_t)t * PK, sb + Q_IMG, wid, lane);
  };
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  GQ_STAMP(1);
#ifdef GEMM_STAMP
  __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) alone: the stamp's scalar loads must not colour the loop's first LDS waits
#endif
  if (wr == 1) __builtin_amdgcn_s_barrier();

  bf8_t fa[8], fb[4];
#define Q_READ(S_)                                                                                     \
  do {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; i++) fa[i] = pp_frag<AKC, QM>(as, wr * 128 + i * 16, S_, lane); \
    _Pragma("unroll") for (int j = 0; j < 4; j++) fb[j] = pp_frag<BKC, QN>(bs, wc * 64 + j * 16, S_, lane);  \
  } while (0)
#define Q_MFMA()                                                                                       \
  do {                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; i++)                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                                  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);      \
    __builtin_amdgcn_s_setprio(0);                                                                     \
  } while (0)
  for (int t = 0; t < nk; t++) {
    const char* as = smem + (t & 1) * Q_SLOT;
    const char* bs = as + Q_IMG;
    // READ k-step 0 (+ DMA of the next stage)
    if (t + 1 < nk) dma(t + 1, (t + 1) & 1);
    Q_READ(0);
    PP_FENCE8("s_waitcnt lgkmcnt(0)", fa); PP_FENCE4("", fb);
    __builtin_amdgcn_s_barrier();
    Q_MFMA();
    __builtin_amdgcn_s_barrier();
    // READ k-step 1
    Q_READ(1);
    if (wr == 1) PP_FENCE8("s_waitcnt vmcnt(0) lgkmcnt(0)", fa);
    else PP_FENCE8("s_waitcnt lgkmcnt(0)", fa);
    PP_FENCE4("", fb);
    __builtin_amdgcn_s_barrier();
    Q_MFMA();
    if (wr == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef Q_READ
#undef Q_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();
  GQ_STAMP(2);

  if (g.split_f32) {
    float* W = (float*)g.C + bz * g.c_bs;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int64_t row = m0 + wr * 128 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int64_t col = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
        *reinterpret_cast<f4_t*>(W + row * g.ldc + col) = acc[i][j];
      }
    }
    return;
  }
  bf16_t* C = (bf16_t*)g.C + bz * g.c_bs;
  const bf16_t* S = g.S ? (const bf16_t*)g.S + bz * g.s_bs : nullptr;
  const float alpha = (float)g.alpha, beta = (float)g.beta;
  const bool s_vec = S && g.s_cs == 1 && (g.s_rs % 4 == 0) && (((uintptr_t)S & 7) == 0);
  // Round 3: the wave's 128 x 64 output tile goes through its own 16 KiB of the (now free) LDS as [row][64 columns] rows and leaves as
  // 16-byte stores, eight lanes per 128-byte row - full lines, half the store instructions.  In-kernel stamps (scripts/gemm_clock_probe.py)
  // had the 8-byte-per-lane epilogue at 9.8 of the kernel's 107 us with nothing to overlap it (one workgroup per CU).  8-byte slot s of
  // row r sits at s ^ ((r & 7) << 1): the 16 rows of a write spread over the banks and 16-byte chunks stay whole.
  const bool c16 = (g.ldc % 8 == 0) && (((uintptr_t)C & 15) == 0);
  char* El = smem + wid * 16384;
  const int q = lane >> 4;
  // the beta operand (dW += ..., dX += ...: S is C itself) comes in the same way: sixteen 16-byte loads per lane in flight together, parked
  // in the tile's LDS image, each lane then picks up its 8 bytes where it will put its result
  const bool s16 = c16 && s_vec && (g.s_rs % 8 == 0) && (((uintptr_t)S & 15) == 0);
  if (s16) {
    const bf16_t* Sw = S + (m0 + wr * 128) * g.s_rs + n0 + wc * 64;
    uint4 sv[16];
#pragma unroll
    for (int it = 0; it < 16; it++) { const int idx = it * 64 + lane; sv[it] = *reinterpret_cast<const uint4*>(Sw + (int64_t)(idx >> 3) * g.s_rs + (idx & 7) * 8); }
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int idx = it * 64 + lane;
      const int lrow = idx >> 3, c = idx & 7;
      *reinterpret_cast<uint4*>(El + lrow * 128 + ((c ^ (lrow & 7)) << 4)) = sv[it];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int64_t row = m0 + wr * 128 + i * 16 + (lane & 15);
    const int lrow = i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t col = n0 + wc * 64 + j * 16 + q * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) { v[r] = alpha * acc[i][j][r]; if (g.round_first) v[r] = (float)bf16_t(v[r]); }
      if (S) {
        if (s_vec) {                                       // beta operand contiguous along n: 8 bytes per lane (from LDS or from memory)
          const uint2 sv = s16 ? *reinterpret_cast<const uint2*>(El + lrow * 128 + (((j * 4 + q) ^ ((lrow & 7) << 1)) << 3))
                               : *reinterpret_cast<const uint2*>(S + row * g.s_rs + col);
          bf16_t e0, e1, e2, e3;
          e0.bits = (unsigned short)(sv.x & 0xffffu); e1.bits = (unsigned short)(sv.x >> 16);
          e2.bits = (unsigned short)(sv.y & 0xffffu); e3.bits = (unsigned short)(sv.y >> 16);
          v[0] += beta * (float)e0; v[1] += beta * (float)e1; v[2] += beta * (float)e2; v[3] += beta * (float)e3;
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] += beta * (float)S[row * g.s_rs + (col + r) * g.s_cs];
        }
      }
      const bf16_t o0(v[0]), o1(v[1]), o2(v[2]), o3(v[3]);
      uint2 pk;
      pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
      pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
      if (c16) *reinterpret_cast<uint2*>(El + lrow * 128 + (((j * 4 + q) ^ ((lrow & 7) << 1)) << 3)) = pk;
      else *reinterpret_cast<uint2*>(C + row * g.ldc + col) = pk;
    }
  }
  if (c16) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the region is private to this wave
    bf16_t* Cw = C + (m0 + wr * 128) * g.ldc + n0 + wc * 64;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int idx = it * 64 + lane;
      const int lrow = idx >> 3, c = idx & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(El + lrow * 128 + ((c ^ (lrow & 7)) << 4));
      *reinterpret_cast<uint4*>(Cw + (int64_t)lrow * g.ldc + c * 8) = v;
    }
  }
  GQ_STAMP(3);
}

// split-K epilogue: C[m][n] = alpha * sum_s W[s][m][n] + beta * S[m][n], four columns per thread (N % 4 == 0), slices summed in order
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ W, int split, int64_t M, int64_t N, bf16_t* __restrict__ C,
                                                                 int64_t ldc, const bf16_t* __restrict__ S, int64_t s_rs, int64_t s_cs, float alpha, float beta, int round_first) {
  const int64_t total = M * N / 4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    f4_t a = *reinterpret_cast<const f4_t*>(W + e * 4);
    for (int s = 1; s < split; s++) a += *reinterpret_cast<const f4_t*>(W + (int64_t)s * M * N + e * 4);
    const int64_t row = (e * 4) / N, col = (e * 4) % N;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      v[r] = alpha * a[r];
      if (round_first) v[r] = (float)bf16_t(v[r]);
      if (S) v[r] += beta * (float)S[row * s_rs + (col + r) * s_cs];
    }
    const bf16_t o0(v[0]), o1(v[1]), o2(v[2]), o3(v[3]);
    bf16_t* c = C + row * ldc + col;
    c[0] = o0; c[1] = o1; c[2] = o2; c[3] = o3;
  }
}

// split-K epilogue of the f32 / f64 kernel: C[m][n] = alpha * sum_z W[z][m][n] + beta * S[m][n], slices summed in order (deterministic)
template <class T>
__global__ __launch_bounds__(256) void gemm_fp_splitk_reduce_kernel(const T* __restrict__ W, int split, int64_t M, int64_t N, T* __restrict__ C, int64_t ldc,
                                                                    const T* __restrict__ S, int64_t s_rs, int64_t s_cs, T alpha, T beta) {
  const int64_t total = M * N;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    T a = W[e];
    for (int z = 1; z < split; z++) a += W[(int64_t)z * total + e];
    const int64_t row = e / N, col = e - row * N;
    T v = alpha * a;
    if (S) v += beta * S[row * s_rs + col * s_cs];
    C[row * ldc + col] = v;
  }
}

// ================================================================================================
// f32 / f64 kernel: 64 x 64 x 16 tile, 256 threads = 4 waves (2 x 2), 32 x 32 per wave,
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64. LDS tiles are k-major [16][64+pad] so a
// fragment read is one conflict-free 4/8-byte read per lane.
// ================================================================================================
template <class T> struct FpTraits;
template <> struct FpTraits<float> {
  using acc4 = f4_t;
  static __device__ __forceinline__ acc4 mfma(float a, float b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) * 4 + r; }
};
template <> struct FpTraits<double> {
  using acc4 = d4_t;
  static __device__ __forceinline__ acc4 mfma(double a, double b, acc4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  // f64 C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};

constexpr int FM = 64, FN = 64, FK = 16, FPAD = 4;

template <class T>
__global__ __launch_bounds__(256) void gemm_fp_kernel(GemmArgs g) {
  using TR = FpTraits<T>;
  using acc4 = typename TR::acc4;
  __shared__ T As[2][FK][FM + FPAD];
  __shared__ T Bs[2][FK][FN + FPAD];
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int64_t m0 = (int64_t)tm * FM, n0 = (int64_t)tn * FN;
  const int bz = blockIdx.z;
  const T* A = (const T*)g.A + bz * g.a_bs;
  const T* B = (const T*)g.B + bz * g.b_bs;
  const int64_t K = g.k_split_total > 0 ? min(g.K, g.k_split_total - (int64_t)bz * g.K) : g.K;

  acc4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = acc4{0, 0, 0, 0};

  // staging: 64 x 16 = 1024 elements per operand, 4 per thread; thread -> (row, k) chosen so the
  // global reads are coalesced along whichever index is contiguous
  const bool a_kc = (g.a_cs == 1), b_kc = (g.b_rs == 1);
  T ra[4], rb[4];
  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int e = tid + i * 256;
      int ar, ak, bn, bk;
      if (a_kc) { ar = e >> 4; ak = e & 15; } else { ak = e >> 6; ar = e & 63; }
      if (b_kc) { bn = e >> 4; bk = e & 15; } else { bk = e >> 6; bn = e & 63; }
      const int64_t gm = m0 + ar, gka = k0 + ak, gn = n0 + bn, gkb = k0 + bk;
      ra[i] = (gm < g.M && gka < K) ? A[gm * g.a_rs + gka * g.a_cs] : T(0);
      rb[i] = (gn < g.N && gkb < K) ? B[gkb * g.b_rs + gn * g.b_cs] : T(0);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int e = tid + i * 256;
      int ar, ak, bn, bk;
      if (a_kc) { ar = e >> 4; ak = e & 15; } else { ak = e >> 6; ar = e & 63; }
      if (b_kc) { bn = e >> 4; bk = e & 15; } else { bk = e >> 6; bn = e & 63; }
      As[buf][ak][ar] = ra[i];
      Bs[buf][bk][bn] = rb[i];
    }
  };

  const int nk = (int)((K + FK - 1) / FK);
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nk; t++) {
    const int cur = t & 1;
    if (t + 1 < nk) load_tile((int64_t)(t + 1) * FK);
#pragma unroll
    for (int s = 0; s < FK / 4; s++) {
      const int k = s * 4 + (lane >> 4);
      T fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) fa[i] = As[cur][k][wr * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; j++) fb[j] = Bs[cur][k][wc * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = TR::mfma(fa[i], fb[j], acc[i][j]);
    }
    if (t + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
  T* C = (T*)g.C + bz * g.c_bs;
  const T* S = g.S ? (const T*)g.S + bz * g.s_bs : nullptr;
  const T alpha = (T)g.alpha, beta = (T)g.beta;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int64_t col = n0 + wc * 32 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t row = m0 + wr * 32 + i * 16 + TR::crow(lane, r);
        if (row < g.M && col < g.N) {
          T v = alpha * acc[i][j][r];
          if (S) v += beta * S[row * g.s_rs + col * g.s_cs];
          if (g.knn_q) {       // the operation order of the separate distance kernel: (qn + dn) - 2 * outer, clamped at 0
            const T o2 = v * T(2);
            const T sn = ((const T*)g.knn_q)[row] + ((const T*)g.knn_d)[col];
            const T dd = sn - o2;
            v = dd > T(0) ? dd : T(0);
          }
          C[row * g.ldc + col] = v;
        }
      }
    }
}

// ================================================================================================
// f32 kernel for larger problems: 128 x 128 x 16 tile, 4 waves (2 x 2, 64 x 64 each), 16-byte global loads along whichever
// index of an operand is contiguous, k-major LDS tiles [16][128 + 16] (the +16 puts the four k rows a fragment read touches on
// disjoint bank ranges), v_mfma_f32_16x16x4_f32: the same exact f32 FMA chain per output as the 64 x 64 kernel, twice the
// FLOP per byte staged.  Needs 16-byte aligned operands with leading dimensions that are multiples of 4.
// ================================================================================================
constexpr int GM = 128, GN = 128, GK = 16, GPAD = 16;

__global__ __launch_bounds__(256) void gemm_f32_big_kernel(GemmArgs g) {
  __shared__ float As[2][GK][GM + GPAD];
  __shared__ float Bs[2][GK][GN + GPAD];
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int64_t m0 = (int64_t)tm * GM, n0 = (int64_t)tn * GN;
  const int bz = blockIdx.z;
  const float* A = (const float*)g.A + bz * g.a_bs;
  const float* B = (const float*)g.B + bz * g.b_bs;
  const bool a_kc = (g.a_cs == 1), b_kc = (g.b_rs == 1);
  const int64_t lda = a_kc ? g.a_rs : g.a_cs, ldb = b_kc ? g.b_cs : g.b_rs;

  f4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  // staging: 128 x 16 floats per operand = 512 packets of 4, two per thread
  float4 ra[2], rb[2];
  auto load_op = [&](float4 (&r)[2], const float* P, bool kc, int64_t ld, int64_t r0, int64_t rows, int64_t k0) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int c = tid + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kc) {                                   // [rows][K]: packet = 4 consecutive k of one row
        const int64_t row = r0 + (c >> 2), k = k0 + ((c & 3) << 2);
        if (row < rows) {
          if (k + 4 <= g.K) v = *reinterpret_cast<const float4*>(P + row * ld + k);
          else { float e[4] = {0, 0, 0, 0}; for (int j = 0; j < 4; j++) if (k + j < g.K) e[j] = P[row * ld + k + j]; v = make_float4(e[0], e[1], e[2], e[3]); }
        }
      } else {                                    // [K][rows]: packet = 4 consecutive rows of one k
        const int64_t k = k0 + (c >> 5), row = r0 + ((c & 31) << 2);
        if (k < g.K) {
          if (row + 4 <= rows) v = *reinterpret_cast<const float4*>(P + k * ld + row);
          else { float e[4] = {0, 0, 0, 0}; for (int j = 0; j < 4; j++) if (row + j < rows) e[j] = P[k * ld + row + j]; v = make_float4(e[0], e[1], e[2], e[3]); }
        }
      }
      r[i] = v;
    }
  };
  auto store_op = [&](const float4 (&r)[2], float (*T)[GM + GPAD], bool kc) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int c = tid + i * 256;
      if (kc) {
        const int row = c >> 2, k = (c & 3) << 2;
        T[k][row] = r[i].x; T[k + 1][row] = r[i].y; T[k + 2][row] = r[i].z; T[k + 3][row] = r[i].w;
      } else {
        const int k = c >> 5, row = (c & 31) << 2;
        *reinterpret_cast<float4*>(&T[k][row]) = r[i];
      }
    }
  };

  const int nk = (int)((g.K + GK - 1) / GK);
  load_op(ra, A, a_kc, lda, m0, g.M, 0);
  load_op(rb, B, b_kc, ldb, n0, g.N, 0);
  store_op(ra, As[0], a_kc);
  store_op(rb, Bs[0], b_kc);
  __syncthreads();
  for (int t = 0; t < nk; t++) {
    const int cur = t & 1;
    if (t + 1 < nk) {
      load_op(ra, A, a_kc, lda, m0, g.M, (int64_t)(t + 1) * GK);
      load_op(rb, B, b_kc, ldb, n0, g.N, (int64_t)(t + 1) * GK);
    }
#pragma unroll
    for (int s4 = 0; s4 < GK / 4; s4++) {
      const int k = s4 * 4 + (lane >> 4);
      float fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = As[cur][k][wr * 64 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; j++) fb[j] = Bs[cur][k][wc * 64 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) {
      store_op(ra, As[cur ^ 1], a_kc);
      store_op(rb, Bs[cur ^ 1], b_kc);
    }
    __syncthreads();
  }
  float* C = (float*)g.C + bz * g.c_bs;
  const float* S = g.S ? (const float*)g.S + bz * g.s_bs : nullptr;
  const float alpha = (float)g.alpha, beta = (float)g.beta;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t col = n0 + wc * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t row = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row < g.M && col < g.N) {
          float v = alpha * acc[i][j][r];
          if (S) v += beta * S[row * g.s_rs + col * g.s_cs];
          if (g.knn_q) {
            const float o2 = v * 2.f;
            const float sn = ((const float*)g.knn_q)[row] + ((const float*)g.knn_d)[col];
            const float dd = sn - o2;
            v = dd > 0.f ? dd : 0.f;
          }
          C[row * g.ldc + col] = v;
        }
      }
    }
}

// ================================================================================================
// host side
// ================================================================================================
struct Operand {
  const Tensor* t;
  Hold owned;  // contiguous copy when the strides are not GEMM-friendly
  int64_t rs, cs, bs;
};

// a 2-D (or batched 3-D) operand whose (row, col) strides have a unit stride somewhere
static void prep_operand(Operand& o, const Tensor* t, bool batched) {
  int nd = t->ndim;
  o.t = t;
  int64_t rs = t->strides[nd - 2], cs = t->strides[nd - 1];
  int64_t rows = t->sizes[nd - 2], cols = t->sizes[nd - 1];
  bool ok = (cs == 1 && (rs >= cols || rows == 1)) || (rs == 1 && (cs >= rows || cols == 1));
  if (rows == 1 && cols == 1) ok = true;
  if (!ok) {
    o.owned = Hold(new_like(t));
    copy_into(o.owned.get(), t);
    o.t = o.owned.get();
    rs = o.t->strides[nd - 2];
    cs = o.t->strides[nd - 1];
  }
  // normalise degenerate strides so exactly one of them is 1
  if (cols == 1 && rs == 1 && cs != 1 && rows > 1) { /* column vector stored densely: M-contiguous */ }
  o.rs = rs;
  o.cs = cs;
  o.bs = batched ? o.t->strides[0] : 0;
}

// transA: use a^T, transB: use b^T.  self may be null (beta ignored) and may alias out.
static void gemm_dispatch(Tensor* out, const Tensor* self, const Tensor* a, const Tensor* b, bool transA, bool transB,
                          double beta, double alpha, bool batched, const Tensor* knn_q = nullptr, const Tensor* knn_d = nullptr, bool round_first = false) {
  check_device_tensor(out, "out"); check_device_tensor(a, "mat1"); check_device_tensor(b, "mat2");
  const int nd = batched ? 3 : 2;
  LAMP_CHECK(a->ndim == nd && b->ndim == nd && out->ndim == nd, "expected " << nd << "-D operands, got " << a->describe() << ", "
             << b->describe() << " -> " << out->describe());
  LAMP_CHECK(a->dtype == b->dtype && a->dtype == out->dtype, "dtype mismatch: " << a->describe() << " x " << b->describe()
             << " -> " << out->describe());
  check_same_device(a, b); check_same_device(a, out);
  Operand oa, ob;
  prep_operand(oa, a, batched);
  prep_operand(ob, b, batched);
  const int64_t ar = oa.t->sizes[nd - 2], ac = oa.t->sizes[nd - 1], br = ob.t->sizes[nd - 2], bc = ob.t->sizes[nd - 1];
  GemmArgs g{};
  g.round_first = round_first ? 1 : 0;
  if (knn_q) {
    LAMP_CHECK(a->dtype == kF32 || a->dtype == kF64, "the fused kNN epilogue exists for f32 / f64");
    g.knn_q = knn_q->data(); g.knn_d = knn_d->data();
  }
  g.M = transA ? ac : ar;
  g.K = transA ? ar : ac;
  const int64_t Kb = transB ? bc : br;
  g.N = transB ? br : bc;
  LAMP_CHECK(g.K == Kb, "shapes cannot be multiplied: " << a->describe() << (transA ? "^T" : "") << " x " << b->describe() << (transB ? "^T" : ""));
  LAMP_CHECK(out->sizes[nd - 2] == g.M && out->sizes[nd - 1] == g.N, "out " << out->describe() << " has the wrong shape for " << g.M << "x" << g.N);
  LAMP_CHECK(out->strides[nd - 1] == 1 || g.N == 1, "out must be row-major");
  g.batch = batched ? (int)out->sizes[0] : 1;
  if (batched) LAMP_CHECK(a->sizes[0] == g.batch && b->sizes[0] == g.batch, "batch size mismatch");
  g.a_rs = transA ? oa.cs : oa.rs;
  g.a_cs = transA ? oa.rs : oa.cs;
  g.b_rs = transB ? ob.cs : ob.rs;
  g.b_cs = transB ? ob.rs : ob.cs;
  g.a_bs = oa.bs; g.b_bs = ob.bs;
  g.A = oa.t->data(); g.B = ob.t->data(); g.C = out->data();
  g.ldc = out->strides[nd - 2];
  g.c_bs = batched ? out->strides[0] : 0;
  g.alpha = alpha; g.beta = beta;
  if (self && beta != 0.0) {
    check_device_tensor(self, "self");
    LAMP_CHECK(self->dtype == out->dtype, "self dtype mismatch");
    // broadcast self to [batch, M, N]
    std::vector<int64_t> oshape = out->shape();
    const Tensor* ops[2] = {out, self};
    LAMP_CHECK(self->ndim <= nd, "self has too many dims");
    int lead = nd - self->ndim;
    int64_t st[3] = {0, 0, 0};
    for (int d = 0; d < nd; d++) {
      if (d < lead) continue;
      int64_t sz = self->sizes[d - lead];
      LAMP_CHECK(sz == oshape[d] || sz == 1, "self " << self->describe() << " does not broadcast to " << out->describe());
      st[d] = (sz == 1 && oshape[d] != 1) ? 0 : self->strides[d - lead];
    }
    g.S = self->data();
    g.s_rs = st[nd - 2]; g.s_cs = st[nd - 1]; g.s_bs = batched ? st[0] : 0;
  } else {
    g.S = nullptr;
  }
  if (g.M == 0 || g.N == 0 || g.batch == 0) return;
  hipStream_t stm = current_stream(out->device());
  if (g.K == 0) {  // out = beta * self
    if (g.S) { LAMP_CHECK(lamp_mul_scalar_(out, 0.0) == 0, lamp_last_error()); LAMP_CHECK(lamp_add_(out, self, beta) == 0, lamp_last_error()); }
    else fill_zero(out);
    return;
  }
  const double g_flops = 2.0 * (double)g.M * (double)g.N * (double)g.K * g.batch;
  const double g_bytes = ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N * (g.S ? 2 : 1)) * g.batch * (double)dtype_size(a->dtype);
  const char* kt_tag = a->dtype == kBF16 ? "gemm_bf16" : (a->dtype == kF16 ? "gemm_f16" : (a->dtype == kF32 ? "gemm_f32" : "gemm_f64"));
  static const bool shape_tags = getenv("LAMP_GEMM_SHAPE_TAGS") != nullptr;   // profiling aid: one timer class per shape and layout
  if (shape_tags) {
    static std::mutex mu;
    static std::map<std::string, std::unique_ptr<std::string>> interned;
    std::ostringstream os;
    os << kt_tag << "[" << g.batch << "x" << g.M << "x" << g.N << "x" << g.K << (g.a_cs == 1 ? ",Akc" : ",Amc") << (g.b_rs == 1 ? ",Bkc" : ",Bnc") << "]";
    std::lock_guard<std::mutex> lock(mu);
    auto& e = interned[os.str()];
    if (!e) e.reset(new std::string(os.str()));
    kt_tag = e->c_str();
  }
  KernelTimer kt(kt_tag, g_flops, g_bytes, stm);
  if (a->dtype == kBF16) {
    const int out_device = out->device();
    std::function<void(GemmArgs)> run = [&](GemmArgs g) {
    const bool akc = (g.a_cs == 1), bkc = (g.b_rs == 1);
    LAMP_CHECK(akc || g.a_rs == 1, "internal: A has no unit stride");
    LAMP_CHECK(bkc || g.b_cs == 1, "internal: B has no unit stride");
    const int64_t lda = akc ? g.a_rs : g.a_cs, ldb = bkc ? g.b_cs : g.b_rs;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    g.a_vec = (lda % 8 == 0) && al16(g.A) && (g.a_bs % 8 == 0);
    g.b_vec = (ldb % 8 == 0) && al16(g.B) && (g.b_bs % 8 == 0);
    // aligned, large problems: 256 x 128 LDS-DMA / ping-pong kernel
    const bool big = g.a_vec && g.b_vec && g.M % PM == 0 && g.N % PN == 0 && g.K % PK == 0 && (g.ldc % 4 == 0) &&
                     (((uintptr_t)g.C & 7) == 0) && (g.c_bs % 4 == 0) && (g.M / PM) * (g.N / PN) * g.batch >= 128;
    bool big2 = big && g.N % QN == 0 && (g.M / QM) * (g.N / QN) * g.batch >= 200;
    if (big && !big2 && g.batch == 1 && g.M % QM == 0 && g.N % QN == 0) {
      // fewer than 200 tiles of 256 x 256: still the better choice when the 256 x 128 kernel would need a second, mostly empty round
      // (rounds x k-steps x ~1.7 / ~1.2 us per 64-deep k-step; 3072 x 3072 x 768: 288 half tiles = 2 rounds, 144 full tiles = 1: 34 -> 25 us)
      const int64_t cusn = num_cus(), nk = g.K / PK;
      const double c2 = (double)(((g.M / QM) * (g.N / QN) + cusn - 1) / cusn) * (double)nk * 1.7;
      const double c1 = (double)(((g.M / PM) * (g.N / PN) + cusn - 1) / cusn) * (double)nk * 1.2;
      if (c2 < c1) big2 = true;
    }
    // few output tiles over a long K (the weight gradients x^T . p of a token batch): split K over blockIdx.z
    static const bool allow_split = !(getenv("LAMP_GEMM_SPLITK") && atoi(getenv("LAMP_GEMM_SPLITK")) == 0);
    if (allow_split && !big2 && g.batch == 1 && g.a_vec && g.b_vec && g.M % QM == 0 && g.N % QN == 0 && g.K % PK == 0 && g.K >= 512) {
      const int64_t tiles = (g.M / QM) * (g.N / QN), nk_total = g.K / PK;
      // one workgroup per CU.  Cost of a split d in us: rounds x k-steps per chunk x ~1.7 us (one 256 x 256 x 64 stage at the
      // kernel's ~4.9 TFLOP/s per CU) + writing and re-reading d f32 slices at ~4 TB/s
      const int64_t cus = num_cus();
      auto cost = [&](int64_t d) {
        const int64_t wgs = tiles * d;
        return (double)((wgs + cus - 1) / cus) * (double)(nk_total / d) * 1.7 + (d > 1 ? (double)d * (double)g.M * (double)g.N * 8.0 / 4.0e6 : 0.0);
      };
      int split = 1;
      double best = cost(1);
      for (int64_t d = 2; d <= nk_total / 4; d++) {
        if (nk_total % d) continue;
        const double c = cost(d);
        if (c < best * 0.97) { best = c; split = (int)d; }
      }
      // ... against the 128 x 128 kernel without a split (two workgroups per CU, ~1.1 us per 64-deep k-step, ~5 us fixed): with a short K
      // and enough 128 x 128 tiles it beats the split, whose second launch and f32 slices cost ~9 us beyond the model above
      // (3072 x 768 x 768, the attention projections of a 3072-token batch: 29.6 -> 18.8 us)
      const int64_t t128 = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
      const double c128 = (double)((t128 + 2 * cus - 1) / (2 * cus)) * (double)nk_total * 1.1 + 5.0;
      if (split > 1 && !big && c128 < best + 9.0) split = 1;
      if (split > 1 && tiles * split >= 64) {
        Hold ws(new_tensor({(int64_t)split, g.M, g.N}, kF32, out_device));
        GemmArgs h = g;
        h.K = g.K / split;
        h.a_bs = h.K * g.a_cs; h.b_bs = h.K * g.b_rs;
        h.C = ws->data(); h.ldc = g.N; h.c_bs = g.M * g.N; h.S = nullptr; h.split_f32 = 1;
        h.tiles_m = (int)(g.M / QM); h.tiles_n = (int)(g.N / QN);
        dim3 grid(h.tiles_m * h.tiles_n, 1, split);
        const size_t lds = 2 * Q_SLOT;
#define PP2S_LAUNCH(A_, B_)                                                                                                     \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    allow_big_lds((const void*)gemm_bf16_pp2_kernel<A_, B_>); \
    hipLaunchKernelGGL((gemm_bf16_pp2_kernel<A_, B_>), grid, dim3(512), lds, stm, h);                                           \
  } while (0)
        if (akc && !bkc) PP2S_LAUNCH(true, false);
        else if (akc && bkc) PP2S_LAUNCH(true, true);
        else if (!akc && !bkc) PP2S_LAUNCH(false, false);
        else PP2S_LAUNCH(false, true);
#undef PP2S_LAUNCH
        LAMP_LAUNCH_CHECK();
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(grid_for(g.M * g.N / 4, 256)), dim3(256), 0, stm, (const float*)ws->data(), split, g.M, g.N,
                           (bf16_t*)g.C, g.ldc, (const bf16_t*)g.S, g.s_rs, g.s_cs, (float)g.alpha, (float)g.beta, g.round_first);
        LAMP_LAUNCH_CHECK();
        return;
      }
    }
    if (big2) {
      g.tiles_m = (int)(g.M / QM);
      g.tiles_n = (int)(g.N / QN);
      dim3 grid(g.tiles_m * g.tiles_n, 1, g.batch);
      const size_t lds = 2 * Q_SLOT;
#define PP2_LAUNCH(A_, B_)                                                                                                      \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    allow_big_lds((const void*)gemm_bf16_pp2_kernel<A_, B_>); \
    hipLaunchKernelGGL((gemm_bf16_pp2_kernel<A_, B_>), grid, dim3(512), lds, stm, g);                                           \
  } while (0)
      if (akc && !bkc) PP2_LAUNCH(true, false);
      else if (akc && bkc) PP2_LAUNCH(true, true);
      else if (!akc && !bkc) PP2_LAUNCH(false, false);
      else PP2_LAUNCH(false, true);
#undef PP2_LAUNCH
      LAMP_LAUNCH_CHECK();
      return;
    }
    if (big) {
      g.tiles_m = (int)(g.M / PM);
      g.tiles_n = (int)(g.N / PN);
      dim3 grid(g.tiles_m * g.tiles_n, 1, g.batch);
      const size_t lds = 3 * P_SLOT;
#define PP_LAUNCH(A_, B_)                                                                                                       \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    allow_big_lds((const void*)gemm_bf16_pp_kernel<A_, B_>); \
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<A_, B_>), grid, dim3(512), lds, stm, g);                                            \
  } while (0)
      if (akc && !bkc) PP_LAUNCH(true, false);
      else if (akc && bkc) PP_LAUNCH(true, true);
      else if (!akc && !bkc) PP_LAUNCH(false, false);
      else PP_LAUNCH(false, true);
#undef PP_LAUNCH
      LAMP_LAUNCH_CHECK();
      return;
    }
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (int)((g.N + BN - 1) / BN);
    dim3 grid(g.tiles_m * g.tiles_n, 1, g.batch);
    size_t lds = 4 * TILE_BYTES;
    if (akc && !bkc) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, dim3(256), lds, stm, g);
    else if (akc && bkc) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, dim3(256), lds, stm, g);
    else if (!akc && !bkc) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, dim3(256), lds, stm, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, dim3(256), lds, stm, g);
    LAMP_LAUNCH_CHECK();
    };
    // The 256 x 256 kernel runs one workgroup per CU, so a tile count just above a multiple of the CU count costs a whole extra round
    // (288 tiles: 155 us, 255 tiles: 101 us at K = 3072).  When the last round would be less than a quarter full, the rows of that
    // remainder become a second product, which the split-K path spreads over the idle CUs.
    static const bool tail_split = !(getenv("LAMP_GEMM_TAIL_SPLIT") && atoi(getenv("LAMP_GEMM_TAIL_SPLIT")) == 0);
    const int64_t cus = num_cus();
    if (tail_split && g.batch == 1 && !g.knn_q && g.M % QM == 0 && g.N % QN == 0 && g.K % PK == 0 && g.K >= 2048) {   // shorter K: the second launch costs more than the round
      const int64_t tn = g.N / QN, t2 = (g.M / QM) * tn, rem = t2 % cus;
      if (t2 > cus && rem > 0 && rem * 4 <= cus) {
        const int64_t rem_rows = (rem + tn - 1) / tn, M2 = rem_rows * QM, M1 = g.M - M2;
        if (M1 > 0 && (M1 / QM) * tn >= 200) {
          GemmArgs g1 = g, g2 = g;
          g1.M = M1;
          g2.M = M2;
          g2.A = (const char*)g.A + M1 * g.a_rs * (int64_t)sizeof(bf16_t);
          g2.C = (char*)g.C + M1 * g.ldc * (int64_t)sizeof(bf16_t);
          if (g.S) g2.S = (const char*)g.S + M1 * g.s_rs * (int64_t)sizeof(bf16_t);
          run(g1);
          run(g2);
          return;
        }
      }
    }
    run(g);
    return;
  } else if (a->dtype == kF16) {
    // half precision: the 128 x 128 x 64 register-staged kernel (every shape, bounds checked) on v_mfma_f32_16x16x32_f16
    const bool akc = (g.a_cs == 1), bkc = (g.b_rs == 1);
    LAMP_CHECK(akc || g.a_rs == 1, "internal: A has no unit stride");
    LAMP_CHECK(bkc || g.b_cs == 1, "internal: B has no unit stride");
    const int64_t lda = akc ? g.a_rs : g.a_cs, ldb = bkc ? g.b_cs : g.b_rs;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    g.a_vec = (lda % 8 == 0) && al16(g.A) && (g.a_bs % 8 == 0);
    g.b_vec = (ldb % 8 == 0) && al16(g.B) && (g.b_bs % 8 == 0);
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (int)((g.N + BN - 1) / BN);
    dim3 grid(g.tiles_m * g.tiles_n, 1, g.batch);
    const size_t lds = 4 * TILE_BYTES;
    if (akc && !bkc) hipLaunchKernelGGL((gemm_bf16_kernel<true, false, f16_t>), grid, dim3(256), lds, stm, g);
    else if (akc && bkc) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, f16_t>), grid, dim3(256), lds, stm, g);
    else if (!akc && !bkc) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, f16_t>), grid, dim3(256), lds, stm, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, true, f16_t>), grid, dim3(256), lds, stm, g);
  } else if (a->dtype == kF32 || a->dtype == kF64) {
    g.tiles_m = (int)((g.M + FM - 1) / FM);
    g.tiles_n = (int)((g.N + FN - 1) / FN);
    dim3 grid(g.tiles_m * g.tiles_n, 1, g.batch);
    const bool a_kc32 = (g.a_cs == 1), b_kc32 = (g.b_rs == 1);
    const int64_t lda32 = a_kc32 ? g.a_rs : g.a_cs, ldb32 = b_kc32 ? g.b_cs : g.b_rs;
    const bool big32 = a->dtype == kF32 && (lda32 % 4 == 0) && (ldb32 % 4 == 0) && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0 &&
                       (g.a_bs % 4 == 0) && (g.b_bs % 4 == 0) && ((g.M + GM - 1) / GM) * ((g.N + GN - 1) / GN) * g.batch >= 128;
    // Few output tiles over a long K (config 1's MLP: 1024 x 784 . 784 x 256 is 64 tiles on 256 CUs, each walking 49 k-steps one global
    // round trip at a time: 35 us for 0.4 GFLOP): K is split over blockIdx.z until every CU has a workgroup (chunks of at least 64), the
    // f32 / f64 slices are summed in order by a second launch.  LAMP_GEMM_SPLITK=0: off.
    // What this changes: a row's dot product is no longer ONE fma chain over K but `split` chains added in slice order, and
    // `split` depends on the tile count, i.e. on M and N - the same input row can give different last bits at another batch size (within
    // the 1e-5 the f32 path is held to; deterministic for a given shape).  The slices carry no epilogue: only (alpha, beta, S) of GemmArgs
    // are applied, by the reduction - any other epilogue field keeps the un-split path (the condition below lists them).
    static const bool allow_fp_split = !(getenv("LAMP_GEMM_SPLITK") && atoi(getenv("LAMP_GEMM_SPLITK")) == 0);
    const int64_t fp_tiles = (int64_t)g.tiles_m * g.tiles_n;
    if (allow_fp_split && g.batch == 1 && !g.knn_q && fp_tiles < 128 && g.K >= 128) {
      static const int64_t fp_split_wgs = [] { const char* e = getenv("LAMP_GEMM_FP_SPLIT_WGS"); return e ? std::max(1, atoi(e)) : 512; }();
      int64_t split = std::min<int64_t>(std::min<int64_t>((fp_split_wgs + fp_tiles - 1) / fp_tiles, g.K / 64), 32);
      if (split > 1) {
        const int64_t kc = ((g.K + split - 1) / split + FK - 1) / FK * FK;
        split = (g.K + kc - 1) / kc;
      }
      if (split > 1) {
        const int64_t kc = ((g.K + split - 1) / split + FK - 1) / FK * FK;
        Hold ws(new_tensor({split, g.M, g.N}, a->dtype, out->device()));
        GemmArgs h = g;
        h.k_split_total = g.K; h.K = kc;
        h.a_bs = kc * g.a_cs; h.b_bs = kc * g.b_rs;
        h.C = ws->data(); h.ldc = g.N; h.c_bs = g.M * g.N; h.S = nullptr; h.alpha = 1.0; h.beta = 0.0;
        dim3 sgrid(g.tiles_m * g.tiles_n, 1, (unsigned)split);
        if (a->dtype == kF32) {
          hipLaunchKernelGGL((gemm_fp_kernel<float>), sgrid, dim3(256), 0, stm, h);
          LAMP_LAUNCH_CHECK();
          hipLaunchKernelGGL((gemm_fp_splitk_reduce_kernel<float>), dim3(grid_for(g.M * g.N, 256)), dim3(256), 0, stm, (const float*)ws->data(), (int)split, g.M, g.N,
                             (float*)g.C, g.ldc, (const float*)g.S, g.s_rs, g.s_cs, (float)g.alpha, (float)g.beta);
        } else {
          hipLaunchKernelGGL((gemm_fp_kernel<double>), sgrid, dim3(256), 0, stm, h);
          LAMP_LAUNCH_CHECK();
          hipLaunchKernelGGL((gemm_fp_splitk_reduce_kernel<double>), dim3(grid_for(g.M * g.N, 256)), dim3(256), 0, stm, (const double*)ws->data(), (int)split, g.M,
                             g.N, (double*)g.C, g.ldc, (const double*)g.S, g.s_rs, g.s_cs, g.alpha, g.beta);
        }
        LAMP_LAUNCH_CHECK();
        return;
      }
    }
    if (big32) {
      g.tiles_m = (int)((g.M + GM - 1) / GM);
      g.tiles_n = (int)((g.N + GN - 1) / GN);
      hipLaunchKernelGGL(gemm_f32_big_kernel, dim3(g.tiles_m * g.tiles_n, 1, g.batch), dim3(256), 0, stm, g);
    } else if (a->dtype == kF32) hipLaunchKernelGGL((gemm_fp_kernel<float>), grid, dim3(256), 0, stm, g);
    else hipLaunchKernelGGL((gemm_fp_kernel<double>), grid, dim3(256), 0, stm, g);
  } else {
    LAMP_CHECK(false, "GEMM supports bf16, f16, f32 and f64, got " << a->describe());
  }
  LAMP_LAUNCH_CHECK();
}

// out[i, j] = max(0, qn[i] + dn[j] - 2 * q[i, :] . x[j, :])  in one kernel (f32 / f64): the kNN distance block
void knn_distance_block(Tensor* out, const Tensor* q, const Tensor* x, const Tensor* qn, const Tensor* dn) {
  LAMP_CHECK(qn->is_contiguous() && dn->is_contiguous() && qn->numel() == q->sizes[0] && dn->numel() == x->sizes[0] && qn->dtype == q->dtype &&
             dn->dtype == q->dtype, "knn_distance_block: bad norm vectors");
  gemm_dispatch(out, nullptr, q, x, false, true, 0.0, 1.0, false, qn, dn);
}

static Tensor* alloc_out(const Tensor* a, const Tensor* b, bool transA, bool transB, bool batched) {
  const int nd = batched ? 3 : 2;
  LAMP_CHECK(a->ndim == nd && b->ndim == nd, "expected " << nd << "-D operands, got " << a->describe() << " and " << b->describe());
  int64_t M = transA ? a->sizes[nd - 1] : a->sizes[nd - 2];
  int64_t N = transB ? b->sizes[nd - 2] : b->sizes[nd - 1];
  std::vector<int64_t> s;
  if (batched) s.push_back(a->sizes[0]);
  s.push_back(M); s.push_back(N);
  return new_tensor(s, a->dtype, a->device());
}

}  // namespace lamp

using namespace lamp;

extern "C" {

// lamp's CPU device: mm of tensors that live in host memory (f32 / f64; f64 accumulation, i-k-j loop) - fixtures and small host
// side products; models multiply on the GPU
static Tensor* host_mm(const Tensor* a, const Tensor* b) {
  LAMP_CHECK(a->ndim == 2 && b->ndim == 2 && a->sizes[1] == b->sizes[0] && a->dtype == b->dtype, "mm: shapes " << a->describe() << " x " << b->describe());
  LAMP_CHECK(a->dtype == kF32 || a->dtype == kF64, "host mm supports f32 and f64, got " << a->describe());
  const int64_t M = a->sizes[0], K = a->sizes[1], N = b->sizes[1];
  int64_t os[2] = {M, N};
  Hold r(new_tensor(os, 2, a->dtype, -1));
  std::vector<double> row((size_t)N);
  for (int64_t i = 0; i < M; i++) {
    std::fill(row.begin(), row.end(), 0.0);
    for (int64_t k = 0; k < K; k++) {
      const double av = a->dtype == kF32 ? (double)a->ptr<float>()[i * a->strides[0] + k * a->strides[1]] : a->ptr<double>()[i * a->strides[0] + k * a->strides[1]];
      for (int64_t j = 0; j < N; j++)
        row[j] += av * (b->dtype == kF32 ? (double)b->ptr<float>()[k * b->strides[0] + j * b->strides[1]] : b->ptr<double>()[k * b->strides[0] + j * b->strides[1]]);
    }
    for (int64_t j = 0; j < N; j++) { if (a->dtype == kF32) r->ptr<float>()[i * N + j] = (float)row[j]; else r->ptr<double>()[i * N + j] = row[j]; }
  }
  return r.take();
}

int lamp_mm(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN
  if (a && b && !a->is_device() && !b->is_device()) { *out = host_mm(a, b); return 0; }
  check_device_tensor(a, "self"); check_device_tensor(b, "mat2");
  Hold r(alloc_out(a, b, false, false, false));
  gemm_dispatch(r.get(), nullptr, a, b, false, false, 0.0, 1.0, false);
  *out = r.take();
  LAMP_API_END
}
int lamp_mm_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN gemm_dispatch(out, nullptr, a, b, false, false, 0.0, 1.0, false); LAMP_API_END
}
int lamp_bmm(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_device_tensor(b, "mat2");
  Hold r(alloc_out(a, b, false, false, true));
  gemm_dispatch(r.get(), nullptr, a, b, false, false, 0.0, 1.0, true);
  *out = r.take();
  LAMP_API_END
}
int lamp_bmm_out(lamp_tensor* out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN gemm_dispatch(out, nullptr, a, b, false, false, 0.0, 1.0, true); LAMP_API_END
}
int lamp_addmm(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha) {
  LAMP_API_BEGIN
  check_device_tensor(a, "mat1"); check_device_tensor(b, "mat2");
  Hold r(alloc_out(a, b, false, false, false));
  gemm_dispatch(r.get(), self, a, b, false, false, beta, alpha, false);
  *out = r.take();
  LAMP_API_END
}
int lamp_addmm_out(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha) {
  LAMP_API_BEGIN gemm_dispatch(out, self, a, b, false, false, beta, alpha, false); LAMP_API_END
}
int lamp_baddbmm(lamp_tensor** out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha) {
  LAMP_API_BEGIN
  check_device_tensor(a, "batch1"); check_device_tensor(b, "batch2");
  Hold r(alloc_out(a, b, false, false, true));
  gemm_dispatch(r.get(), self, a, b, false, false, beta, alpha, true);
  *out = r.take();
  LAMP_API_END
}
int lamp_addmm_out_transposed1(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha) {
  LAMP_API_BEGIN gemm_dispatch(out, self, a, b, true, false, beta, alpha, false); LAMP_API_END
}
int lamp_addmm_out_transposed2(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha) {
  LAMP_API_BEGIN gemm_dispatch(out, self, a, b, false, true, beta, alpha, false); LAMP_API_END
}
int lamp_baddbmm_out_transposed1(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha) {
  LAMP_API_BEGIN gemm_dispatch(out, self, a, b, true, false, beta, alpha, true); LAMP_API_END
}
int lamp_baddbmm_out_transposed2(lamp_tensor* out, const lamp_tensor* self, const lamp_tensor* a, const lamp_tensor* b, double beta, double alpha) {
  LAMP_API_BEGIN gemm_dispatch(out, self, a, b, false, true, beta, alpha, true); LAMP_API_END
}
int lamp_linear_bias(lamp_tensor** out, const lamp_tensor* x, const lamp_tensor* w, const lamp_tensor* bias) {
  LAMP_API_BEGIN
  check_device_tensor(x, "x"); check_device_tensor(w, "w");
  Hold r(alloc_out(x, w, false, false, false));
  gemm_dispatch(r.get(), bias, x, w, false, false, bias ? 1.0 : 0.0, 1.0, false, nullptr, nullptr, /*round_first: x.mm(w) + bias, rounded as the chain*/ true);
  *out = r.take();
  LAMP_API_END
}
// matmul: 1-D/2-D/3-D combinations lamp uses (STen.scala:1223)
int lamp_matmul(lamp_tensor** out, const lamp_tensor* a, const lamp_tensor* b) {
  LAMP_API_BEGIN
  check_device_tensor(a, "self"); check_device_tensor(b, "other");
  if (a->ndim == 2 && b->ndim == 2) return lamp_mm(out, a, b);
  if (a->ndim == 3 && b->ndim == 3) return lamp_bmm(out, a, b);
  if (a->ndim == 1 && b->ndim == 1) {
    lamp_tensor *a2 = nullptr, *b2 = nullptr, *r = nullptr;
    int64_t s1[2] = {1, a->sizes[0]}, s2[2] = {b->sizes[0], 1};
    LAMP_CHECK(lamp_view(&a2, a, s1, 2) == 0, lamp_last_error()); Hold ha(a2);
    LAMP_CHECK(lamp_reshape(&b2, b, s2, 2) == 0, lamp_last_error()); Hold hb(b2);
    LAMP_CHECK(lamp_mm(&r, a2, b2) == 0, lamp_last_error()); Hold hr(r);
    return lamp_view(out, r, nullptr, 0);
  }
  if (a->ndim == 2 && b->ndim == 1) {
    lamp_tensor *b2 = nullptr, *r = nullptr;
    int64_t s2[2] = {b->sizes[0], 1};
    LAMP_CHECK(lamp_reshape(&b2, b, s2, 2) == 0, lamp_last_error()); Hold hb(b2);
    LAMP_CHECK(lamp_mm(&r, a, b2) == 0, lamp_last_error()); Hold hr(r);
    int64_t so[1] = {a->sizes[0]};
    return lamp_view(out, r, so, 1);
  }
  if (a->ndim == 1 && b->ndim == 2) {
    lamp_tensor *a2 = nullptr, *r = nullptr;
    int64_t s1[2] = {1, a->sizes[0]};
    LAMP_CHECK(lamp_reshape(&a2, a, s1, 2) == 0, lamp_last_error()); Hold ha(a2);
    LAMP_CHECK(lamp_mm(&r, a2, b) == 0, lamp_last_error()); Hold hr(r);
    int64_t so[1] = {b->sizes[1]};
    return lamp_view(out, r, so, 1);
  }
  if (a->ndim == 3 && b->ndim == 2) {  // [B,M,K] x [K,N] -> fold the batch into M
    lamp_tensor *a2 = nullptr, *r = nullptr;
    int64_t s1[2] = {a->sizes[0] * a->sizes[1], a->sizes[2]};
    LAMP_CHECK(lamp_reshape(&a2, a, s1, 2) == 0, lamp_last_error()); Hold ha(a2);
    LAMP_CHECK(lamp_mm(&r, a2, b) == 0, lamp_last_error()); Hold hr(r);
    int64_t so[3] = {a->sizes[0], a->sizes[1], b->sizes[1]};
    return lamp_view(out, r, so, 3);
  }
  LAMP_CHECK(false, "matmul: unsupported operand ranks " << a->ndim << " and " << b->ndim);
  LAMP_API_END
}

}  // extern "C"
