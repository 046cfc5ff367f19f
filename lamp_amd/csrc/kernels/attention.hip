// Fused scaled-dot-product attention forward (flash form) for bf16, head dim 64 or 128.
//
// Replaces the CUDA-only fused path of lamp's ScaledDotProductAttention op (reference:
// lamp-core/src/main/scala/lamp/autograd/ops.scala:2342-2390, STen.scala:501-584 -
// _scaled_dot_product_cudnn_attention) whose semantics are softmax(Q K^T * scale + causal mask) V with a
// per-row logsumexp as the saved tensor.  The S x S score matrix is never written: a workgroup owns 128 query
// rows of one (batch, head) - 32 per wave, so that every K / V fragment read from LDS feeds two MFMAs - and streams the
// keys / values in tiles of 64.
//
// Everything is computed TRANSPOSED so that no register shuffle is needed between the two matrix products:
//   S^T[key][q]  = K[key][:] . Q[q][:]      A = K tile (LDS, ds_read_b128), B = Q fragments (registers)
//     D layout: a lane holds ONE query (lane & 15) and keys 16*mt + 4*(lane >> 4) + reg
//     -> the online-softmax row statistics need two xor-shuffles (16, 32), the running rescale is per lane
//   O^T[d][q]   += V^T[d][key] . P^T[key][q]   B = P^T: the lane's own eight probabilities of two key tiles, packed
//     to bf16 in place (the MFMA k-slot <-> key assignment is chosen to match), A = V^T read from the row-major
//     V tile with the transposing ds_read_b64_tr_b16.
//   O^T's D layout gives a lane four consecutive d of its query: 8-byte stores.
// K / V tiles arrive by LDS-DMA into two buffers (swizzled images, swizzle on the per-lane source address), the
// next tile is in flight while the current one is consumed.
#include "device_utils.h"
#include <type_traits>

namespace lamp {

typedef short at_s8 __attribute__((ext_vector_type(8)));
typedef short at_s4 __attribute__((ext_vector_type(4)));
typedef __bf16 at_bf8 __attribute__((ext_vector_type(8)));
typedef float at_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char at_lds_t;
typedef const __attribute__((address_space(1))) char at_glb_t;

constexpr int AT_QT = 2;                     // 16-query tiles per wave
constexpr int AT_BQ = 64 * AT_QT, AT_BK = 64;   // queries per workgroup (4 waves), keys per tile

// K tile: DH/64 sub-images [64 keys][64 d], 128-byte rows, chunk' = chunk ^ (row & 7)
__device__ __forceinline__ int at_k_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// V tile: [64 keys][DH d], 32-byte pairs of a row XOR (key & (DH/16 - 1)): the 8 consecutive keys one 32-lane half of a
// transposing read touches land in 8 (DH = 128) / 4 (DH = 64) different 32-byte slots
template <int DH> __device__ __forceinline__ int at_v_off(int key, int col8) {
  constexpr int PAIRS = DH / 16;
  return key * (DH * 2) + ((((col8 >> 1) ^ (key & (PAIRS - 1)))) << 5) + ((col8 & 1) << 4);
}

// ---- LDS reads are issued from inline asm: hipcc cannot tell that a C++ LDS load does not alias the LDS-DMA in flight and
// drains it (s_waitcnt vmcnt(0)) in front of the first read of every tile, which exposes the whole K / V prefetch.  The
// fences below are the s_waitcnt that "produce" the registers for the compiler's scheduler.
template <int OFF> __device__ __forceinline__ void at_read128(at_s8& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void at_read128f(at_f4& d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void at_read_tr(at_s4& d, unsigned a) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
#define AT_FENCE2(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]) : : "memory")
#define AT_FENCE4(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : : "memory")
#define AT_FENCE8(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]) : : "memory")
template <int I, int N, class F> __device__ __forceinline__ void at_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); at_static_for<I + 1, N>(f); }
}
// xor-16 / xor-32 butterflies on the VALU (gfx950 v_permlane*_swap) instead of ds_bpermute round trips through the LDS pipe.
// With both operands holding x, the swap leaves (rows 0,0,2,2 | rows 1,1,3,3) resp. (low half twice | high half twice).
__device__ __forceinline__ float at_max_x16_x32(float x) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a = fmaxf(a, b); b = a;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ float at_sum_x16_x32(float x) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a = a + b; b = a;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// Inline-asm VALU ops are invisible to hipcc's hazard recogniser: an asm instruction that reads an MFMA result too early gets stale
// data (no hardware interlock).  These "settle" fences sit between the MFMAs and their first asm consumer: they depend on the
// accumulators (so they are scheduled after the MFMAs) and spend the required wait states (11 for an 8-pass MFMA).
#define AT_SETTLE "s_nop 7\n\ts_nop 7"
__device__ __forceinline__ void at_settle2(at_f4& a, at_f4& b) { asm volatile(AT_SETTLE : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void at_settle4(at_f4& a, at_f4& b, at_f4& c, at_f4& d) { asm volatile(AT_SETTLE : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void at_settle8(at_f4& a, at_f4& b, at_f4& c, at_f4& d, at_f4& e, at_f4& f, at_f4& g, at_f4& h) {
  asm volatile(AT_SETTLE : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ float at_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float at_fma1(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float at_add1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Workgroup -> (batch x head, query block) for the query-block kernels.  gridDim = (query blocks, BH), walked x-fastest by the
// dispatcher.  Non-causal: the identity.  Causal: the blocks of AT_CH consecutive (batch, head) pairs form a chunk that is
// walked heaviest query block first, (batch, head) fastest - the K / V of a chunk (AT_CH x 2 MB at S = 4096, d = 128) stay in L2 as
// before, but the LAST workgroups of the launch are now the lightest ones of every head instead of whole heads (the tail of a
// launch with only ~8 workgroup generations was 15 % of it).
constexpr int AT_CH = 16;
// Where row `r` of (batch b, head h) starts, in elements: b * sb + h * sh + r * sr.  Contiguous (B, heads, S, d): {S d heads, S d, d};
// a (B, heads, S, d) VIEW of projections stored (B, S, heads, d) - what lamp's multi-head attention holds after `mm1` + `view`
// (Transformer.scala:925-945) - : {S heads d, d, heads d}.  The kernels read and write such views in place: no transposed copies.
struct AtLay { int64_t sb, sh, sr; };
__device__ __forceinline__ int64_t at_base(const AtLay& l, int bh, int H) { const int b = bh / H, h = bh - b * H; return b * l.sb + h * l.sh; }
__device__ __forceinline__ void at_block_of(int causal, int& bh, int& qb) {
  const int nqb = gridDim.x, BH = gridDim.y;
  if (!causal) { bh = blockIdx.y; qb = blockIdx.x; return; }
  const int id = blockIdx.y * nqb + blockIdx.x;
  const int chunk = id / (AT_CH * nqb), r = id - chunk * (AT_CH * nqb);
  const int chl = min(AT_CH, BH - chunk * AT_CH);
  bh = chunk * AT_CH + r % chl;
  qb = nqb - 1 - r / chl;
}

template <int DH>
__global__ __launch_bounds__(256, 2) void sdpa_flash_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                bf16_t* __restrict__ o, float* __restrict__ lse, int Sq, int Sk, float scale, int causal,
                                                                int H, AtLay lq, AtLay lk, AtLay lv, AtLay lo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 32;                 // k-steps of the S^T product
  constexpr int DT = DH / 16;                 // 16-row tiles of O^T
  constexpr int KIMG = AT_BK * DH * 2;        // bytes of one K (or V) tile
  constexpr int VPITCH = DH * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  int bh_, qb_;
  at_block_of(causal, bh_, qb_);                // causal: the query blocks with the most keys go first
  const int64_t bh = bh_;
  const int q0 = qb_ * AT_BQ;
  const bf16_t* qp = q + at_base(lq, bh_, H);
  const bf16_t* kp = k + at_base(lk, bh_, H);
  const bf16_t* vp = v + at_base(lv, bh_, H);
  char* Vl = smem;                            // [2][KIMG]  (first: the immediate offsets of its reads stay below 64 KiB)
  char* Kl = smem + 2 * KIMG;                 // [2][KIMG]

  // this lane's AT_QT queries and their Q fragments (B operand: 8 consecutive d per k-step)
  const int qw0 = q0 + wid * (16 * AT_QT);    // first query of the wave
  int qi[AT_QT];
  at_bf8 qf[AT_QT][KS];
#pragma unroll
  for (int t = 0; t < AT_QT; t++) {
    qi[t] = qw0 + t * 16 + (lane & 15);
    const int qrow = qi[t] < Sq ? qi[t] : Sq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) qf[t][ks] = __builtin_bit_cast(at_bf8, *reinterpret_cast<const at_s8*>(qp + (int64_t)qrow * lq.sr + ks * 32 + g * 8));
  }

  auto dma_tile = [&](int kt, int buf) {
    const int key0 = kt * AT_BK;
    // K: DH/64 sub-images of 8 pieces (1 KiB = 8 rows x 128 B); V: KIMG / 1024 pieces
    constexpr int KP = DH / 64 * 8, VP = KIMG / 1024;
#pragma unroll
    for (int i = 0; i < KP / 4; i++) {
      const int piece = wid * (KP / 4) + i;
      const int sub = piece >> 3, pp = (piece & 7) * 64 + lane;
      const int row = pp >> 3, chunk = (pp & 7) ^ (row & 7);
      int key = key0 + row; key = key < Sk ? key : Sk - 1;
      __builtin_amdgcn_global_load_lds((at_glb_t*)(kp + (int64_t)key * lk.sr + sub * 64 + chunk * 8), (at_lds_t*)(Kl + buf * KIMG + piece * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < VP / 4; i++) {
      const int piece = wid * (VP / 4) + i;
      const int pp = piece * 64 + lane;
      constexpr int CPR = DH / 8;             // 16-byte chunks per key row
      const int row = pp / CPR, c = pp % CPR;
      const int col8 = ((((c >> 1) ^ (row & (DH / 16 - 1)))) << 1) | (c & 1);
      int key = key0 + row; key = key < Sk ? key : Sk - 1;
      __builtin_amdgcn_global_load_lds((at_glb_t*)(vp + (int64_t)key * lv.sr + col8 * 8), (at_lds_t*)(Vl + buf * KIMG + piece * 1024), 16, 0, 0);
    }
  };

  // per-lane LDS byte addresses (buffer 0); everything else about a read is an immediate offset
  //   K fragment (key tile mt, k-step ks):  kbase[ks & 1] + mt * 2048 + (ks >> 1) * 8192
  //   V^T fragment (d tile dt, 32-key step tp, half hi): vbase[dt] + tp * 32 * VPITCH + hi * 16 * VPITCH
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned kbase[2], vbase[DT];
#pragma unroll
  for (int h = 0; h < 2; h++) { kbase[h] = lds0 + 2 * KIMG + at_k_off(lane & 15, h * 4 + g); asm volatile("" : "+v"(kbase[h])); }
  {
    const int qq = (lane >> 2) & 3, pc = lane & 3;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { vbase[dt] = lds0 + at_v_off<DH>(4 * g + qq, dt * 2 + (pc >> 1)) + ((pc & 1) << 3); asm volatile("" : "+v"(vbase[dt])); }
  }

  at_f4 acc_o[AT_QT][DT];
  float m_run[AT_QT], l_run[AT_QT];           // l_run: this lane's share of the row sum (its 16 keys of every tile)
#pragma unroll
  for (int t = 0; t < AT_QT; t++) {
    m_run[t] = -INFINITY; l_run[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) acc_o[t][dt] = at_f4{0.f, 0.f, 0.f, 0.f};
  }
  const float c2 = scale * 1.44269504088896340736f;   // scores are kept in the log2 domain: p = exp2(s c2 - m)

  int nkt = (Sk + AT_BK - 1) / AT_BK;
  if (causal) { const int last_q = min(q0 + AT_BQ, Sq) - 1; nkt = min(nkt, last_q / AT_BK + 1); }
  if (nkt > 0) dma_tile(0, 0);
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // tile kt is in LDS; every wave is done with tile kt-1
    if (kt + 1 < nkt) dma_tile(kt + 1, buf ^ 1);
    // causal: a wave whose last query precedes this tile's first key has nothing to add (wave-uniform)
    if (causal && kt * AT_BK > qw0 + 16 * AT_QT - 1) continue;
    const unsigned boff = buf * KIMG;
    const unsigned ka[2] = {kbase[0] + boff, kbase[1] + boff};
    // ---- S^T = K Q^T: the K fragments of key tile mt+1 are in flight while tile mt multiplies; every fragment feeds both
    //      query tiles of the wave
    at_f4 s[AT_QT][4];
    at_s8 kf[2][KS];
    at_s4 vlo[2][DT], vhi[2][DT];
    auto k_issue = [&](auto mtc, at_s8* dst) {
      constexpr int mt = decltype(mtc)::value;
      at_static_for<0, KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; at_read128<mt * 2048 + (ks >> 1) * 8192>(dst[ks], ka[ks & 1]); });
    };
    auto v_issue = [&](auto tpc, at_s4* lo, at_s4* hi) {
      constexpr int tp = decltype(tpc)::value;
      at_static_for<0, DT>([&](auto dtc) {
        constexpr int dt = decltype(dtc)::value;
        at_read_tr<tp * 32 * VPITCH>(lo[dt], vbase[dt] + boff);
        at_read_tr<tp * 32 * VPITCH + 16 * VPITCH>(hi[dt], vbase[dt] + boff);
      });
    };
    k_issue(std::integral_constant<int, 0>{}, kf[0]);
    at_static_for<0, 4>([&](auto mtc) {
      constexpr int mt = decltype(mtc)::value;
      if constexpr (mt < 3) k_issue(std::integral_constant<int, mt + 1>{}, kf[(mt + 1) & 1]);
      if constexpr (KS == 4) { if constexpr (mt < 3) AT_FENCE4("s_waitcnt lgkmcnt(4)", kf[mt & 1]); else AT_FENCE4("s_waitcnt lgkmcnt(0)", kf[mt & 1]); }
      else                   { if constexpr (mt < 3) AT_FENCE2("s_waitcnt lgkmcnt(2)", kf[mt & 1]); else AT_FENCE2("s_waitcnt lgkmcnt(0)", kf[mt & 1]); }
      // the V^T fragments of the first 32 keys fly during the last key tile's MFMAs and the softmax
      if constexpr (mt == 3) v_issue(std::integral_constant<int, 0>{}, vlo[0], vhi[0]);
#pragma unroll
      for (int t = 0; t < AT_QT; t++) s[t][mt] = at_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int t = 0; t < AT_QT; t++) s[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, kf[mt & 1][ks]), qf[t][ks], s[t][mt], 0, 0, 0);
    });
    static_assert(AT_QT == 2, "the settle fence below lists the score tiles of two query tiles");
    at_settle8(s[0][0], s[0][1], s[0][2], s[0][3], s[1][0], s[1][1], s[1][2], s[1][3]);
    // ---- online softmax: a lane owns 16 keys of each of its queries
    const bool need_mask = (kt * AT_BK + AT_BK > Sk) || (causal && kt * AT_BK + AT_BK - 1 > qw0);
    at_bf8 pf[AT_QT][2];
#pragma unroll
    for (int t = 0; t < AT_QT; t++) {
      if (need_mask) {
        // key 64 kt + 16 mt + 4 g + r is visible iff 16 mt + r < lim
        int lim = (causal ? min(Sk, qi[t] + 1) : Sk) - kt * AT_BK - 4 * g;
        asm volatile("" : "+v"(lim));          // keeps the 16 compares inside this branch
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
          for (int r = 0; r < 4; r++) s[t][mt][r] = (mt * 16 + r < lim) ? s[t][mt][r] : -INFINITY;
      }
      // v_max3 from asm: hipcc puts a canonicalising v_max in front of every fmaxf on an MFMA result
      float m_loc = at_max3(s[t][0][0], s[t][0][1], s[t][0][2]);
      m_loc = at_max3(m_loc, s[t][0][3], s[t][1][0]);
#pragma unroll
      for (int mt = 1; mt < 4; mt++) {
        m_loc = at_max3(m_loc, s[t][mt][1], s[t][mt][2]);
        if (mt < 3) m_loc = at_max3(m_loc, s[t][mt][3], s[t][mt + 1][0]); else m_loc = at_max3(m_loc, s[t][mt][3], s[t][mt][3]);
      }
      m_loc = at_max_x16_x32(m_loc) * c2;       // c2 > 0: max commutes with the scaling
      // deferred max: the reference point moves only for a jump of 2^8 (p stays below 2^8: harmless in f32 / bf16), so the
      // rescale of O is rare; -inf + 8 < anything finite starts it
      const float m_new = (m_loc > m_run[t] + 8.f) ? m_loc : m_run[t];
      const float m_sub = (m_new == -INFINITY) ? 0.f : m_new;   // rows with no visible key yet: every p is exp2(-inf) = 0
      const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_sub);
      at_f4 psum = {0.f, 0.f, 0.f, 0.f};
      // single-issue f32 ops (asm): hipcc would pack them into v_pk_*_f32, which costs more beside MFMAs than it saves
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          s[t][mt][r] = __builtin_amdgcn_exp2f(at_fma1(s[t][mt][r], c2, -m_sub));
          psum[r] = at_add1(psum[r], s[t][mt][r]);
        }
      l_run[t] = l_run[t] * alpha + ((psum[0] + psum[1]) + (psum[2] + psum[3]));
      m_run[t] = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
        const at_f4 av = {alpha, alpha, alpha, alpha};
#pragma unroll
        for (int dt = 0; dt < DT; dt++) acc_o[t][dt] *= av;
      }
#pragma unroll
      for (int tp = 0; tp < 2; tp++)
#pragma unroll
        for (int j = 0; j < 4; j++) { pf[t][tp][j] = (__bf16)s[t][2 * tp][j]; pf[t][tp][4 + j] = (__bf16)s[t][2 * tp + 1][j]; }
    }
    // ---- O^T += V^T P^T; k-slot (g, j) of k-step tp <-> key 32*tp + 16*(j >> 2) + 4*g + (j & 3)
    if constexpr (DT == 8) { AT_FENCE8("s_waitcnt lgkmcnt(0)", vlo[0]); AT_FENCE8("", vhi[0]); }
    else                   { AT_FENCE4("s_waitcnt lgkmcnt(0)", vlo[0]); AT_FENCE4("", vhi[0]); }
    v_issue(std::integral_constant<int, 1>{}, vlo[1], vhi[1]);
#pragma unroll
    for (int tp = 0; tp < 2; tp++) {
      if (tp == 1) {
        if constexpr (DT == 8) { AT_FENCE8("s_waitcnt lgkmcnt(0)", vlo[1]); AT_FENCE8("", vhi[1]); }
        else                   { AT_FENCE4("s_waitcnt lgkmcnt(0)", vlo[1]); AT_FENCE4("", vhi[1]); }
      }
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const at_s4 lo = vlo[tp][dt], hi = vhi[tp][dt];
        const at_s8 x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int t = 0; t < AT_QT; t++) acc_o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, x), pf[t][tp], acc_o[t][dt], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: O[q][16 dt + 4 g + 0..3] = O^T / l ; lse = (m + log2 l) ln 2
#pragma unroll
  for (int t = 0; t < AT_QT; t++) {
    const float l_tot = at_sum_x16_x32(l_run[t]);
    if (qi[t] >= Sq) continue;
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    bf16_t* op = o + at_base(lo, bh_, H) + (int64_t)qi[t] * lo.sr;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      const bf16_t o0(acc_o[t][dt][0] * inv), o1(acc_o[t][dt][1] * inv), o2(acc_o[t][dt][2] * inv), o3(acc_o[t][dt][3] * inv);
      uint2 pk;
      pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
      pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
      *reinterpret_cast<uint2*>(op + dt * 16 + g * 4) = pk;
    }
    if (g == 0) lse[bh * (int64_t)Sq + qi[t]] = l_tot > 0.f ? (m_run[t] + __log2f(l_tot)) * 0.69314718055994530942f : -INFINITY;
  }
}

// q, k, v: contiguous [BH, S, D] bf16; lse: f32 [BH, Sq] (as ATen's logsumexp).  Returns false when the shape is not covered (the caller composes the op instead).
bool small_attention_fwd(const Tensor* q, const Tensor* k, const Tensor* v, Tensor* out, Tensor* lse, int64_t BH, int64_t Sq, int64_t Sk, int64_t D,
                         int64_t Dv, int is_causal, double scale, hipStream_t st);   // attention_small.hip: S <= 16, d = 64
bool small_attention_bwd(const Tensor* go, const Tensor* q, const Tensor* k, const Tensor* v, const Tensor* out, const Tensor* lse, Tensor* dq, Tensor* dk,
                         Tensor* dv, int64_t BH, int64_t Sq, int64_t Sk, int64_t D, int64_t Dv, int is_causal, double scale, hipStream_t st);

// 4-D (B, heads, S, d) tensor whose rows are d contiguous elements: contiguous tensors and (B, heads, S, d) views of (B, S, heads, d)
// storage alike.  16-byte packets need every stride to be a multiple of 8 elements and an aligned base.
bool at_layout_of(const Tensor* t, AtLay* l) {
  if (t->ndim != 4 || (t->sizes[3] != 1 && t->strides[3] != 1)) return false;
  l->sb = t->sizes[0] > 1 ? t->strides[0] : 0;
  l->sh = t->sizes[1] > 1 ? t->strides[1] : 0;
  l->sr = t->sizes[2] > 1 ? t->strides[2] : t->sizes[3];
  if ((l->sb | l->sh | l->sr) & 7) return false;
  return (((uintptr_t)t->raw()) & 15) == 0;
}
// result buffer for a (B, heads, S, d) operator whose input `like` has that shape: laid out as `like` is - a view over
// (B, S, heads, d) storage when `like` is one (so that lamp's transpose(1, 2) + flatten back to (B, S, heads * d) is free)
Tensor* at_new_like_layout(const Tensor* like, int64_t B, int64_t H, int64_t S, int64_t D, int dtype) {
  const bool permuted = like->ndim == 4 && H > 1 && S > 1 && like->strides[1] < like->strides[2];
  if (!permuted) { int64_t sz[4] = {B, H, S, D}; return new_tensor(sz, 4, dtype, like->device()); }
  int64_t sz[4] = {B, S, H, D};
  Hold buf(new_tensor(sz, 4, dtype, like->device()));
  int64_t vs[4] = {B, H, S, D}, vst[4] = {S * H * D, D, H * D, 1};
  return new_view(buf.get(), vs, vst, 4, buf->offset);
}

// q, k, v, out: (B, heads, S, d) bf16 in any AtLay layout; lse: contiguous f32 [B, heads, Sq] (as ATen's logsumexp).
// Returns false when the shape / layout is not covered (the caller composes the op instead).
bool flash_attention_fwd(const Tensor* q, const Tensor* k, const Tensor* v, Tensor* out, Tensor* lse, int is_causal, double scale, hipStream_t st) {
  const int64_t B = q->sizes[0], H = q->sizes[1], Sq = q->sizes[2], D = q->sizes[3], Sk = k->sizes[2], Dv = v->sizes[3], BH = B * H;
  if (q->is_contiguous() && k->is_contiguous() && v->is_contiguous() && out->is_contiguous() &&
      small_attention_fwd(q, k, v, out, lse, BH, Sq, Sk, D, Dv, is_causal, scale, st)) return true;
  static const bool enabled = [] { const char* e = getenv("LAMP_FLASH_ATTENTION"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  if (q->dtype != kBF16 || lse->dtype != kF32 || D != Dv || !(D == 64 || D == 128) || Sq < 1 || Sk < 1 || Sq > (1 << 30) || Sk > (1 << 30)) return false;
  AtLay lq, lk, lv, lo;
  if (!at_layout_of(q, &lq) || !at_layout_of(k, &lk) || !at_layout_of(v, &lv) || !at_layout_of(out, &lo) || !lse->is_contiguous()) return false;
  KernelTimer kt("sdpa_flash_fwd", 4.0 * (double)BH * Sq * Sk * D * (is_causal ? 0.5 : 1.0), (double)BH * (2.0 * Sq + 2.0 * Sk) * D * 2, st);
  const dim3 grid((unsigned)((Sq + AT_BQ - 1) / AT_BQ), (unsigned)BH);
  const size_t lds = (size_t)4 * AT_BK * D * 2;
  if (D == 128) {
    allow_big_lds((const void*)sdpa_flash_fwd_kernel<128>);
    hipLaunchKernelGGL((sdpa_flash_fwd_kernel<128>), grid, dim3(256), lds, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(), v->ptr<bf16_t>(), out->ptr<bf16_t>(),
                       lse->ptr<float>(), (int)Sq, (int)Sk, (float)scale, is_causal, (int)H, lq, lk, lv, lo);
  } else {
    hipLaunchKernelGGL((sdpa_flash_fwd_kernel<64>), grid, dim3(256), lds, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(), v->ptr<bf16_t>(), out->ptr<bf16_t>(),
                       lse->ptr<float>(), (int)Sq, (int)Sk, (float)scale, is_causal, (int)H, lq, lk, lv, lo);
  }
  LAMP_LAUNCH_CHECK();
  return true;
}

// =====================================================================================================================
// Backward (flash form): P is recomputed from Q, K and the forward's logsumexp; nothing of size S x S is stored.
//   D_i  = sum_d dO[i][d] O[i][d]                               (sdpa_bwd_dsum_kernel, f32 [BH, Sq])
//   dQ   = scale * (P o (dO V^T - D)) K                         (sdpa_flash_bwd_dq_kernel: workgroup = 128 queries, streams K / V)
//   dK^T = scale * Q^T (P o (dO V^T - D)),  dV^T = dO^T P       (sdpa_flash_bwd_dkv_kernel: workgroup owns keys, streams Q / dO)
// S and dP are computed in both kernels (7 instead of 5 matrix products per tile pair) so that neither needs a sum across
// workgroups: no atomics, results independent of the schedule.
// Every LDS tile is a [64 rows][DH] row-major image with the 32-byte-pair swizzle of at_v_off: it serves both the row reads
// (ds_read_b128: rows x 8 consecutive d) and the transposing reads (ds_read_b64_tr_b16: d x 8 rows) without bank conflicts.
// As in the forward, each product is oriented so that the contraction index of the NEXT product is the in-lane dimension of
// this one's result: the dq kernel computes S^T / dP^T (lane = one query, 4 keys per tile), the dkv kernel S / dP (lane = one
// key, 4 queries per tile), and the packed bf16 probabilities feed the next MFMA's B operand with no cross-lane traffic.
// =====================================================================================================================
template <int DH> __device__ __forceinline__ void at_dma_rows(const bf16_t* base, int64_t sr, int row0, int nrows, char* lds, int wid, int lane) {
  constexpr int VP = AT_BK * DH * 2 / 1024, CPR = DH / 8;
#pragma unroll
  for (int i = 0; i < VP / 4; i++) {
    const int piece = wid * (VP / 4) + i;
    const int pp = piece * 64 + lane;
    const int row = pp / CPR, c = pp % CPR;
    const int col8 = ((((c >> 1) ^ (row & (DH / 16 - 1)))) << 1) | (c & 1);
    int r = row0 + row; r = r < nrows ? r : nrows - 1;
    __builtin_amdgcn_global_load_lds((at_glb_t*)(base + (int64_t)r * sr + col8 * 8), (at_lds_t*)(lds + piece * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256) void sdpa_bwd_dsum_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ o, float* __restrict__ dsum, int64_t rows, int D,
                                                            int Sq, int H, AtLay lg, AtLay lo) {
  // 16 lanes per row, 8-element packets
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  float acc = 0.f;
  if (row < rows) {
    const int bh = (int)(row / Sq), r = (int)(row - (int64_t)bh * Sq);
    const bf16_t* gp = dO + at_base(lg, bh, H) + (int64_t)r * lg.sr;
    const bf16_t* op = o + at_base(lo, bh, H) + (int64_t)r * lo.sr;
    for (int c = l * 8; c < D; c += 128) {
      const at_s8 a = *reinterpret_cast<const at_s8*>(gp + c), b = *reinterpret_cast<const at_s8*>(op + c);
#pragma unroll
      for (int j = 0; j < 8; j++) acc += __uint_as_float((unsigned)(unsigned short)a[j] << 16) * __uint_as_float((unsigned)(unsigned short)b[j] << 16);
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (row < rows && l == 0) dsum[row] = acc;
}

template <int DH>
__global__ __launch_bounds__(256, 2) void sdpa_flash_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                   const bf16_t* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                   bf16_t* __restrict__ dq, int Sq, int Sk, float scale, int causal,
                                                                   int H, AtLay lq, AtLay lk, AtLay lv, AtLay lg, AtLay ldq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 32, DT = DH / 16, KIMG = AT_BK * DH * 2, PITCH = DH * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  int bh_, qb_;
  at_block_of(causal, bh_, qb_);
  const int64_t bh = bh_;
  const int q0 = qb_ * AT_BQ;
  const bf16_t* qp = q + at_base(lq, bh_, H);
  const bf16_t* dop = dO + at_base(lg, bh_, H);
  const bf16_t* kp = k + at_base(lk, bh_, H);
  const bf16_t* vp = v + at_base(lv, bh_, H);
  char* Kl = smem;                            // [2][KIMG]
  char* Vl = smem + 2 * KIMG;                 // [2][KIMG]

  const int qw0 = q0 + wid * (16 * AT_QT);
  int qi[AT_QT];
  at_bf8 qf[AT_QT][KS], dof[AT_QT][KS];
  float lse2[AT_QT], dsm[AT_QT];
#pragma unroll
  for (int t = 0; t < AT_QT; t++) {
    qi[t] = qw0 + t * 16 + (lane & 15);
    const int qrow = qi[t] < Sq ? qi[t] : Sq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      qf[t][ks] = __builtin_bit_cast(at_bf8, *reinterpret_cast<const at_s8*>(qp + (int64_t)qrow * lq.sr + ks * 32 + g * 8));
      dof[t][ks] = __builtin_bit_cast(at_bf8, *reinterpret_cast<const at_s8*>(dop + (int64_t)qrow * lg.sr + ks * 32 + g * 8));
    }
    lse2[t] = lse[bh * (int64_t)Sq + qrow] * 1.44269504088896340736f;
    dsm[t] = dsum[bh * (int64_t)Sq + qrow];
  }
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned rbase[KS], tbase[DT];               // row-read / transposing-read byte addresses in tile 0 of K
#pragma unroll
  for (int ks = 0; ks < KS; ks++) { rbase[ks] = lds0 + at_v_off<DH>(lane & 15, ks * 4 + g); asm volatile("" : "+v"(rbase[ks])); }
  {
    const int qq = (lane >> 2) & 3, pc = lane & 3;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { tbase[dt] = lds0 + at_v_off<DH>(4 * g + qq, dt * 2 + (pc >> 1)) + ((pc & 1) << 3); asm volatile("" : "+v"(tbase[dt])); }
  }
  at_f4 acc[AT_QT][DT];
#pragma unroll
  for (int t = 0; t < AT_QT; t++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) acc[t][dt] = at_f4{0.f, 0.f, 0.f, 0.f};
  const float c2 = scale * 1.44269504088896340736f;

  int nkt = (Sk + AT_BK - 1) / AT_BK;
  if (causal) { const int last_q = min(q0 + AT_BQ, Sq) - 1; nkt = min(nkt, last_q / AT_BK + 1); }
  if (nkt > 0) { at_dma_rows<DH>(kp, lk.sr, 0, Sk, Kl, wid, lane); at_dma_rows<DH>(vp, lv.sr, 0, Sk, Vl, wid, lane); }
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nkt) { at_dma_rows<DH>(kp, lk.sr, (kt + 1) * AT_BK, Sk, Kl + (buf ^ 1) * KIMG, wid, lane); at_dma_rows<DH>(vp, lv.sr, (kt + 1) * AT_BK, Sk, Vl + (buf ^ 1) * KIMG, wid, lane); }
    if (causal && kt * AT_BK > qw0 + 16 * AT_QT - 1) continue;
    const unsigned koff = buf * KIMG, voff = 2 * KIMG + buf * KIMG;
    const bool need_mask = (kt * AT_BK + AT_BK > Sk) || (causal && kt * AT_BK + AT_BK - 1 > qw0);
    int lim[AT_QT];
#pragma unroll
    for (int t = 0; t < AT_QT; t++) lim[t] = (causal ? min(Sk, qi[t] + 1) : Sk) - kt * AT_BK - 4 * g;
    // the 64 keys are processed as two halves of 32 (one k-step of the dQ product each): keeps 32 instead of 64 score registers live
    at_static_for<0, 2>([&](auto tpc) {
      constexpr int tp = decltype(tpc)::value;
      at_f4 s[AT_QT][2], dp[AT_QT][2];
      at_s8 fr[2][KS];
      auto r_issue = [&](auto mtc, unsigned off, at_s8* dst) {
        constexpr int mt = decltype(mtc)::value;
        at_static_for<0, KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; at_read128<mt * 16 * PITCH>(dst[ks], rbase[ks] + off); });
      };
      // ---- S^T = K Q^T and dP^T = V dO^T for key tiles 2 tp, 2 tp + 1: 4 fragment sets stream through two register buffers
      r_issue(std::integral_constant<int, 2 * tp>{}, koff, fr[0]);
      at_static_for<0, 4>([&](auto ic) {
        constexpr int i = decltype(ic)::value, h = i & 1;
        if constexpr (i == 0) r_issue(std::integral_constant<int, 2 * tp + 1>{}, koff, fr[1]);
        if constexpr (i == 1) r_issue(std::integral_constant<int, 2 * tp>{}, voff, fr[0]);
        if constexpr (i == 2) r_issue(std::integral_constant<int, 2 * tp + 1>{}, voff, fr[1]);
        if constexpr (KS == 4) { if constexpr (i < 3) AT_FENCE4("s_waitcnt lgkmcnt(4)", fr[i & 1]); else AT_FENCE4("s_waitcnt lgkmcnt(0)", fr[i & 1]); }
        else                   { if constexpr (i < 3) AT_FENCE2("s_waitcnt lgkmcnt(2)", fr[i & 1]); else AT_FENCE2("s_waitcnt lgkmcnt(0)", fr[i & 1]); }
#pragma unroll
        for (int t = 0; t < AT_QT; t++) { if constexpr (i < 2) s[t][h] = at_f4{0.f, 0.f, 0.f, 0.f}; else dp[t][h] = at_f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
          for (int t = 0; t < AT_QT; t++) {
            if constexpr (i < 2) s[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, fr[i & 1][ks]), qf[t][ks], s[t][h], 0, 0, 0);
            else dp[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, fr[i & 1][ks]), dof[t][ks], dp[t][h], 0, 0, 0);
          }
      });
      at_settle4(s[0][0], s[0][1], s[1][0], s[1][1]);
      // ---- K^T fragments of these 32 keys fly during the elementwise part
      at_s4 klo[DT], khi[DT];
      at_static_for<0, DT>([&](auto dtc) {
        constexpr int dt = decltype(dtc)::value;
        at_read_tr<tp * 32 * PITCH>(klo[dt], tbase[dt] + koff);
        at_read_tr<tp * 32 * PITCH + 16 * PITCH>(khi[dt], tbase[dt] + koff);
      });
      // ---- dS^T = P^T o (dP^T - D) * scale
      at_bf8 dsf[AT_QT];
#pragma unroll
      for (int t = 0; t < AT_QT; t++) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float p = __builtin_amdgcn_exp2f(at_fma1(s[t][h][r], c2, -lse2[t]));
            if (need_mask) p = ((2 * tp + h) * 16 + r < lim[t]) ? p : 0.f;
            dsf[t][4 * h + r] = (__bf16)(p * (dp[t][h][r] - dsm[t]) * scale);
          }
      }
      // ---- dQ^T += K^T dS^T
      if constexpr (DT == 8) { AT_FENCE8("s_waitcnt lgkmcnt(0)", klo); AT_FENCE8("", khi); }
      else                   { AT_FENCE4("s_waitcnt lgkmcnt(0)", klo); AT_FENCE4("", khi); }
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const at_s4 lo = klo[dt], hi = khi[dt];
        const at_s8 x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int t = 0; t < AT_QT; t++) acc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, x), dsf[t], acc[t][dt], 0, 0, 0);
      }
    });
  }
#pragma unroll
  for (int t = 0; t < AT_QT; t++) {
    if (qi[t] >= Sq) continue;
    bf16_t* op = dq + at_base(ldq, bh_, H) + (int64_t)qi[t] * ldq.sr;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      const bf16_t o0(acc[t][dt][0]), o1(acc[t][dt][1]), o2(acc[t][dt][2]), o3(acc[t][dt][3]);
      uint2 pk;
      pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
      pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
      *reinterpret_cast<uint2*>(op + dt * 16 + g * 4) = pk;
    }
  }
}

// dK, dV: a wave owns NK key tiles of 16 (its K / V fragments stay in registers), the workgroup streams Q / dO / lse / D in
// 64-query tiles and consumes them 32 queries at a time (one MFMA k-step of the dK / dV products).
template <int DH, int NK>
__global__ __launch_bounds__(256, 2) void sdpa_flash_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                 const bf16_t* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                 bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int Sq, int Sk, float scale, int causal,
                                                                 int H, AtLay lq, AtLay lk, AtLay lv, AtLay lg, AtLay ldk, AtLay ldv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 32, DT = DH / 16, KIMG = AT_BK * DH * 2, PITCH = DH * 2;
  constexpr int BKW = 4 * 16 * NK;             // keys per workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int64_t bh = blockIdx.y;
  const int bh_ = blockIdx.y;
  const int k0 = blockIdx.x * BKW;
  const bf16_t* qp = q + at_base(lq, bh_, H);
  const bf16_t* dop = dO + at_base(lg, bh_, H);
  const bf16_t* kp = k + at_base(lk, bh_, H);
  const bf16_t* vp = v + at_base(lv, bh_, H);
  const float* lsep = lse + bh * (int64_t)Sq;
  const float* dsp = dsum + bh * (int64_t)Sq;
  char* Ql = smem;                            // [2][KIMG]
  char* Ol = smem + 2 * KIMG;                 // [2][KIMG]  (dO)
  char* Sl = smem + 4 * KIMG;                 // [2][2][64] f32: lse, D of the tile's 64 queries

  const int kw0 = k0 + wid * (16 * NK);
  int ki[NK];
  at_bf8 kfr[NK][KS], vfr[NK][KS];
#pragma unroll
  for (int u = 0; u < NK; u++) {
    ki[u] = kw0 + u * 16 + (lane & 15);
    const int krow = ki[u] < Sk ? ki[u] : Sk - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      kfr[u][ks] = __builtin_bit_cast(at_bf8, *reinterpret_cast<const at_s8*>(kp + (int64_t)krow * lk.sr + ks * 32 + g * 8));
      vfr[u][ks] = __builtin_bit_cast(at_bf8, *reinterpret_cast<const at_s8*>(vp + (int64_t)krow * lv.sr + ks * 32 + g * 8));
    }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned rbase[KS], tbase[DT];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) { rbase[ks] = lds0 + at_v_off<DH>(lane & 15, ks * 4 + g); asm volatile("" : "+v"(rbase[ks])); }
  {
    const int qq = (lane >> 2) & 3, pc = lane & 3;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { tbase[dt] = lds0 + at_v_off<DH>(4 * g + qq, dt * 2 + (pc >> 1)) + ((pc & 1) << 3); asm volatile("" : "+v"(tbase[dt])); }
  }
  at_f4 acck[NK][DT], accv[NK][DT];
#pragma unroll
  for (int u = 0; u < NK; u++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { acck[u][dt] = at_f4{0.f, 0.f, 0.f, 0.f}; accv[u][dt] = at_f4{0.f, 0.f, 0.f, 0.f}; }
  const float c2 = scale * 1.44269504088896340736f;

  const int nqt = (Sq + AT_BK - 1) / AT_BK;
  const int qt0 = causal ? min(k0 / AT_BK, nqt) : 0;    // queries before the block's first key see none of its keys
  auto dma_stats = [&](int qt, int buf) {     // wave 0: lse, wave 1: D; one 4-byte element per lane
    if (wid < 2) {
      int r = qt * AT_BK + lane; r = r < Sq ? r : Sq - 1;
      __builtin_amdgcn_global_load_lds((at_glb_t*)((wid == 0 ? lsep : dsp) + r), (at_lds_t*)(Sl + buf * 512 + wid * 256), 4, 0, 0);
    }
  };
  const unsigned sbase = lds0 + 4 * KIMG + 16 * g;   // + buf * 512 + array * 256 + m * 64
  if (qt0 < nqt) { at_dma_rows<DH>(qp, lq.sr, qt0 * AT_BK, Sq, Ql, wid, lane); at_dma_rows<DH>(dop, lg.sr, qt0 * AT_BK, Sq, Ol, wid, lane); dma_stats(qt0, 0); }
  for (int qt = qt0; qt < nqt; qt++) {
    const int buf = (qt - qt0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (qt + 1 < nqt) {
      at_dma_rows<DH>(qp, lq.sr, (qt + 1) * AT_BK, Sq, Ql + (buf ^ 1) * KIMG, wid, lane);
      at_dma_rows<DH>(dop, lg.sr, (qt + 1) * AT_BK, Sq, Ol + (buf ^ 1) * KIMG, wid, lane);
      dma_stats(qt + 1, buf ^ 1);
    }
    const unsigned qoff = buf * KIMG, ooff = 2 * KIMG + buf * KIMG;
    // wave-uniform: does this (query tile, key range of the wave) touch the sequence ends or the diagonal?
    const bool need_mask = (qt * AT_BK + AT_BK > Sq) || (kw0 + 16 * NK > Sk) || (causal && kw0 + 16 * NK - 1 > qt * AT_BK);
    // the lane's rows of a 16-query tile are 4 g .. 4 g + 3: its lse / D values are one 16-byte LDS read per tile
    at_f4 lq[4], dq_[4];
    at_static_for<0, 4>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      at_read128f<m * 64>(lq[m], sbase + buf * 512);
      at_read128f<256 + m * 64>(dq_[m], sbase + buf * 512);
    });
    AT_FENCE4("s_waitcnt lgkmcnt(0)", lq); AT_FENCE4("", dq_);
#pragma unroll
    for (int m = 0; m < 4; m++) lq[m] *= 1.44269504088896340736f;
#pragma unroll
    for (int half = 0; half < 2; half++) {      // 32 queries: tiles m = 2 half, 2 half + 1
      at_f4 s[2][NK], dp[2][NK];
      // ---- S = Q K^T, dP = dO V^T (A = rows of the Q / dO tile, B = the wave's resident K / V fragments)
      at_static_for<0, 2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        const int m = 2 * half + h;
        at_s8 fq[KS], fo[KS];
        at_static_for<0, KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; at_read128<0>(fq[ks], rbase[ks] + qoff + m * 16 * PITCH); });
        at_static_for<0, KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; at_read128<0>(fo[ks], rbase[ks] + ooff + m * 16 * PITCH); });
        if constexpr (KS == 4) { AT_FENCE4("s_waitcnt lgkmcnt(4)", fq); } else { AT_FENCE2("s_waitcnt lgkmcnt(2)", fq); }
#pragma unroll
        for (int u = 0; u < NK; u++) {
          s[h][u] = at_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ks++) s[h][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, fq[ks]), kfr[u][ks], s[h][u], 0, 0, 0);
        }
        if constexpr (KS == 4) { AT_FENCE4("s_waitcnt lgkmcnt(0)", fo); } else { AT_FENCE2("s_waitcnt lgkmcnt(0)", fo); }
#pragma unroll
        for (int u = 0; u < NK; u++) {
          dp[h][u] = at_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ks++) dp[h][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, fo[ks]), vfr[u][ks], dp[h][u], 0, 0, 0);
        }
      });
      if constexpr (NK == 1) at_settle2(s[0][0], s[1][0]); else at_settle4(s[0][0], s[0][NK - 1], s[1][0], s[1][NK - 1]);
      // ---- transposed dO fragments fly during the elementwise part
      at_s4 tlo[DT], thi[DT];
      auto t_issue = [&](unsigned off) {
        at_static_for<0, DT>([&](auto dtc) {
          constexpr int dt = decltype(dtc)::value;
          at_read_tr<0>(tlo[dt], tbase[dt] + off + half * 32 * PITCH);
          at_read_tr<16 * PITCH>(thi[dt], tbase[dt] + off + half * 32 * PITCH);
        });
      };
      auto t_fence = [&]() {
        if constexpr (DT == 8) { AT_FENCE8("s_waitcnt lgkmcnt(0)", tlo); AT_FENCE8("", thi); }
        else                   { AT_FENCE4("s_waitcnt lgkmcnt(0)", tlo); AT_FENCE4("", thi); }
      };
      t_issue(ooff);
      // ---- P and dS for the lane's key (column) and 8 queries; both become B operands: k-slot (g, j) <-> query 16 (j >> 2) + 4 g + (j & 3)
      at_bf8 pb[NK], dsb[NK];
#pragma unroll
      for (int u = 0; u < NK; u++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int m = 2 * half + h;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float p = __builtin_amdgcn_exp2f(at_fma1(s[h][u][r], c2, -lq[m][r]));
            if (need_mask) {
              const int qidx = qt * AT_BK + m * 16 + 4 * g + r;
              p = (qidx < Sq && ki[u] < Sk && (!causal || ki[u] <= qidx)) ? p : 0.f;
            }
            pb[u][4 * h + r] = (__bf16)p;
            dsb[u][4 * h + r] = (__bf16)(p * (dp[h][u][r] - dq_[m][r]) * scale);
          }
        }
      }
      // ---- dV^T += dO^T P ; dK^T += Q^T dS
      t_fence();
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const at_s8 x = {tlo[dt][0], tlo[dt][1], tlo[dt][2], tlo[dt][3], thi[dt][0], thi[dt][1], thi[dt][2], thi[dt][3]};
#pragma unroll
        for (int u = 0; u < NK; u++) accv[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, x), pb[u], accv[u][dt], 0, 0, 0);
      }
      t_issue(qoff);
      t_fence();
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const at_s8 x = {tlo[dt][0], tlo[dt][1], tlo[dt][2], tlo[dt][3], thi[dt][0], thi[dt][1], thi[dt][2], thi[dt][3]};
#pragma unroll
        for (int u = 0; u < NK; u++) acck[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, x), dsb[u], acck[u][dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NK; u++) {
    if (ki[u] >= Sk) continue;
    bf16_t* okp = dk + at_base(ldk, bh_, H) + (int64_t)ki[u] * ldk.sr;
    bf16_t* ovp = dv + at_base(ldv, bh_, H) + (int64_t)ki[u] * ldv.sr;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      {
        const bf16_t o0(acck[u][dt][0]), o1(acck[u][dt][1]), o2(acck[u][dt][2]), o3(acck[u][dt][3]);
        uint2 pk; pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16); pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
        *reinterpret_cast<uint2*>(okp + dt * 16 + g * 4) = pk;
      }
      {
        const bf16_t o0(accv[u][dt][0]), o1(accv[u][dt][1]), o2(accv[u][dt][2]), o3(accv[u][dt][3]);
        uint2 pk; pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16); pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
        *reinterpret_cast<uint2*>(ovp + dt * 16 + g * 4) = pk;
      }
    }
  }
}

// dq, dk, dv <- grad_out, q, k, v, out, lse (f32).  Returns false when the shape is not covered.
bool flash_attention_bwd(const Tensor* go, const Tensor* q, const Tensor* k, const Tensor* v, const Tensor* out, const Tensor* lse, Tensor* dq, Tensor* dk,
                         Tensor* dv, Tensor* dsum /* f32 [BH * Sq] scratch */, int is_causal, double scale, hipStream_t st) {
  const int64_t B = q->sizes[0], H = q->sizes[1], Sq = q->sizes[2], D = q->sizes[3], Sk = k->sizes[2], Dv = v->sizes[3], BH = B * H;
  const Tensor* all[9] = {go, q, k, v, out, lse, dq, dk, dv};
  bool contig = true;
  for (auto* t : all) contig = contig && t->is_contiguous();
  if (contig && small_attention_bwd(go, q, k, v, out, lse, dq, dk, dv, BH, Sq, Sk, D, Dv, is_causal, scale, st)) return true;
  static const bool enabled = [] { const char* e = getenv("LAMP_FLASH_ATTENTION"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  if (q->dtype != kBF16 || lse->dtype != kF32 || D != Dv || !(D == 64 || D == 128) || Sq < 1 || Sk < 1 || Sq > (1 << 30) || Sk > (1 << 30)) return false;
  AtLay lq, lk, lv, lg, lo, ldq, ldk, ldv;
  if (!at_layout_of(q, &lq) || !at_layout_of(k, &lk) || !at_layout_of(v, &lv) || !at_layout_of(go, &lg) || !at_layout_of(out, &lo) ||
      !at_layout_of(dq, &ldq) || !at_layout_of(dk, &ldk) || !at_layout_of(dv, &ldv) || !lse->is_contiguous()) return false;
  const int64_t rows = BH * Sq;
  hipLaunchKernelGGL(sdpa_bwd_dsum_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, go->ptr<bf16_t>(), out->ptr<bf16_t>(), dsum->ptr<float>(),
                     rows, (int)D, (int)Sq, (int)H, lg, lo);
  const size_t lds = (size_t)4 * AT_BK * D * 2 + 1024;
  {
    KernelTimer kt("sdpa_flash_bwd_dq", 6.0 * (double)BH * Sq * Sk * D * (is_causal ? 0.5 : 1.0), (double)BH * (3.0 * Sq + 2.0 * Sk) * D * 2, st);
    const dim3 grid((unsigned)((Sq + AT_BQ - 1) / AT_BQ), (unsigned)BH);
    if (D == 128) {
      allow_big_lds((const void*)sdpa_flash_bwd_dq_kernel<128>);
      hipLaunchKernelGGL((sdpa_flash_bwd_dq_kernel<128>), grid, dim3(256), lds, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(), v->ptr<bf16_t>(), go->ptr<bf16_t>(),
                         lse->ptr<float>(), dsum->ptr<float>(), dq->ptr<bf16_t>(), (int)Sq, (int)Sk, (float)scale, is_causal, (int)H, lq, lk, lv, lg, ldq);
    } else {
      hipLaunchKernelGGL((sdpa_flash_bwd_dq_kernel<64>), grid, dim3(256), lds, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(), v->ptr<bf16_t>(), go->ptr<bf16_t>(),
                         lse->ptr<float>(), dsum->ptr<float>(), dq->ptr<bf16_t>(), (int)Sq, (int)Sk, (float)scale, is_causal, (int)H, lq, lk, lv, lg, ldq);
    }
    LAMP_LAUNCH_CHECK();
  }
  {
    KernelTimer kt("sdpa_flash_bwd_dkv", 8.0 * (double)BH * Sq * Sk * D * (is_causal ? 0.5 : 1.0), (double)BH * (2.0 * Sq + 4.0 * Sk) * D * 2, st);
#define AT_DKV(DHV, NKV)                                                                                                                             \
  do {                                                                                                                                               \
    allow_big_lds((const void*)sdpa_flash_bwd_dkv_kernel<DHV, NKV>);                                                                                 \
    const dim3 grid((unsigned)((Sk + 64 * NKV - 1) / (64 * NKV)), (unsigned)BH);                                                                     \
    hipLaunchKernelGGL((sdpa_flash_bwd_dkv_kernel<DHV, NKV>), grid, dim3(256), lds, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(), v->ptr<bf16_t>(),        \
                       go->ptr<bf16_t>(), lse->ptr<float>(), dsum->ptr<float>(), dk->ptr<bf16_t>(), dv->ptr<bf16_t>(), (int)Sq, (int)Sk, (float)scale, \
                       is_causal, (int)H, lq, lk, lv, lg, ldk, ldv);                                                                                 \
  } while (0)
    // keys per wave, measured (B 8, h 16, S 4096): 32 keys need one wave per SIMD at d = 128 (3.45 vs 2.53 ms for 16 keys at two
    // waves per SIMD); at d = 64, 32 keys win without a causal mask (1.35 vs 1.46 ms) and lose with one (1.28 vs 1.04 ms)
    if (D == 128) AT_DKV(128, 1);
    else if (is_causal) AT_DKV(64, 1);
    else AT_DKV(64, 2);
#undef AT_DKV
    LAMP_LAUNCH_CHECK();
  }
  return true;
}

}  // namespace lamp
