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

__device__ __forceinline__ float at_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float at_fma1(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float at_add1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int DH>
__global__ __launch_bounds__(256, 2) void sdpa_flash_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                bf16_t* __restrict__ o, bf16_t* __restrict__ lse, int Sq, int Sk, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 32;                 // k-steps of the S^T product
  constexpr int DT = DH / 16;                 // 16-row tiles of O^T
  constexpr int KIMG = AT_BK * DH * 2;        // bytes of one K (or V) tile
  constexpr int VPITCH = DH * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int64_t bh = blockIdx.y;
  // causal: the query blocks with the most keys go first
  const int q0 = (causal ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * AT_BQ;
  const bf16_t* qp = q + bh * (int64_t)Sq * DH;
  const bf16_t* kp = k + bh * (int64_t)Sk * DH;
  const bf16_t* vp = v + bh * (int64_t)Sk * DH;
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
    for (int ks = 0; ks < KS; ks++) qf[t][ks] = __builtin_bit_cast(at_bf8, *reinterpret_cast<const at_s8*>(qp + (int64_t)qrow * DH + ks * 32 + g * 8));
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
      __builtin_amdgcn_global_load_lds((at_glb_t*)(kp + (int64_t)key * DH + sub * 64 + chunk * 8), (at_lds_t*)(Kl + buf * KIMG + piece * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < VP / 4; i++) {
      const int piece = wid * (VP / 4) + i;
      const int pp = piece * 64 + lane;
      constexpr int CPR = DH / 8;             // 16-byte chunks per key row
      const int row = pp / CPR, c = pp % CPR;
      const int col8 = ((((c >> 1) ^ (row & (DH / 16 - 1)))) << 1) | (c & 1);
      int key = key0 + row; key = key < Sk ? key : Sk - 1;
      __builtin_amdgcn_global_load_lds((at_glb_t*)(vp + (int64_t)key * DH + col8 * 8), (at_lds_t*)(Vl + buf * KIMG + piece * 1024), 16, 0, 0);
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
    bf16_t* op = o + (bh * (int64_t)Sq + qi[t]) * DH;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      const bf16_t o0(acc_o[t][dt][0] * inv), o1(acc_o[t][dt][1] * inv), o2(acc_o[t][dt][2] * inv), o3(acc_o[t][dt][3] * inv);
      uint2 pk;
      pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
      pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
      *reinterpret_cast<uint2*>(op + dt * 16 + g * 4) = pk;
    }
    if (g == 0) lse[bh * (int64_t)Sq + qi[t]] = bf16_t(l_tot > 0.f ? (m_run[t] + __log2f(l_tot)) * 0.69314718055994530942f : -INFINITY);
  }
}

// q, k, v: contiguous [BH, S, D] bf16.  Returns false when the shape is not covered (the caller composes the op instead).
bool flash_attention_fwd(const Tensor* q, const Tensor* k, const Tensor* v, Tensor* out, Tensor* lse, int64_t BH, int64_t Sq, int64_t Sk, int64_t D,
                         int64_t Dv, int is_causal, double scale, hipStream_t st) {
  static const bool enabled = [] { const char* e = getenv("LAMP_FLASH_ATTENTION"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  if (q->dtype != kBF16 || D != Dv || !(D == 64 || D == 128) || Sq < 1 || Sk < 1 || Sq > (1 << 30) || Sk > (1 << 30)) return false;
  if ((((uintptr_t)q->data() | (uintptr_t)k->data() | (uintptr_t)v->data() | (uintptr_t)out->data()) & 15) != 0) return false;
  KernelTimer kt("sdpa_flash_fwd", 4.0 * (double)BH * Sq * Sk * D * (is_causal ? 0.5 : 1.0), (double)BH * (2.0 * Sq + 2.0 * Sk) * D * 2, st);
  const dim3 grid((unsigned)((Sq + AT_BQ - 1) / AT_BQ), (unsigned)BH);
  const size_t lds = (size_t)4 * AT_BK * D * 2;
  if (D == 128) {
    static bool attr = false;
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void*)sdpa_flash_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((sdpa_flash_fwd_kernel<128>), grid, dim3(256), lds, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(), v->ptr<bf16_t>(), out->ptr<bf16_t>(),
                       lse->ptr<bf16_t>(), (int)Sq, (int)Sk, (float)scale, is_causal);
  } else {
    hipLaunchKernelGGL((sdpa_flash_fwd_kernel<64>), grid, dim3(256), lds, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(), v->ptr<bf16_t>(), out->ptr<bf16_t>(),
                       lse->ptr<bf16_t>(), (int)Sq, (int)Sk, (float)scale, is_causal);
  }
  LAMP_LAUNCH_CHECK();
  return true;
}

}  // namespace lamp
