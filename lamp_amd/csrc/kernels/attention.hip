// Fused scaled-dot-product attention forward (flash form) for bf16, head dim 64 or 128.
//
// Replaces the CUDA-only fused path of lamp's ScaledDotProductAttention op (reference:
// lamp-core/src/main/scala/lamp/autograd/ops.scala:2342-2390, STen.scala:501-584 -
// _scaled_dot_product_cudnn_attention) whose semantics are softmax(Q K^T * scale + causal mask) V with a
// per-row logsumexp as the saved tensor.  The S x S score matrix is never written: a workgroup owns 64 query
// rows of one (batch, head) and streams the keys / values in tiles of 64.
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

namespace lamp {

typedef short at_s8 __attribute__((ext_vector_type(8)));
typedef short at_s4 __attribute__((ext_vector_type(4)));
typedef __bf16 at_bf8 __attribute__((ext_vector_type(8)));
typedef float at_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char at_lds_t;
typedef const __attribute__((address_space(1))) char at_glb_t;

constexpr int AT_BQ = 64, AT_BK = 64;

// K tile: DH/64 sub-images [64 keys][64 d], 128-byte rows, chunk' = chunk ^ (row & 7)
__device__ __forceinline__ int at_k_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// V tile: [64 keys][DH d], 32-byte pairs of a row XOR (key & (DH/16 - 1)): the 8 consecutive keys one 32-lane half of a
// transposing read touches land in 8 (DH = 128) / 4 (DH = 64) different 32-byte slots
template <int DH> __device__ __forceinline__ int at_v_off(int key, int col8) {
  constexpr int PAIRS = DH / 16;
  return key * (DH * 2) + ((((col8 >> 1) ^ (key & (PAIRS - 1)))) << 5) + ((col8 & 1) << 4);
}

template <int DH>
__global__ __launch_bounds__(256) void sdpa_flash_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                             bf16_t* __restrict__ o, bf16_t* __restrict__ lse, int Sq, int Sk, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 32;                 // k-steps of the S^T product
  constexpr int DT = DH / 16;                 // 16-row tiles of O^T
  constexpr int KIMG = AT_BK * DH * 2;        // bytes of one K (or V) tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4;
  const int64_t bh = blockIdx.y;
  const int q0 = blockIdx.x * AT_BQ;
  const bf16_t* qp = q + bh * (int64_t)Sq * DH;
  const bf16_t* kp = k + bh * (int64_t)Sk * DH;
  const bf16_t* vp = v + bh * (int64_t)Sk * DH;
  char* Kl = smem;                            // [2][KIMG]
  char* Vl = smem + 2 * KIMG;                 // [2][KIMG]

  // this lane's query and its Q fragments (B operand: 8 consecutive d per k-step)
  const int qi = q0 + wid * 16 + (lane & 15);
  const int qrow = qi < Sq ? qi : Sq - 1;
  at_bf8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) qf[ks] = __builtin_bit_cast(at_bf8, *reinterpret_cast<const at_s8*>(qp + (int64_t)qrow * DH + ks * 32 + g * 8));

  auto dma_tile = [&](int kt, int buf) {
    const int key0 = kt * AT_BK;
    // K: DH/64 sub-images of 8 pieces (1 KiB = 8 rows x 128 B); V: DH/8 ... pieces of 1 KiB
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

  at_f4 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; dt++) acc_o[dt] = at_f4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  int nkt = (Sk + AT_BK - 1) / AT_BK;
  if (causal) { const int last_q = min(q0 + AT_BQ, Sq) - 1; nkt = min(nkt, last_q / AT_BK + 1); }
  if (nkt > 0) dma_tile(0, 0);
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                           // tile kt is in LDS; every wave is done with tile kt-1
    if (kt + 1 < nkt) dma_tile(kt + 1, buf ^ 1);
    const char* kb = Kl + buf * KIMG;
    const char* vb = Vl + buf * KIMG;
    // ---- S^T = K Q^T
    at_f4 s[4];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
      s[mt] = at_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const at_s8 kf = *reinterpret_cast<const at_s8*>(kb + (ks >> 1) * (AT_BK * 128) + at_k_off(mt * 16 + (lane & 15), (ks & 1) * 4 + g));
        s[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, kf), qf[ks], s[mt], 0, 0, 0);
      }
    }
    // ---- online softmax over this lane's 16 keys of its query
    float m_loc = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int kk = kt * AT_BK + mt * 16 + g * 4 + r;
        const bool ok = kk < Sk && (!causal || kk <= qi);
        const float val = ok ? s[mt][r] * scale : -INFINITY;
        s[mt][r] = val;
        m_loc = fmaxf(m_loc, val);
      }
    m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16, 64));
    m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));
    const float m_new = fmaxf(m_run, m_loc);
    const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
    float l_loc = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float p = (m_new == -INFINITY) ? 0.f : __expf(s[mt][r] - m_new);
        s[mt][r] = p;
        l_loc += p;
      }
    l_loc += __shfl_xor(l_loc, 16, 64);
    l_loc += __shfl_xor(l_loc, 32, 64);
    l_run = l_run * alpha + l_loc;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) { acc_o[dt][0] *= alpha; acc_o[dt][1] *= alpha; acc_o[dt][2] *= alpha; acc_o[dt][3] *= alpha; }
    // ---- O^T += V^T P^T; k-slot (g, j) of k-step tp <-> key 32*tp + 16*(j >> 2) + 4*g + (j & 3)
#pragma unroll
    for (int tp = 0; tp < 2; tp++) {
      at_bf8 pf;
#pragma unroll
      for (int j = 0; j < 4; j++) { pf[j] = (__bf16)s[2 * tp][j]; pf[4 + j] = (__bf16)s[2 * tp + 1][j]; }
      const int qq = (lane >> 2) & 3, pc = lane & 3;
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const int col8 = dt * 2 + (pc >> 1);
        const unsigned a0 = (unsigned)(uintptr_t)(vb + at_v_off<DH>(32 * tp + 4 * g + qq, col8) + ((pc & 1) << 3));
        const unsigned a1 = (unsigned)(uintptr_t)(vb + at_v_off<DH>(32 * tp + 16 + 4 * g + qq, col8) + ((pc & 1) << 3));
        at_s4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
        const at_s8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc_o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(at_bf8, vf), pf, acc_o[dt], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: O[q][16 dt + 4 g + 0..3] = O^T / l ; lse = m + log l
  if (qi < Sq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    bf16_t* op = o + (bh * (int64_t)Sq + qi) * DH;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      const bf16_t o0(acc_o[dt][0] * inv), o1(acc_o[dt][1] * inv), o2(acc_o[dt][2] * inv), o3(acc_o[dt][3] * inv);
      uint2 pk;
      pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
      pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
      *reinterpret_cast<uint2*>(op + dt * 16 + g * 4) = pk;
    }
    if (g == 0) lse[bh * (int64_t)Sq + qi] = bf16_t(l_run > 0.f ? m_run + __logf(l_run) : -INFINITY);
  }
}

// q, k, v: contiguous [BH, S, D] bf16.  Returns false when the shape is not covered (the caller composes the op instead).
bool flash_attention_fwd(const Tensor* q, const Tensor* k, const Tensor* v, Tensor* out, Tensor* lse, int64_t BH, int64_t Sq, int64_t Sk, int64_t D,
                         int64_t Dv, int is_causal, double scale, hipStream_t st) {
  static const bool enabled = [] { const char* e = getenv("LAMP_FLASH_ATTENTION"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  if (q->dtype != kBF16 || D != Dv || !(D == 64 || D == 128) || Sq < 1 || Sk < 1 || Sq > (1 << 30) || Sk > (1 << 30)) return false;
  if ((((uintptr_t)q->data() | (uintptr_t)k->data() | (uintptr_t)v->data() | (uintptr_t)out->data()) & 15) != 0) return false;
  const dim3 grid((unsigned)((Sq + AT_BQ - 1) / AT_BQ), (unsigned)BH);
  const size_t lds = (size_t)4 * AT_BK * D * 2;
  KernelTimer kt("sdpa_flash_fwd", 4.0 * (double)BH * Sq * Sk * D * (is_causal ? 0.5 : 1.0), (double)BH * (2.0 * Sq + 2.0 * Sk) * D * 2, st);
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
