// Scaled-dot-product attention for very short sequences (S <= 16, head width 64, bf16): one wavefront per (batch, head) problem.
//
// Reference: lamp-core/src/main/scala/lamp/autograd/ops.scala:2342-2390 (ScaledDotProductAttention over ATen's fused operator).
// lamp's MultiheadAttention hands that operator (batch, sequence, heads, d) views (nn/Transformer.scala:930-945), so the axis it
// attends over is the head axis: example-autoregressivelm runs batch x 384 problems of 12 x 12 scores per block.  The tiled
// flash kernels (attention.hip: 128 queries x 64 keys per step) spend such a problem's time on padding; here a problem's
// q / k / v (3 x 1.5 KB) are read once, the 16 x 16 score block lives in registers / LDS, and the kernel is bound by HBM:
//   forward : reads q, k, v, writes out + f32 logsumexp        = 4 x S x 64 x 2 bytes per problem
//   backward: reads q, k, v, out, dout, lse, writes dq, dk, dv  = 8 x S x 64 x 2 bytes per problem
// Lane (i = lane & 15, g = lane >> 4) owns query i; scores and dP for keys j = g + 4c; outputs for the 16 features g * 16 ....
// Softmax statistics in f32 with the same definition as the flash kernels (lse = ln sum exp(scale * s)); P stays f32.
#include "device_utils.h"
#include "../core/tensor.h"

namespace lamp {

namespace {
constexpr int SA_D = 64;      // head width
constexpr int SA_S = 16;      // maximum sequence length
constexpr int SA_WAVES = 4;   // problems per workgroup

__device__ __forceinline__ float sa_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float sa_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// stage one [S][64] bf16 tensor of a problem into its LDS image (row-major, 128-byte rows), 16-byte packets
__device__ __forceinline__ void sa_stage(const bf16_t* __restrict__ src, uint4* dst, int S, int lane) {
  const uint4* s = reinterpret_cast<const uint4*>(src);
  const int npk = S * (SA_D / 8);
  if (lane < npk) dst[lane] = s[lane];
  if (lane + 64 < npk) dst[lane + 64] = s[lane + 64];
}
// dot product of the 64-feature row held in registers (32 packed pairs) with an LDS row
__device__ __forceinline__ float sa_dot(const unsigned (&a)[32], const uint4* row) {
  float acc = 0.f;
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const uint4 b = row[p];
    const unsigned bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int w = 0; w < 4; w++) {
      acc = fmaf(sa_lo(a[p * 4 + w]), sa_lo(bw[w]), acc);
      acc = fmaf(sa_hi(a[p * 4 + w]), sa_hi(bw[w]), acc);
    }
  }
  return acc;
}
__device__ __forceinline__ void sa_load_row(const uint4* row, unsigned (&a)[32]) {
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const uint4 b = row[p];
    a[p * 4 + 0] = b.x; a[p * 4 + 1] = b.y; a[p * 4 + 2] = b.z; a[p * 4 + 3] = b.w;
  }
}
// acc[0..16) += w * row[g * 16 .. g * 16 + 16)
__device__ __forceinline__ void sa_axpy16(float (&acc)[16], float w, const uint4* row, int g) {
  const uint4 b0 = row[g * 2], b1 = row[g * 2 + 1];
  const unsigned bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int t = 0; t < 8; t++) {
    acc[2 * t] = fmaf(w, sa_lo(bw[t]), acc[2 * t]);
    acc[2 * t + 1] = fmaf(w, sa_hi(bw[t]), acc[2 * t + 1]);
  }
}
__device__ __forceinline__ void sa_store16(bf16_t* dst, const float (&acc)[16]) {
  unsigned w[8];
#pragma unroll
  for (int t = 0; t < 8; t++) w[t] = (unsigned)bf16_t(acc[2 * t]).bits | ((unsigned)bf16_t(acc[2 * t + 1]).bits << 16);
  uint4* d = reinterpret_cast<uint4*>(dst);
  d[0] = make_uint4(w[0], w[1], w[2], w[3]);
  d[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ float sa_group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float sa_group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// LDS per wavefront bounds the resident waves (160 KB per CU): forward 7.2 KB -> 20 waves, backward 9.3 KB -> 16 waves
struct SaLdsFwd {
  uint4 q[SA_S * 8], k[SA_S * 8], v[SA_S * 8];                // [S][64] bf16 images
  float p[SA_S][SA_S + 1];
};
struct SaLds {
  uint4 q[SA_S * 8], k[SA_S * 8], v[SA_S * 8], g[SA_S * 8];   // [S][64] bf16 images (g = dout)
  float p[SA_S][SA_S + 1];                                     // P, then dS
};
}  // namespace

__global__ __launch_bounds__(64 * SA_WAVES) void sdpa_small_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                                      bf16_t* __restrict__ O, float* __restrict__ LSE, int64_t nprob, int S, float scale,
                                                                      int causal) {
  __shared__ SaLdsFwd lds[SA_WAVES];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t prob = (int64_t)blockIdx.x * SA_WAVES + wid;
  const bool live = prob < nprob;
  const int64_t base = (live ? prob : 0) * S * SA_D;
  SaLdsFwd& L = lds[wid];
  const int i = lane & 15, g = lane >> 4;
  if (live) {
    sa_stage(Q + base, L.q, S, lane);
    sa_stage(K + base, L.k, S, lane);
    sa_stage(V + base, L.v, S, lane);
  }
  __syncthreads();
  if (!live) return;
  const bool row = i < S;
  unsigned qa[32];
  sa_load_row(L.q + (row ? i : 0) * 8, qa);
  float s[4], m = -INFINITY;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int j = g + 4 * c;
    const bool ok = row && j < S && !(causal && j > i);
    s[c] = ok ? sa_dot(qa, L.k + (j < S ? j : 0) * 8) * scale : -INFINITY;
    m = fmaxf(m, s[c]);
  }
  m = sa_group_max(m);
  float l = 0.f, p[4];
#pragma unroll
  for (int c = 0; c < 4; c++) { p[c] = s[c] == -INFINITY ? 0.f : __expf(s[c] - m); l += p[c]; }
  l = sa_group_sum(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int c = 0; c < 4; c++) L.p[i][g + 4 * c] = p[c] * inv;
  if (row && g == 0) LSE[prob * S + i] = l > 0.f ? m + __logf(l) : -INFINITY;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; t++) acc[t] = 0.f;
  for (int j = 0; j < S; j++) sa_axpy16(acc, L.p[i][j], L.v + j * 8, g);
  if (row) sa_store16(O + base + (int64_t)i * SA_D + g * 16, acc);
}

__global__ __launch_bounds__(64 * SA_WAVES) void sdpa_small_bwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                                      const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO, const float* __restrict__ LSE,
                                                                      bf16_t* __restrict__ dQ, bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, int64_t nprob,
                                                                      int S, float scale, int causal) {
  __shared__ SaLds lds[SA_WAVES];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t prob = (int64_t)blockIdx.x * SA_WAVES + wid;
  const bool live = prob < nprob;
  const int64_t base = (live ? prob : 0) * S * SA_D;
  SaLds& L = lds[wid];
  const int i = lane & 15, g = lane >> 4;
  if (live) {
    sa_stage(Q + base, L.q, S, lane);
    sa_stage(K + base, L.k, S, lane);
    sa_stage(V + base, L.v, S, lane);
    sa_stage(dO + base, L.g, S, lane);
  }
  __syncthreads();
  if (!live) return;
  const bool row = i < S;
  // D_i = sum_d dO[i][d] * O[i][d]: this lane's 16 features, then across the four lane groups
  float Di = 0.f;
  if (row) {
    const uint4* op = reinterpret_cast<const uint4*>(O + base + (int64_t)i * SA_D + g * 16);
    const uint4 o0 = op[0], o1 = op[1];
    const uint4 g0 = L.g[i * 8 + g * 2], g1 = L.g[i * 8 + g * 2 + 1];
    const unsigned ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w}, gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
    for (int t = 0; t < 8; t++) Di += sa_lo(ow[t]) * sa_lo(gw[t]) + sa_hi(ow[t]) * sa_hi(gw[t]);
  }
  Di = sa_group_sum(Di);
  const float lse = row ? LSE[prob * S + i] : 0.f;
  unsigned qa[32], ga[32];
  sa_load_row(L.q + (row ? i : 0) * 8, qa);
  sa_load_row(L.g + (row ? i : 0) * 8, ga);
  float dsv[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int j = g + 4 * c;
    const bool ok = row && j < S && !(causal && j > i) && lse != -INFINITY;
    float p = 0.f, ds = 0.f;
    if (ok) {
      p = __expf(sa_dot(qa, L.k + j * 8) * scale - lse);
      ds = p * (sa_dot(ga, L.v + j * 8) - Di) * scale;
    }
    L.p[i][j] = p;
    dsv[c] = ds;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float acc[16];
  // dV[j] = sum_i P[i][j] dO[i]   (this lane: key j = lane & 15)
#pragma unroll
  for (int t = 0; t < 16; t++) acc[t] = 0.f;
  for (int r = 0; r < S; r++) sa_axpy16(acc, L.p[r][i], L.g + r * 8, g);
  if (row) sa_store16(dV + base + (int64_t)i * SA_D + g * 16, acc);
  // the block now holds dS
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < 4; c++) L.p[i][g + 4 * c] = dsv[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // dQ[i] = sum_j dS[i][j] K[j]
#pragma unroll
  for (int t = 0; t < 16; t++) acc[t] = 0.f;
  for (int j = 0; j < S; j++) sa_axpy16(acc, L.p[i][j], L.k + j * 8, g);
  if (row) sa_store16(dQ + base + (int64_t)i * SA_D + g * 16, acc);
  // dK[j] = sum_i dS[i][j] Q[i]
#pragma unroll
  for (int t = 0; t < 16; t++) acc[t] = 0.f;
  for (int r = 0; r < S; r++) sa_axpy16(acc, L.p[r][i], L.q + r * 8, g);
  if (row) sa_store16(dK + base + (int64_t)i * SA_D + g * 16, acc);
}

static bool small_attention_enabled() {
  static const bool enabled = [] { const char* e = getenv("LAMP_SMALL_ATTENTION"); return !(e && e[0] == '0'); }();
  return enabled;
}
static bool small_attention_fits(const Tensor* q, int64_t Sq, int64_t Sk, int64_t D, int64_t Dv) {
  return small_attention_enabled() && q->dtype == kBF16 && D == SA_D && Dv == SA_D && Sq == Sk && Sq >= 1 && Sq <= SA_S;
}

bool small_attention_fwd(const Tensor* q, const Tensor* k, const Tensor* v, Tensor* out, Tensor* lse, int64_t BH, int64_t Sq, int64_t Sk, int64_t D,
                         int64_t Dv, int is_causal, double scale, hipStream_t st) {
  if (!small_attention_fits(q, Sq, Sk, D, Dv) || lse->dtype != kF32) return false;
  if ((((uintptr_t)q->data() | (uintptr_t)k->data() | (uintptr_t)v->data() | (uintptr_t)out->data()) & 15) != 0) return false;
  KernelTimer kt("sdpa_small_fwd", 4.0 * (double)BH * Sq * Sk * D, (double)BH * 4.0 * Sq * D * 2, st);
  hipLaunchKernelGGL(sdpa_small_fwd_kernel, dim3((unsigned)((BH + SA_WAVES - 1) / SA_WAVES)), dim3(64 * SA_WAVES), 0, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(),
                     v->ptr<bf16_t>(), out->ptr<bf16_t>(), lse->ptr<float>(), BH, (int)Sq, (float)scale, is_causal);
  LAMP_LAUNCH_CHECK();
  return true;
}

bool small_attention_bwd(const Tensor* go, const Tensor* q, const Tensor* k, const Tensor* v, const Tensor* out, const Tensor* lse, Tensor* dq, Tensor* dk,
                         Tensor* dv, int64_t BH, int64_t Sq, int64_t Sk, int64_t D, int64_t Dv, int is_causal, double scale, hipStream_t st) {
  if (!small_attention_fits(q, Sq, Sk, D, Dv) || lse->dtype != kF32) return false;
  const uintptr_t all = (uintptr_t)q->data() | (uintptr_t)k->data() | (uintptr_t)v->data() | (uintptr_t)out->data() | (uintptr_t)go->data() |
                        (uintptr_t)dq->data() | (uintptr_t)dk->data() | (uintptr_t)dv->data();
  if (all & 15) return false;
  KernelTimer kt("sdpa_small_bwd", 10.0 * (double)BH * Sq * Sk * D, (double)BH * 8.0 * Sq * D * 2, st);
  hipLaunchKernelGGL(sdpa_small_bwd_kernel, dim3((unsigned)((BH + SA_WAVES - 1) / SA_WAVES)), dim3(64 * SA_WAVES), 0, st, q->ptr<bf16_t>(), k->ptr<bf16_t>(),
                     v->ptr<bf16_t>(), out->ptr<bf16_t>(), go->ptr<bf16_t>(), lse->ptr<float>(), dq->ptr<bf16_t>(), dk->ptr<bf16_t>(), dv->ptr<bf16_t>(), BH,
                     (int)Sq, (float)scale, is_causal);
  LAMP_LAUNCH_CHECK();
  return true;
}

}  // namespace lamp
