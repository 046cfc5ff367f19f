// The second half of the matrix-core convolutions' weight gradients: f32 partial sums per image range -> one (rounded bf16 or f32) tensor, summed in
// a fixed order (deterministic).  The two bodies below are used by (a) the per-layer kernels of conv_igemm.hip / conv_narrow.hip and
// (b) the multi-tensor kernel of wgrad_reduce.hip, which runs ALL reductions a backward pass has registered in one launch:
// the ResNet step had 13 of these ~5 us launches, each at the ~4.5 us launch floor.
#pragma once
#include "device_utils.h"

namespace lamp {

struct WgradReduceArgs {
  int kind;                 // 0: implicit GEMM v2 layout [split][tap][COP][CIP], 1: narrow layout [block][O]
  const float* partial;
  void* dw;                 // bf16 (the bf16 convolutions) or f32 (conv_igemm_f32.hip: dw_f32 = 1)
  int dw_f32;
  int CO, CI, CIP, COP, RS; // kind 0 (COP x CIP = the padded tile the kernel wrote per tap)
  int nsplit;               // kind 0: image ranges; kind 1: blocks
  int O;                    // kind 1: outputs
  int blocks;               // workgroups (256 threads) this reduction needs
  int cached;               // 1: the reduction runs right behind the kernel that wrote the partial sums (they are read from the caches: plain loads)
};

// kind 0.  Thread (q, sg) sums float4 column q of this block over the splits sg, sg+8, ... (independent 16-byte loads in flight), the 8
// split groups are then combined through LDS in a fixed order.
__device__ __forceinline__ void wgrad_reduce_igemm(const WgradReduceArgs& a, int block, float4 (*red)[32]) {
  const int M = a.COP;
  const int q = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int per_split = a.RS * M * a.CIP / 4;                // float4 elements of one split (host: < 2^31)
  const int col = block * 32 + q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const int c4 = a.CIP / 4;                                  // a power of two (CIP = 16 or a multiple of 32 ... 128)
  const int row = col / c4;
  const int ci = (col - row * c4) * 4, rs = row / M, co = row - rs * M;
  const bool real = col < per_split && co < a.CO && ci < a.CI;   // padding of the tile: not written by the kernels that skip it, never read here
  if (real) {
    const float4* p4 = reinterpret_cast<const float4*>(a.partial) + col;
#pragma unroll 4
    for (int sp = sg; sp < a.nsplit; sp += 8) {
      // read exactly once: streamed past the caches (with the eight-wave kernel's streaming stores: -14 us per step; the same hints on the
      // small partial sums of the v2 / narrow kernels cost +12 us - those still sit in the Infinity Cache when the reduction runs)
      typedef float nt_f4 __attribute__((ext_vector_type(4)));
      float4 v;
      if (a.cached) v = p4[(int64_t)sp * per_split];
      else { const nt_f4 nv = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(&p4[(int64_t)sp * per_split])); v = make_float4(nv[0], nv[1], nv[2], nv[3]); }
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[sg][q] = s;
  __syncthreads();
  if (sg == 0 && real) {
#pragma unroll
    for (int g = 1; g < 8; g++) { const float4 v = red[g][q]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    const float r[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (ci + k < a.CI) {
        const int64_t o = ((int64_t)co * a.CI + ci + k) * a.RS + rs;
        if (a.dw_f32) static_cast<float*>(a.dw)[o] = r[k]; else static_cast<bf16_t*>(a.dw)[o] = bf16_t(r[k]);
      }
  }
}

// kind 1.  Wave per output element, lanes over the blocks, butterfly sum.
__device__ __forceinline__ void wgrad_reduce_narrow(const WgradReduceArgs& a, int block) {
  const int lane = threadIdx.x & 63;
  const int o = (int)(((int64_t)block * 256 + threadIdx.x) >> 6);
  if (o >= a.O) return;
  float s = 0.f;
  for (int b = lane; b < a.nsplit; b += 64) s += a.partial[(int64_t)b * a.O + o];
  s = wave_sum(s);
  if (lane == 0) { if (a.dw_f32) static_cast<float*>(a.dw)[o] = s; else static_cast<bf16_t*>(a.dw)[o] = bf16_t(s); }
}

// Registers the reduction (wgrad_reduce.hip): it runs with every other pending one at the next flush_deferred() - the end of
// backprop - or the moment anything asks for a pointer into dw's storage (Tensor::raw()), whichever comes first.
// LAMP_DEFER_WGRAD_REDUCE=0: launch it now.
void wgrad_reduce_enqueue(const WgradReduceArgs& a, lamp_tensor* partial, lamp_tensor* dw, hipStream_t st);
bool wgrad_reduce_deferred();      // false: LAMP_DEFER_WGRAD_REDUCE=0 (the producers then leave their partial sums in the caches)

}  // namespace lamp
