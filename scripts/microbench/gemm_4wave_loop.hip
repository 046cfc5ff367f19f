// A 256 x 256 x 64 GEMM main loop with FOUR waves of 128 x 128 (one per SIMD) instead of gemm_bf16_pp2_kernel's eight of 128 x 64: 16 fragment reads
// per 64 MFMAs instead of 24 (64 KiB of LDS reads per k-step and workgroup instead of 96), each wave pipelining its own reads under its MFMAs
// (two fragment sets), ONE barrier per stage.  EXPERIMENTS 61 found the eight-wave loop bound by its LDS port (DMA writes + fragment reads = the
// whole stage at the pipes' full rate); this measures whether the four-wave form gets closer to the 0.72 of 2.5 PF the MFMAs alone sustain on random data.
//   MODE 0  MFMAs only            MODE 1  + fragment reads (no DMA)       MODE 2  + DMA of L2-resident bytes      MODE 3  + DMA of the 4096^3 operands
//   hipcc -O3 --offload-arch=gfx950 gemm_4wave_loop.hip -o gemm_4wave_loop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_t;
typedef const __attribute__((address_space(1))) char glb_t;
constexpr int IMG = 256 * 64 * 2, SLOT = 2 * IMG;

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ u4v frag_kc(const char* img, int row0, int s, int lane) {
  u4v v;
  const unsigned a = (unsigned)(uintptr_t)(img + kc_off(row0 + (lane & 15), s * 4 + (lane >> 4)));
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}
#define FENCE8(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]) : : "memory")

template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(const char* __restrict__ src, float* out, unsigned long long* cyc, int stages, int rnd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  for (int o = tid * 16; o < 2 * SLOT; o += 256 * 16) {
    u4v v = u4v{0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    if (rnd) {
      unsigned x = (unsigned)(o + 1) * 2654435761u + blockIdx.x * 40503u, w[4];
      for (int j = 0; j < 4; j++) {
        unsigned h[2];
        for (int e = 0; e < 2; e++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; const float f = (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f; h[e] = __float_as_uint(f) >> 16; }
        w[j] = h[0] | (h[1] << 16);
      }
      v = u4v{w[0], w[1], w[2], w[3]};
    }
    *reinterpret_cast<u4v*>(smem + o) = v;
  }
  __syncthreads();
  f4v acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = f4v{0, 0, 0, 0};
  int tm = 0, tn = 0;
  { int bid = blockIdx.x; const int xcd = bid % 8; bid = xcd * 32 + bid / 8; tm = (bid / 64) * 4 + (bid % 64) % 4; tn = (bid % 64) / 4; }
  const char* Ag = src;
  const char* Bg = src + (size_t)4096 * 4096 * 2;
  // a quarter (q = 0 .. 3) of this wave's 16 pieces of stage t: pieces wid * 16 + 4 q .. + 3 (A: pieces 0 - 31, B: 32 - 63; both K-contiguous)
  auto dma_quarter = [&](int t, int slot, int q) {
    char* sb = smem + slot * SLOT;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int piece = wid * 16 + q * 4 + i;
      if (MODE == 2) {
        __builtin_amdgcn_global_load_lds((glb_t*)(src + (size_t)(t & 3) * SLOT + piece * 1024 + lane * 16), (lds_t*)(sb + piece * 1024), 16, 0, 0);
      } else {
        const int pp = (piece & 31) * 64 + lane, row = pp >> 3, chunk = (pp & 7) ^ (row & 7), k0 = (t & 63) * 64;
        const char* g = piece < 32 ? Ag + ((size_t)(tm * 256 + row) * 4096 + k0 + chunk * 8) * 2 : Bg + ((size_t)(tn * 256 + row) * 4096 + k0 + chunk * 8) * 2;
        __builtin_amdgcn_global_load_lds((glb_t*)g, (lds_t*)(sb + piece * 1024), 16, 0, 0);
      }
    }
  };
  u4v fa[2][8], fb[2][8];
  auto rd_half = [&](int set, const char* as, const char* bs, int s, int h) {       // half of a k-step's 16 fragment reads (h = 0: A, 1: B)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (h == 0) fa[set][i] = frag_kc(as, wr * 128 + i * 16, s, lane);
      else fb[set][i] = frag_kc(bs, wc * 128 + i * 16, s, lane);
    }
  };
  auto mfma_rows = [&](int set, int i0, int i1) {
#pragma unroll
    for (int i = i0; i < i1; i++)
#pragma unroll
      for (int j = 0; j < 8; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, fb[set][j]), __builtin_bit_cast(bf8v, fa[set][i]), acc[i][j], 0, 0, 0);
  };
  rd_half(0, smem, smem + IMG, 0, 0); rd_half(0, smem, smem + IMG, 0, 1);
  FENCE8("s_waitcnt lgkmcnt(0)", fa[0]); FENCE8("", fb[0]);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < stages; t++) {
    const char* as = smem + (t & 1) * SLOT;
    const char* bs = as + IMG;
    const char* an = smem + ((t + 1) & 1) * SLOT;
    const char* bn = an + IMG;
    if (MODE == 0) {
      asm volatile("" : "+v"(fa[0][0]), "+v"(fb[0][0]));
      __builtin_amdgcn_sched_barrier(0); mfma_rows(0, 0, 8); __builtin_amdgcn_sched_barrier(0); mfma_rows(0, 0, 8); __builtin_amdgcn_sched_barrier(0);
      continue;
    }
    // ---- k-step 0 (set 0) with the reads of k-step 1 (set 1) under it, in four quarters of 16 MFMAs
    __builtin_amdgcn_sched_barrier(0);
    rd_half(1, as, bs, 1, 0); __builtin_amdgcn_sched_barrier(0); mfma_rows(0, 0, 4); __builtin_amdgcn_sched_barrier(0);
    rd_half(1, as, bs, 1, 1); __builtin_amdgcn_sched_barrier(0); mfma_rows(0, 4, 8); __builtin_amdgcn_sched_barrier(0);
    FENCE8("s_waitcnt vmcnt(0) lgkmcnt(0)", fa[1]); FENCE8("", fb[1]);     // stage t + 1 (requested a stage ago) has landed; all reads of slot t are done
    __builtin_amdgcn_s_barrier();
    // ---- k-step 1 (set 1): the DMA of stage t + 2 into slot t, and the reads of stage t + 1's k-step 0 (set 0)
    if (MODE >= 2 && t + 2 < stages) { dma_quarter(t + 2, t & 1, 0); dma_quarter(t + 2, t & 1, 1); }
    rd_half(0, an, bn, 0, 0); __builtin_amdgcn_sched_barrier(0); mfma_rows(1, 0, 4); __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 2 && t + 2 < stages) { dma_quarter(t + 2, t & 1, 2); dma_quarter(t + 2, t & 1, 3); }
    rd_half(0, an, bn, 0, 1); __builtin_amdgcn_sched_barrier(0); mfma_rows(1, 4, 8); __builtin_amdgcn_sched_barrier(0);
    FENCE8("s_waitcnt lgkmcnt(0)", fa[0]); FENCE8("", fb[0]);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
}

template <int MODE>
static void run(const char* src, float* out, unsigned long long* cyc, int rnd, const char* what) {
  const int stages = 2048;
  (void)hipFuncSetAttribute((const void*)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(256), dim3(256), 2 * SLOT, 0, src, out, cyc, stages, rnd);
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  hipError_t err = hipGetLastError();
  const double flops = 256.0 * 4 * stages * 128.0 * 16384.0;
  printf("%s 4-wave mode %d (%s): launch %.3f ms = %.0f TFLOP/s = %.3f of 2.5 PF%s\n", rnd ? "RANDOM" : "const", MODE, what, ms, flops / (ms * 1e-3) * 1e-12,
         flops / (ms * 1e-3) * 1e-12 / 2500.0, err == hipSuccess ? "" : "  [HIP ERROR]");
}

int main() {
  float* out; unsigned long long* cyc; char* src;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
  const size_t bytes = (size_t)2 * 4096 * 4096 * 2;
  (void)hipMalloc(&src, bytes);
  for (int pass = 0; pass < 4; pass++) {
    const int rnd = pass & 1;
    if (rnd) {
      std::vector<unsigned short> h(bytes / 2);
      unsigned x = 12345u;
      for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; float f = (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
      (void)hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    } else (void)hipMemset(src, 0x3f, bytes);
    run<0>(src, out, cyc, rnd, "MFMAs only, one wave per SIMD");
    run<1>(src, out, cyc, rnd, "+ 16 fragment reads per k-step under the MFMAs, one barrier per stage");
    run<2>(src, out, cyc, rnd, "+ DMA of L2-resident bytes");
    run<3>(src, out, cyc, rnd, "+ DMA of the 4096^3 operands (p.W^T form)");
  }
  return 0;
}
