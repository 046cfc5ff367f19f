// The k-step of gemm_bf16_pp2_kernel (kernels/gemm.hip: 256 x 256 x 64 tile, 8 waves of 128 x 64, v_mfma_f32_16x16x32_bf16) as a BARE LOOP:
// what bounds the 4096^3 product once global memory is taken away ("settle the 60 % question by measurement").
// One workgroup of 512 threads per CU, 128 KiB of LDS holding two stages of the kernel's swizzled operand images, `stages` stages of two
// k-steps each; per k-step and wave: 8 A + 4 B fragment reads (12 x 1 KiB) and 32 MFMAs.
//   MODE 0  MFMAs only (fragments resident)                                   - the matrix pipe's own rate at two waves per SIMD
//   MODE 1  the kernel's schedule: READ | barrier | MFMA | barrier, waves 4 - 7 one phase behind (no DMA)
//   MODE 2  MODE 1 + the LDS-DMA of the next stage (64 KiB per stage from an L2-resident buffer), as the kernel issues and retires it
//   MODE 3  no phase barriers: every wave requests k-step s + 1's fragments (second register set) before k-step s's MFMAs; one barrier per stage
//   MODE 4  MODE 3 + the DMA of the next stage
//   MODE 5  MODE 4 with the 12 reads spread between the MFMAs (sched_group_barrier: 1 read per ~3 MFMAs)
// BT = 1: B fragments through ds_read_b64_tr_b16 (the forward x.W, B K-strided); BT = 0: ds_read_b128 (p.W^T).
//   hipcc -O3 --offload-arch=gfx950 gemm_kstep_loop.hip -o gemm_kstep_loop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_t;
typedef const __attribute__((address_space(1))) char glb_t;

constexpr int IMG = 256 * 64 * 2, SLOT = 2 * IMG;          // 32 KiB per operand image, 64 KiB per stage

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int ks_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int ks_off_p(int k, int col8, int pitch) { return k * pitch + ((((col8 >> 1) ^ ks_swz(k))) << 5) + ((col8 & 1) << 4); }

__device__ __forceinline__ u4v frag_kc(const char* img, int row0, int s, int lane) {
  u4v v;
  const unsigned a = (unsigned)(uintptr_t)(img + kc_off(row0 + (lane & 15), s * 4 + (lane >> 4)));
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}
__device__ __forceinline__ u4v frag_ks(const char* img, int row0, int s, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int k = s * 32 + g * 8 + q;
  const int col8 = (row0 >> 3) + (p >> 1);
  u2v lo, hi;
  const unsigned a0 = (unsigned)(uintptr_t)(img + ks_off_p(k, col8, 512) + ((p & 1) << 3));
  const unsigned a1 = (unsigned)(uintptr_t)(img + ks_off_p(k + 4, col8, 512) + ((p & 1) << 3));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
  return u4v{lo[0], lo[1], hi[0], hi[1]};
}
#define FENCE4(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : : "memory")
#define FENCE8(WAIT, F) asm volatile(WAIT : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]) : : "memory")

// SRC (modes with DMA): 0 = every workgroup fills its stages from the same 256 KiB (L2 hits only: what the DMA itself costs the loop - issue, LDS
// write port), 1 = the operands of a 4096^3 product with the kernel's own addressing and tile order (A and B shared between workgroups as there)
template <int MODE, int BT, int SRC>
__global__ __launch_bounds__(512) void loop_kernel(const char* __restrict__ src, float* out, unsigned long long* cyc, int stages, int ld, int group_m, int rot_mode, int rnd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  // operand values: RANDOM bf16 uniform in [-1, 1) (rnd = 1: the matrix pipe's power, and with it the clock, depends on the data - the benchmark's
  // operands are random) or the constants 1, 0.5, 0.25 (rnd = 0)
  for (int o = tid * 16; o < 2 * SLOT; o += 512 * 16) {
    u4v v = u4v{0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    if (rnd) {
      unsigned x = (unsigned)(o + 1) * 2654435761u + blockIdx.x * 40503u;
      unsigned w[4];
      for (int j = 0; j < 4; j++) {
        unsigned h[2];
        for (int e = 0; e < 2; e++) {
          x ^= x << 13; x ^= x >> 17; x ^= x << 5;
          const float f = (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;      // [-1, 1)
          h[e] = __float_as_uint(f) >> 16;
        }
        w[j] = h[0] | (h[1] << 16);
      }
      v = u4v{w[0], w[1], w[2], w[3]};
    }
    *reinterpret_cast<u4v*>(smem + o) = v;
  }
  __syncthreads();
  f4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4v{0, 0, 0, 0};
  // the next stage by LDS-DMA: 64 pieces of 1 KiB, 8 per wave (the kernel: 4 of A + 4 of B); source = a 64 KiB window of an L2-resident buffer
  int tm = 0, tn = 0;
  {
    int bid = blockIdx.x;
    const int ntiles = 256, q = ntiles / 8, xcd = bid % 8;
    bid = xcd * q + bid / 8;
    const int per_group = group_m * 16, group = bid / per_group, first_m = group * group_m;      // group_m tile rows x all 16 tile columns, column-major inside
    tm = first_m + (bid % per_group) % group_m; tn = (bid % per_group) / group_m;
  }
  const char* Ag = src;                                     // A [4096][4096] bf16, K contiguous
  const char* Bg = src + (size_t)4096 * 4352 * 2;           // B: BT = 1 [K][N] (N contiguous), BT = 0 [N][K]
  auto dma = [&](int t, int slot) {
    char* sb = smem + slot * SLOT;
    if (SRC == 0) {
      const char* g = src + (size_t)(t & 3) * SLOT;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int piece = wid * 8 + i;
        __builtin_amdgcn_global_load_lds((glb_t*)(g + piece * 1024 + lane * 16), (lds_t*)(sb + piece * 1024), 16, 0, 0);
      }
    } else {
      const int k0 = (t & 63) * 64;
#pragma unroll
      for (int i = 0; i < 4; i++) {                         // A: image [256 rows][8 chunks], chunk' = chunk ^ (row & 7)
        const int piece = wid * 4 + i, p = piece * 64 + lane;
        const int row = p >> 3, chunk = (p & 7) ^ (row & 7);
        __builtin_amdgcn_global_load_lds((glb_t*)(Ag + ((size_t)(tm * 256 + row) * ld + k0 + chunk * 8) * 2), (lds_t*)(sb + piece * 1024), 16, 0, 0);
      }
      // rot: the ORDER in which a workgroup requests the 32 pieces of B's stage (any order fills the same image): workgroups that walk the same
      // rows in the same order hammer the same few memory channels at the same time
      const int rot = rot_mode == 1 ? tn * 4 : rot_mode == 2 ? tn * 4 + tm : rot_mode == 3 ? tm * 8 : rot_mode == 4 ? tn * 2 + tm * 8 : 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int piece = (wid * 4 + i + rot) & 31, p = piece * 64 + lane;
        if (BT == 2) {                                      // B K-strided, image cut into four [64 k][64 columns] sub-images: a piece = 8 k rows x 128 bytes
          const int sub = piece >> 3, k = ((piece & 7) << 3) + (lane >> 3), c = lane & 7;
          __builtin_amdgcn_global_load_lds((glb_t*)(Bg + ((size_t)(k0 + k) * ld + tn * 256 + sub * 64 + c * 8) * 2), (lds_t*)(sb + IMG + piece * 1024), 16, 0, 0);
        } else if (BT) {                                    // B: image [64 k][32 chunks], 32-byte pairs XOR ks_swz(k)
          const int k = p >> 5, c = p & 31;
          const int col8 = ((((c >> 1) ^ ks_swz(k))) << 1) | (c & 1);
          __builtin_amdgcn_global_load_lds((glb_t*)(Bg + ((size_t)(k0 + k) * ld + tn * 256 + col8 * 8) * 2), (lds_t*)(sb + IMG + piece * 1024), 16, 0, 0);
        } else {
          const int row = p >> 3, chunk = (p & 7) ^ (row & 7);
          __builtin_amdgcn_global_load_lds((glb_t*)(Bg + ((size_t)(tn * 256 + row) * ld + k0 + chunk * 8) * 2), (lds_t*)(sb + IMG + piece * 1024), 16, 0, 0);
        }
      }
    }
  };
  u4v fa[2][8], fb[2][4];
  auto rd = [&](int set, const char* as, const char* bs, int s) {
#pragma unroll
    for (int i = 0; i < 8; i++) fa[set][i] = frag_kc(as, wr * 128 + i * 16, s, lane);
#pragma unroll
    for (int j = 0; j < 4; j++) fb[set][j] = BT ? frag_ks(bs, wc * 64 + j * 16, s, lane) : frag_kc(bs, wc * 64 + j * 16, s, lane);
  };
  auto mfma = [&](int set) {
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, fb[set][j]), __builtin_bit_cast(bf8v, fa[set][i]), acc[i][j], 0, 0, 0);
  };
  rd(0, smem, smem + IMG, 0);
  FENCE8("s_waitcnt lgkmcnt(0)", fa[0]); FENCE4("", fb[0]);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 6) {
    // the same flops per k-step from v_mfma_f32_32x32x16_bf16 (a 128 x 64 wave tile = 4 x 2 tiles of 32 x 32; 16 MFMAs of 32768 flop per 32-deep
    // k-step): does the wider instruction, which reads a quarter of the operand registers per flop, sustain more on random data?
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v c[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) c[i][j][e] = 0.f;
    for (int t = 0; t < stages; t++) {
      asm volatile("" : "+v"(fa[0][0]), "+v"(fb[0][0]));
#pragma unroll
      for (int rep = 0; rep < 2; rep++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, fb[0][j * 2 + kk]), __builtin_bit_cast(bf8v, fa[0][i * 2 + kk]), c[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][j][e & 3] += c[i][j][e];
  } else if (MODE == 7) {
    // the legacy K = 16 instruction (v_mfma_f32_16x16x16_bf16): half the flops of the K = 32 one - in half its time?  (a 16-wide tail chunk for the
    // 100-channel layers of the ResNet only pays if so).  64 MFMAs per k-step here, the same flops as the other modes
    typedef short s4v __attribute__((ext_vector_type(4)));
    for (int t = 0; t < stages; t++) {
      asm volatile("" : "+v"(fa[0][0]), "+v"(fb[0][0]));
#pragma unroll
      for (int rep = 0; rep < 2; rep++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const s4v bb = __builtin_bit_cast(s4v, u2v{fb[0][j][2 * h], fb[0][j][2 * h + 1]}), aa = __builtin_bit_cast(s4v, u2v{fa[0][i][2 * h], fa[0][i][2 * h + 1]});
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bb, aa, acc[i][j], 0, 0, 0);
            }
      }
    }
  } else if (MODE == 0) {
    for (int t = 0; t < stages; t++) {
      asm volatile("" : "+v"(fa[0][0]), "+v"(fb[0][0]));
      __builtin_amdgcn_sched_barrier(0); mfma(0); __builtin_amdgcn_sched_barrier(0); mfma(0); __builtin_amdgcn_sched_barrier(0);
    }
  } else if (MODE <= 2) {
    if (wr == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < stages; t++) {
      const char* as = smem + (t & 1) * SLOT;
      const char* bs = as + IMG;
      if (MODE == 2 && t + 1 < stages) dma(t + 1, (t + 1) & 1);
      rd(0, as, bs, 0);
      FENCE8("s_waitcnt lgkmcnt(0)", fa[0]); FENCE4("", fb[0]);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1); mfma(0); __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      rd(0, as, bs, 1);
      if (MODE == 2 && wr == 1) FENCE8("s_waitcnt vmcnt(0) lgkmcnt(0)", fa[0]);
      else FENCE8("s_waitcnt lgkmcnt(0)", fa[0]);
      FENCE4("", fb[0]);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1); mfma(0); __builtin_amdgcn_s_setprio(0);
      if (MODE == 2 && wr == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
  } else {
    // software pipeline inside every wave: set 0 holds k-step 0 of the stage, set 1 k-step 1; the reads of the NEXT k-step are requested in
    // front of this k-step's MFMAs.  One barrier per stage: after it nobody reads slot t any more (its DMA refill is issued a stage later,
    // behind that barrier), and the DMA of stage t + 1 has landed (each wave waited for its own pieces in front of the barrier)
    for (int t = 0; t < stages; t++) {
      const char* as = smem + (t & 1) * SLOT;
      const char* bs = as + IMG;
      const char* an = smem + ((t + 1) & 1) * SLOT;
      const char* bn = an + IMG;
      if (MODE >= 4 && t + 1 < stages) dma(t + 1, (t + 1) & 1);
      // k-step 0: request k-step 1's fragments, multiply set 0
      rd(1, as, bs, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma(0);
      if (MODE == 5) {
#pragma unroll
        for (int k = 0; k < 10; k++) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
      FENCE8("s_waitcnt lgkmcnt(0)", fa[1]); FENCE4("", fb[1]);
      if (MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // stage t + 1 is complete in LDS for everybody
      // k-step 1: request the next stage's k-step 0, multiply set 1
      rd(0, an, bn, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma(1);
      __builtin_amdgcn_sched_barrier(0);
      FENCE8("s_waitcnt lgkmcnt(0)", fa[0]); FENCE4("", fb[0]);
      __builtin_amdgcn_s_barrier();                        // nobody reads slot t any more: the next iteration's DMA may overwrite it
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

int g_rnd = 0;
template <int MODE, int BT, int SRC = 0>
static void run(const char* src, float* out, unsigned long long* cyc, const char* what, int ld = 4096, int group_m = 4, int rot_mode = 0) {
  extern int g_rnd; const int rnd = g_rnd;
  const int stages = 2048;
  (void)hipFuncSetAttribute((const void*)loop_kernel<MODE, BT, SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop_kernel<MODE, BT, SRC>), dim3(256), dim3(512), 2 * SLOT, 0, src, out, cyc, stages, ld, group_m, rot_mode, rnd);
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  hipError_t err = hipGetLastError();
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (int b = 0; b < 256; b++) for (int w = 0; w < 8; w++) c += (double)h[b * 8 + w];
  c /= 256.0 * 8 * stages * 2;
  const double flops = 256.0 * 8 * stages * 64.0 * 16384.0;
  printf("%s mode %d BT %d SRC %d ld %d group_m %d rot %d (%s): %.0f ticks per k-step and wave (32 MFMAs) = %.1f per MFMA and SIMD; launch %.3f ms = %.0f TFLOP/s = %.3f of 2.5 PF; %.2f ticks/ns%s\n", rnd ? "RANDOM" : "const", MODE, BT, SRC, ld, group_m, rot_mode, what, c,
         c / 32.0 / 2.0, ms, flops / (ms * 1e-3) * 1e-12, flops / (ms * 1e-3) * 1e-12 / 2500.0, c * stages * 2 / (ms * 1e6), err == hipSuccess ? "" : "  [HIP ERROR]");
}

int main() {
  float* out; unsigned long long* cyc; char* src;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  const size_t bytes = (size_t)2 * 4096 * 4352 * 2;
  (void)hipMalloc(&src, bytes);
  (void)hipMemset(src, 0x3f, bytes);
  for (int pass = 0; pass < 4; pass++) {
    g_rnd = pass & 1;
    if (g_rnd) {                                            // the global operands too
      std::vector<unsigned short> h(bytes / 2);
      unsigned x = 12345u;
      for (size_t i = 0; i < h.size(); i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; float f = (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
      (void)hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    } else (void)hipMemset(src, 0x3f, bytes);
    run<0, 0>(src, out, cyc, "MFMAs only");
    run<6, 0>(src, out, cyc, "MFMAs only, v_mfma_f32_32x32x16_bf16");
    run<7, 0>(src, out, cyc, "MFMAs only, v_mfma_f32_16x16x16_bf16 (64 per k-step)");
    run<1, 0>(src, out, cyc, "kernel schedule, B by ds_read_b128");
    run<1, 1>(src, out, cyc, "kernel schedule, B by ds_read_b64_tr_b16");
    run<2, 0, 0>(src, out, cyc, "kernel schedule + DMA of L2-resident bytes, b128");
    run<2, 1, 0>(src, out, cyc, "kernel schedule + DMA of L2-resident bytes, tr");
    run<2, 0, 1>(src, out, cyc, "kernel schedule + DMA of the 4096^3 operands, b128 (the kernel's dX form without prologue / epilogue)");
    run<2, 1, 1>(src, out, cyc, "kernel schedule + DMA of the 4096^3 operands, tr (the forward)");
    run<3, 0>(src, out, cyc, "wave-pipelined reads, one barrier per k-step, b128");
  }
  return 0;
}
