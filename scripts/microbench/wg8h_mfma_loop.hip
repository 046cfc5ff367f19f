// The k-step of ig_wgrad8h_kernel<2> (conv_igemm.hip, rd / mul) as a bare loop: 18 v_mfma_f32_16x16x32_bf16 on 9 x 2 accumulators, with
//   MODE 0  all operands resident, no vector work
//   MODE 1  + the shifted X fragments cut in registers (3 perm + 2 shifts + 3 moves per half), spread between the MFMAs as the kernel does
//   MODE 2  + the five fragment reads of the NEXT k-step (ds_read_b128, issued before this k-step's MFMAs)
//   MODE 3  + a workgroup barrier every fourth k-step (the kernel's image pair)
// at one and two waves per SIMD (256 / 512 threads, one workgroup per CU).  Prints s_memtime ticks per k-step and wave: 288 = the matrix
// pipe's share of one wave's k-step, 576 per SIMD at two waves.   hipcc -O3 --offload-arch=gfx950 wg8h_mfma_loop.hip -o wg8h_mfma_loop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

struct Frag { u4v fa[3]; u4v xc[2]; };

template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(float* out, unsigned long long* cyc, int ksteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int o = tid * 16; o < 64 * 1024; o += blockDim.x * 16) *reinterpret_cast<u4v*>(smem + o) = u4v{0x3f803f80u + tid, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
  __syncthreads();
  f4v acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; t++) { acc[t][0] = f4v{0, 0, 0, 0}; acc[t][1] = f4v{0, 0, 0, 0}; }
  auto rd = [&](Frag& f, int it) {
    const char* st = smem + (it & 3) * 16384;
#pragma unroll
    for (int r = 0; r < 3; r++) f.fa[r] = *reinterpret_cast<const u4v*>(st + (wid * 16 + (lane & 15)) * 128 + ((((lane >> 4) + r) ^ (lane & 7)) << 4));
#pragma unroll
    for (int i = 0; i < 2; i++) f.xc[i] = *reinterpret_cast<const u4v*>(st + 8192 + (i * 16 + (lane & 15)) * 176 + ((lane >> 4) + 1) * 16);
  };
  auto mul = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bf8v fc = __builtin_bit_cast(bf8v, f.xc[i]);
#pragma unroll
      for (int r = 0; r < 3; r++) acc[r * 3 + 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc, __builtin_bit_cast(bf8v, f.fa[r]), acc[r * 3 + 1][i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const u4v c = f.xc[i];
#pragma unroll
      for (int s_ = 0; s_ < 3; s_ += 2) {
        u4v sh;
        if (MODE == 0) sh = c;
        else sh = s_ == 0 ? u4v{c[0] << 16, (c[1] << 16) | (c[0] >> 16), (c[2] << 16) | (c[1] >> 16), (c[3] << 16) | (c[2] >> 16)}
                          : u4v{(c[0] >> 16) | (c[1] << 16), (c[1] >> 16) | (c[2] << 16), (c[2] >> 16) | (c[3] << 16), c[3] >> 16};
        const bf8v fb = __builtin_bit_cast(bf8v, sh);
#pragma unroll
        for (int r = 0; r < 3; r++) acc[r * 3 + s_][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, __builtin_bit_cast(bf8v, f.fa[r]), acc[r * 3 + s_][i], 0, 0, 0);
      }
    }
    if (MODE >= 1) {
#pragma unroll
      for (int k = 0; k < 9; k++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
    }
  };
  Frag f0, f1;
  rd(f0, 0); rd(f1, 1);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < ksteps; it += 2) {
    if (MODE >= 2) {
      __builtin_amdgcn_sched_barrier(0); mul(f0); __builtin_amdgcn_sched_barrier(0); rd(f0, it + 2); __builtin_amdgcn_sched_barrier(0);
      mul(f1); __builtin_amdgcn_sched_barrier(0); rd(f1, it + 3); __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 3 && (it & 2)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    } else {
      // (the empty asm makes the fragments "new" every k-step: the shifts are recomputed, as they are on fresh LDS data)
      asm volatile("" : "+v"(f0.xc[0]), "+v"(f0.xc[1]), "+v"(f1.xc[0]), "+v"(f1.xc[1]));
      __builtin_amdgcn_sched_barrier(0); mul(f0); __builtin_amdgcn_sched_barrier(0); mul(f1); __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int t = 0; t < 9; t++) s += acc[t][0][0] + acc[t][0][1] + acc[t][0][2] + acc[t][0][3] + acc[t][1][0] + acc[t][1][1] + acc[t][1][2] + acc[t][1][3];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE>
static void run(float* out, unsigned long long* cyc, int threads, const char* what) {
  const int ksteps = 4000, waves = threads / 64;
  (void)hipFuncSetAttribute((const void*)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(256), dim3(threads), 64 * 1024, 0, out, cyc, ksteps);
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) c += (double)h[b * 8 + w];
  c /= 256.0 * waves * ksteps;
  // wall clock of the launch (events; includes ~10 us of prologue): MFMA rate of the chip and the tick rate it implies
  const double flops = 256.0 * waves * ksteps * 18.0 * 16384.0;
  printf("mode %d (%s), %d wave(s) per SIMD: %.0f ticks per k-step and wave = %.1f per MFMA and SIMD; launch %.3f ms = %.0f TFLOP/s, %.2f ticks/ns\n", MODE, what,
         waves / 4, c, c / 18.0 / (waves / 4), ms, flops / (ms * 1e-3) * 1e-12, c * ksteps / (ms * 1e6));
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  for (int threads : {256, 512}) {
    run<0>(out, cyc, threads, "MFMAs only");
    run<1>(out, cyc, threads, "+ shifts in registers");
    run<2>(out, cyc, threads, "+ next k-step's five LDS reads");
    run<3>(out, cyc, threads, "+ barrier every 4 k-steps");
  }
  return 0;
}
