// cycles per v_mfma_f32_16x16x4_f32 / 32x32x2 on one SIMD with 1, 2 and 4 waves per SIMD (independent accumulators, operands in registers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k16(float* out, unsigned long long* cyc, int iters) {
  f4v acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = f4v{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, unsigned long long* cyc, int iters) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  const int iters = 2000;
  for (int threads : {256, 512, 1024}) {
    std::vector<unsigned long long> h(256 * 16);
    hipLaunchKernelGGL(k16<8>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize();
    hipLaunchKernelGGL(k16<8>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256 * (threads / 64); i++) c += h[i]; c /= 256 * (threads / 64);
    const double per_simd = c / (iters * 8.0 * (threads / 256));
    printf("16x16x4 f32: %4d threads (%d waves/SIMD): %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD -> %.0f FLOP/clk/CU\n", threads, threads / 256,
           c / (iters * 8.0), per_simd, 4 * 2048.0 / per_simd);
    hipLaunchKernelGGL(k32<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize();
    hipLaunchKernelGGL(k32<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
    c = 0; for (int i = 0; i < 256 * (threads / 64); i++) c += h[i]; c /= 256 * (threads / 64);
    const double per_simd32 = c / (iters * 4.0 * (threads / 256));
    printf("32x32x2 f32: %4d threads (%d waves/SIMD): %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD -> %.0f FLOP/clk/CU\n", threads, threads / 256,
           c / (iters * 4.0), per_simd32, 4 * 4096.0 / per_simd32);
  }
  return 0;
}
