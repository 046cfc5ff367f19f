// LDS read throughput vs alignment on gfx950: each lane reads a 16-byte window at byte offset lane*stride + shift.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s8v __attribute__((ext_vector_type(8)));
typedef s8v __attribute__((aligned(2))) s8v_u;
typedef int i2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef i2v __attribute__((aligned(2))) i2v_u;
typedef int __attribute__((aligned(2))) i1_u;

template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int stride, int shift, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<int*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned base = (unsigned)(size_t)(lds) + (lane & 15) * stride + (lane >> 4) * 2048 + shift;
  int acc = 0;
  for (int it = 0; it < iters; it++) {
    const unsigned q = base + ((it & 7) << 4);
    if (MODE == 0) {
      i4v v0, v1, v2, v3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:256\n ds_read_b128 %2, %4 offset:512\n ds_read_b128 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(q) : "memory");
      acc += v0[0] + v1[1] + v2[2] + v3[3];
    }
    if (MODE == 1) {
      i2v v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:256\n ds_read_b64 %3, %8 offset:264\n"
                   "ds_read_b64 %4, %8 offset:512\n ds_read_b64 %5, %8 offset:520\n ds_read_b64 %6, %8 offset:768\n ds_read_b64 %7, %8 offset:776\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(q) : "memory");
      acc += v0[0] + v1[1] + v2[0] + v3[1] + v4[0] + v5[1] + v6[0] + v7[1];
    }
    if (MODE == 2) {
      int v[16];
      asm volatile("ds_read_b32 %0, %16\n ds_read_b32 %1, %16 offset:4\n ds_read_b32 %2, %16 offset:8\n ds_read_b32 %3, %16 offset:12\n"
                   "ds_read_b32 %4, %16 offset:256\n ds_read_b32 %5, %16 offset:260\n ds_read_b32 %6, %16 offset:264\n ds_read_b32 %7, %16 offset:268\n"
                   "ds_read_b32 %8, %16 offset:512\n ds_read_b32 %9, %16 offset:516\n ds_read_b32 %10, %16 offset:520\n ds_read_b32 %11, %16 offset:524\n"
                   "ds_read_b32 %12, %16 offset:768\n ds_read_b32 %13, %16 offset:772\n ds_read_b32 %14, %16 offset:776\n ds_read_b32 %15, %16 offset:780\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]), "=v"(v[8]), "=v"(v[9]), "=v"(v[10]),
                     "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]) : "v"(q) : "memory");
#pragma unroll
      for (int j = 0; j < 16; j++) acc += v[j];
    }
    if (MODE == 3) {
      int v[8];
      asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:2\n ds_read_u16 %2, %8 offset:4\n ds_read_u16 %3, %8 offset:6\n"
                   "ds_read_u16 %4, %8 offset:8\n ds_read_u16 %5, %8 offset:10\n ds_read_u16 %6, %8 offset:12\n ds_read_u16 %7, %8 offset:14\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(q) : "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) acc += v[j];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const char* name, int* out, int stride, int shift) {
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(out, stride, shift, 10);
  hipEventRecord(a);
  k<MODE><<<blocks, 256>>>(out, stride, shift, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * 256 * iters * (MODE == 3 ? 1 : 4) * 16;
  printf("%-10s stride %3d shift %2d : %8.3f ms  %8.1f GB/s LDS window bytes\n", name, stride, shift, ms, bytes / ms * 1e-6);
}
int main() {
  int* out; hipMalloc(&out, 1024 * 256 * 4 * 4);
  for (int stride : {16, 2, 4}) for (int shift : {0, 2, 4, 8}) {
    if (stride == 16 || shift == 0) {
      run<0>("b128", out, stride, shift); run<1>("2xb64", out, stride, shift); run<2>("4xb32", out, stride, shift); run<3>("8xu16", out, stride, shift);
    }
  }
  return 0;
}
