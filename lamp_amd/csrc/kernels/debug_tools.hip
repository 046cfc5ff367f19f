// Test tools behind the C ABI (no arithmetic of the path): a kernel that takes compute units away for a chosen time, so that the
// kernels whose workgroups wait for each other can be exercised on a device they do not have to themselves.
#include "device_utils.h"
#include "../core/tensor.h"
#include "lamp_hip.h"

namespace lamp {

// one wave per workgroup, spinning on the constant 100 MHz clock.  The training step's big kernels fill a CU's register file, so a CU
// that hosts one of these waves cannot take one of their workgroups until the wave leaves.
__global__ __launch_bounds__(64) void occupy_cu_kernel(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned n = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); n++; }
  if (sink && n == 0xffffffffu) *sink = n;                  // never true: keeps the loop
}

}  // namespace lamp

using namespace lamp;

extern "C" int lamp_debug_occupy_cus(int workgroups, double microseconds, lamp_stream* s) {
  LAMP_API_BEGIN
  LAMP_CHECK(workgroups >= 1 && workgroups <= 4096, "lamp_debug_occupy_cus: 1 ... 4096 workgroups");
  LAMP_CHECK(microseconds >= 0 && microseconds <= 5e6, "lamp_debug_occupy_cus: at most five seconds");
  hipStream_t st = current_stream();
  if (s) { void* native = nullptr; LAMP_CHECK(lamp_stream_native(s, &native) == 0, "bad stream handle"); st = (hipStream_t)native; }
  hipLaunchKernelGGL(occupy_cu_kernel, dim3((unsigned)workgroups), dim3(64), 0, st, (unsigned long long)(microseconds * 100.0), (unsigned*)nullptr);
  LAMP_LAUNCH_CHECK();
  LAMP_API_END
}
