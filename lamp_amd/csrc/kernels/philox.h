// Philox4x32-10 counter-based generator (shared by the RNG kernels of index.hip and the kernels that draw their own samples, fused_apps.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lamp {

struct Philox {
  uint32_t key[2];
  uint32_t ctr[4];
  __device__ Philox(uint64_t seed, uint64_t subsequence, uint64_t offset) {
    key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
    ctr[0] = (uint32_t)offset; ctr[1] = (uint32_t)(offset >> 32);
    ctr[2] = (uint32_t)subsequence; ctr[3] = (uint32_t)(subsequence >> 32);
  }
  __device__ uint4 next() {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
      c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    if (++ctr[0] == 0) ++ctr[1];
    return make_uint4(c0, c1, c2, c3);
  }
};
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {  // 53-bit uniform in [0,1)
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

}  // namespace lamp
