// What a device-side phase barrier costs against the kernel boundary of a replayed HIP graph ("build the narrow chain as ONE
// persistent launch per direction with device-side phase barriers ... or the barrier cost that rules it out").
// PHASES dependent phases; in each, every workgroup reads a slice of the tensor the PREVIOUS phase wrote (all of it was written by other
// workgroups: a 1-D "layer" whose every output depends on a neighbourhood that crosses workgroup slices) and writes its slice of the next one.
//   graph      one kernel per phase, captured into a HIP graph, replayed (what the step does today at small batches)
//   persist_sc one launch; between phases a counter barrier (agent-scope relaxed atomic add + spinning on an agent-scope load); the handed-off
//              bytes leave with sc0 sc1 stores and are read with sc0 sc1 loads (no fence: the per-XCD L2s are bypassed)
//   persist_f  one launch; plain loads and stores, a release fence before the arrival and an acquire fence after the wait
// for tensors of 0 (the bare synchronisation), 256 KiB, 1 MiB and 4 MiB per phase (B = 32 .. 256 activations of the CIFAR ResNet are 0.2 .. 4 MB),
// with 256 workgroups (one per CU) of 256 threads.   hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u4v ld_sc(const u4v* p) { u4v v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_sc(u4v* p, u4v v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory"); }

// one phase's work for workgroup b of nb: packets [b * per, (b + 1) * per) of dst from the packets of src shifted by a quarter of the tensor
// (always another workgroup's output) - 16 bytes per thread and round
template <int MODE>
__device__ __forceinline__ void phase_work(const u4v* src, u4v* dst, int packets, int b, int nb, int tid) {
  const int per = packets / nb;
  for (int i = tid; i < per; i += 256) {
    const int o = b * per + i;
    const int s = (o + packets / 4 + 7) % packets;
    u4v v = MODE == 1 ? ld_sc(src + s) : src[s];
    v[0] += 1u; v[1] ^= v[0];
    if (MODE == 1) st_sc(dst + o, v); else dst[o] = v;
  }
}

__global__ __launch_bounds__(256) void phase_kernel(const u4v* src, u4v* dst, int packets) { phase_work<0>(src, dst, packets, blockIdx.x, gridDim.x, threadIdx.x); }

// MODE 3: as 1, but no shared counter (256 workgroups adding to ONE address serialise at the memory side): every workgroup stores the phase number
// into its own 4-byte flag and all 256 threads of a workgroup poll one flag each (one 1 KiB sc1 load per wave and round) until every flag shows
// the phase - the exchange form of bn_bwd_fused_kernel (norm.hip) with 256 participants
template <int MODE>   // 1: sc loads / stores, no fences; 2: plain + fences
__global__ __launch_bounds__(256) void persistent_kernel(u4v* a, u4v* b, int packets, int phases, unsigned* counter, unsigned long long* stamps) {
  const int tid = threadIdx.x, nb = gridDim.x;
  unsigned long long t_arrive = 0, t_leave = 0, acc_wait = 0;
  for (int p = 0; p < phases; p++) {
    phase_work<(MODE == 3 ? 1 : MODE)>((p & 1) ? b : a, (p & 1) ? a : b, packets, blockIdx.x, nb, tid);
    // ---- barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 3) {
      if (tid == 0) { t_arrive = __builtin_amdgcn_s_memtime(); __hip_atomic_store(counter + 16 + blockIdx.x, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      const unsigned want = (unsigned)(p + 1);
      bool ok = false;
      while (!ok) {
        const unsigned v = tid < nb ? __hip_atomic_load(counter + 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
        ok = __syncthreads_and(v >= want);
      }
      if (tid == 0) { t_leave = __builtin_amdgcn_s_memtime(); acc_wait += t_leave - t_arrive; }
      continue;
    }
    if (tid == 0) {
      t_arrive = __builtin_amdgcn_s_memtime();
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(p + 1) * (unsigned)nb;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      t_leave = __builtin_amdgcn_s_memtime();
      acc_wait += t_leave - t_arrive;
    }
    __syncthreads();
  }
  if (tid == 0) stamps[blockIdx.x] = acc_wait;
}

static float time_it(hipStream_t st, int reps, auto&& f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ms;
  for (int r = 0; r < reps; r++) {
    (void)hipEventRecord(e0, st); f(); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float m = 0; (void)hipEventElapsedTime(&m, e0, e1); ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main() {
  const int PHASES = 50, NB = 256;
  hipStream_t st; (void)hipStreamCreate(&st);
  u4v *a, *b; unsigned* counter; unsigned long long* stamps;
  (void)hipMalloc(&a, 8 << 20); (void)hipMalloc(&b, 8 << 20); (void)hipMalloc(&counter, 4096); (void)hipMalloc(&stamps, NB * 8);
  (void)hipMemset(a, 1, 8 << 20); (void)hipMemset(b, 2, 8 << 20);
  for (int bytes : {0, 256 << 10, 1 << 20, 4 << 20}) {
    const int packets = std::max(bytes / 16, 0);
    // ---- graph of PHASES kernels
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int p = 0; p < PHASES; p++) hipLaunchKernelGGL(phase_kernel, dim3(NB), dim3(256), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, packets);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
    const float t_graph = time_it(st, 21, [&] { (void)hipGraphLaunch(ge, st); });
    // ---- eager launches
    const float t_eager = time_it(st, 21, [&] { for (int p = 0; p < PHASES; p++) hipLaunchKernelGGL(phase_kernel, dim3(NB), dim3(256), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, packets); });
    // ---- persistent
    float t_p[4] = {0, 0, 0, 0}; double wait_ticks[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; mode++) {
      auto launch = [&] {
        (void)hipMemsetAsync(counter, 0, 4096, st);
        if (mode == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(NB), dim3(256), 0, st, a, b, packets, PHASES, counter, stamps);
        else if (mode == 2) hipLaunchKernelGGL(persistent_kernel<2>, dim3(NB), dim3(256), 0, st, a, b, packets, PHASES, counter, stamps);
        else hipLaunchKernelGGL(persistent_kernel<3>, dim3(NB), dim3(256), 0, st, a, b, packets, PHASES, counter, stamps);
      };
      launch(); (void)hipStreamSynchronize(st);
      t_p[mode] = time_it(st, 21, launch);
      std::vector<unsigned long long> h(NB);
      (void)hipMemcpy(h.data(), stamps, NB * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += (double)v;
      wait_ticks[mode] = s / NB / PHASES;
    }
    printf("%7d bytes per phase, %d phases, %d workgroups: graph replay %.2f us per phase, eager launches %.2f, persistent sc1 %.2f (in the barrier: %.0f ticks per phase and workgroup), "
           "persistent with fences %.2f (%.0f ticks), persistent sc1 with per-workgroup flags %.2f (%.0f ticks)\n", bytes, PHASES, NB, t_graph * 1e3 / PHASES, t_eager * 1e3 / PHASES, t_p[1] * 1e3 / PHASES, wait_ticks[1], t_p[2] * 1e3 / PHASES, wait_ticks[2], t_p[3] * 1e3 / PHASES, wait_ticks[3]);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  }
  return 0;
}
