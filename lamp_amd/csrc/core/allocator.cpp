// Caching device allocator.
//
// Why it exists: lamp's Scope releases hundreds of tensors at the end of every batch scope
// and `Variable.apply` allocates a zeroed gradient buffer for every op output (reference:
// lamp-sten/src/main/scala/lamp/Scope.scala:234-506, lamp-core/.../autograd.scala:89-96), so a
// training step is an allocation storm.  hipMalloc/hipFree are device-synchronising, so blocks
// are cached in size-class free lists keyed by (device, stream): a block is only re-used on the
// stream it was last used on, which keeps re-use stream-ordered without events.
//
// Sized for 288 GB of HBM3E: nothing is ever returned to the driver unless lamp_allocator_trim
// is called or an allocation fails (then everything cached is freed and the malloc retried).
//
// HIP-graph capture: while a capture is open, blocks come from (and return to) a private pool
// that is never handed to eager work afterwards, so addresses baked into a captured graph
// stay valid for every replay.
#include "tensor.h"

#include <map>
#include <mutex>
#include <unordered_map>

namespace lamp {

namespace {

struct Block {
  void* ptr;
  size_t size;
  hipStream_t stream;
  bool capture;  // belongs to the graph-private pool
};

struct DevicePool {
  std::mutex mu;
  // (stream, capture-pool flag) -> size -> blocks
  std::map<std::pair<hipStream_t, bool>, std::multimap<size_t, Block*>> free_lists;
  int64_t reserved = 0, in_use = 0, n_malloc = 0;
};

DevicePool g_pools[16];
thread_local bool tl_capturing = false;

size_t round_size(size_t n) {
  if (n == 0) n = 1;
  if (n <= (64u << 10)) return (n + 511) & ~size_t(511);
  if (n <= (1u << 20)) return (n + 4095) & ~size_t(4095);
  if (n <= (64u << 20)) return (n + (256u << 10) - 1) & ~size_t((256u << 10) - 1);
  return (n + (2u << 20) - 1) & ~size_t((2u << 20) - 1);
}

void free_all_cached(DevicePool& p, bool include_capture) {
  for (auto& kv : p.free_lists) {
    if (kv.first.second && !include_capture) continue;
    for (auto& sb : kv.second) {
      (void)hipFree(sb.second->ptr);
      p.reserved -= (int64_t)sb.second->size;
      delete sb.second;
    }
    kv.second.clear();
  }
}

}  // namespace

void allocator_begin_capture_pool() { tl_capturing = true; }
void allocator_end_capture_pool() { tl_capturing = false; }

void* device_alloc(int device, size_t bytes, void** cookie) {
  LAMP_CHECK(device >= 0 && device < 16, "bad device");
  DevicePool& p = g_pools[device];
  size_t sz = round_size(bytes);
  hipStream_t stream = current_stream(device);
  bool cap = tl_capturing;
  std::lock_guard<std::mutex> lk(p.mu);
  auto& fl = p.free_lists[{stream, cap}];
  auto it = fl.lower_bound(sz);
  // accept a cached block if it wastes at most 25 % (or 1 MB for small requests)
  if (it != fl.end() && (it->first <= sz + sz / 4 || it->first <= sz + (1u << 20) / 4)) {
    Block* b = it->second;
    fl.erase(it);
    p.in_use += (int64_t)b->size;
    *cookie = b;
    return b->ptr;
  }
  void* ptr = nullptr;
  int prev = current_device();
  if (prev != device) HIP_CHECK(hipSetDevice(device));
  hipError_t e = hipMalloc(&ptr, sz);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    free_all_cached(p, false);
    e = hipMalloc(&ptr, sz);
  }
  if (prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess)
    throw Error("out of HBM: hipMalloc(" + std::to_string(sz) + " bytes) failed: " + hipGetErrorString(e));
  Block* b = new Block{ptr, sz, stream, cap};
  p.reserved += (int64_t)sz;
  p.in_use += (int64_t)sz;
  p.n_malloc++;
  *cookie = b;
  return ptr;
}

void device_free(int device, void* ptr, void* cookie) {
  if (!ptr || !cookie) return;
  DevicePool& p = g_pools[device];
  Block* b = (Block*)cookie;
  std::lock_guard<std::mutex> lk(p.mu);
  p.in_use -= (int64_t)b->size;
  p.free_lists[{b->stream, b->capture}].emplace(b->size, b);
}

void allocator_stats(int device, int64_t* reserved, int64_t* in_use, int64_t* n_malloc) {
  DevicePool& p = g_pools[device];
  std::lock_guard<std::mutex> lk(p.mu);
  if (reserved) *reserved = p.reserved;
  if (in_use) *in_use = p.in_use;
  if (n_malloc) *n_malloc = p.n_malloc;
}

void allocator_trim(int device) {
  DevicePool& p = g_pools[device];
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(p.mu);
  free_all_cached(p, false);
}

}  // namespace lamp

extern "C" {
int lamp_allocator_stats(int device, int64_t* reserved_bytes, int64_t* in_use_bytes, int64_t* n_device_mallocs) {
  LAMP_API_BEGIN
  lamp::allocator_stats(device, reserved_bytes, in_use_bytes, n_device_mallocs);
  LAMP_API_END
}
int lamp_allocator_trim(int device) {
  LAMP_API_BEGIN
  lamp::allocator_trim(device);
  LAMP_API_END
}
}
