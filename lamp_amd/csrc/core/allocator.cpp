// Caching device allocator.
//
// Why it exists: lamp's Scope releases hundreds of tensors at the end of every batch scope
// and `Variable.apply` allocates a zeroed gradient buffer for every op output (reference:
// lamp-sten/src/main/scala/lamp/Scope.scala:234-506, lamp-core/.../autograd.scala:89-96), so a
// training step is an allocation storm.  hipMalloc/hipFree are device-synchronising, so blocks
// are cached in size-class free lists keyed by (device, stream): a block is only re-used on the
// stream it was allocated on, which keeps re-use stream-ordered without events.
//
// A block that was also used on ANOTHER stream (lamp's `withOtherStream`, device.scala:199-213; the gradient exchange of the
// data-parallel step; peer copies issued on the destination device's stream) carries that stream in `used_on`
// (device_record_stream): when it is freed it waits in a pending list until an event recorded on every such stream at the time of
// the free has completed, and only then returns to its free list (polled at the next allocation - no host synchronisation).
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
  std::vector<std::pair<int, hipStream_t>> used_on;   // (device of the stream, stream) other than `stream`
};
struct Pending { Block* b; std::vector<hipEvent_t> events; };

struct DevicePool {
  std::mutex mu;
  // (stream, capture-pool flag) -> size -> blocks
  std::map<std::pair<hipStream_t, bool>, std::multimap<size_t, Block*>> free_lists;
  std::vector<Pending> pending;   // freed, but another stream may still be using them
  int64_t reserved = 0, in_use = 0, n_malloc = 0, n_deferred = 0;
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

// blocks whose cross-stream users have finished go back to their free lists (caller holds p.mu)
void reap_pending(DevicePool& p, bool wait) {
  for (size_t i = 0; i < p.pending.size();) {
    Pending& pe = p.pending[i];
    bool done = true;
    for (hipEvent_t e : pe.events) {
      if (wait) (void)hipEventSynchronize(e);
      else if (hipEventQuery(e) != hipSuccess) { (void)hipGetLastError(); done = false; break; }
    }
    if (!done) { i++; continue; }
    for (hipEvent_t e : pe.events) (void)hipEventDestroy(e);
    p.free_lists[{pe.b->stream, pe.b->capture}].emplace(pe.b->size, pe.b);
    p.pending[i] = std::move(p.pending.back());
    p.pending.pop_back();
  }
}

}  // namespace

void allocator_begin_capture_pool() { tl_capturing = true; }
void allocator_end_capture_pool() { tl_capturing = false; }
bool allocator_capturing() { return tl_capturing; }

void* device_alloc(int device, size_t bytes, void** cookie) {
  LAMP_CHECK(device >= 0 && device < 16, "bad device");
  DevicePool& p = g_pools[device];
  size_t sz = round_size(bytes);
  hipStream_t stream = current_stream(device);
  bool cap = tl_capturing;
  std::lock_guard<std::mutex> lk(p.mu);
  if (!p.pending.empty()) reap_pending(p, false);
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
    reap_pending(p, true);
    free_all_cached(p, false);
    e = hipMalloc(&ptr, sz);
  }
  if (prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess)
    throw Error("out of HBM: hipMalloc(" + std::to_string(sz) + " bytes) failed: " + hipGetErrorString(e));
  Block* b = new Block{ptr, sz, stream, cap, {}};
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
  if (!b->used_on.empty() && !b->capture) {
    // the other streams may still have work on this block in flight: mark "now" on each of them
    Pending pe{b, {}};
    int prev = -1;
    (void)hipGetDevice(&prev);
    int cur = prev;
    bool ok = true;
    for (auto& ds : b->used_on) {
      if (ds.first != cur) { if (hipSetDevice(ds.first) != hipSuccess) { ok = false; break; } cur = ds.first; }
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, ds.second) != hipSuccess) { ok = false; if (ev) (void)hipEventDestroy(ev); break; }
      pe.events.push_back(ev);
    }
    if (cur != prev && prev >= 0) (void)hipSetDevice(prev);
    b->used_on.clear();
    if (!ok) {                      // could not fence: fall back to the one thing that is always safe
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      for (hipEvent_t e : pe.events) (void)hipEventDestroy(e);
      p.free_lists[{b->stream, b->capture}].emplace(b->size, b);
      return;
    }
    p.n_deferred++;
    p.pending.push_back(std::move(pe));
    return;
  }
  b->used_on.clear();
  p.free_lists[{b->stream, b->capture}].emplace(b->size, b);
}

// the block behind `cookie` (device `device`) is read or written by work queued on `stream` of `stream_device`
void device_record_stream(int device, void* cookie, int stream_device, hipStream_t stream) {
  if (!cookie) return;
  DevicePool& p = g_pools[device];
  Block* b = (Block*)cookie;
  std::lock_guard<std::mutex> lk(p.mu);
  if (stream_device == device && stream == b->stream) return;
  for (auto& ds : b->used_on) if (ds.first == stream_device && ds.second == stream) return;
  b->used_on.emplace_back(stream_device, stream);
}
int64_t allocator_deferred_frees(int device) {
  DevicePool& p = g_pools[device];
  std::lock_guard<std::mutex> lk(p.mu);
  return p.n_deferred;
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
  reap_pending(p, true);
  free_all_cached(p, false);
}

}  // namespace lamp

extern "C" {
int lamp_allocator_stats(int device, int64_t* reserved_bytes, int64_t* in_use_bytes, int64_t* n_device_mallocs) {
  LAMP_API_BEGIN
  lamp::allocator_stats(device, reserved_bytes, in_use_bytes, n_device_mallocs);
  LAMP_API_END
}
int lamp_allocator_deferred_frees(int device, int64_t* out) {
  LAMP_API_BEGIN
  *out = lamp::allocator_deferred_frees(device);
  LAMP_API_END
}
int lamp_allocator_trim(int device) {
  LAMP_API_BEGIN
  lamp::allocator_trim(device);
  LAMP_API_END
}
}
