// Tensor handles: creation, release, metadata, views, host<->device copies, strided copy/cast.
//
// Replaces the `aten.Tensor` instance surface lamp-sten drives (reference:
// lamp-sten/src/main/scala/lamp/STen.scala:845-1000, TensorHelpers.scala:44-345,
// device.scala:62-114,221-225).
#include <map>
#include <mutex>
#include <algorithm>
#include "tensor.h"
#include <cerrno>
#include <cstring>
#include "strided.h"
#include "../kernels/device_utils.h"

#include <climits>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace lamp {

static std::atomic<int64_t> g_live_tensors{0};

// ---- live-handle registry (aten.TensorTrace: TensorLogger.scala:13-62) -----------------------------------------------------------
// lamp's leak detector enables the trace, lists the live tensors periodically (shape, type, device, birth time; the stack trace is
// captured on the JVM side) and its tests assert that nothing is alive after a scope closes (mlp.test.scala:180-188).  The count is
// always maintained; the per-handle records only while the trace is on (a mutex + map operation per handle otherwise).
static std::atomic<int> g_trace_on{0};
static std::mutex g_trace_mu;
static std::unordered_map<const Tensor*, int64_t> g_trace;   // handle -> birth (ns, steady clock)
static inline void track(const Tensor* t) {
  g_live_tensors++;
  if (g_trace_on.load(std::memory_order_relaxed)) {
    const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace[t] = now;
  }
}
static inline void untrack(const Tensor* t) {
  g_live_tensors--;
  if (g_trace_on.load(std::memory_order_relaxed) || !g_trace.empty()) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace.erase(t);
  }
}

// Page-locked host blocks are recycled instead of returned to the driver: a hipHostFree / hipHostMalloc cycle costs ~0.2 ms per MB and,
// measured, hands back pages that the GPU then reads at anything between 19 and 53 GB/s (EXPERIMENTS (17)).  A freed block waits in a small
// cache (LAMP_PINNED_CACHE_MB, default 2048; 0: off); it is handed out again for a request it fits with at most 25 % of slack, after a
// synchronise of EVERY device this process has used (from any thread: a loader thread that never selected a device sits on device 0 while the
// step runs on device N) - asynchronous copies that still read or write the old contents (lamp_to / copy_ with non_blocking)
// finish first, which is what hipHostFree's implicit synchronisation guaranteed.  The wait happens outside the cache's lock and its result
// is checked; a failed hipHostMalloc empties the cache and tries once more.
namespace {
struct PinnedCache {
  std::mutex mu;
  std::multimap<size_t, void*> blocks;      // capacity -> pointer
  size_t cached = 0;
  size_t limit = [] { const char* e = getenv("LAMP_PINNED_CACHE_MB"); return (size_t)(e ? std::max(0, atoi(e)) : 2048) << 20; }();
};
PinnedCache& pinned_cache() { static PinnedCache* c = new PinnedCache(); return *c; }      // never destroyed: storages may outlive static destructors
}  // namespace
static void pinned_cache_flush() {
  PinnedCache& c = pinned_cache();
  std::multimap<size_t, void*> victims;
  { std::lock_guard<std::mutex> lk(c.mu); victims.swap(c.blocks); c.cached = 0; }
  for (auto& kv : victims) (void)hipHostFree(kv.second);      // (hipHostFree waits for the device itself)
}
static void* pinned_alloc(size_t bytes, size_t* capacity) {
  PinnedCache& c = pinned_cache();
  const size_t want = bytes ? bytes : 1;
  if (!allocator_capturing()) {                               // (a device synchronise would break a stream capture: allocate fresh there)
    void* p = nullptr;
    size_t cap = 0;
    {
      std::lock_guard<std::mutex> lk(c.mu);
      auto it = c.blocks.lower_bound(want);
      if (it != c.blocks.end() && it->first <= want + want / 4) {
        p = it->second; cap = it->first;
        c.cached -= cap;
        c.blocks.erase(it);
      }
    }
    if (p) {
      try {
        synchronize_all_used_devices();                       // outside the lock; throws when a device reports an error
      } catch (...) {
        (void)hipHostFree(p);                                 // the block's old readers may not be done: never hand it out
        throw;
      }
      *capacity = cap;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
  if (e != hipSuccess) {                                      // the cache may be what holds the memory: give it back and try once more
    (void)hipGetLastError();
    pinned_cache_flush();
    HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
  }
  *capacity = want;
  return p;
}
static void pinned_free(void* p, size_t capacity) {
  PinnedCache& c = pinned_cache();
  {
    std::lock_guard<std::mutex> lk(c.mu);
    if (capacity >= (1u << 16) && c.cached + capacity <= c.limit) { c.blocks.emplace(capacity, p); c.cached += capacity; return; }
  }
  (void)hipHostFree(p);
}

static Storage* new_storage(size_t bytes, int device, bool pinned = false) {
  Storage* s = new Storage();
  s->bytes = bytes;
  s->device = device;
  s->pinned = pinned;
  if (device >= 0) {
    s->ptr = device_alloc(device, bytes, &s->pool);
  } else if (pinned) {
    s->ptr = pinned_alloc(bytes, &s->pinned_capacity);
  } else {
    s->ptr = malloc(bytes ? bytes : 1);
    LAMP_CHECK(s->ptr, "host malloc of " << bytes << " bytes failed");
  }
  return s;
}
static void storage_unref(Storage* s) {
  if (!s) return;
  if (s->refs.fetch_sub(1) == 1) {
    if (s->map_base) {
      (void)munmap(s->map_base, s->map_len);
    } else if (s->owned) {
      if (s->device >= 0) device_free(s->device, s->ptr, s->pool);
      else if (s->pinned) pinned_free(s->ptr, s->pinned_capacity);
      else free(s->ptr);
    }
    delete s;
  }
}

Tensor* new_tensor(const int64_t* sizes, int ndim, int dtype, int device) {
  LAMP_CHECK(ndim >= 0 && ndim <= kMaxDims, "ndim " << ndim << " out of range");
  Tensor* t = new Tensor();
  t->ndim = ndim;
  t->dtype = dtype;
  int64_t n = 1;
  for (int i = ndim - 1; i >= 0; i--) {
    LAMP_CHECK(sizes[i] >= 0, "negative size");
    t->sizes[i] = sizes[i];
    t->strides[i] = n;
    n *= sizes[i];
  }
  try {
    t->st = new_storage((size_t)n * dtype_size(dtype), device);
  } catch (...) {
    delete t;
    throw;
  }
  track(t);
  return t;
}
Tensor* new_like(const Tensor* t) { return new_tensor(t->sizes, t->ndim, t->dtype, t->device()); }
Tensor* new_like(const Tensor* t, int dtype) { return new_tensor(t->sizes, t->ndim, dtype, t->device()); }

Tensor* new_view(const Tensor* base, const int64_t* sizes, const int64_t* strides, int ndim, int64_t offset) {
  LAMP_CHECK(ndim >= 0 && ndim <= kMaxDims, "ndim out of range");
  Tensor* t = new Tensor();
  t->st = base->st;
  base->st->refs.fetch_add(1);
  t->dtype = base->dtype;
  t->ndim = ndim;
  t->offset = offset;
  for (int i = 0; i < ndim; i++) { t->sizes[i] = sizes[i]; t->strides[i] = strides[i]; }
  track(t);
  return t;
}
Tensor* retain(const Tensor* t) { return new_view(t, t->sizes, t->strides, t->ndim, t->offset); }
void release(Tensor* t) {
  if (!t) return;
  untrack(t);
  storage_unref(t->st);
  t->st = nullptr;
  delete t;
}

// ---- strided copy with conversion ------------------------------------------------------------
template <class D, class S>
__global__ void copy_strided_kernel(D* __restrict__ dst, const S* __restrict__ src, int64_t n, IterArgs it) {
  using A = acc_t<S>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t off[2];
    iter_offsets<2>(it, i, off);
    dst[off[0]] = store_as<D>(load_as<A>(src[off[1]]));
  }
}
template <class D, class S>
__global__ void copy_contig_kernel(D* __restrict__ dst, const S* __restrict__ src, int64_t n) {
  using A = acc_t<S>;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = store_as<D>(load_as<A>(src[i]));
}

template <class D> struct conv_helper {
  template <class S> static void run(Tensor* dst, const Tensor* src, const IterSpace& it, hipStream_t st) {
    int64_t n = it.numel;
    if (n == 0) return;
    int grid = grid_for(n, 256);
    if (it.all_contiguous) {
      hipLaunchKernelGGL((copy_contig_kernel<D, S>), dim3(grid), dim3(256), 0, st, dst->ptr<D>(), src->ptr<S>(), n);
    } else {
      hipLaunchKernelGGL((copy_strided_kernel<D, S>), dim3(grid), dim3(256), 0, st, dst->ptr<D>(), src->ptr<S>(), n,
                         to_args(it));
    }
    LAMP_LAUNCH_CHECK();
  }
};

static void device_copy(Tensor* dst, const Tensor* src) {
  hipStream_t st = current_stream(dst->device());
  const Tensor* ops[2] = {dst, src};
  IterSpace it = make_iter(dst->shape(), ops, 2);
  if (it.numel == 0) return;
  if (dst->dtype == src->dtype && it.all_contiguous) {
    HIP_CHECK(hipMemcpyAsync(dst->data(), src->data(), (size_t)it.numel * dst->itemsize(), hipMemcpyDeviceToDevice, st));
    return;
  }
  LAMP_DISPATCH_ALL(dst->dtype, D, LAMP_DISPATCH_ALL(src->dtype, S, (conv_helper<D>::template run<S>(dst, src, it, st))));
}

// host side strided/converting copy (staging only; used for host<->host and as the slow path
// when a host<->device copy needs a layout or dtype change)
template <class D, class S> static void host_copy_t(Tensor* dst, const Tensor* src, const IterSpace& it) {
  using A = acc_t<S>;
  D* d = dst->ptr<D>();
  const S* s = src->ptr<S>();
  int64_t idx[kMaxDims] = {0};
  for (int64_t i = 0; i < it.numel; i++) {
    int64_t od = 0, os = 0;
    for (int k = 0; k < it.ndim; k++) { od += idx[k] * it.strides[0][k]; os += idx[k] * it.strides[1][k]; }
    d[od] = store_as<D>(load_as<A>(s[os]));
    for (int k = it.ndim - 1; k >= 0; k--) {
      if (++idx[k] < it.sizes[k]) break;
      idx[k] = 0;
    }
  }
}
template <class D> struct host_conv_helper {
  template <class S> static void run(Tensor* dst, const Tensor* src, const IterSpace& it) { host_copy_t<D, S>(dst, src, it); }
};
static void host_copy(Tensor* dst, const Tensor* src) {
  const Tensor* ops[2] = {dst, src};
  IterSpace it = make_iter(dst->shape(), ops, 2);
  if (it.numel == 0) return;
  if (dst->dtype == src->dtype && it.all_contiguous) {
    memcpy(dst->data(), src->data(), (size_t)it.numel * dst->itemsize());
    return;
  }
  LAMP_DISPATCH_ALL(dst->dtype, D, LAMP_DISPATCH_ALL(src->dtype, S, (host_conv_helper<D>::template run<S>(dst, src, it))));
}

void copy_into(Tensor* dst, const Tensor* src) {
  // src broadcasts into dst
  bool dd = dst->is_device(), sd = src->is_device();
  if (dd && sd) {
    if (dst->device() == src->device()) { device_copy(dst, src); return; }
    // peer copy: stage through a contiguous same-dtype buffer on the source device
    Hold sc(contiguous(src));
    Hold tmp(new_tensor(sc->sizes, sc->ndim, sc->dtype, dst->device()));
    // the copy runs on the DESTINATION device's stream: it must see what the source device's stream has written (src itself, or
    // the contiguous staging copy just queued there), and the staging block must not be recycled on its own stream while the copy
    // is still reading it
    hipStream_t dstream = current_stream(dst->device()), sstream = current_stream(src->device());
    {
      const int prev = current_device();
      HIP_CHECK(hipSetDevice(src->device()));
      hipEvent_t ev = nullptr;
      HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(ev, sstream));
      HIP_CHECK(hipSetDevice(dst->device()));
      HIP_CHECK(hipStreamWaitEvent(dstream, ev, 0));
      HIP_CHECK(hipEventDestroy(ev));
      HIP_CHECK(hipSetDevice(prev));
    }
    record_stream(sc.get(), dst->device(), dstream);
    HIP_CHECK(hipMemcpyPeerAsync(tmp->data(), dst->device(), sc->data(), src->device(),
                                 (size_t)sc->numel() * sc->itemsize(), dstream));
    device_copy(dst, tmp.get());
    return;
  }
  if (!dd && !sd) { host_copy(dst, src); return; }
  bool simple = dst->dtype == src->dtype && dst->is_contiguous() && src->is_contiguous() && dst->numel() == src->numel();
  if (dd) {  // host -> device
    hipStream_t st = current_stream(dst->device());
    if (simple) {
      HIP_CHECK(hipMemcpyAsync(dst->data(), src->data(), (size_t)dst->numel() * dst->itemsize(), hipMemcpyHostToDevice, st));
      if (!src->st->pinned) HIP_CHECK(hipStreamSynchronize(st));
      return;
    }
    Hold sc(contiguous(src));
    Hold tmp(new_tensor(sc->sizes, sc->ndim, sc->dtype, dst->device()));
    HIP_CHECK(hipMemcpyAsync(tmp->data(), sc->data(), (size_t)sc->numel() * sc->itemsize(), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    device_copy(dst, tmp.get());
    return;
  }
  // device -> host
  hipStream_t st = current_stream(src->device());
  if (simple) {
    HIP_CHECK(hipMemcpyAsync(dst->data(), src->data(), (size_t)dst->numel() * dst->itemsize(), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return;
  }
  Hold tmp(new_tensor(dst->sizes, dst->ndim, dst->dtype, src->device()));
  device_copy(tmp.get(), src);
  Hold htmp(new_tensor(dst->sizes, dst->ndim, dst->dtype, -1));
  HIP_CHECK(hipMemcpyAsync(htmp->data(), tmp->data(), (size_t)tmp->numel() * tmp->itemsize(), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  host_copy(dst, htmp.get());
}

Tensor* contiguous(const Tensor* t) {
  if (t->is_contiguous()) return retain(t);
  Tensor* c = new_like(t);
  try { copy_into(c, t); } catch (...) { release(c); throw; }
  return c;
}

void fill_zero(Tensor* t) {
  if (t->numel() == 0) return;
  if (t->is_contiguous()) {
    if (t->is_device()) HIP_CHECK(hipMemsetAsync(t->data(), 0, (size_t)t->numel() * t->itemsize(), current_stream(t->device())));
    else memset(t->data(), 0, (size_t)t->numel() * t->itemsize());
    return;
  }
  int64_t one[1] = {1};
  Hold z(new_tensor(one, 0, t->dtype, t->device()));
  fill_zero(z.get());
  copy_into(t, z.get());
}

// ---- fills -------------------------------------------------------------------------------------
template <class T> __global__ void arange_kernel(T* p, int64_t n, double start, double step) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = store_as<T>((acc_t<T>)(start + (double)i * step));
}
template <class T> __global__ void fill_kernel(T* p, int64_t n, T v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
static void fill_value(Tensor* t, double value) {
  if (t->numel() == 0) return;
  if (value == 0.0) { fill_zero(t); return; }
  if (!t->is_contiguous()) {
    Hold s(new_tensor(nullptr, 0, t->dtype, t->device()));
    fill_value(s.get(), value);
    copy_into(t, s.get());
    return;
  }
  int64_t n = t->numel();
  if (t->is_device()) {
    hipStream_t st = current_stream(t->device());
    LAMP_DISPATCH_ALL(t->dtype, T,
                      hipLaunchKernelGGL((fill_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, st, t->ptr<T>(), n,
                                         store_as<T>((acc_t<T>)value)));
    LAMP_LAUNCH_CHECK();
  } else {
    LAMP_DISPATCH_ALL(t->dtype, T, { T v = store_as<T>((acc_t<T>)value); T* p = t->ptr<T>(); for (int64_t i = 0; i < n; i++) p[i] = v; });
  }
}

// ---- view helpers ------------------------------------------------------------------------------
static std::vector<int64_t> infer_size(const int64_t* sizes, int ndim, int64_t numel) {
  std::vector<int64_t> out(sizes, sizes + ndim);
  int64_t known = 1;
  int infer = -1;
  for (int i = 0; i < ndim; i++) {
    if (out[i] == -1) { LAMP_CHECK(infer < 0, "only one dimension can be inferred"); infer = i; }
    else { LAMP_CHECK(out[i] >= 0, "invalid shape dimension " << out[i]); known *= out[i]; }
  }
  if (infer >= 0) {
    LAMP_CHECK(known > 0 && numel % known == 0, "shape is invalid for input of size " << numel);
    out[infer] = numel / known;
  } else {
    LAMP_CHECK(known == numel, "shape is invalid for input of size " << numel);
  }
  return out;
}

// can `t` be viewed with `shape` without a copy? (same algorithm idea as ATen's computeStride)
static bool compute_view_strides(const Tensor* t, const std::vector<int64_t>& shape, std::vector<int64_t>& out) {
  out.assign(shape.size(), 0);
  int64_t numel = t->numel();
  if (numel == 0) {
    int64_t s = 1;
    for (int i = (int)shape.size() - 1; i >= 0; i--) { out[i] = s; s *= std::max<int64_t>(shape[i], 1); }
    return true;
  }
  int view_d = (int)shape.size() - 1;
  int64_t chunk_base_stride = t->ndim ? t->strides[t->ndim - 1] : 1;
  int64_t tensor_numel = 1, view_numel = 1;
  for (int td = t->ndim - 1; td >= 0; td--) {
    tensor_numel *= t->sizes[td];
    if (td == 0 || (t->sizes[td - 1] != 1 && t->strides[td - 1] != tensor_numel * chunk_base_stride)) {
      while (view_d >= 0 && (view_numel < tensor_numel || shape[view_d] == 1)) {
        out[view_d] = view_numel * chunk_base_stride;
        view_numel *= shape[view_d];
        view_d--;
      }
      if (view_numel != tensor_numel) return false;
      if (td > 0) {
        chunk_base_stride = t->strides[td - 1];
        tensor_numel = 1;
        view_numel = 1;
      }
    }
  }
  if (view_d != -1) {
    // remaining leading size-1 dims
    for (; view_d >= 0; view_d--) {
      if (shape[view_d] != 1) return false;
      out[view_d] = 0;
    }
  }
  return true;
}

}  // namespace lamp

using namespace lamp;

#define NOT_NULL(p) LAMP_CHECK((p) != nullptr, #p " is null")

__global__ void timer_null_kernel(int* p) { if (p) *p = 0; }

extern "C" {

int lamp_live_tensor_count(int64_t* out) { *out = g_live_tensors.load(); return 0; }

// TensorTrace.enable / disable / list
int lamp_tensor_trace_enable(int on) {
  LAMP_API_BEGIN
  g_trace_on.store(on ? 1 : 0);
  if (!on) { std::lock_guard<std::mutex> lk(g_trace_mu); g_trace.clear(); }
  LAMP_API_END
}
// one record of LAMP_TRACE_RECORD int64 per live handle created since the trace was enabled:
//   [0] handle id, [1] birth (ns, monotonic clock), [2] scalar type byte, [3] device (-1 host), [4] ndim, [5..12] sizes, [13] bytes of the view
int lamp_tensor_trace_list(int64_t* records, int64_t capacity, int64_t* count) {
  LAMP_API_BEGIN
  std::lock_guard<std::mutex> lk(g_trace_mu);
  int64_t k = 0;
  for (auto& kv : g_trace) {
    if (records && k < capacity) {
      const Tensor* t = kv.first;
      int64_t* r = records + k * LAMP_TRACE_RECORD;
      r[0] = (int64_t)(intptr_t)t; r[1] = kv.second; r[2] = t->dtype; r[3] = t->device(); r[4] = t->ndim;
      for (int d = 0; d < 8; d++) r[5 + d] = d < t->ndim ? t->sizes[d] : 0;
      r[13] = t->numel() * (int64_t)t->itemsize();
    }
    k++;
  }
  *count = k;
  LAMP_API_END
}

// Tensor.from_file / tensors_from_file (STen.scala:115-194): the byte range [offset, offset + length) of `path` is mmap'ed (private,
// copy-on-write; mlock'ed when pin) and n one-dimensional host tensors are cut from it at (types[i], offsets[i], lengths[i] bytes).  Nothing is
// copied: checkpoints and data sets reach the GPU with one lamp_to per tensor straight from the page cache.
int lamp_tensors_from_file(lamp_tensor** outs, const char* path, int64_t offset, int64_t length, int pin, const int64_t* types,
                           const int64_t* offsets, const int64_t* lengths, int n) {
  LAMP_API_BEGIN
  LAMP_CHECK(path && outs, "null argument");
  LAMP_CHECK(offset % 4096 == 0, "Offset must be multiple of 4096. Got " << offset << ". Tried to create tensor from " << path << ".");
  LAMP_CHECK(length >= 0, "negative length");
  for (int i = 0; i < n; i++) {
    LAMP_CHECK(offsets[i] % 8 == 0, "Some tensor offsets within the list is not aligned to 8");
    LAMP_CHECK(offsets[i] >= 0 && lengths[i] >= 0 && offsets[i] <= length && lengths[i] <= length - offsets[i],   // no int64 overflow
               "Some tensor offset +length is out of bounds");
    LAMP_CHECK(lengths[i] % (int64_t)dtype_size((int)types[i]) == 0, "tensor " << i << ": byte length " << lengths[i] << " is not a multiple of the element size");
  }
  Storage* st = nullptr;
  if (length > 0) {
    const int fd = open(path, O_RDONLY);
    LAMP_CHECK(fd >= 0, "cannot open " << path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || offset > (int64_t)sb.st_size || length > (int64_t)sb.st_size - offset) { close(fd); LAMP_CHECK(false, path << " is shorter than offset + length = " << offset + length); }
    // private and WRITABLE (copy on write, the file is never touched), as ATen's non-shared from_file maps it: in-place host
    // arithmetic and copy_ into such a tensor work instead of faulting
    void* base = mmap(nullptr, (size_t)length, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, (off_t)offset);
    close(fd);
    LAMP_CHECK(base != MAP_FAILED, "mmap of " << path << " failed");
    if (pin && mlock(base, (size_t)length) != 0) {
      const int err = errno;
      munmap(base, (size_t)length);
      LAMP_CHECK(false, "cannot pin " << length << " bytes of " << path << " (mlock: " << strerror(err) << "; RLIMIT_MEMLOCK?)");
    }
    st = new Storage();
    st->ptr = base; st->bytes = (size_t)length; st->device = -1; st->owned = false; st->map_base = base; st->map_len = (size_t)length;
  }
  for (int i = 0; i < n; i++) {
    Tensor* t = new Tensor();
    t->dtype = (int)types[i];
    t->ndim = 1;
    t->sizes[0] = lengths[i] / (int64_t)dtype_size((int)types[i]);
    t->strides[0] = 1;
    if (st) { t->st = st; if (i > 0) st->refs.fetch_add(1); t->offset = offsets[i] / (int64_t)dtype_size((int)types[i]); }
    else { t->st = new Storage(); t->st->ptr = malloc(1); t->st->device = -1; }
    track(t);
    outs[i] = t;
  }
  if (st && n == 0) storage_unref(st);
  LAMP_API_END
}

int lamp_tensor_release(lamp_tensor* t) {
  LAMP_API_BEGIN
  release(t);
  LAMP_API_END
}
int lamp_tensor_release_all(lamp_tensor** ts, int n) {
  LAMP_API_BEGIN
  for (int i = 0; i < n; i++) release(ts[i]);
  LAMP_API_END
}
int lamp_tensor_retain(const lamp_tensor* t, lamp_tensor** out) {
  LAMP_API_BEGIN
  NOT_NULL(t);
  *out = retain(t);
  LAMP_API_END
}
int lamp_tensor_ndim(const lamp_tensor* t, int* out) { LAMP_API_BEGIN NOT_NULL(t); *out = t->ndim; LAMP_API_END }
int lamp_tensor_sizes(const lamp_tensor* t, int64_t* out) {
  LAMP_API_BEGIN NOT_NULL(t);
  for (int i = 0; i < t->ndim; i++) out[i] = t->sizes[i];
  LAMP_API_END
}
int lamp_tensor_strides(const lamp_tensor* t, int64_t* out) {
  LAMP_API_BEGIN NOT_NULL(t);
  for (int i = 0; i < t->ndim; i++) out[i] = t->strides[i];
  LAMP_API_END
}
int lamp_tensor_numel(const lamp_tensor* t, int64_t* out) { LAMP_API_BEGIN NOT_NULL(t); *out = t->numel(); LAMP_API_END }
int lamp_tensor_element_size(const lamp_tensor* t, int64_t* out) { LAMP_API_BEGIN NOT_NULL(t); *out = (int64_t)t->itemsize(); LAMP_API_END }
int lamp_tensor_scalar_type(const lamp_tensor* t, int* out) { LAMP_API_BEGIN NOT_NULL(t); *out = t->dtype; LAMP_API_END }
int lamp_tensor_device(const lamp_tensor* t, int* out) { LAMP_API_BEGIN NOT_NULL(t); *out = t->device(); LAMP_API_END }
int lamp_tensor_is_contiguous(const lamp_tensor* t, int* out) { LAMP_API_BEGIN NOT_NULL(t); *out = t->is_contiguous(); LAMP_API_END }
int lamp_tensor_is_pinned(const lamp_tensor* t, int* out) { LAMP_API_BEGIN NOT_NULL(t); *out = t->st->pinned; LAMP_API_END }
// the caller receives a MUTABLE pointer: counts as a write for the storage version
int lamp_tensor_data_ptr(const lamp_tensor* t, void** out) { LAMP_API_BEGIN NOT_NULL(t); *out = const_cast<lamp_tensor*>(t)->data(); LAMP_API_END }
int lamp_tensor_storage_id(const lamp_tensor* t, uint64_t* out) { LAMP_API_BEGIN NOT_NULL(t); *out = (uint64_t)(uintptr_t)t->st; LAMP_API_END }

int lamp_empty(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device) {
  LAMP_API_BEGIN
  dtype_size(dtype);
  *out = new_tensor(sizes, ndim, dtype, device);
  LAMP_API_END
}
int lamp_full(lamp_tensor** out, const int64_t* sizes, int ndim, double value, int dtype, int device) {
  LAMP_API_BEGIN
  dtype_size(dtype);
  Hold t(new_tensor(sizes, ndim, dtype, device));
  fill_value(t.get(), value);
  *out = t.take();
  LAMP_API_END
}
int lamp_zeros(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device) {
  return lamp_full(out, sizes, ndim, 0.0, dtype, device);
}
int lamp_ones(lamp_tensor** out, const int64_t* sizes, int ndim, int dtype, int device) {
  return lamp_full(out, sizes, ndim, 1.0, dtype, device);
}
int lamp_zeros_like(lamp_tensor** out, const lamp_tensor* t) {
  LAMP_API_BEGIN NOT_NULL(t);
  Hold r(new_like(t));
  fill_zero(r.get());
  *out = r.take();
  LAMP_API_END
}
int lamp_ones_like(lamp_tensor** out, const lamp_tensor* t) {
  LAMP_API_BEGIN NOT_NULL(t);
  Hold r(new_like(t));
  fill_value(r.get(), 1.0);
  *out = r.take();
  LAMP_API_END
}
int lamp_scalar_tensor(lamp_tensor** out, double value, int dtype, int device) {
  return lamp_full(out, nullptr, 0, value, dtype, device);
}
int lamp_scalar_tensor_l(lamp_tensor** out, int64_t value, int dtype, int device) {
  LAMP_API_BEGIN
  Hold h(new_tensor(nullptr, 0, dtype, -1));
  LAMP_DISPATCH_ALL(dtype, T, *h->ptr<T>() = store_as<T>((acc_t<T>)value));
  if (device < 0) { *out = h.take(); }
  else {
    Hold d(new_tensor(nullptr, 0, dtype, device));
    copy_into(d.get(), h.get());
    *out = d.take();
  }
  LAMP_API_END
}
int lamp_arange(lamp_tensor** out, double start, double end, double step, int dtype, int device) {
  LAMP_API_BEGIN
  LAMP_CHECK(step != 0, "step must be non-zero");
  int64_t n = (int64_t)std::ceil((end - start) / step);
  if (n < 0) n = 0;
  int64_t sz[1] = {n};
  if (device >= 0) {   // filled on the device with the same f64 expression (no staging copy: legal inside a graph capture)
    Hold d(new_tensor(sz, 1, dtype, device));
    if (n) {
      LAMP_DISPATCH_ALL(dtype, T, hipLaunchKernelGGL((arange_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, current_stream(device), d->ptr<T>(), n, start, step));
      LAMP_LAUNCH_CHECK();
    }
    *out = d.take();
    return 0;
  }
  Hold h(new_tensor(sz, 1, dtype, -1));
  LAMP_DISPATCH_ALL(dtype, T, { T* p = h->ptr<T>(); for (int64_t i = 0; i < n; i++) p[i] = store_as<T>((acc_t<T>)(start + i * step)); });
  *out = h.take();
  LAMP_API_END
}
int lamp_eye(lamp_tensor** out, int64_t n, int64_t m, int dtype, int device) {
  LAMP_API_BEGIN
  int64_t sz[2] = {n, m};
  Hold h(new_tensor(sz, 2, dtype, -1));
  fill_zero(h.get());
  LAMP_DISPATCH_ALL(dtype, T, { T* p = h->ptr<T>(); for (int64_t i = 0; i < std::min(n, m); i++) p[i * m + i] = store_as<T>((acc_t<T>)1); });
  if (device < 0) *out = h.take();
  else {
    Hold d(new_tensor(sz, 2, dtype, device));
    copy_into(d.get(), h.get());
    *out = d.take();
  }
  LAMP_API_END
}
int lamp_from_blob(lamp_tensor** out, void* data, const int64_t* sizes, const int64_t* strides, int ndim, int dtype, int device) {
  LAMP_API_BEGIN
  LAMP_CHECK(ndim >= 0 && ndim <= kMaxDims, "ndim out of range");
  Tensor* t = new Tensor();
  t->st = new Storage();
  t->st->ptr = data;
  t->st->device = device;
  t->st->owned = false;
  t->dtype = dtype;
  t->ndim = ndim;
  int64_t n = 1;
  for (int i = ndim - 1; i >= 0; i--) {
    t->sizes[i] = sizes[i];
    t->strides[i] = strides ? strides[i] : n;
    n *= sizes[i];
  }
  t->st->bytes = (size_t)n * dtype_size(dtype);
  track(t);
  *out = t;
  LAMP_API_END
}
int lamp_copy_from_host(lamp_tensor* dst, const void* src, size_t nbytes) {
  LAMP_API_BEGIN NOT_NULL(dst);
  LAMP_CHECK(nbytes == (size_t)dst->numel() * dst->itemsize(), "byte count " << nbytes << " does not match tensor " << dst->describe());
  int64_t sz[kMaxDims];
  for (int i = 0; i < dst->ndim; i++) sz[i] = dst->sizes[i];
  lamp_tensor* wrap = nullptr;
  LAMP_CHECK(lamp_from_blob(&wrap, (void*)src, sz, nullptr, dst->ndim, dst->dtype, -1) == 0, lamp_last_error());
  Hold h(wrap);
  copy_into(dst, h.get());
  if (dst->is_device()) HIP_CHECK(hipStreamSynchronize(current_stream(dst->device())));
  LAMP_API_END
}
int lamp_copy_to_host(const lamp_tensor* src, void* dst, size_t nbytes) {
  LAMP_API_BEGIN NOT_NULL(src);
  LAMP_CHECK(nbytes == (size_t)src->numel() * src->itemsize(), "byte count " << nbytes << " does not match tensor " << src->describe());
  int64_t sz[kMaxDims];
  for (int i = 0; i < src->ndim; i++) sz[i] = src->sizes[i];
  lamp_tensor* wrap = nullptr;
  LAMP_CHECK(lamp_from_blob(&wrap, dst, sz, nullptr, src->ndim, src->dtype, -1) == 0, lamp_last_error());
  Hold h(wrap);
  copy_into(h.get(), src);
  if (src->is_device()) check_device_asserts(src->device());
  LAMP_API_END
}
int lamp_clone(lamp_tensor** out, const lamp_tensor* t) {
  LAMP_API_BEGIN NOT_NULL(t);
  Hold r(new_like(t));
  copy_into(r.get(), t);
  *out = r.take();
  LAMP_API_END
}
int lamp_contiguous(lamp_tensor** out, const lamp_tensor* t) {
  LAMP_API_BEGIN NOT_NULL(t);
  *out = contiguous(t);
  LAMP_API_END
}
int lamp_copy_(lamp_tensor* dst, const lamp_tensor* src, int non_blocking) {
  LAMP_API_BEGIN NOT_NULL(dst); NOT_NULL(src);
  copy_into(dst, src);
  (void)non_blocking;
  LAMP_API_END
}
int lamp_to(lamp_tensor** out, const lamp_tensor* t, int dtype, int device, int non_blocking, int copy) {
  LAMP_API_BEGIN NOT_NULL(t);
  (void)non_blocking;
  dtype_size(dtype);
  if (!copy && dtype == t->dtype && device == t->device()) { *out = retain(t); }
  else {
    Hold r(new_tensor(t->sizes, t->ndim, dtype, device));
    copy_into(r.get(), t);
    *out = r.take();
  }
  LAMP_API_END
}
int lamp_cast(lamp_tensor** out, const lamp_tensor* t, int dtype) {
  LAMP_API_BEGIN NOT_NULL(t);
  dtype_size(dtype);
  if (dtype == t->dtype) { *out = retain(t); }
  else {
    Hold r(new_tensor(t->sizes, t->ndim, dtype, t->device()));
    copy_into(r.get(), t);
    *out = r.take();
  }
  LAMP_API_END
}
int lamp_pin_memory(lamp_tensor** out, const lamp_tensor* t) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(!t->is_device(), "pin_memory expects a host tensor");
  Tensor* r = new Tensor();
  r->ndim = t->ndim;
  r->dtype = t->dtype;
  int64_t n = 1;
  for (int i = t->ndim - 1; i >= 0; i--) { r->sizes[i] = t->sizes[i]; r->strides[i] = n; n *= t->sizes[i]; }
  try { r->st = new_storage((size_t)n * dtype_size(t->dtype), -1, true); } catch (...) { delete r; throw; }
  track(r);
  Hold h(r);
  copy_into(r, t);
  *out = h.take();
  LAMP_API_END
}
int lamp_item(const lamp_tensor* t, double* out) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(t->numel() == 1, "item() needs a one element tensor, got " << t->describe());
  int64_t one[1] = {1};
  Hold h(new_tensor(one, 0, kF64, -1));
  Hold flat(new_view(t, one, one, 0, t->offset));
  copy_into(h.get(), flat.get());
  if (t->is_device()) check_device_asserts(t->device());
  *out = *h->ptr<double>();
  LAMP_API_END
}

int lamp_fill_(lamp_tensor* t, double value) {
  LAMP_API_BEGIN NOT_NULL(t);
  fill_value(t, value);
  LAMP_API_END
}
int lamp_zero_(lamp_tensor* t) {
  LAMP_API_BEGIN NOT_NULL(t);
  fill_zero(t);
  LAMP_API_END
}

// ---- views -------------------------------------------------------------------------------------
int lamp_view(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, int ndim) {
  LAMP_API_BEGIN NOT_NULL(t);
  auto shape = infer_size(sizes, ndim, t->numel());
  std::vector<int64_t> st;
  LAMP_CHECK(compute_view_strides(t, shape, st), "view size is not compatible with input tensor's size and stride; use reshape");
  *out = new_view(t, shape.data(), st.data(), (int)shape.size(), t->offset);
  LAMP_API_END
}
int lamp_reshape(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, int ndim) {
  LAMP_API_BEGIN NOT_NULL(t);
  auto shape = infer_size(sizes, ndim, t->numel());
  std::vector<int64_t> st;
  if (compute_view_strides(t, shape, st)) { *out = new_view(t, shape.data(), st.data(), (int)shape.size(), t->offset); }
  else {
    Hold c(contiguous(t));
    LAMP_CHECK(compute_view_strides(c.get(), shape, st), "internal: reshape of a contiguous tensor failed");
    *out = new_view(c.get(), shape.data(), st.data(), (int)shape.size(), c->offset);
  }
  LAMP_API_END
}
int lamp_flatten(lamp_tensor** out, const lamp_tensor* t, int64_t start_dim, int64_t end_dim) {
  LAMP_API_BEGIN NOT_NULL(t);
  int64_t s = wrap_dim(start_dim, t->ndim), e = wrap_dim(end_dim, t->ndim);
  LAMP_CHECK(s <= e, "flatten: start_dim after end_dim");
  std::vector<int64_t> shape;
  for (int i = 0; i < s; i++) shape.push_back(t->sizes[i]);
  int64_t m = 1;
  for (int64_t i = s; i <= e && i < t->ndim; i++) m *= t->sizes[i];
  shape.push_back(m);
  for (int i = (int)e + 1; i < t->ndim; i++) shape.push_back(t->sizes[i]);
  return lamp_reshape(out, t, shape.data(), (int)shape.size());
  LAMP_API_END
}
int lamp_transpose(lamp_tensor** out, const lamp_tensor* t, int64_t dim0, int64_t dim1) {
  LAMP_API_BEGIN NOT_NULL(t);
  int64_t a = wrap_dim(dim0, t->ndim), b = wrap_dim(dim1, t->ndim);
  Tensor* v = retain(t);
  if (t->ndim > 0) { std::swap(v->sizes[a], v->sizes[b]); std::swap(v->strides[a], v->strides[b]); }
  *out = v;
  LAMP_API_END
}
int lamp_t(lamp_tensor** out, const lamp_tensor* t) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(t->ndim <= 2, "t() expects a tensor with <= 2 dimensions");
  if (t->ndim < 2) { *out = retain(t); return 0; }
  return lamp_transpose(out, t, 0, 1);
  LAMP_API_END
}
int lamp_select(lamp_tensor** out, const lamp_tensor* t, int64_t dim, int64_t index) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(t->ndim > 0, "select on a 0-dim tensor");
  int64_t d = wrap_dim(dim, t->ndim);
  int64_t n = t->sizes[d];
  LAMP_CHECK(index >= -n && index < n, "select: index " << index << " out of range for size " << n);
  if (index < 0) index += n;
  int64_t sz[kMaxDims], st[kMaxDims];
  int k = 0;
  for (int i = 0; i < t->ndim; i++) if (i != d) { sz[k] = t->sizes[i]; st[k] = t->strides[i]; k++; }
  *out = new_view(t, sz, st, k, t->offset + index * t->strides[d]);
  LAMP_API_END
}
int lamp_slice(lamp_tensor** out, const lamp_tensor* t, int64_t dim, int64_t start, int64_t end, int64_t step) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(t->ndim > 0, "slice on a 0-dim tensor");
  LAMP_CHECK(step > 0, "slice step must be positive");
  int64_t d = wrap_dim(dim, t->ndim);
  int64_t n = t->sizes[d];
  if (start < 0) start += n;
  if (end < 0) end += n;
  start = std::min(std::max<int64_t>(start, 0), n);
  end = std::min(std::max<int64_t>(end, start), n);
  Tensor* v = retain(t);
  v->offset += start * t->strides[d];
  v->sizes[d] = (end - start + step - 1) / step;
  v->strides[d] = t->strides[d] * step;
  *out = v;
  LAMP_API_END
}
int lamp_narrow(lamp_tensor** out, const lamp_tensor* t, int64_t dim, int64_t start, int64_t length) {
  LAMP_API_BEGIN NOT_NULL(t);
  int64_t d = wrap_dim(dim, t->ndim);
  if (start < 0) start += t->sizes[d];
  LAMP_CHECK(start >= 0 && length >= 0 && start + length <= t->sizes[d], "narrow: start " << start << " + length " << length << " exceeds size " << t->sizes[d]);
  return lamp_slice(out, t, d, start, start + length, 1);
  LAMP_API_END
}
// ATen's as_strided: a view of t's STORAGE with the given geometry (storage_offset in elements from the start of the storage; every
// element it can address must lie inside the storage)
int lamp_as_strided(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, const int64_t* strides, int ndim, int64_t storage_offset) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(ndim >= 0 && ndim <= kMaxDims && storage_offset >= 0, "as_strided: bad rank or offset");
  int64_t last = storage_offset;
  bool empty = false;
  for (int i = 0; i < ndim; i++) {
    LAMP_CHECK(sizes[i] >= 0 && strides[i] >= 0, "as_strided: negative size or stride");
    if (sizes[i] == 0) empty = true;
    else last += (sizes[i] - 1) * strides[i];
  }
  LAMP_CHECK(empty || (uint64_t)(last + 1) * t->itemsize() <= t->st->bytes, "as_strided: the view reaches element " << last << ", beyond the storage of " << t->describe());
  *out = new_view(t, sizes, strides, ndim, storage_offset);
  LAMP_API_END
}
int lamp_expand(lamp_tensor** out, const lamp_tensor* t, const int64_t* sizes, int ndim) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(ndim >= t->ndim && ndim <= kMaxDims, "expand: target has fewer dims than the tensor");
  int64_t sz[kMaxDims], st[kMaxDims];
  int lead = ndim - t->ndim;
  for (int i = 0; i < ndim; i++) {
    int64_t want = sizes[i];
    if (i < lead) { LAMP_CHECK(want >= 0, "expand: -1 not allowed in a leading new dimension"); sz[i] = want; st[i] = 0; continue; }
    int64_t have = t->sizes[i - lead];
    if (want == -1) want = have;
    if (have == want) { sz[i] = have; st[i] = t->strides[i - lead]; }
    else { LAMP_CHECK(have == 1, "expand: size " << have << " cannot expand to " << want << " at dim " << i); sz[i] = want; st[i] = 0; }
  }
  *out = new_view(t, sz, st, ndim, t->offset);
  LAMP_API_END
}
int lamp_expand_as(lamp_tensor** out, const lamp_tensor* t, const lamp_tensor* other) {
  LAMP_API_BEGIN NOT_NULL(t); NOT_NULL(other);
  return lamp_expand(out, t, other->sizes, other->ndim);
  LAMP_API_END
}
int lamp_squeeze(lamp_tensor** out, const lamp_tensor* t, int64_t dim) {
  LAMP_API_BEGIN NOT_NULL(t);
  int64_t sz[kMaxDims], st[kMaxDims];
  int k = 0;
  if (dim == INT64_MIN) {
    for (int i = 0; i < t->ndim; i++) if (t->sizes[i] != 1) { sz[k] = t->sizes[i]; st[k] = t->strides[i]; k++; }
  } else {
    int64_t d = wrap_dim(dim, t->ndim);
    for (int i = 0; i < t->ndim; i++) if (!(i == d && t->sizes[i] == 1)) { sz[k] = t->sizes[i]; st[k] = t->strides[i]; k++; }
  }
  *out = new_view(t, sz, st, k, t->offset);
  LAMP_API_END
}
int lamp_unsqueeze(lamp_tensor** out, const lamp_tensor* t, int64_t dim) {
  LAMP_API_BEGIN NOT_NULL(t);
  LAMP_CHECK(t->ndim < kMaxDims, "too many dims");
  int64_t d = wrap_dim(dim, t->ndim, true);
  int64_t sz[kMaxDims], st[kMaxDims];
  int k = 0;
  for (int i = 0; i <= t->ndim; i++) {
    if (i == d) { sz[k] = 1; st[k] = (i < t->ndim) ? t->sizes[i] * t->strides[i] : 1; k++; }
    if (i < t->ndim) { sz[k] = t->sizes[i]; st[k] = t->strides[i]; k++; }
  }
  *out = new_view(t, sz, st, k, t->offset);
  LAMP_API_END
}

// up to eight contiguous row blocks side by side in ONE launch (a copy per input costs a launch each: the packed projection weights of
// the language model are three 768 x 768 blocks, 4.7 us per launch against 0.3 us of data).  dir 0: out[r][off_i + c] = part_i[r][c]
// (cat along the last dimension), dir 1: part_i[r][c] = whole[r][off_i + c] (the inverse).  16-byte packets.
struct ColBlocks { char* part[8]; int64_t width[8], off[8]; int n; int64_t rows, row_bytes; };   // widths / offsets in 16-byte packets
__global__ __launch_bounds__(256) void col_blocks_kernel(ColBlocks a, char* whole, int dir) {
  const int64_t per_row = a.row_bytes / 16, total = a.rows * per_row;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / per_row, c = e - r * per_row;
    int i = 0;
    while (i + 1 < a.n && c >= a.off[i + 1]) i++;
    uint4* w = reinterpret_cast<uint4*>(whole) + e;
    uint4* p = reinterpret_cast<uint4*>(a.part[i]) + r * a.width[i] + (c - a.off[i]);
    if (dir == 0) *w = *p; else *p = *w;
  }
}
// the 2-D view (rows, bytes per row) of a device tensor cut at dimension d, or false
static bool col_block_of(const Tensor* t, int64_t d, int64_t* rows, int64_t* row_bytes) {
  if (!t->is_device() || !t->is_contiguous() || t->numel() == 0) return false;
  int64_t r = 1, w = (int64_t)t->itemsize();
  for (int k = 0; k < t->ndim; k++) { if (k < d) r *= t->sizes[k]; else w *= t->sizes[k]; }
  if (w % 16 != 0 || ((uintptr_t)t->raw() & 15) != 0) return false;
  *rows = r; *row_bytes = w;
  return true;
}
static bool cat_into_one_launch(Tensor* out, lamp_tensor* const* ts, int n, int64_t d) {
  if (n < 2 || n > 8) return false;
  ColBlocks a{};
  int64_t rows = 0, wb = 0, orows = 0, owb = 0, pos = 0;
  if (!col_block_of(out, d, &orows, &owb)) return false;
  for (int i = 0; i < n; i++) {
    if (ts[i]->dtype != out->dtype || ts[i]->device() != out->device() || !col_block_of(ts[i], d, &rows, &wb) || rows != orows) return false;
    a.part[i] = const_cast<char*>(static_cast<const char*>(static_cast<const Tensor*>(ts[i])->raw()));
    a.width[i] = wb / 16; a.off[i] = pos; pos += wb / 16;
  }
  if (pos * 16 != owb) return false;
  a.n = n; a.rows = orows; a.row_bytes = owb;
  hipLaunchKernelGGL(col_blocks_kernel, dim3(grid_for(orows * (owb / 16), 256)), dim3(256), 0, current_stream(out->device()), a, static_cast<char*>(out->raw()), 0);
  LAMP_LAUNCH_CHECK();
  return true;
}
static void cat_into(Tensor* out, lamp_tensor* const* ts, int n, int64_t d) {
  if (cat_into_one_launch(out, ts, n, d)) return;
  int64_t pos = 0;
  for (int i = 0; i < n; i++) {
    if (ts[i]->ndim == 1 && ts[i]->sizes[0] == 0 && out->ndim != 1) continue;  // legacy empty tensor
    Tensor* v = retain(out);
    Hold hv(v);
    v->offset += pos * out->strides[d];
    v->sizes[d] = ts[i]->sizes[d];
    copy_into(v, ts[i]);
    pos += ts[i]->sizes[d];
  }
}
static std::vector<int64_t> cat_shape(lamp_tensor* const* ts, int n, int64_t dim, int64_t* dout) {
  LAMP_CHECK(n > 0, "cat of an empty list");
  const Tensor* ref = nullptr;
  for (int i = 0; i < n; i++) { LAMP_CHECK(ts[i], "null tensor in list"); if (!(ts[i]->ndim == 1 && ts[i]->sizes[0] == 0)) { ref = ts[i]; break; } }
  if (!ref) ref = ts[0];
  int64_t d = wrap_dim(dim, ref->ndim);
  std::vector<int64_t> shape = ref->shape();
  int64_t total = 0;
  for (int i = 0; i < n; i++) {
    const Tensor* t = ts[i];
    if (t->ndim == 1 && t->sizes[0] == 0 && ref->ndim != 1) continue;
    LAMP_CHECK(t->ndim == ref->ndim, "cat: tensors must have the same number of dimensions");
    LAMP_CHECK(t->dtype == ref->dtype, "cat: dtype mismatch");
    for (int k = 0; k < t->ndim; k++) if (k != d) LAMP_CHECK(t->sizes[k] == ref->sizes[k], "cat: sizes must match except in dimension " << d);
    total += t->sizes[d];
  }
  shape[d] = total;
  *dout = d;
  return shape;
}
int lamp_cat(lamp_tensor** out, lamp_tensor* const* ts, int n, int64_t dim) {
  LAMP_API_BEGIN
  int64_t d;
  auto shape = cat_shape(ts, n, dim, &d);
  Hold r(new_tensor(shape, ts[0]->dtype, ts[0]->device()));
  cat_into(r.get(), ts, n, d);
  *out = r.take();
  LAMP_API_END
}
int lamp_cat_out(lamp_tensor* out, lamp_tensor* const* ts, int n, int64_t dim) {
  LAMP_API_BEGIN NOT_NULL(out);
  int64_t d;
  auto shape = cat_shape(ts, n, dim, &d);
  LAMP_CHECK(out->shape() == shape, "cat_out: output shape mismatch");
  cat_into(out, ts, n, d);
  LAMP_API_END
}
// outs[i] = x.chunk(n, dim)[i].contiguous() (equal chunks) - one launch where the blocks are whole 16-byte packets
int lamp_chunk_contiguous(lamp_tensor** outs, const lamp_tensor* x, int n, int64_t dim) {
  LAMP_API_BEGIN NOT_NULL(x);
  LAMP_CHECK(n >= 1, "chunk_contiguous: n must be positive");
  const int64_t d = wrap_dim(dim, x->ndim);
  LAMP_CHECK(x->sizes[d] % n == 0, "chunk_contiguous: dimension " << d << " of " << x->describe() << " does not divide into " << n << " equal chunks");
  const int64_t len = x->sizes[d] / n;
  std::vector<int64_t> shape = x->shape();
  shape[d] = len;
  std::vector<Hold> parts;
  for (int i = 0; i < n; i++) parts.emplace_back(new_tensor(shape, x->dtype, x->device()));
  bool done = false;
  int64_t rows = 0, wb = 0, prow = 0, pwb = 0;
  if (n >= 2 && n <= 8 && col_block_of(x, d, &rows, &wb) && col_block_of(parts[0].get(), d, &prow, &pwb) && prow == rows && pwb * n == wb) {
    ColBlocks a{};
    for (int i = 0; i < n; i++) { a.part[i] = static_cast<char*>(parts[i]->raw()); a.width[i] = pwb / 16; a.off[i] = i * (pwb / 16); }
    a.n = n; a.rows = rows; a.row_bytes = wb;
    hipLaunchKernelGGL(col_blocks_kernel, dim3(grid_for(rows * (wb / 16), 256)), dim3(256), 0, current_stream(x->device()), a,
                       const_cast<char*>(static_cast<const char*>(x->raw())), 1);
    LAMP_LAUNCH_CHECK();
    done = true;
  }
  if (!done) {
    for (int i = 0; i < n; i++) {
      lamp_tensor* v = nullptr;
      LAMP_CHECK(lamp_narrow(&v, x, d, i * len, len) == 0, lamp_last_error());
      Hold hv(v);
      copy_into(parts[i].get(), v);
    }
  }
  for (int i = 0; i < n; i++) outs[i] = parts[i].take();
  LAMP_API_END
}
int lamp_stack(lamp_tensor** out, lamp_tensor* const* ts, int n, int64_t dim) {
  LAMP_API_BEGIN
  LAMP_CHECK(n > 0, "stack of an empty list");
  int64_t d = wrap_dim(dim, ts[0]->ndim, true);
  std::vector<lamp_tensor*> us(n);
  std::vector<Hold> holds;
  holds.reserve(n);
  for (int i = 0; i < n; i++) {
    LAMP_CHECK(ts[i]->shape() == ts[0]->shape(), "stack expects each tensor to be equal size");
    lamp_tensor* u = nullptr;
    LAMP_CHECK(lamp_unsqueeze(&u, ts[i], d) == 0, lamp_last_error());
    holds.emplace_back(u);
    us[i] = u;
  }
  return lamp_cat(out, us.data(), n, d);
  LAMP_API_END
}

// Event-bracket overhead of the kernel timers: median elapsed time of (event, empty kernel, event) on the current
// stream, in microseconds.  bench.py subtracts it from every timed launch (a bracket costs several us of packet
// processing that a rocprofv3 kernel trace does not see).
int lamp_kernel_timer_calibrate(double* out_us) {
  LAMP_API_BEGIN
  NOT_NULL(out_us);
  hipStream_t st = current_stream();
  const int R = 33;
  std::vector<hipEvent_t> ev(2 * R);
  for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
  for (int w = 0; w < 4; w++) hipLaunchKernelGGL(timer_null_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
  for (int i = 0; i < R; i++) {
    HIP_CHECK(hipEventRecord(ev[2 * i], st));
    hipLaunchKernelGGL(timer_null_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr);
    HIP_CHECK(hipEventRecord(ev[2 * i + 1], st));
  }
  HIP_CHECK(hipStreamSynchronize(st));
  std::vector<float> ms(R);
  for (int i = 0; i < R; i++) HIP_CHECK(hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(ms.begin(), ms.end());
  *out_us = (double)ms[R / 2] * 1e3;
  LAMP_API_END
}

}  // extern "C"
