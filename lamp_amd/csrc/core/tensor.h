// Internal C++ view of the runtime behind include/lamp_hip.h.
//
// A lamp_tensor is what lamp's `aten.Tensor` JVM object points at
// (reference: lamp-sten/src/main/scala/lamp/STen.scala:845 wraps exactly one
// such native handle): a strided view (sizes/strides/offset, row-major by
// default, NCHW for images) over a reference-counted storage block that lives
// either in HBM (hipMalloc'ed through the caching allocator) or in host memory
// (lamp's CPU device: staging, views, casts, element-wise / reduction / mm arithmetic where the tensor lives - see the note on device
// types in include/lamp_hip.h; there is no CPU fallback for GPU tensors).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <sstream>

#include "../../../include/lamp_hip.h"

namespace lamp {

constexpr int kMaxDims = LAMP_MAX_DIMS;

// ATen scalar-type bytes, the values lamp passes around (STen.scala:726-731).
enum DType : int {
  kU8 = 0, kI8 = 1, kI16 = 2, kI32 = 3, kI64 = 4, kF16 = 5, kF32 = 6, kF64 = 7, kBool = 11, kBF16 = 15
};

inline size_t dtype_size(int dt) {
  switch (dt) {
    case kU8: case kI8: case kBool: return 1;
    case kI16: case kF16: case kBF16: return 2;
    case kI32: case kF32: return 4;
    case kI64: case kF64: return 8;
  }
  throw std::runtime_error("unknown scalar type byte " + std::to_string(dt));
}
inline const char* dtype_name(int dt) {
  switch (dt) {
    case kU8: return "u8"; case kI8: return "i8"; case kI16: return "i16"; case kI32: return "i32";
    case kI64: return "i64"; case kF16: return "f16"; case kF32: return "f32"; case kF64: return "f64";
    case kBool: return "bool"; case kBF16: return "bf16";
  }
  return "?";
}
inline bool is_float(int dt) { return dt == kF32 || dt == kF64 || dt == kBF16 || dt == kF16; }

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// a GPU-only operator met a host tensor (check_device_tensor): status LAMP_STATUS_HOST_TENSOR at the C ABI, where the generated
// staging layer (core/host_staging.cpp) runs the operator on GPU copies when ALL tensor arguments are host tensors
struct HostTensorError : Error {
  using Error::Error;
};

#define LAMP_CHECK(cond, ...)                                                     \
  do {                                                                            \
    if (!(cond)) {                                                                \
      std::ostringstream _os;                                                     \
      _os << __VA_ARGS__;                                                         \
      throw ::lamp::Error(std::string(__func__) + ": " + _os.str());              \
    }                                                                             \
  } while (0)

#define HIP_CHECK(expr)                                                           \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      throw ::lamp::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e)); \
    }                                                                             \
  } while (0)

void set_last_error(const std::string& msg);

// every extern "C" body is wrapped in these: exceptions never cross the C ABI
// (the JNI shim turns a non-zero status + lamp_last_error() into a JVM exception,
// which is what lamp's Scope relies on - Scope.scala:394-421).
#define LAMP_STATUS_HOST_TENSOR 2   /* lamp::HostTensorError: the generated staging layer (core/host_staging.cpp) retries on the GPU */
#define LAMP_API_BEGIN try {
#define LAMP_API_END                                                              \
  return 0;                                                                       \
  }                                                                               \
  catch (const ::lamp::HostTensorError& e) {                                      \
    ::lamp::set_last_error(e.what());                                             \
    return LAMP_STATUS_HOST_TENSOR;                                               \
  }                                                                               \
  catch (const std::exception& e) {                                               \
    ::lamp::set_last_error(e.what());                                             \
    return 1;                                                                     \
  }                                                                               \
  catch (...) {                                                                   \
    ::lamp::set_last_error("unknown C++ exception");                              \
    return 1;                                                                     \
  }

// ---------------------------------------------------------------------------------------------
struct Storage {
  void* ptr = nullptr;
  size_t bytes = 0;
  int device = -1;          // -1 = host, >=0 = HIP device ordinal
  bool pinned = false;      // host only
  bool owned = true;        // false: wraps caller memory (lamp_tensor_from_blob)
  void* pool = nullptr;     // allocator block cookie
  void* map_base = nullptr; // host storage that is a window of an mmap'ed file (Tensor.from_file): base and length of the mapping,
  size_t map_len = 0;       // shared by every tensor cut from it (they share this Storage); munmap'ed with the last handle
  size_t pinned_capacity = 0;   // pinned host storage: the size of the page-locked block behind it (blocks are recycled, tensor.hip)
  std::atomic<int> refs{1};
  // Bumped whenever a MUTABLE pointer into the storage is handed out (data() / ptr<T>() on a non-const handle): every kernel
  // that writes a tensor has to go through one of those, so "version unchanged" proves "contents unchanged" - the packed-weight
  // cache of the implicit-GEMM convolutions relies on it.  Spurious bumps (a mutable pointer used for reading) only cost a miss.
  std::atomic<uint64_t> version{0};
  uint64_t uid = next_uid();    // never reused (a freed Storage's address can be)
  static uint64_t next_uid() { static std::atomic<uint64_t> c{1}; return c.fetch_add(1, std::memory_order_relaxed); }
  // Set while a deferred kernel still has to produce the contents (kernels/wgrad_reduce.hip: the reductions of a backward pass's
  // weight-gradient partial sums run batched in one launch).  Every pointer into the storage goes through lamp_tensor::raw(), which
  // resolves the deferral first - so nobody can observe the tensor before it is complete.
  std::atomic<uint32_t> pending{0};
  // a strided filter's contiguous copy made inside a convolution entry point (kernels/conv.hip): it dies with the call, so no packed-weight
  // cache keeps an image of it (an entry under its never-reused uid could only be evicted, and a pair image would outlive one of its sources)
  bool scratch = false;
};
void resolve_deferred(Storage* st);   // runs every pending deferred kernel now (on the streams they were registered on)
void flush_deferred();                // the same, called at the natural batching points (end of backprop, lamp_flush_deferred)

}  // namespace lamp

// The opaque handle of the C ABI.
struct lamp_tensor {
  lamp::Storage* st = nullptr;
  int64_t offset = 0;  // in elements
  int ndim = 0;
  int64_t sizes[lamp::kMaxDims] = {0};
  int64_t strides[lamp::kMaxDims] = {0};
  int dtype = lamp::kF32;

  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < ndim; i++) n *= sizes[i];
    return n;
  }
  int device() const { return st ? st->device : -1; }
  bool is_device() const { return st && st->device >= 0; }
  size_t itemsize() const { return lamp::dtype_size(dtype); }
  void* raw() const {                                                                            // no version bump: read-only uses
    if (!st) return nullptr;
    if (st->pending.load(std::memory_order_acquire)) lamp::resolve_deferred(st);
    return (char*)st->ptr + offset * (int64_t)itemsize();
  }
  void* data() { if (st) st->version.fetch_add(1, std::memory_order_relaxed); return raw(); }
  const void* data() const { return raw(); }
  template <class T> T* ptr() { return (T*)data(); }
  template <class T> const T* ptr() const { return (const T*)raw(); }
  bool is_contiguous() const {
    int64_t expect = 1;
    for (int i = ndim - 1; i >= 0; i--) {
      if (sizes[i] == 1) continue;
      if (strides[i] != expect) return false;
      expect *= sizes[i];
    }
    return true;
  }
  std::vector<int64_t> shape() const { return std::vector<int64_t>(sizes, sizes + ndim); }
  std::string describe() const {
    std::ostringstream os;
    os << lamp::dtype_name(dtype) << "[";
    for (int i = 0; i < ndim; i++) os << (i ? "," : "") << sizes[i];
    os << "]@" << (is_device() ? "gpu" : "cpu");
    return os.str();
  }
};

namespace lamp {

using Tensor = lamp_tensor;

// ---- allocator (core/allocator.cpp) ----
void* device_alloc(int device, size_t bytes, void** cookie);
void device_free(int device, void* ptr, void* cookie);
void device_record_stream(int device, void* cookie, int stream_device, hipStream_t stream);   // block is in use on that stream too
inline void record_stream(const lamp_tensor* t, int stream_device, hipStream_t stream) {
  if (t && t->st && t->st->device >= 0 && t->st->owned) device_record_stream(t->st->device, t->st->pool, stream_device, stream);
}
int64_t allocator_deferred_frees(int device);
void allocator_stats(int device, int64_t* reserved, int64_t* in_use, int64_t* n_malloc);
void allocator_trim(int device);
void allocator_begin_capture_pool();
void allocator_end_capture_pool();
bool allocator_capturing();                 // this thread is between lamp_graph_begin_capture and lamp_graph_end_capture

// ---- runtime (core/runtime.cpp) ----
int current_device();
void set_device(int d);
uint32_t devices_used_mask();           // bit d: some thread of this process has selected device d
void synchronize_all_used_devices();    // hipDeviceSynchronize on each of them (checked); the caller's current device is restored
hipStream_t current_stream();           // thread-local current stream of current device
hipStream_t current_stream(int device);
int num_cus();
int kernel_occupancy(const void* kernel, int threads, size_t dynamic_lds);   // co-resident workgroups per CU (cached)
void allow_big_lds(const void* kernel);   // opt a kernel into 160 KiB of dynamic LDS on the current device (once per device)
// Batch-norm statistics computed by a convolution's epilogue: Welford triples [P][C][3] (f32) over P disjoint slices of the output,
// keyed by the output's storage (uid, offset) and valid while its version is unchanged.  The batch norm that consumes the tensor
// picks them up instead of re-reading it (norm.hip).  A small ring: the consumer runs right after the producer.
void conv_stats_publish(const lamp_tensor* y, lamp_tensor* partial, int P, uint64_t producer = 0);
bool conv_stats_wanted(uint64_t producer);   // false: this producer's last statistics were never taken (it probes again every 64th call)
lamp_tensor* conv_stats_lookup(const lamp_tensor* x, int64_t C, int* P);   // +1 handle or nullptr
// device-side assertions (runtime.cpp): a kernel stores a code into *device_assert_word(dev); the next host wait raises
enum DeviceAssert : int { kAssertNllTarget = 1, kAssertIndexRange = 2, kAssertBnExchangeTimeout = 3, kAssertMultinomial = 4 };
// "this device is also running kernels the library does not schedule" (an RCCL collective on the exchange stream while backward
// continues, a caller's own side-stream work): kernels whose workgroups wait for each other take their non-waiting form meanwhile.
void device_shared_add(int device, int delta);
int device_shared(int device);
int* device_assert_word(int device);
void check_device_asserts(int device);
uint64_t next_philox_offset(uint64_t n);  // advances the generator state by n draws
uint64_t philox_seed();

// ---- tensor construction (core/tensor.cpp) ----
Tensor* new_tensor(const int64_t* sizes, int ndim, int dtype, int device);  // uninitialised, contiguous
inline Tensor* new_tensor(const std::vector<int64_t>& s, int dtype, int device) {
  return new_tensor(s.data(), (int)s.size(), dtype, device);
}
Tensor* new_like(const Tensor* t);  // contiguous, same shape/dtype/device
Tensor* new_like(const Tensor* t, int dtype);
Tensor* new_view(const Tensor* base, const int64_t* sizes, const int64_t* strides, int ndim, int64_t offset);
Tensor* retain(const Tensor* t);  // new handle on the same view
void release(Tensor* t);
Tensor* contiguous(const Tensor* t);  // +1 handle; a copy only if needed
void copy_into(Tensor* dst, const Tensor* src);  // handles dtype conversion, strides, host<->device
void fill_zero(Tensor* t);

// ---- per-kernel-class timing with HIP events (bench.py roofline section) ----------------------
// Disabled unless lamp_kernel_timer_enable(1): then every tagged launch site records a pair of
// events on ITS stream; lamp_kernel_timer_report() sums elapsed times per tag together with the
// algorithmic flops / bytes the launcher declared.
struct KernelTimer {
  KernelTimer(const char* tag, double flops, double bytes, hipStream_t stream);
  ~KernelTimer();
  void* slot;
  hipStream_t stream;
};

// RAII holder for temporaries inside API functions.
struct Hold {
  Tensor* t;
  explicit Hold(Tensor* t_ = nullptr) : t(t_) {}
  ~Hold() { if (t) release(t); }
  Hold(const Hold&) = delete;
  Hold& operator=(const Hold&) = delete;
  Hold(Hold&& o) : t(o.t) { o.t = nullptr; }
  Hold& operator=(Hold&& o) { if (this != &o) { if (t) release(t); t = o.t; o.t = nullptr; } return *this; }
  Tensor* get() const { return t; }
  Tensor* operator->() const { return t; }
  Tensor* take() { Tensor* r = t; t = nullptr; return r; }
};

inline void check_same_device(const Tensor* a, const Tensor* b) {
  LAMP_CHECK(a->device() == b->device(), "tensors on different devices: " << a->describe() << " vs " << b->describe());
}
inline void check_device_tensor(const Tensor* a, const char* what) {
  LAMP_CHECK(a != nullptr, what << " is null");
  if (!a->is_device()) {
    // status LAMP_STATUS_HOST_TENSOR at the C ABI: when ALL tensor arguments are host tensors the staging layer runs the kernel on
    // copies (core/host_staging.cpp); with mixed devices this message reaches the caller
    std::ostringstream os;
    os << what << " " << a->describe() << " is a host tensor while other arguments live on the GPU: this operator exists only as a GPU kernel - "
       << "pass all tensors from one device (all-host arguments are staged through the GPU automatically)";
    throw HostTensorError(os.str());
  }
}

inline int grid_for(int64_t work_items, int block, int max_blocks_per_cu = 8) {
  int64_t g = (work_items + block - 1) / block;
  int64_t cap = (int64_t)num_cus() * max_blocks_per_cu;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace lamp
