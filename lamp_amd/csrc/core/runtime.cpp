// Devices, streams, errors, RNG state and HIP-graph capture.
//
// Mirrors what lamp expects from aten.CudaStream / aten.Tensor statics
// (reference: lamp-sten/src/main/scala/lamp/device.scala:119-129,178-217):
// the current device and the current stream are PER OS THREAD, kernels go to the
// calling thread's current stream.
#include "tensor.h"

#include <unordered_map>
#include <atomic>
#include <mutex>
#include <set>
#include <map>

struct lamp_stream {
  hipStream_t s = nullptr;
  int device = 0;
  bool is_default = false;
  std::atomic<int> refs{1};
};

struct lamp_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t stream = nullptr;
  int device = 0;
};

namespace lamp {

// Before the HIP runtime reads its environment (first HIP call): kernel arguments in device memory instead of host memory - ~100 short
// launches per training step fetch them (1 % of the graph-replayed ResNet step, 10 % of the eager one).  An explicit setting wins.
namespace {
struct EarlyEnv { EarlyEnv() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); } };
EarlyEnv g_early_env;
}  // namespace

static thread_local std::string tl_error;
void set_last_error(const std::string& msg) { tl_error = msg; }

static thread_local int tl_device = 0;
static thread_local bool tl_device_set = false;
static thread_local hipStream_t tl_streams[16] = {nullptr};
static thread_local bool tl_stream_set[16] = {false};

static std::mutex g_mu;
static int g_num_cus = -1;
static std::atomic<uint64_t> g_seed{0x5eed1234abcdULL};
static std::atomic<uint64_t> g_philox_offset{0};

// every device some thread of this process has selected (bit d): what "all work that may still touch a recycled pinned block" ranges over
static std::atomic<uint32_t> g_devices_used{0};
uint32_t devices_used_mask() { return g_devices_used.load(std::memory_order_acquire); }
// Waits for all queued work on every device this process has used, from any thread (the calling thread's current device is restored).
// hipHostFree waits like this implicitly; a recycled pinned block needs the same guarantee: a loader thread that never selected a device
// sits on device 0 while training runs on device N, and one process may drive several GPUs (DataParallel.scala:195-311).
void synchronize_all_used_devices() {
  const uint32_t mask = devices_used_mask();
  if (!mask) return;
  int prev = -1;
  HIP_CHECK(hipGetDevice(&prev));
  // the error path too leaves the thread on the device tl_device names (a throw used to leave HIP on device d)
  struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
  for (int d = 0; d < 16; d++) {
    if (!(mask & (1u << d))) continue;
    HIP_CHECK(hipSetDevice(d));
    HIP_CHECK(hipDeviceSynchronize());
  }
}
int current_device() {
  if (!tl_device_set) {
    // adopt the process default (LOCAL_RANK based selection happens via lamp_set_device)
    tl_device = 0;
    tl_device_set = true;
    hipError_t e = hipSetDevice(0);
    if (e != hipSuccess) throw Error(std::string("hipSetDevice(0) failed: ") + hipGetErrorString(e) +
                                     " - no usable MI355X device; this library has no CPU fallback");
    g_devices_used.fetch_or(1u, std::memory_order_acq_rel);
  }
  return tl_device;
}
void set_device(int d) {
  HIP_CHECK(hipSetDevice(d));
  tl_device = d;
  tl_device_set = true;
  if (d >= 0 && d < 16) g_devices_used.fetch_or(1u << d, std::memory_order_acq_rel);
}
hipStream_t current_stream(int device) {
  LAMP_CHECK(device >= 0 && device < 16, "bad device " << device);
  if (!tl_stream_set[device]) return nullptr;  // the null (default) stream
  return tl_streams[device];
}
hipStream_t current_stream() { return current_stream(current_device()); }

int num_cus() {
  if (g_num_cus < 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_num_cus < 0) {
      hipDeviceProp_t p;
      hipError_t e = hipGetDeviceProperties(&p, current_device());
      g_num_cus = (e == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
  }
  return g_num_cus;
}
// kernels that use more than 64 KiB of dynamic LDS opt in once per (function, device): the attribute belongs to the device's copy of
// the code object, and one process may drive several GPUs (single-process data parallel, one host thread per GPU)
void allow_big_lds(const void* fn) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({fn, dev})) return;
  HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  done.insert({fn, dev});
}
// co-resident workgroups per CU of (kernel, threads, dynamic LDS) on the current device, as the runtime computes it from the code
// object's register / LDS use.  Launchers that size a persistent grid as CUs x workgroups-per-CU ask here instead of assuming: a
// grid one workgroup per CU too large runs a second, nearly empty round.
int kernel_occupancy(const void* fn, int threads, size_t lds) {
  static std::mutex mu;
  static std::map<std::tuple<const void*, int, size_t, int>, int> cache;
  const int dev = current_device();
  const auto key = std::make_tuple(fn, threads, lds, dev);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int n = 0;
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds));
  n = std::max(1, n);
  cache[key] = n;
  return n;
}
uint64_t philox_seed() { return g_seed.load(); }
uint64_t next_philox_offset(uint64_t n) { return g_philox_offset.fetch_add(n); }

// ---- device-side assertions ---------------------------------------------------------------------
// ATen raises a device assert when nll_loss meets a class index outside [0, C); silently skipping the row would train on a subset
// with a shrunken total_weight.  Kernels report such conditions into one host-mapped word per device; the library looks at it at
// every point where the host waits for the device anyway (item, copy to host, synchronize) and raises there.
namespace {
std::mutex g_assert_mu;
int* g_assert_words = nullptr;     // [16], pinned + mapped + coherent: the device writes, the host reads without a copy
const char* assert_text(int code) {
  switch (code) {
    case kAssertNllTarget: return "nll_loss: a target class index is outside [0, numClasses) and is not ignore_index";
    case kAssertIndexRange: return "index out of range";
    case kAssertMultinomial: return "multinomial: invalid distribution (a weight is negative, infinite or NaN, or a row sums to zero)";
    case kAssertBnExchangeTimeout: return "batch-norm backward (one pass): a workgroup waited two minutes for the partial sums of its channel - "
                                          "another kernel is holding the compute units; the gradients of that launch are invalid (LAMP_BN_FUSED_BWD=0 selects the two-kernel form)";
  }
  return "device-side assertion";
}
}  // namespace
namespace { std::atomic<int> g_device_shared[16]; }
void device_shared_add(int device, int delta) { if (device >= 0 && device < 16) g_device_shared[device].fetch_add(delta, std::memory_order_acq_rel); }
int device_shared(int device) { return device >= 0 && device < 16 ? g_device_shared[device].load(std::memory_order_acquire) : 0; }

int* device_assert_word(int device) {
  std::lock_guard<std::mutex> lk(g_assert_mu);
  if (!g_assert_words) {
    void* p = nullptr;
    HIP_CHECK(hipHostMalloc(&p, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable));
    memset(p, 0, 16 * sizeof(int));
    g_assert_words = (int*)p;
  }
  return g_assert_words + device;
}
void check_device_asserts(int device) {
  if (!g_assert_words) return;
  volatile int* w = g_assert_words + device;
  const int code = *w;
  if (code == 0) return;
  *w = 0;
  throw Error(std::string("device-side assertion failed: ") + assert_text(code));
}

// ---- kernel timers ------------------------------------------------------------------------------
namespace {
struct TimerEntry { std::string tag; double flops, bytes; hipEvent_t a, b; };
std::mutex g_timer_mu;
std::vector<TimerEntry*> g_timer_entries;
std::vector<hipEvent_t> g_event_pool;      // recycled events: creation is the expensive part of a timed launch
std::atomic<int> g_timer_on{0};
std::string g_timer_filter;                // empty = every tagged launch, else only this tag (read under g_timer_on transitions)
hipEvent_t pooled_event() {
  {
    std::lock_guard<std::mutex> lk(g_timer_mu);
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

KernelTimer::KernelTimer(const char* tag, double flops, double bytes, hipStream_t st) : slot(nullptr), stream(st) {
  if (!g_timer_on.load(std::memory_order_relaxed)) return;
  if (!g_timer_filter.empty() && g_timer_filter != tag) return;
  auto* e = new TimerEntry{tag, flops, bytes, pooled_event(), pooled_event()};
  if (!e->a || !e->b) { delete e; return; }
  (void)hipEventRecord(e->a, st);
  slot = e;
}
KernelTimer::~KernelTimer() {
  if (!slot) return;
  auto* e = (TimerEntry*)slot;
  (void)hipEventRecord(e->b, stream);
  std::lock_guard<std::mutex> lk(g_timer_mu);
  g_timer_entries.push_back(e);
}

}  // namespace lamp

using namespace lamp;

extern "C" int lamp_kernel_timer_enable(int on) {
  g_timer_on.store(on);
  return 0;
}
// restrict timing to one tag (NULL or "" = all tags); call while the timers are disabled
extern "C" int lamp_kernel_timer_filter(const char* tag) {
  g_timer_filter = tag ? tag : "";
  return 0;
}
// writes lines "tag count total_ms flops_per_launch bytes_per_launch\n" and clears the log
extern "C" int lamp_kernel_timer_report(char* buf, int buflen) {
  LAMP_API_BEGIN
  std::vector<TimerEntry*> es;
  { std::lock_guard<std::mutex> lk(g_timer_mu); es.swap(g_timer_entries); }
  struct Agg { int64_t n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto* e : es) {
    (void)hipEventSynchronize(e->b);
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->a, e->b) == hipSuccess) {
      auto& a = agg[e->tag];
      a.n++; a.ms += ms; a.flops += e->flops; a.bytes += e->bytes;
    }
    { std::lock_guard<std::mutex> lk(g_timer_mu); g_event_pool.push_back(e->a); g_event_pool.push_back(e->b); }
    delete e;
  }
  std::string out;
  for (auto& kv : agg) {
    char line[256];
    snprintf(line, sizeof line, "%s %lld %.6f %.6e %.6e\n", kv.first.c_str(), (long long)kv.second.n, kv.second.ms,
             kv.second.flops / kv.second.n, kv.second.bytes / kv.second.n);
    out += line;
  }
  snprintf(buf, buflen, "%s", out.c_str());
  LAMP_API_END
}

extern "C" {

const char* lamp_last_error(void) { return tl_error.c_str(); }
const char* lamp_version(void) { return "lamp_hip 0.1 (gfx950)"; }

int lamp_has_gpu(int* out) {
  LAMP_API_BEGIN
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *out = (e == hipSuccess && n > 0) ? 1 : 0;
  LAMP_API_END
}
int lamp_get_num_gpus(int* out) {
  LAMP_API_BEGIN
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *out = (e == hipSuccess) ? n : 0;
  LAMP_API_END
}
int lamp_get_device(int* out) {
  LAMP_API_BEGIN
  *out = current_device();
  LAMP_API_END
}
int lamp_set_device(int device) {
  LAMP_API_BEGIN
  set_device(device);
  LAMP_API_END
}
int lamp_flush_deferred(void) {
  LAMP_API_BEGIN
  flush_deferred();
  LAMP_API_END
}
int lamp_device_synchronize(void) {
  LAMP_API_BEGIN
  flush_deferred();
  const int dev = current_device();
  HIP_CHECK(hipDeviceSynchronize());
  check_device_asserts(dev);
  LAMP_API_END
}
int lamp_device_shared_hint(int device, int delta) {
  LAMP_API_BEGIN
  LAMP_CHECK(device >= 0 && device < 16, "device " << device << " out of range");
  device_shared_add(device, delta);
  LAMP_CHECK(device_shared(device) >= 0, "lamp_device_shared_hint: more releases than acquisitions on device " << device);
  LAMP_API_END
}
int lamp_device_name(char* buf, int buflen) {
  LAMP_API_BEGIN
  hipDeviceProp_t p;
  HIP_CHECK(hipGetDeviceProperties(&p, current_device()));
  snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  LAMP_API_END
}
int lamp_device_num_cus(int* out) {
  LAMP_API_BEGIN
  *out = num_cus();
  LAMP_API_END
}

int lamp_stream_get_current(int device, lamp_stream** out) {
  LAMP_API_BEGIN
  auto* s = new lamp_stream();
  s->device = device;
  s->s = current_stream(device);
  s->is_default = (s->s == nullptr);
  *out = s;
  LAMP_API_END
}
int lamp_stream_get_default(int device, lamp_stream** out) {
  LAMP_API_BEGIN
  auto* s = new lamp_stream();
  s->device = device;
  s->s = nullptr;
  s->is_default = true;
  *out = s;
  LAMP_API_END
}

// a small round-robin pool per device, like c10's getStreamFromPool
static std::mutex g_pool_mu;
static std::map<std::pair<int, int>, std::vector<hipStream_t>> g_pool;
static std::map<std::pair<int, int>, size_t> g_pool_next;

int lamp_stream_get_from_pool(int high_priority, int device, lamp_stream** out) {
  LAMP_API_BEGIN
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto key = std::make_pair(device, high_priority ? 1 : 0);
  auto& v = g_pool[key];
  if (v.empty()) {
    int prev = current_device();
    HIP_CHECK(hipSetDevice(device));
    int lo = 0, hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int i = 0; i < 8; i++) {
      hipStream_t s;
      HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo));
      v.push_back(s);
    }
    HIP_CHECK(hipSetDevice(prev));
  }
  size_t& nx = g_pool_next[key];
  auto* s = new lamp_stream();
  s->device = device;
  s->s = v[nx % v.size()];
  nx++;
  *out = s;
  LAMP_API_END
}
int lamp_stream_set_current(lamp_stream* s) {
  LAMP_API_BEGIN
  LAMP_CHECK(s, "null stream");
  tl_streams[s->device] = s->s;
  tl_stream_set[s->device] = !s->is_default;
  LAMP_API_END
}
int lamp_stream_synchronize(lamp_stream* s) {
  LAMP_API_BEGIN
  LAMP_CHECK(s, "null stream");
  flush_deferred();
  HIP_CHECK(hipStreamSynchronize(s->s));
  check_device_asserts(s->device);
  LAMP_API_END
}
int lamp_stream_wait_stream(lamp_stream* waiter, lamp_stream* on) {
  LAMP_API_BEGIN
  hipEvent_t ev;
  HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(ev, on->s));
  HIP_CHECK(hipStreamWaitEvent(waiter->s, ev, 0));
  HIP_CHECK(hipEventDestroy(ev));
  LAMP_API_END
}
// `t`'s storage is (also) in use by work queued on `s` (NULL: the calling thread's current stream of t's device): when the last
// handle is released the block is not recycled before that work has finished (at::Tensor::record_stream; lamp consumes tensors
// allocated inside `withOtherStream` on the default stream, device.scala:199-213)
int lamp_tensor_record_stream(const lamp_tensor* t, lamp_stream* s) {
  LAMP_API_BEGIN
  LAMP_CHECK(t, "null tensor");
  if (!t->is_device()) return 0;
  if (s) record_stream(t, s->device, s->s);
  else record_stream(t, t->device(), current_stream(t->device()));
  LAMP_API_END
}
int lamp_stream_release(lamp_stream* s) {
  LAMP_API_BEGIN
  delete s;  // pool streams stay alive for the process lifetime
  LAMP_API_END
}
int lamp_stream_native(lamp_stream* s, void** out) {
  LAMP_API_BEGIN
  *out = (void*)s->s;
  LAMP_API_END
}

int lamp_manual_seed(uint64_t seed) {
  LAMP_API_BEGIN
  g_seed.store(seed);
  g_philox_offset.store(0);
  LAMP_API_END
}
int lamp_allow_tf32(int) { return 0; }

// ---- HIP graphs ------------------------------------------------------------------------------
int lamp_graph_begin_capture(void) {
  LAMP_API_BEGIN
  hipStream_t s = current_stream();
  LAMP_CHECK(s != nullptr, "graph capture needs a non-default current stream "
                           "(lamp_stream_get_from_pool + lamp_stream_set_current first)");
  // deferred kernels registered BEFORE the capture run now, eagerly: flushed by the first raw() inside the capture they would only be
  // recorded and the eager pass would never produce their tensors
  flush_deferred();
  allocator_begin_capture_pool();
  hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    allocator_end_capture_pool();
    HIP_CHECK(e);
  }
  LAMP_API_END
}
int lamp_graph_end_capture(lamp_graph** out) {
  LAMP_API_BEGIN
  hipStream_t s = current_stream();
  hipGraph_t g = nullptr;
  // deferred kernels registered INSIDE the capture belong to the graph: queue them on the capturing stream before it ends, otherwise
  // they would run once, eagerly, after the capture and every replay would leave their tensors stale
  try { flush_deferred(); } catch (...) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(s, &dead); if (dead) (void)hipGraphDestroy(dead); allocator_end_capture_pool(); throw; }
  hipError_t e = hipStreamEndCapture(s, &g);
  allocator_end_capture_pool();
  HIP_CHECK(e);
  auto* lg = new lamp_graph();
  lg->graph = g;
  lg->stream = s;
  lg->device = current_device();
  HIP_CHECK(hipGraphInstantiate(&lg->exec, g, nullptr, nullptr, 0));
  *out = lg;
  LAMP_API_END
}
int lamp_graph_is_capturing(int* out) {
  LAMP_API_BEGIN
  *out = allocator_capturing() ? 1 : 0;
  LAMP_API_END
}
int lamp_graph_launch(lamp_graph* g) {
  LAMP_API_BEGIN
  LAMP_CHECK(g && g->exec, "null graph");
  HIP_CHECK(hipGraphLaunch(g->exec, current_stream()));
  LAMP_API_END
}
int lamp_graph_release(lamp_graph* g) {
  LAMP_API_BEGIN
  if (g) {
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
  }
  LAMP_API_END
}

}  // extern "C"

// ---- convolution -> batch-norm statistics hand-off ------------------------------------------------------------------------
namespace lamp {
namespace {
struct ConvStatsEntry {
  uint64_t uid = 0, version = 0; int64_t offset = 0, N = 0, C = 0, HW = 0; lamp_tensor* partial = nullptr; int P = 0;
  uint64_t producer = 0;                    // who computed them (the filter's storage uid; 0: not tracked)
  bool consumed = false;                    // a batch norm has taken them
  uint64_t hits_then = 0;                   // g_conv_stats_hits when they were published
};
uint64_t g_conv_stats_hits = 0;             // hand-offs taken so far
// producers whose last statistics nobody took (a convolution that is not followed by a batch norm: the stem of Cnn.resnet): they stop
// computing them, and probe again every 64th call
struct ConvStatsUse { bool unused = false; unsigned calls = 0; };
std::unordered_map<uint64_t, ConvStatsUse> g_conv_stats_use;
inline int64_t spatial_of(const lamp_tensor* t) { int64_t hw = 1; for (int i = 2; i < t->ndim; i++) hw *= t->sizes[i]; return hw; }
constexpr int kConvStatsRing = 32;
ConvStatsEntry g_conv_stats[kConvStatsRing];
int g_conv_stats_next = 0;
std::mutex g_conv_stats_mu;
}  // namespace

void conv_stats_publish(const lamp_tensor* y, lamp_tensor* partial, int P, uint64_t producer) {
  std::lock_guard<std::mutex> lock(g_conv_stats_mu);
  ConvStatsEntry& e = g_conv_stats[g_conv_stats_next];
  g_conv_stats_next = (g_conv_stats_next + 1) % kConvStatsRing;
  if (e.partial) lamp_tensor_release(e.partial);
  e.uid = y->st->uid; e.version = y->st->version.load(std::memory_order_relaxed); e.offset = y->offset;
  e.N = y->sizes[0]; e.C = y->sizes[1]; e.HW = spatial_of(y);
  e.partial = nullptr; e.P = P; e.producer = producer; e.consumed = false; e.hits_then = g_conv_stats_hits;
  lamp_tensor_retain(partial, &e.partial);
}
bool conv_stats_wanted(uint64_t producer) {
  std::lock_guard<std::mutex> lock(g_conv_stats_mu);
  // its previous output's fate: taken, or passed over - later hand-offs were taken while this one was not (two calls in a row with no
  // batch norm in between say nothing)
  for (auto& o : g_conv_stats)
    if (o.partial && o.producer == producer) {
      if (o.consumed) g_conv_stats_use[producer].unused = false;
      else if (g_conv_stats_hits > o.hits_then) g_conv_stats_use[producer].unused = true;
      o.producer = 0;
    }
  if (g_conv_stats_use.size() > 4096) g_conv_stats_use.clear();          // (filters come and go: forget rather than grow)
  auto it = g_conv_stats_use.find(producer);
  if (it == g_conv_stats_use.end() || !it->second.unused) return true;
  return (++it->second.calls & 63u) == 0;
}

lamp_tensor* conv_stats_lookup(const lamp_tensor* x, int64_t C, int* P) {
  if (!x || !x->st) return nullptr;
  std::lock_guard<std::mutex> lock(g_conv_stats_mu);
  for (auto& e : g_conv_stats) {
    // the whole tensor the convolution wrote, unchanged since: same storage, offset, shape and version, and dense
    if (e.partial && e.uid == x->st->uid && e.offset == x->offset && x->ndim >= 3 && e.N == x->sizes[0] && e.C == C && x->sizes[1] == C &&
        e.HW == spatial_of(x) && x->is_contiguous() && e.version == x->st->version.load(std::memory_order_relaxed)) {
      lamp_tensor* r = nullptr;
      lamp_tensor_retain(e.partial, &r);
      *P = e.P;
      e.consumed = true;
      g_conv_stats_hits++;
      return r;
    }
  }
  return nullptr;
}
}  // namespace lamp
