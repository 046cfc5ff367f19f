// Deferred, batched reduction of the convolutions' weight-gradient partial sums (see wgrad_reduce.h).
#include "wgrad_reduce.h"
#include <mutex>
#include <thread>
#include <vector>

namespace lamp {

constexpr int WR_MAX = 16;
struct WgradReduceMany { WgradReduceArgs e[WR_MAX]; };

__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(WgradReduceMany m) {
  __shared__ float4 red[8][32];
  const WgradReduceArgs& a = m.e[blockIdx.y];
  if ((int)blockIdx.x >= a.blocks) return;          // uniform per workgroup
  if (a.kind == 0) wgrad_reduce_igemm(a, blockIdx.x, red);
  else wgrad_reduce_narrow(a, blockIdx.x);
}

namespace {
// Ownership: an entry belongs to the host thread that registered it.  flush_deferred() - the natural batching point - only
// takes the CALLING thread's entries, so the replica threads of the single-process data-parallel step never flush (or clear the flags
// of) each other's reductions; resolve_deferred(storage), reached from raw() of any thread, takes the entries of the thread that owns
// that storage's entry.  g_mu is held from the moment entries leave the list until their kernels are queued and only then are the
// `pending` flags cleared: a concurrent raw() of a pending tensor blocks on the mutex and returns with the reduction already queued
// in front of whatever the caller launches next.
struct Pending { WgradReduceArgs a; Tensor* partial; Tensor* dw; hipStream_t st; int device; std::thread::id owner; };
std::mutex g_mu;
std::vector<Pending> g_pending;

// called with g_mu held.  Pointers are taken from the storages directly (raw() would recurse into resolve_deferred).
void launch_batch(const std::vector<Pending>& v) {
  // one launch per (device, stream) group of at most WR_MAX reductions; order of registration kept
  std::vector<bool> done(v.size(), false);
  for (size_t i = 0; i < v.size(); i++) {
    if (done[i]) continue;
    WgradReduceMany m;
    int cnt = 0, maxb = 0;
    for (size_t j = i; j < v.size() && cnt < WR_MAX; j++) {
      if (done[j] || v[j].st != v[i].st || v[j].device != v[i].device) continue;
      m.e[cnt] = v[j].a;
      m.e[cnt].partial = reinterpret_cast<const float*>(static_cast<const char*>(v[j].partial->st->ptr)) + v[j].partial->offset;
      m.e[cnt].dw = static_cast<char*>(v[j].dw->st->ptr) + v[j].dw->offset * (v[j].a.dw_f32 ? 4 : 2);
      v[j].dw->st->version.fetch_add(1, std::memory_order_relaxed);   // a writer like any other: bumps dw's version
      maxb = std::max(maxb, v[j].a.blocks);
      done[j] = true;
      cnt++;
    }
    const int prev = current_device();
    if (prev != v[i].device) set_device(v[i].device);
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((unsigned)maxb, (unsigned)cnt), dim3(256), 0, v[i].st, m);
    hipError_t e = hipGetLastError();
    if (prev != v[i].device) set_device(prev);
    if (e != hipSuccess) throw Error(std::string("wgrad reduce launch failed: ") + hipGetErrorString(e));
    // a reader on another stream of that device must see the result: order it behind the reduction
    hipStream_t cur = current_stream(v[i].device);
    if (cur != v[i].st) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (allocator_capturing()) HIP_CHECK(hipStreamIsCapturing(v[i].st, &cs));
      if (allocator_capturing() && cs != hipStreamCaptureStatusActive) {
        // the caller is recording a graph on `cur` and the reduction's stream is outside that capture: an event from outside the capture
        // cannot be waited for inside it - wait on the host.  (A stream forked INTO the capture - the weight gradients' second stream,
        // host/ops.cpp - is joined with an event like any other: both ends are nodes of the same graph.)
        HIP_CHECK(hipStreamSynchronize(v[i].st));
      } else {
        hipEvent_t ev;
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev, v[i].st));
        HIP_CHECK(hipStreamWaitEvent(cur, ev, 0));
        HIP_CHECK(hipEventDestroy(ev));
      }
    }
  }
}

// takes the entries of `owner` out of the list, queues their kernels and clears their flags - all under g_mu
void flush_owner_locked(std::thread::id owner) {
  std::vector<Pending> v;
  size_t keep = 0;
  for (size_t i = 0; i < g_pending.size(); i++) {
    if (g_pending[i].owner == owner) v.push_back(g_pending[i]);
    else g_pending[keep++] = g_pending[i];
  }
  g_pending.resize(keep);
  if (v.empty()) return;
  struct Finish {
    std::vector<Pending>& v;
    ~Finish() {
      for (auto& p : v) p.dw->st->pending.store(0, std::memory_order_release);   // after the launches are queued (or have failed)
      for (auto& p : v) { release(p.partial); release(p.dw); }
    }
  } fin{v};
  launch_batch(v);
}
}  // namespace

void igemm_wgrad_flush_parked(bool all);     // conv_igemm.hip: weight-gradient LAUNCHES that wait for a second layer to share the chip with (they register their reductions here)
void flush_deferred() {
  igemm_wgrad_flush_parked(false);
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_pending.empty()) return;
  flush_owner_locked(std::this_thread::get_id());
}

void resolve_deferred(Storage* st) {
  igemm_wgrad_flush_parked(true);
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& p : g_pending)
    if (p.dw->st == st) { flush_owner_locked(p.owner); return; }
  // not in the list any more: another thread queued it while this one waited for the mutex
}

bool wgrad_reduce_deferred() {
  static const bool defer = [] { const char* e = getenv("LAMP_DEFER_WGRAD_REDUCE"); return !(e && e[0] == '0'); }();
  return defer;
}
void wgrad_reduce_enqueue(const WgradReduceArgs& a, lamp_tensor* partial, lamp_tensor* dw, hipStream_t st) {
  const bool defer = wgrad_reduce_deferred();
  if (!defer || !dw->st->owned) {
    WgradReduceMany m;
    m.e[0] = a;
    m.e[0].cached = defer ? 0 : 1;
    m.e[0].partial = partial->ptr<float>();
    m.e[0].dw = dw->data();
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((unsigned)a.blocks, 1), dim3(256), 0, st, m);
    LAMP_LAUNCH_CHECK();
    return;
  }
  Pending p{a, retain(partial), retain(dw), st, dw->device(), std::this_thread::get_id()};
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending.push_back(p);
  dw->st->pending.store(1, std::memory_order_release);
}

}  // namespace lamp
