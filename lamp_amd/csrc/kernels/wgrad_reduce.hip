// Deferred, batched reduction of the convolutions' weight-gradient partial sums (see wgrad_reduce.h).
#include "wgrad_reduce.h"
#include <mutex>
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
struct Pending { WgradReduceArgs a; Tensor* partial; Tensor* dw; hipStream_t st; int device; };
std::mutex g_mu;
std::vector<Pending> g_pending;

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
      m.e[cnt].partial = static_cast<const float*>(v[j].partial->raw());
      m.e[cnt].dw = v[j].dw->ptr<bf16_t>();               // mutable pointer: bumps dw's version like any other writer
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
      hipEvent_t ev;
      HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(ev, v[i].st));
      HIP_CHECK(hipStreamWaitEvent(cur, ev, 0));
      HIP_CHECK(hipEventDestroy(ev));
    }
  }
}
}  // namespace

void flush_deferred() {
  std::vector<Pending> v;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pending.empty()) return;
    v.swap(g_pending);
    for (auto& p : v) p.dw->st->pending.store(0, std::memory_order_release);   // before any pointer is taken below
  }
  struct Releaser { std::vector<Pending>& v; ~Releaser() { for (auto& p : v) { release(p.partial); release(p.dw); } } } rel{v};
  launch_batch(v);
}

void resolve_deferred(Storage*) { flush_deferred(); }

void wgrad_reduce_enqueue(const WgradReduceArgs& a, lamp_tensor* partial, lamp_tensor* dw, hipStream_t st) {
  static const bool defer = [] { const char* e = getenv("LAMP_DEFER_WGRAD_REDUCE"); return !(e && e[0] == '0'); }();
  if (!defer || !dw->st->owned) {
    WgradReduceMany m;
    m.e[0] = a;
    m.e[0].partial = partial->ptr<float>();
    m.e[0].dw = dw->ptr<bf16_t>();
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((unsigned)a.blocks, 1), dim3(256), 0, st, m);
    LAMP_LAUNCH_CHECK();
    return;
  }
  Pending p{a, retain(partial), retain(dw), st, dw->device()};
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending.push_back(p);
  dw->st->pending.store(1, std::memory_order_release);
}

}  // namespace lamp
