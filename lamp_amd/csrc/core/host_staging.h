// Host tensors on operators that exist only as GPU kernels (design in scripts/gen_host_staging.py).
// lamp's default device for knnSearch / Umap.umap is CPU (knn/package.scala:145, umap.scala:357) and the reference's gradient suite
// runs its CPU variant first (autograd.test.scala:117-133): a caller that hands such an operator tensors that ALL live in host memory
// gets them copied to the calling thread's current GPU, the same kernel run on the current stream, and host tensors back.
#pragma once
#include "tensor.h"
#include "lamp_hip.h"

#include <vector>

namespace lamp {
namespace staging {

class Stager {
 public:
  void see(const lamp_tensor* t) { if (t) seen_.push_back(t); }
  // every tensor argument lives in host memory (and there is at least one): otherwise the kernel's own error stands
  bool all_host() const {
    if (seen_.empty()) return false;
    for (const lamp_tensor* t : seen_) if (t->is_device()) return false;
    return true;
  }
  const lamp_tensor* in(const lamp_tensor* t) { return t ? pair_of(const_cast<lamp_tensor*>(t), false) : nullptr; }
  lamp_tensor* inout(lamp_tensor* t) { return t ? pair_of(t, true) : nullptr; }
  void ran(int rc) { if (rc != 0) throw Error(lamp_last_error()); }
  // a device result -> a host tensor of the same shape and dtype (the device handle is released)
  lamp_tensor* out(lamp_tensor* dev) {
    if (!dev) return nullptr;
    Hold d(dev);
    Hold h(new_tensor(d->sizes, d->ndim, d->dtype, -1));
    if (d->numel() > 0) copy_into(h.get(), d.get());
    return h.take();
  }
  // tensors the operator wrote in place: back into the caller's host tensors (through their strides)
  void finish() {
    for (Pair& p : pairs_)
      if (p.writeback && p.host->numel() > 0) copy_into(p.host, p.dev);
  }
  ~Stager() { for (Pair& p : pairs_) release(p.dev); }

 private:
  struct Pair { lamp_tensor* host; lamp_tensor* dev; bool writeback; };
  lamp_tensor* pair_of(lamp_tensor* host, bool writeback) {
    for (Pair& p : pairs_)
      if (p.host == host) { p.writeback |= writeback; return p.dev; }     // the same handle passed twice: one copy
    Hold d(new_tensor(host->sizes, host->ndim, host->dtype, current_device()));
    if (host->numel() > 0) copy_into(d.get(), host);
    pairs_.push_back(Pair{host, d.get(), writeback});
    return d.take();
  }
  std::vector<const lamp_tensor*> seen_;
  std::vector<Pair> pairs_;
};

}  // namespace staging
}  // namespace lamp
