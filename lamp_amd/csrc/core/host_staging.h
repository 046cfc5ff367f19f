// Host tensors on operators that exist only as GPU kernels (design in scripts/gen_host_staging.py).
// lamp's default device for knnSearch / Umap.umap is CPU (knn/package.scala:145, umap.scala:357) and the reference's gradient suite
// runs its CPU variant first (autograd.test.scala:117-133): a caller that hands such an operator tensors that ALL live in host memory
// gets them copied to the calling thread's current GPU, the same kernel run on the current stream, and host tensors back.
#pragma once
#include "tensor.h"
#include "lamp_hip.h"

#include <algorithm>
#include <tuple>
#include <vector>

namespace lamp {
namespace staging {

// One distinct view of a host storage (the common case) is staged as before: a fresh, compact, aligned device tensor, written back through
// the host view's strides if the operator may write it.  When the arguments of ONE call view the same host storage in two or more
// different ways (an `_out` form whose destination aliases an input through another handle, a slice next to its
// parent) the storage is staged as a BLOCK: the byte range its views span is copied to the GPU once and each argument becomes a view
// of that block with the same sizes, strides and relative offset - they alias on the GPU exactly as on the host - and a block that
// any argument may write is copied back once, whole.
class Stager {
 public:
  void see(const lamp_tensor* t) {
    if (!t) return;
    seen_.push_back(t);
    if (t->is_device() || !t->st || t->numel() == 0) return;
    const int64_t isz = (int64_t)t->itemsize();
    int64_t lo = t->offset, hi = t->offset;
    for (int i = 0; i < t->ndim; i++) {
      const int64_t span = (t->sizes[i] - 1) * t->strides[i];
      if (span < 0) lo += span; else hi += span;
    }
    const int64_t b0 = (lo * isz) & ~(int64_t)15, b1 = (hi + 1) * isz;      // 16-byte aligned start: every dtype's offsets stay whole
    for (Block& b : blocks_)
      if (b.host == t->st) {
        if (!same_view(b.first, t)) b.shared = true;
        b.lo = std::min(b.lo, b0); b.hi = std::max(b.hi, b1);
        return;
      }
    blocks_.push_back(Block{t->st, t, b0, b1, nullptr, false, false});
  }
  // every tensor argument lives in host memory (and there is at least one): otherwise the kernel's own error stands
  bool all_host() const {
    if (seen_.empty()) return false;
    for (const lamp_tensor* t : seen_) if (t->is_device()) return false;
    return true;
  }
  const lamp_tensor* in(const lamp_tensor* t) { return t ? view_of(const_cast<lamp_tensor*>(t), false) : nullptr; }
  lamp_tensor* inout(lamp_tensor* t) { return t ? view_of(t, true) : nullptr; }
  void ran(int rc) { if (rc != 0) throw Error(lamp_last_error()); }
  // a device result -> a host tensor of the same shape and dtype (the device handle is released)
  lamp_tensor* out(lamp_tensor* dev) {
    if (!dev) return nullptr;
    Hold d(dev);
    Hold h(new_tensor(d->sizes, d->ndim, d->dtype, -1));
    if (d->numel() > 0) copy_into(h.get(), d.get());
    return h.take();
  }
  // what the operator may have written: back into the caller's host memory
  void finish() {
    for (View& v : views_)
      if (v.writeback && v.host->numel() > 0) copy_into(v.host, v.dev);          // compact copies, through the host view's strides
    for (Block& b : blocks_)
      if (b.shared && b.writeback && b.dev) { Hold hv(bytes_of(b)); copy_into(hv.get(), b.dev); }
  }
  ~Stager() {
    for (View& v : views_) release(v.dev);
    for (Block& b : blocks_) if (b.dev) release(b.dev);
  }

 private:
  // [lo, hi): bytes of the host storage its views span; shared: viewed in more than one way by this call
  struct Block { Storage* host; const lamp_tensor* first; int64_t lo, hi; lamp_tensor* dev; bool writeback, shared; };
  struct View { lamp_tensor* host; lamp_tensor* dev; bool writeback; };
  static bool same_view(const lamp_tensor* a, const lamp_tensor* b) {
    if (a->offset != b->offset || a->ndim != b->ndim || a->dtype != b->dtype) return false;
    for (int i = 0; i < a->ndim; i++) if (a->sizes[i] != b->sizes[i] || a->strides[i] != b->strides[i]) return false;
    return true;
  }
  // the staged byte range of a host storage as a u8 host tensor (a view: no copy)
  lamp_tensor* bytes_of(const Block& b) {
    const int64_t n[1] = {b.hi - b.lo}, one[1] = {1};
    lamp_tensor* t = new_view(b.first, n, one, 1, b.lo);
    t->dtype = kU8;
    return t;
  }
  lamp_tensor* view_of(lamp_tensor* host, bool writeback) {
    Block* blk = nullptr;
    if (host->st && host->numel() > 0)
      for (Block& b : blocks_) if (b.host == host->st) blk = &b;
    for (View& v : views_)
      if (v.host == host || (blk && !blk->shared && v.host->st == host->st)) {     // the same handle, or another handle on the same view
        if (!(blk && blk->shared)) v.writeback |= writeback;
        else blk->writeback |= writeback;
        return v.dev;
      }
    if (!blk || !blk->shared) {                                  // the only view of its storage (or empty): a compact device copy
      Hold d(new_tensor(host->sizes, host->ndim, host->dtype, current_device()));
      if (host->numel() > 0) copy_into(d.get(), host);
      views_.push_back(View{host, d.get(), writeback});
      return d.take();
    }
    blk->writeback |= writeback;
    if (!blk->dev) {
      const int64_t n[1] = {blk->hi - blk->lo};
      Hold d(new_tensor(n, 1, kU8, current_device()));
      Hold hv(bytes_of(*blk));
      copy_into(d.get(), hv.get());
      blk->dev = d.take();
    }
    const int64_t isz = (int64_t)host->itemsize();
    lamp_tensor* d = new_view(blk->dev, host->sizes, host->strides, host->ndim, (host->offset * isz - blk->lo) / isz);
    d->dtype = host->dtype;
    views_.push_back(View{host, d, false});
    return d;
  }
  std::vector<const lamp_tensor*> seen_;
  std::vector<Block> blocks_;
  std::vector<View> views_;
};

// ---- one template for every staged entry point ------------------------------------------------------------------------------------
// scripts/gen_host_staging.py writes ONE line per C-ABI function: `return staging::call(<name>__dev, role{arg}...)`, the role saying what
// the header's parameter convention means (round 4; until then the generator spelled these ~18 lines out per function: 3.9 k lines).
struct Out { lamp_tensor** p; };                                  // lamp_tensor** name: one output handle (NULL: not wanted)
template <int K> struct OutK { lamp_tensor** p; };               // lamp_tensor* name[K]: K output handles
struct In { const lamp_tensor* t; };                              // const lamp_tensor* name: read (NULL allowed)
struct InOut { lamp_tensor* t; };                                 // lamp_tensor* name: written in place
struct Arr { lamp_tensor* const* a; int n; bool read_only; };     // lamp_tensor* const* name + n (a NULL array is passed through)

// first attempt: the caller's own arguments
template <class T> inline T raw(T v) { return v; }
inline lamp_tensor** raw(Out o) { return o.p; }
template <int K> inline lamp_tensor** raw(OutK<K> o) { return o.p; }
inline const lamp_tensor* raw(In i) { return i.t; }
inline lamp_tensor* raw(InOut i) { return i.t; }
inline lamp_tensor* const* raw(Arr a) { return a.a; }

template <class T> inline void see(Stager&, const T&) {}
inline void see(Stager& s, In i) { s.see(i.t); }
inline void see(Stager& s, InOut i) { s.see(i.t); }
inline void see(Stager& s, Arr a) { if (a.a) for (int i = 0; i < a.n; i++) s.see(a.a[i]); }

// second attempt: what each argument becomes on the GPU, and what happens to it afterwards
template <class T> struct Held { T v; Held(Stager&, T x) : v(x) {} T arg() { return v; } void done(Stager&) {} };
template <> struct Held<Out> {
  lamp_tensor** p; lamp_tensor* d = nullptr;
  Held(Stager&, Out o) : p(o.p) {}
  lamp_tensor** arg() { return p ? &d : nullptr; }
  void done(Stager& s) { if (p) *p = s.out(d); }
};
template <int K> struct Held<OutK<K>> {
  lamp_tensor** p; lamp_tensor* d[K] = {};
  Held(Stager&, OutK<K> o) : p(o.p) {}
  lamp_tensor** arg() { return d; }
  void done(Stager& s) { for (int i = 0; i < K; i++) p[i] = s.out(d[i]); }
};
template <> struct Held<In> { const lamp_tensor* d; Held(Stager& s, In i) : d(s.in(i.t)) {} const lamp_tensor* arg() { return d; } void done(Stager&) {} };
template <> struct Held<InOut> { lamp_tensor* d; Held(Stager& s, InOut i) : d(s.inout(i.t)) {} lamp_tensor* arg() { return d; } void done(Stager&) {} };
template <> struct Held<Arr> {
  std::vector<lamp_tensor*> d; bool null;
  Held(Stager& s, Arr a) : d(a.a && a.n > 0 ? a.n : 0), null(!a.a) {
    if (a.a) for (int i = 0; i < a.n; i++) d[i] = a.read_only ? const_cast<lamp_tensor*>(s.in(a.a[i])) : s.inout(a.a[i]);
  }
  lamp_tensor* const* arg() { return null ? nullptr : d.data(); }
  void done(Stager&) {}
};

// rc = fn(args); host tensors everywhere?  stage -> fn on the copies -> outputs and in-place results back to the host
template <class Fn, class... A>
int call(Fn fn, A... a) {
  const int rc = fn(raw(a)...);
  if (rc != LAMP_STATUS_HOST_TENSOR) return rc;
  LAMP_API_BEGIN
  Stager s;
  (see(s, a), ...);
  if (!s.all_host()) return rc;          // mixed devices (or nothing to stage): the kernel's own message stands
  std::tuple<Held<A>...> h{Held<A>(s, a)...};                     // (braced: built left to right, the order of the parameters)
  s.ran(std::apply([&](auto&... x) { return fn(x.arg()...); }, h));
  std::apply([&](auto&... x) { (x.done(s), ...); }, h);
  s.finish();
  LAMP_API_END
}

}  // namespace staging
}  // namespace lamp
