// Broadcast / stride bookkeeping shared by copy, element-wise and reduction launchers.
#pragma once
#include "tensor.h"
#include <algorithm>

namespace lamp {

constexpr int kMaxOperands = 4;

// N operands iterated over one common (broadcast, dimension-collapsed) index space.
struct IterSpace {
  int ndim = 0;
  int nops = 0;
  int64_t sizes[kMaxDims] = {0};
  int64_t strides[kMaxOperands][kMaxDims] = {{0}};  // in elements
  int64_t numel = 1;
  bool all_contiguous = false;  // every operand dense in iteration order (stride pattern of a contiguous tensor)
};

inline std::vector<int64_t> broadcast_shapes(const std::vector<int64_t>& a, const std::vector<int64_t>& b) {
  size_t n = std::max(a.size(), b.size());
  std::vector<int64_t> out(n);
  for (size_t i = 0; i < n; i++) {
    int64_t da = i < n - a.size() ? 1 : a[i - (n - a.size())];
    int64_t db = i < n - b.size() ? 1 : b[i - (n - b.size())];
    LAMP_CHECK(da == db || da == 1 || db == 1, "shapes are not broadcastable (" << da << " vs " << db << " at dim " << i << ")");
    out[i] = da == 1 ? db : da;
  }
  return out;
}

// ops[i] are broadcast against `shape` (ops[0] is normally the output and must match it exactly).
inline IterSpace make_iter(const std::vector<int64_t>& shape, const Tensor* const* ops, int nops) {
  LAMP_CHECK(nops <= kMaxOperands, "too many operands");
  LAMP_CHECK((int)shape.size() <= kMaxDims, "too many dims");
  int nd = (int)shape.size();
  int64_t st[kMaxOperands][kMaxDims];
  for (int o = 0; o < nops; o++) {
    const Tensor* t = ops[o];
    LAMP_CHECK(t->ndim <= nd, "operand has more dims than the iteration shape");
    int lead = nd - t->ndim;
    for (int d = 0; d < nd; d++) {
      if (d < lead) { st[o][d] = 0; continue; }
      int64_t sz = t->sizes[d - lead];
      if (sz == shape[d]) st[o][d] = (sz == 1) ? 0 : t->strides[d - lead];
      else {
        LAMP_CHECK(sz == 1, "operand " << o << " " << t->describe() << " does not broadcast to the iteration shape");
        st[o][d] = 0;
      }
    }
  }
  // drop size-1 dims, then merge adjacent dims where every operand allows it
  IterSpace it;
  it.nops = nops;
  it.numel = 1;
  for (int d = 0; d < nd; d++) it.numel *= shape[d];
  int k = 0;
  for (int d = 0; d < nd; d++) {
    if (shape[d] == 1) continue;
    if (k > 0) {
      bool merge = true;
      for (int o = 0; o < nops; o++)
        if (it.strides[o][k - 1] != st[o][d] * shape[d]) { merge = false; break; }
      if (merge) {
        it.sizes[k - 1] *= shape[d];
        for (int o = 0; o < nops; o++) it.strides[o][k - 1] = st[o][d];
        continue;
      }
    }
    it.sizes[k] = shape[d];
    for (int o = 0; o < nops; o++) it.strides[o][k] = st[o][d];
    k++;
  }
  if (k == 0) {  // scalar
    it.sizes[0] = 1;
    for (int o = 0; o < nops; o++) it.strides[o][0] = 0;
    k = 1;
  }
  it.ndim = k;
  it.all_contiguous = (k == 1);
  if (it.all_contiguous)
    for (int o = 0; o < nops; o++)
      if (it.strides[o][0] != 1 && it.numel != 1) it.all_contiguous = false;
  return it;
}

// device-side copy of the index space (passed by value as a kernel argument)
struct IterArgs {
  int ndim;
  int64_t sizes[kMaxDims];
  int64_t strides[kMaxOperands][kMaxDims];
};
inline IterArgs to_args(const IterSpace& it) {
  IterArgs a;
  a.ndim = it.ndim;
  for (int d = 0; d < kMaxDims; d++) {
    a.sizes[d] = d < it.ndim ? it.sizes[d] : 1;
    for (int o = 0; o < kMaxOperands; o++) a.strides[o][d] = (d < it.ndim && o < it.nops) ? it.strides[o][d] : 0;
  }
  return a;
}

template <int NOPS>
__host__ __device__ __forceinline__ void iter_offsets(const IterArgs& a, int64_t linear, int64_t (&off)[NOPS]) {
#pragma unroll
  for (int o = 0; o < NOPS; o++) off[o] = 0;
  for (int d = a.ndim - 1; d >= 0; d--) {
    int64_t q = linear / a.sizes[d];
    int64_t r = linear - q * a.sizes[d];
    linear = q;
#pragma unroll
    for (int o = 0; o < NOPS; o++) off[o] += r * a.strides[o][d];
  }
}

inline int64_t wrap_dim(int64_t dim, int ndim, bool allow_end = false) {
  int64_t lim = ndim + (allow_end ? 1 : 0);
  if (ndim == 0 && !allow_end) lim = 1;
  LAMP_CHECK(dim >= -lim && dim < lim, "dimension " << dim << " out of range for " << ndim << " dims");
  return dim < 0 ? dim + lim : dim;
}

}  // namespace lamp
