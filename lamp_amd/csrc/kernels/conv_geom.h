// Convolution geometry shared by the direct and implicit-GEMM kernels.
// Always expressed in "regular" terms: (N, Cin, H, W) is the input side of a regular convolution,
// (N, Cout, Ho, Wo) its output side; a transposed convolution simply runs the dgrad direction.
#pragma once
#include <cstdint>

namespace lamp {

struct ConvGeom {
  int64_t N, Cin, H, W, Cout, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  int64_t groups;
  int transposed;
};

// algorithmic work of one convolution pass (any of fprop / dgrad / wgrad): 2*MACs, and one read of
// each operand plus one write of the result
inline double conv_flops(const ConvGeom& g) {
  return 2.0 * (double)g.N * (double)g.Cout * (double)g.Ho * (double)g.Wo * (double)(g.Cin / g.groups) * g.kh * g.kw;
}
inline double conv_bytes(const ConvGeom& g, size_t elt) {
  return ((double)g.N * g.Cin * g.H * g.W + (double)g.N * g.Cout * g.Ho * g.Wo + (double)g.Cout * (g.Cin / g.groups) * g.kh * g.kw) * (double)elt;
}

}  // namespace lamp
