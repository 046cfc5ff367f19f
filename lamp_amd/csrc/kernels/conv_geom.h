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

}  // namespace lamp
