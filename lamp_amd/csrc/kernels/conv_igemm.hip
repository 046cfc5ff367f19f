// Implicit-GEMM MFMA convolution kernels (placeholder dispatch: the direct kernels in conv.hip
// handle every geometry until the qualifying fast paths below are enabled).
#include "device_utils.h"
#include "conv_geom.h"

namespace lamp {

bool igemm_conv_fwd(const Tensor*, const Tensor*, const Tensor*, Tensor*, const ConvGeom&, hipStream_t) { return false; }
bool igemm_conv_dgrad(const Tensor*, const Tensor*, Tensor*, const ConvGeom&, hipStream_t) { return false; }
bool igemm_conv_wgrad(const Tensor*, const Tensor*, Tensor*, const ConvGeom&, hipStream_t) { return false; }

}  // namespace lamp
