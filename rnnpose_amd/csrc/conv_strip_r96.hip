// The strip convolution kernels with 96-row strips (3 32-row MFMA tiles per wave; r06): launches whose 160-row strips leave most SIMDs without a
// wave (the 128-column layers of a half-batch chain: 120 workgroups) -- conv_strip.hip, strip_rows().  Its own translation unit so that the
// strip heights compile in parallel.
#include "conv_strip_kernel.cuh"

namespace rpconv {

int strip_launch_r96(const KParams& p, int nw, int ni, bool spatial, bool hlin, bool norm, unsigned nwg, hipStream_t st) {
  return strip_launch_height<3>(p, nw, ni, spatial, hlin, norm, nwg, st);
}

}  // namespace rpconv
