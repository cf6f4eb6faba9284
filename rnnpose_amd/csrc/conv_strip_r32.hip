// The strip convolution kernels with 32-row strips (1 32-row MFMA tile per wave): launches that would not fill the chip with 160-row
// strips (conv_strip.hip, strip_rows()).  Its own translation unit so that the two strip heights compile in parallel.
#include "conv_strip_kernel.cuh"

namespace rpconv {

int strip_launch_r32(const KParams& p, int nw, int ni, bool spatial, bool hlin, bool norm, unsigned nwg, hipStream_t st) {
  return strip_launch_height<1>(p, nw, ni, spatial, hlin, norm, nwg, st);
}

}  // namespace rpconv
