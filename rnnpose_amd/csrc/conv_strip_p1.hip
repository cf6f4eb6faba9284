// The strip convolution kernels (160-row strips) in their single-product form: cfg.raft.mixed_precision (conv_strip_kernel.cuh, P1).
// Its own translation unit so that it compiles next to the three-product kernels.
#include "conv_strip_kernel.cuh"

namespace rpconv {

int strip_launch_p1(const KParams& p, int nw, int ni, bool spatial, bool hlin, bool norm, unsigned nwg, hipStream_t st) {
  return strip_launch_height<5, true>(p, nw, ni, spatial, hlin, norm, nwg, st);
}

}  // namespace rpconv
