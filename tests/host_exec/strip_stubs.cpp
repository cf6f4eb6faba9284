// TEST INFRASTRUCTURE: csrc/conv_strip*.hip (LDS-DMA inline assembly) cannot be compiled for the host; with these stand-ins the
// convolution dispatcher of csrc/conv_igemm.hip never chooses a strip kernel (strip_rows() = 0: "not a strip launch") and the
// strip copy of the packed weights is left unwritten.  A forced strip request (tile 5 / 6) fails like an unsupported shape.
#include "harness.hpp"
#include "conv_common.cuh"

namespace rpconv {
int strip_waves(int) { return 0; }
void strip_allow_two_wave(int) {}
void strip_force_ni(int) {}
void strip_allow_small(int) {}
void strip_allow_s2(int) {}
long long strip_s2_halfs(int, int, int, int, int, int) { return 0; }
int strip_rows(int, int, int, int, int, int, int, int) { return 0; }
int strip_tiles_per_image(int, int, int, int, int) { return 0; }
int strip_launch(KParams&, int, int, int, int, bool, bool, int, hipStream_t) { return 1; }
void strip_pack(const float*, _Float16*, const PackParams&, hipStream_t) {}
}  // namespace rpconv
