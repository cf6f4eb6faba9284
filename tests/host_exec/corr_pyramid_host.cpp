// csrc/corr_pyramid.hip compiled for the host (see harness.hpp): the volume / pyramid build, MFMA collectives emulated per wave.
#include "harness.hpp"
#include "corr_pyramid.hip"
