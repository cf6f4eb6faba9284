// csrc/pointwise.hip compiled for the host (see harness.hpp): its C-ABI entry points run their kernels on host pointers.
#include "harness.hpp"
#include "pointwise.hip"
