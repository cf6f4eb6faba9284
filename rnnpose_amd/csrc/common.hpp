// Shared helpers for librnnpose_hip.so (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "rnnpose_hip.h"

namespace rp {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* fmt, ...);
int fail_arg(const char* fn, const char* what);
int check_launch(const char* fn);
int cu_count();                      // compute units of the current device (cached per device; 256 on MI355X)
unsigned long long* sat_counter();   // fp16x3 range guard: device counter of the current device, NULL while the check is off

inline hipStream_t as_stream(rnnpose_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int cdiv(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

#define RP_REQUIRE(cond, fn, msg)          \
  do {                                      \
    if (!(cond)) return rp::fail_arg(fn, msg); \
  } while (0)

// constants of the reference, with their sources
constexpr float kMinDepthValid = 0.1f;   // geometry/transformation.py:16
constexpr float kMinDepthProj = 0.01f;   // geometry/projective_ops.py:9
constexpr float kMinTheta = 1e-4f;       // geometry/se3.py:10

__device__ __forceinline__ double shfl_xor_f64(double v, int o) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, o);
  hi = __shfl_xor(hi, o);
  return __hiloint2double(hi, lo);
}

struct Intr {
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Intr load_intr(const float* __restrict__ K, int b) {
  const float* k = K + 9 * b;
  return Intr{k[0], k[4], k[2], k[5]};
}

struct Pose {
  float r[12];  // rows of [R|t]
};

__device__ __forceinline__ Pose load_pose(const float* __restrict__ G, int b) {
  Pose p;
#pragma unroll
  for (int i = 0; i < 12; ++i) p.r[i] = G[16 * b + i];
  return p;
}

}  // namespace rp
