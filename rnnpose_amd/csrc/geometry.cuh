// Per-pixel projective geometry shared by the induced-flow and LM kernels.
// Mirrors the fp32 operation order of geometry/projective_ops.py:68-114 and
// geometry/transformation.py:78-86 (einsum "aijk,ai...k->ai...j" on homogeneous points).
#pragma once
#include "common.hpp"

// keep the reference's rounding points: no fma contraction in the geometric chain -- for the whole including file (pointwise.hip,
// lm.hip: their own expressions rely on it), or, with RP_CONTRACT_LOCAL defined by the including file, inside the functions of this
// header and of induced.cuh only (corr_lookup.hip, nhwc_ops.hip since r06: their own multiply-adds keep contracting)
#ifndef RP_CONTRACT_LOCAL
#pragma clang fp contract(off)
#endif

namespace rp {

struct Reproj {
  float X0, Y0, Z0;   // back-projected point
  float X1, Y1, Z1;   // transformed point (unclamped)
  float Zc;           // max(Z1, 0.01)
  float u, v;         // projection
};

__device__ __forceinline__ Reproj reproject(float Z, float x, float y, const Intr& k, const Pose& g) {
#pragma clang fp contract(off)
  Reproj r;
  r.Z0 = Z;
  r.X0 = Z * (x - k.cx) / k.fx;                                   // projective_ops.py:87-88
  r.Y0 = Z * (y - k.cy) / k.fy;
  r.X1 = g.r[0] * r.X0 + g.r[1] * r.Y0 + g.r[2] * r.Z0 + g.r[3];  // transformation.py:83-85
  r.Y1 = g.r[4] * r.X0 + g.r[5] * r.Y0 + g.r[6] * r.Z0 + g.r[7];
  r.Z1 = g.r[8] * r.X0 + g.r[9] * r.Y0 + g.r[10] * r.Z0 + g.r[11];
  r.Zc = fmaxf(r.Z1, kMinDepthProj);                              // projective_ops.py:107
  r.u = k.fx * (r.X1 / r.Zc) + k.cx;                              // :112-113
  r.v = k.fy * (r.Y1 / r.Zc) + k.cy;
  return r;
}

}  // namespace rp
