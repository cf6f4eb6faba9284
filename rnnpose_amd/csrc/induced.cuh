// a7 + a5 at ONE 1/8-resolution pixel: the pose-induced flow re-projected at the 4 full-resolution taps the align_corners
// down-sampling needs, divided by the scale, interpolated, added to the pixel grid (geometry/transformation.py:184-198,
// model/PoseRefiner.py:324-328, model/CFNet.py:136-144).  Shared by induced_coords_lowres_kernel (pointwise.hip) and -- r06 -- by the two
// kernels that consume those coordinates first in an iteration, the window lookup (corr_lookup.hip) and the flow-feature convolution
// (nhwc_ops.hip), which evaluate it themselves instead of reading the output of a 5-us launch: same function, same operation order (no
// fma contraction inside, whatever the including file's setting), so the coordinates are bit-identical wherever they are formed.
#pragma once
#include "geometry.cuh"

namespace rp {

// bilinear resize, align_corners=True: src = dst * (in-1)/(out-1)   (F.interpolate semantics)
struct AcTap {
  int i0, i1;
  float f;
};
__device__ __forceinline__ AcTap ac_tap(int dst, int in, int out) {
#pragma clang fp contract(off)
  const float scale = out > 1 ? static_cast<float>(in - 1) / static_cast<float>(out - 1) : 0.f;
  const float s = scale * static_cast<float>(dst);
  int i0 = static_cast<int>(s);            // s >= 0
  if (i0 > in - 1) i0 = in - 1;
  const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
  return AcTap{i0, i1, s - static_cast<float>(i0)};
}

__device__ __forceinline__ float2 flow_at(const float* __restrict__ depth_b, int x, int y, int W, float eps, const Intr& k,
                                          const Pose& g) {
#pragma clang fp contract(off)
  const float Z = depth_b[static_cast<long long>(y) * W + x] + eps;
  const Reproj r = reproject(Z, static_cast<float>(x), static_cast<float>(y), k, g);
  const float fg = Z > eps ? 1.f : 0.f;
  return make_float2((r.u - static_cast<float>(x)) * fg, (r.v - static_cast<float>(y)) * fg);
}

// coords1 of low-resolution pixel (X, Y) of an (h, w) map over an (H, W) depth map of one image
__device__ __forceinline__ float2 induced_coords_at(const float* __restrict__ depth_b, int X, int Y, int H, int W, int h, int w, float eps,
                                                    const Intr& k, const Pose& g) {
#pragma clang fp contract(off)
  const float ds = static_cast<float>(W / w);
  const AcTap ty = ac_tap(Y, H, h), tx = ac_tap(X, W, w);
  const float2 f00 = flow_at(depth_b, tx.i0, ty.i0, W, eps, k, g), f01 = flow_at(depth_b, tx.i1, ty.i0, W, eps, k, g);
  const float2 f10 = flow_at(depth_b, tx.i0, ty.i1, W, eps, k, g), f11 = flow_at(depth_b, tx.i1, ty.i1, W, eps, k, g);
  const float topx = (f00.x / ds) * (1.f - tx.f) + (f01.x / ds) * tx.f, botx = (f10.x / ds) * (1.f - tx.f) + (f11.x / ds) * tx.f;
  const float topy = (f00.y / ds) * (1.f - tx.f) + (f01.y / ds) * tx.f, boty = (f10.y / ds) * (1.f - tx.f) + (f11.y / ds) * tx.f;
  return make_float2(static_cast<float>(X) + (topx * (1.f - ty.f) + botx * ty.f), static_cast<float>(Y) + (topy * (1.f - ty.f) + boty * ty.f));
}

// where the coordinates of a launch come from when its first consumer forms them itself: depth (B,1,H,W), K (B,3,3), G (B,4,4) of the
// launch's images (sub-batch pointers), depth == nullptr: the launch reads a coordinate tensor as before
struct InducedSrc {
  const float* depth;
  const float* K;
  const float* G;
  int H, W;
  float eps;
};

}  // namespace rp
