// SURVEY.md section 8(f4) "next" row: the zoom-crop of every outer refinement iteration, entirely on device.
//   model/PoseRefiner.py:145-203  get_affine_transformation  (mask bounding box -> crop window -> affine matrices)
//   model/PoseRefiner.py:205-218  gen_zoom_crop_grids        (projected model centre, F.affine_grid, cropped intrinsics)
//   model/PoseRefiner.py:286-291  F.grid_sample(image, grids), F.grid_sample(geofea_2d, grids)
// The reference pulls the foreground mask and the crop centre to the host (two D2H syncs per outer iteration,
// PoseRefiner.py:154,213), loops over the batch in numpy and calls cv2.getAffineTransform on axis-aligned point
// triples.  Here: a min/max reduction of the mask (integer atomics, order independent -> deterministic), one thread per
// image for the window / affine / intrinsics arithmetic (cv2's 3-point solve in closed form, same float32 rounding
// points as the reference: centre in fp32, window in fp64, the point triples rounded to fp32 before the solve), and a
// fused affine_grid + bilinear grid_sample (align_corners=False, zero padding: the defaults the reference relies on).
#include "common.hpp"

#include <climits>

#pragma clang fp contract(off)

namespace {

// bbox: (B,4) int32 [xmin, ymin, xmax, ymax], initialised to [INT_MAX, INT_MAX, -1, -1] by bbox_init_kernel
__global__ void bbox_init_kernel(int* __restrict__ bbox, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 4) bbox[i] = (i & 3) < 2 ? INT_MAX : -1;
}

__global__ __launch_bounds__(256) void mask_bbox_kernel(const float* __restrict__ depth, int* __restrict__ bbox, int H, int W,
                                                        int rows_per_block) {
  const int b = blockIdx.y;
  const int y0 = blockIdx.x * rows_per_block;
  const int y1 = min(H, y0 + rows_per_block);
  int xmin = INT_MAX, ymin = INT_MAX, xmax = -1, ymax = -1;
  const float* d = depth + static_cast<long long>(b) * H * W;
  for (int i = y0 * W + threadIdx.x; i < y1 * W; i += 256) {
    if (d[i] > 0.f) {                              // fg_mask = pc_depth > 0 (PoseRefiner.py:259)
      const int y = i / W, x = i - y * W;
      xmin = min(xmin, x); xmax = max(xmax, x);
      ymin = min(ymin, y); ymax = max(ymax, y);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    xmin = min(xmin, __shfl_xor(xmin, o)); ymin = min(ymin, __shfl_xor(ymin, o));
    xmax = max(xmax, __shfl_xor(xmax, o)); ymax = max(ymax, __shfl_xor(ymax, o));
  }
  if ((threadIdx.x & 63) == 0 && xmax >= 0) {
    atomicMin(bbox + b * 4 + 0, xmin); atomicMin(bbox + b * 4 + 1, ymin);
    atomicMax(bbox + b * 4 + 2, xmax); atomicMax(bbox + b * 4 + 3, ymax);
  }
}

// one thread per image
__global__ void zoom_params_kernel(const int* __restrict__ bbox, const float* __restrict__ K, const float* __restrict__ T,
                                   int B, int H, int W, int Hc, int Wc, float margin_ratio, float* __restrict__ theta,
                                   float* __restrict__ K_crop) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* k = K + b * 9;
  const float* t = T + b * 16;
  // crop_center = (K @ T[:, :3, 3:])[:2] / [2]   (fp32, PoseRefiner.py:207-208)
  const float tx = t[3], ty = t[7], tz = t[11];
  const float px = k[0] * tx + k[1] * ty + k[2] * tz;
  const float py = k[3] * tx + k[4] * ty + k[5] * tz;
  const float pz = k[6] * tx + k[7] * ty + k[8] * tz;
  const double cx = static_cast<double>(px / pz), cy = static_cast<double>(py / pz);
  int x0 = bbox[b * 4 + 0], y0 = bbox[b * 4 + 1], x1 = bbox[b * 4 + 2], y1 = bbox[b * 4 + 3];
  if (x1 < 0) { x0 = y0 = x1 = y1 = 0; }           // empty mask (PoseRefiner.py:160-164)
  const double ratio = static_cast<double>(H) / static_cast<double>(W);
  const double left = cx - x0, right = x1 - cx, up = cy - y0, down = y1 - cy;
  const double crop_h = fmax(fmax(ratio * right, ratio * left), fmax(up, down)) * 2.0 * (1.0 + static_cast<double>(margin_ratio));
  const double crop_w = crop_h / ratio;
  {  // normalised window -> theta of F.affine_grid: cv2.getAffineTransform([-1,-1],[-1,1],[1,-1] -> fp32 corner points)
    const double nx1 = static_cast<double>(static_cast<float>((cx - crop_w / 2) * 2 / W - 1));
    const double nx2 = static_cast<double>(static_cast<float>((cx + crop_w / 2) * 2 / W - 1));
    const double ny1 = static_cast<double>(static_cast<float>((cy - crop_h / 2) * 2 / H - 1));
    const double ny2 = static_cast<double>(static_cast<float>((cy + crop_h / 2) * 2 / H - 1));
    float* th = theta + b * 6;
    th[0] = static_cast<float>((nx2 - nx1) / 2); th[1] = 0.f; th[2] = static_cast<float>((nx1 + nx2) / 2);
    th[3] = 0.f; th[4] = static_cast<float>((ny2 - ny1) / 2); th[5] = static_cast<float>((ny1 + ny2) / 2);
  }
  {  // pixel window -> intrinsics of the crop: inverse(pad(getAffineTransform([0,0],[0,Hc-1],[Wc-1,0] -> corners))) @ K
    const double wx1 = static_cast<double>(static_cast<float>(cx - crop_w / 2));
    const double wx2 = static_cast<double>(static_cast<float>(cx + crop_w / 2));
    const double wy1 = static_cast<double>(static_cast<float>(cy - crop_h / 2));
    const double wy2 = static_cast<double>(static_cast<float>(cy + crop_h / 2));
    const float a = static_cast<float>((wx2 - wx1) / (Wc - 1)), d = static_cast<float>((wy2 - wy1) / (Hc - 1));
    const float ox = static_cast<float>(wx1), oy = static_cast<float>(wy1);
    // inverse of [[a,0,ox],[0,d,oy],[0,0,1]] (fp32, as torch.inverse on the fp32 matrix), then @ K
    const float ia = 1.f / a, id = 1.f / d, iox = -ox / a, ioy = -oy / d;
    float* o = K_crop + b * 9;
    for (int j = 0; j < 3; ++j) {
      o[0 + j] = ia * k[0 + j] + iox * k[6 + j];
      o[3 + j] = id * k[3 + j] + ioy * k[6 + j];
      o[6 + j] = k[6 + j];
    }
  }
}

// out[b,c,y,x] = bilinear(in[b,c], grid(theta_b; x, y)), zero padding, align_corners=False for both the grid and the
// sampler (torch defaults).  One thread per output pixel, channel loop inside (the 4 taps + weights are shared).
__global__ __launch_bounds__(256) void zoom_crop_kernel(const float* __restrict__ in, const float* __restrict__ theta,
                                                        float* __restrict__ out, float* __restrict__ grid_out, int C, int H,
                                                        int W, int Hc, int Wc) {
  const int b = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= Wc || y >= Hc) return;
  const float* th = theta + b * 6;
  const float bx = (2.f * x + 1.f) / Wc - 1.f, by = (2.f * y + 1.f) / Hc - 1.f;     // affine_grid base, align_corners=False
  const float gx = th[0] * bx + th[1] * by + th[2];
  const float gy = th[3] * bx + th[4] * by + th[5];
  if (grid_out) {
    float* g = grid_out + ((static_cast<long long>(b) * Hc + y) * Wc + x) * 2;
    g[0] = gx;
    g[1] = gy;
  }
  if (!out) return;
  const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;   // grid_sample unnormalise
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
  const bool sane = fabsf(ix) < 1.0e8f && fabsf(iy) < 1.0e8f;
  const int x0 = sane ? static_cast<int>(fx0) : -10, y0 = sane ? static_cast<int>(fy0) : -10;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  // weights in the operation order of torch's grid_sampler (nw, ne, sw, se)
  const float w00 = (fx1 - ix) * (fy1 - iy), w10 = (ix - fx0) * (fy1 - iy), w01 = (fx1 - ix) * (iy - fy0), w11 = (ix - fx0) * (iy - fy0);
  const long long plane = static_cast<long long>(H) * W;
  const float* src = in + static_cast<long long>(b) * C * plane;
  float* dst = out + (static_cast<long long>(b) * C * Hc + y) * Wc + x;
  const long long o00 = static_cast<long long>(y0) * W + x0;
  for (int c = 0; c < C; ++c) {
    const float* p = src + c * plane;
    float v = 0.f;
    if (vy0 && vx0) v += p[o00] * w00;
    if (vy0 && vx1) v += p[o00 + 1] * w10;
    if (vy1 && vx0) v += p[o00 + W] * w01;
    if (vy1 && vx1) v += p[o00 + W + 1] * w11;
    dst[static_cast<long long>(c) * Hc * Wc] = v;
  }
}

// ---- point-cloud depth splat: DiffRender.render_pointcloud (geometry/diff_render_optim.py:369-401) ---------------------
// Every model vertex is projected (X_cam = R v + t, x = K X_cam, fp32, no fma contraction), its pixel is
// round-half-even(x/z, y/z) CLAMPED into the image (the reference clamps, so off-screen vertices land on the border) and
// receives the vertex depth.  Which of several vertices on one pixel wins is unspecified in the reference (torch's indexed
// assignment with repeated indices); here: pass 1 elects the highest vertex index per pixel (atomicMax, order
// independent), pass 2 lets that vertex write its depth -- deterministic, = sequential last-writer-wins.  Only `depth > 0` (the foreground mask of the
// zoom crop, PoseRefiner.py:259) is consumed downstream.
__device__ __forceinline__ bool pc_project(const float* __restrict__ v, const float* __restrict__ T, const float* __restrict__ K,
                                           int H, int W, int& px, int& py, float& depth) {
  const float X = (v[0] * T[0] + v[1] * T[1]) + v[2] * T[2] + T[3];
  const float Y = (v[0] * T[4] + v[1] * T[5]) + v[2] * T[6] + T[7];
  const float Z = (v[0] * T[8] + v[1] * T[9]) + v[2] * T[10] + T[11];
  const float x = (X * K[0] + Y * K[1]) + Z * K[2];
  const float y = (X * K[3] + Y * K[4]) + Z * K[5];
  const float z = (X * K[6] + Y * K[7]) + Z * K[8];
  depth = z;
  const float u = rintf(x / z), w = rintf(y / z);          // rintf = round half to even (torch.round)
  // non-finite -> the reference's .long() yields INT64_MIN, which the clamp turns into 0
  px = (u == u && fabsf(u) < 1.0e9f) ? min(max(static_cast<int>(u), 0), W - 1) : 0;
  py = (w == w && fabsf(w) < 1.0e9f) ? min(max(static_cast<int>(w), 0), H - 1) : 0;
  return true;
}

__global__ __launch_bounds__(256) void pc_owner_kernel(const float* __restrict__ verts, const int* __restrict__ offs,
                                                       const float* __restrict__ T, const float* __restrict__ K,
                                                       int* __restrict__ owner, int H, int W) {
  const int b = blockIdx.y;
  const int p0 = offs[b], n = offs[b + 1] - p0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int px, py;
  float d;
  pc_project(verts + 3LL * (p0 + i), T + b * 16, K + b * 9, H, W, px, py, d);
  atomicMax(owner + (static_cast<long long>(b) * H + py) * W + px, i + 1);
}

__global__ __launch_bounds__(256) void pc_write_kernel(const float* __restrict__ verts, const int* __restrict__ offs,
                                                       const float* __restrict__ T, const float* __restrict__ K,
                                                       const int* __restrict__ owner, float* __restrict__ out, int H, int W) {
  const int b = blockIdx.y;
  const int p0 = offs[b], n = offs[b + 1] - p0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int px, py;
  float d;
  pc_project(verts + 3LL * (p0 + i), T + b * 16, K + b * 9, H, W, px, py, d);
  const long long o = (static_cast<long long>(b) * H + py) * W + px;
  if (owner[o] == i + 1) out[o] = d;
}

}  // namespace

extern "C" {

int rnnpose_mask_bbox_f32(const float* depth, int B, int H, int W, int* bbox, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_mask_bbox_f32";
  RP_REQUIRE(depth && bbox, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && static_cast<long long>(H) * W < (1LL << 31), fn, "bad size");
  hipLaunchKernelGGL(bbox_init_kernel, dim3(rp::cdiv(B * 4, 256)), dim3(256), 0, rp::as_stream(stream), bbox, B);
  const int rows = 16;
  hipLaunchKernelGGL(mask_bbox_kernel, dim3(rp::cdiv(H, rows), B), dim3(256), 0, rp::as_stream(stream), depth, bbox, H, W, rows);
  return rp::check_launch(fn);
}

int rnnpose_zoom_crop_params_f32(const int* bbox, const float* K, const float* T, int B, int H, int W, int crop_h, int crop_w,
                                 float margin_ratio, float* theta, float* K_crop, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_zoom_crop_params_f32";
  RP_REQUIRE(bbox && K && T && theta && K_crop, fn, "null pointer");
  RP_REQUIRE(B > 0 && H > 0 && W > 0 && crop_h > 1 && crop_w > 1, fn, "bad size (crop sizes must be > 1)");
  hipLaunchKernelGGL(zoom_params_kernel, dim3(rp::cdiv(B, 64)), dim3(64), 0, rp::as_stream(stream), bbox, K, T, B, H, W, crop_h,
                     crop_w, margin_ratio, theta, K_crop);
  return rp::check_launch(fn);
}

int rnnpose_zoom_crop_f32(const float* in, const float* theta, int B, int C, int H, int W, int crop_h, int crop_w, float* out,
                          float* grid_out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_zoom_crop_f32";
  RP_REQUIRE(theta && (out || grid_out) && (in || !out), fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && C >= 0 && H > 0 && W > 0 && crop_h > 0 && crop_w > 0 && crop_h < 262144, fn, "bad size");
  hipLaunchKernelGGL(zoom_crop_kernel, dim3(rp::cdiv(crop_w, 64), rp::cdiv(crop_h, 4), B), dim3(256), 0, rp::as_stream(stream),
                     in, theta, out, grid_out, C, H, W, crop_h, crop_w);
  return rp::check_launch(fn);
}

size_t rnnpose_pointcloud_depth_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return static_cast<size_t>(B) * H * W * sizeof(int);
}

int rnnpose_pointcloud_depth_f32(const float* verts, const int* vert_offsets, int max_verts, const float* T, const float* K, int B,
                                 int H, int W, void* workspace, size_t workspace_bytes, float* out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_pointcloud_depth_f32";
  RP_REQUIRE(verts && vert_offsets && T && K && workspace && out, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && max_verts > 0, fn, "bad size");
  RP_REQUIRE(workspace_bytes >= rnnpose_pointcloud_depth_workspace_bytes(B, H, W), fn, "workspace too small");
  hipStream_t st = rp::as_stream(stream);
  if (hipMemsetAsync(workspace, 0, rnnpose_pointcloud_depth_workspace_bytes(B, H, W), st) != hipSuccess ||
      hipMemsetAsync(out, 0, static_cast<size_t>(B) * H * W * sizeof(float), st) != hipSuccess)
  {
    rp::set_error("%s: hipMemsetAsync failed", fn);
    return 2;
  }
  const dim3 grid(rp::cdiv(max_verts, 256), B);
  hipLaunchKernelGGL(pc_owner_kernel, grid, dim3(256), 0, st, verts, vert_offsets, T, K, static_cast<int*>(workspace), H, W);
  hipLaunchKernelGGL(pc_write_kernel, grid, dim3(256), 0, st, verts, vert_offsets, T, K, static_cast<const int*>(workspace), out, H, W);
  return rp::check_launch(fn);
}

}  // extern "C"
