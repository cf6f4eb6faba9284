// f4 (second half): triangle-mesh rasteriser for the render hand-off of every outer refinement iteration -- depth and
// per-vertex attribute maps of the object under the current pose.  Replaces what the reference gets from PyTorch3D through
// geometry/diff_render_optim.py:283-367 (`DiffRender.forward`: MeshRasterizer with faces_per_pixel = 1, blur 0,
// perspective-correct barycentrics + interpolate_face_attributes; `render_depth`: nearest-vertex depth).
// PARITY UNPINNED: PyTorch3D is not in this image and the reference holds no fixture for it; semantics follow PyTorch3D's
// documented ones (pixel centres at +0.5, nearest face by interpolated camera z, ties -> lower face index) and are checked
// against oracle/raster_oracle.py and through properties (tests/test_raster.py).
//
// Two passes, no sorting, no per-pixel face lists:
//   1. raster_faces_kernel: one thread per (image, face).  The face is projected, its screen bounding box walked, and for
//      every covered pixel centre a 64-bit key (depth bits << 32 | face index) is atomicMin-ed into the z-buffer: depth
//      is positive, so the float's bit pattern orders like the float, and equal depths resolve to the lower face index.
//      LINEMOD-size meshes project to triangles of a few pixels: the box walk is short and the atomics rarely collide.
//   2. raster_resolve_kernel: one thread per pixel reads its key, recomputes the barycentrics of the winning face with the
//      SAME arithmetic as pass 1 and writes depth, nearest-vertex depth and the interpolated vertex attributes
//      ([shaded colour |] C channels, NCHW planes: consecutive lanes are consecutive x -> coalesced plane writes; every
//      lane walks its three attribute rows 16 bytes at a time).
#include "common.hpp"

namespace {

struct Cam {
  float r[12];
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ T, const float* __restrict__ K, int b) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 12; ++i) c.r[i] = T[16 * b + i];
  const float* k = K + 9 * b;
  c.fx = k[0]; c.fy = k[4]; c.cx = k[2]; c.cy = k[5];
  return c;
}

struct Tri {
  float x[3], y[3], z[3];   // screen x, y (pixels) and camera z of the three vertices
  float area;               // signed doubled area in screen space
  bool ok;
};

// shared by both passes: identical arithmetic -> identical coverage decisions
__device__ __forceinline__ Tri project_face(const float* __restrict__ verts, const int* __restrict__ faces, long long f,
                                            int vbase, const Cam& c, float near) {
#pragma clang fp contract(off)
  Tri t;
  t.ok = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float* v = verts + 3LL * (vbase + faces[3 * f + i]);
    const float X = c.r[0] * v[0] + c.r[1] * v[1] + c.r[2] * v[2] + c.r[3];
    const float Y = c.r[4] * v[0] + c.r[5] * v[1] + c.r[6] * v[2] + c.r[7];
    const float Z = c.r[8] * v[0] + c.r[9] * v[1] + c.r[10] * v[2] + c.r[11];
    t.z[i] = Z;
    t.ok = t.ok && (Z > near);                     // faces that reach behind the near plane are dropped whole
    const float iz = 1.0f / (Z > near ? Z : near);
    t.x[i] = c.fx * X * iz + c.cx;
    t.y[i] = c.fy * Y * iz + c.cy;
  }
  t.area = (t.x[1] - t.x[0]) * (t.y[2] - t.y[0]) - (t.x[2] - t.x[0]) * (t.y[1] - t.y[0]);
  t.ok = t.ok && (fabsf(t.area) > 1e-8f) && (t.area == t.area);
  return t;
}

// barycentric weights of pixel centre (px, py); inside <=> all three >= 0 (either winding).  -> depth of the face there
__device__ __forceinline__ bool bary_at(const Tri& t, float px, float py, int perspective, float& w0, float& w1, float& w2,
                                        float& z) {
#pragma clang fp contract(off)
  const float e0 = (t.x[2] - t.x[1]) * (py - t.y[1]) - (t.y[2] - t.y[1]) * (px - t.x[1]);
  const float e1 = (t.x[0] - t.x[2]) * (py - t.y[2]) - (t.y[0] - t.y[2]) * (px - t.x[2]);
  const float e2 = (t.x[1] - t.x[0]) * (py - t.y[0]) - (t.y[1] - t.y[0]) * (px - t.x[0]);
  const float ia = 1.0f / t.area;
  w0 = e0 * ia; w1 = e1 * ia; w2 = e2 * ia;
  if (!(w0 >= 0.f && w1 >= 0.f && w2 >= 0.f)) return false;
  if (perspective) {                               // PyTorch3D BarycentricPerspectiveCorrection: b_i = (w_i / z_i) / sum
    const float q0 = w0 / t.z[0], q1 = w1 / t.z[1], q2 = w2 / t.z[2];
    const float s = (q0 + q1) + q2;
    w0 = q0 / s; w1 = q1 / s; w2 = q2 / s;
  }
  z = (w0 * t.z[0] + w1 * t.z[1]) + w2 * t.z[2];
  return z > 0.f;
}

constexpr unsigned long long kEmpty = 0xffffffffffffffffull;

__global__ __launch_bounds__(256) void raster_clear_kernel(unsigned long long* __restrict__ zb, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) zb[i] = kEmpty;
}

__global__ __launch_bounds__(128) void raster_faces_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                           const int* __restrict__ vert_off, const int* __restrict__ face_off,
                                                           const int* __restrict__ face_cnt, const float* __restrict__ T,
                                                           const float* __restrict__ K, int H, int W, float near,
                                                           float pix_center, int perspective,
                                                           unsigned long long* __restrict__ zb) {
  const int b = blockIdx.y;
  const int fl = blockIdx.x * 128 + threadIdx.x;
  if (fl >= face_cnt[b]) return;
  const Cam c = load_cam(T, K, b);
  const long long f = static_cast<long long>(face_off[b]) + fl;
  const Tri t = project_face(verts, faces, f, vert_off[b], c, near);
  if (!t.ok) return;
  const float xmin = fminf(fminf(t.x[0], t.x[1]), t.x[2]), xmax = fmaxf(fmaxf(t.x[0], t.x[1]), t.x[2]);
  const float ymin = fminf(fminf(t.y[0], t.y[1]), t.y[2]), ymax = fmaxf(fmaxf(t.y[0], t.y[1]), t.y[2]);
  if (!(xmax >= 0.f && ymax >= 0.f && xmin < static_cast<float>(W) && ymin < static_cast<float>(H))) return;
  const int x0 = max(0, static_cast<int>(floorf(xmin - pix_center))), x1 = min(W - 1, static_cast<int>(ceilf(xmax - pix_center)));
  const int y0 = max(0, static_cast<int>(floorf(ymin - pix_center))), y1 = min(H - 1, static_cast<int>(ceilf(ymax - pix_center)));
  unsigned long long* img = zb + static_cast<long long>(b) * H * W;
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      float w0, w1, w2, z;
      if (bary_at(t, static_cast<float>(x) + pix_center, static_cast<float>(y) + pix_center, perspective, w0, w1, w2, z)) {
        const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(z)) << 32) | static_cast<unsigned>(fl);
        atomicMin(img + static_cast<long long>(y) * W + x, key);
      }
    }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const float* __restrict__ verts, const int* __restrict__ faces,
                                                             const int* __restrict__ vert_off, const int* __restrict__ face_off,
                                                             const float* __restrict__ T, const float* __restrict__ K, int H,
                                                             int W, float near, float pix_center, int perspective,
                                                             const unsigned long long* __restrict__ zb,
                                                             const float* __restrict__ attr, const long long* __restrict__ attr_off,
                                                             int C, const float* __restrict__ colors, int n_col, int shade,
                                                             float empty_depth, float* __restrict__ out_attr,
                                                             float* __restrict__ out_zbuf, float* __restrict__ out_vdepth) {
  const int b = blockIdx.y;
  const long long P = static_cast<long long>(H) * W;
  const long long pix = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (pix >= P) return;
  const int x = static_cast<int>(pix % W), y = static_cast<int>(pix / W);
  const unsigned long long key = zb[b * P + pix];
  const bool hit = key != kEmpty;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, z = 0.f;
  int v0 = 0, v1 = 0, v2 = 0;
  Tri t{};
  const int vb = vert_off[b];
  if (hit) {
    const Cam c = load_cam(T, K, b);
    const long long f = static_cast<long long>(face_off[b]) + static_cast<unsigned>(key & 0xffffffffull);
    t = project_face(verts, faces, f, vb, c, near);
    bary_at(t, static_cast<float>(x) + pix_center, static_cast<float>(y) + pix_center, perspective, w0, w1, w2, z);
    v0 = faces[3 * f + 0]; v1 = faces[3 * f + 1]; v2 = faces[3 * f + 2];
  }
  if (out_zbuf) out_zbuf[b * P + pix] = hit ? z : empty_depth;
  if (out_vdepth) {       // render_depth: per-vertex camera z with the barycentrics snapped to the nearest vertex (argmax)
    float vz = 0.f;
    if (hit) vz = (w0 >= w1 && w0 >= w2) ? t.z[0] : (w1 >= w2 ? t.z[1] : t.z[2]);
    out_vdepth[b * P + pix] = vz;
  }
  if (!out_attr) return;
  float* o = out_attr + static_cast<long long>(b) * (n_col + C) * P + pix;
  if (n_col) {
    float col[3] = {0.f, 0.f, 0.f};
    if (hit) {
      float lit = 1.f, spec = 0.f;
      const float* p0 = verts + 3LL * (vb + v0);
      const float* p1 = verts + 3LL * (vb + v1);
      const float* p2 = verts + 3LL * (vb + v2);
      if (shade) {        // Phong terms of PyTorch3D's defaults with shininess 0, flat normal, point light at (1, 1, -1)
        const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
        const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
        float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
        const float nn = rsqrtf(fmaxf(nx * nx + ny * ny + nz * nz, 1e-30f));
        nx *= nn; ny *= nn; nz *= nn;
        const float qx = w0 * p0[0] + w1 * p1[0] + w2 * p2[0], qy = w0 * p0[1] + w1 * p1[1] + w2 * p2[1],
                    qz = w0 * p0[2] + w1 * p1[2] + w2 * p2[2];
        float lx = 1.f - qx, ly = 1.f - qy, lz = -1.f - qz;
        const float ln = rsqrtf(fmaxf(lx * lx + ly * ly + lz * lz, 1e-30f));
        const float cosv = fabsf((nx * lx + ny * ly + nz * lz) * ln);      // two-sided: the winding of scanned meshes varies
        lit = 0.5f + 0.3f * cosv;
        spec = 0.2f;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float a = 1.f;
        if (colors) a = w0 * colors[3LL * (vb + v0) + k] + w1 * colors[3LL * (vb + v1) + k] + w2 * colors[3LL * (vb + v2) + k];
        col[k] = a * lit + spec;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k * P] = col[k];
    o += 3 * P;
  }
  if (!attr || C <= 0) return;
  const float* a0 = attr + attr_off[b] + static_cast<long long>(v0) * C;
  const float* a1 = attr + attr_off[b] + static_cast<long long>(v1) * C;
  const float* a2 = attr + attr_off[b] + static_cast<long long>(v2) * C;
  int c = 0;
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(attr) & 15) == 0 && (attr_off[b] & 3) == 0) {
    for (; c < C; c += 4) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hit) {
        const float4 q0 = *reinterpret_cast<const float4*>(a0 + c), q1 = *reinterpret_cast<const float4*>(a1 + c),
                     q2 = *reinterpret_cast<const float4*>(a2 + c);
        r = make_float4(w0 * q0.x + w1 * q1.x + w2 * q2.x, w0 * q0.y + w1 * q1.y + w2 * q2.y,
                        w0 * q0.z + w1 * q1.z + w2 * q2.z, w0 * q0.w + w1 * q1.w + w2 * q2.w);
      }
      o[(c + 0) * P] = r.x; o[(c + 1) * P] = r.y; o[(c + 2) * P] = r.z; o[(c + 3) * P] = r.w;
    }
  }
  for (; c < C; ++c) o[c * P] = hit ? w0 * a0[c] + w1 * a1[c] + w2 * a2[c] : 0.f;
}

}  // namespace

extern "C" {

size_t rnnpose_raster_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return static_cast<size_t>(B) * H * W * sizeof(unsigned long long);
}

int rnnpose_raster_mesh_f32(const float* verts, const int* faces, const int* vert_off, const int* face_off,
                            const int* face_cnt, int max_faces, const float* T, const float* K, int B, int H, int W,
                            float near, float pixel_center, int perspective_correct, void* workspace,
                            size_t workspace_bytes, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_raster_mesh_f32";
  RP_REQUIRE(verts && faces && vert_off && face_off && face_cnt && T && K && workspace, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && max_faces > 0 && near > 0.f, fn, "bad size");
  RP_REQUIRE(workspace_bytes >= rnnpose_raster_workspace_bytes(B, H, W), fn, "workspace too small");
  hipStream_t st = rp::as_stream(stream);
  unsigned long long* zb = static_cast<unsigned long long*>(workspace);
  const long long n = static_cast<long long>(B) * H * W;
  hipLaunchKernelGGL(raster_clear_kernel, dim3(rp::cdiv(n, 256)), dim3(256), 0, st, zb, n);
  hipLaunchKernelGGL(raster_faces_kernel, dim3(rp::cdiv(max_faces, 128), B), dim3(128), 0, st, verts, faces, vert_off, face_off,
                     face_cnt, T, K, H, W, near, pixel_center, perspective_correct, zb);
  return rp::check_launch(fn);
}

int rnnpose_raster_resolve_f32(const float* verts, const int* faces, const int* vert_off, const int* face_off, const float* T,
                               const float* K, int B, int H, int W, float near, float pixel_center, int perspective_correct,
                               const void* workspace, const float* attr, const long long* attr_off, int C,
                               const float* colors, int with_color, int shade, float empty_depth, float* out_attr,
                               float* out_zbuf, float* out_vdepth, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_raster_resolve_f32";
  RP_REQUIRE(verts && faces && vert_off && face_off && T && K && workspace, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && C >= 0, fn, "bad size");
  RP_REQUIRE(!(attr && C > 0) || attr_off, fn, "attr needs attr_off");
  RP_REQUIRE(out_attr || out_zbuf || out_vdepth, fn, "nothing to write");
  const long long P = static_cast<long long>(H) * W;
  hipLaunchKernelGGL(raster_resolve_kernel, dim3(rp::cdiv(P, 256), B), dim3(256), 0, rp::as_stream(stream), verts, faces, vert_off,
                     face_off, T, K, H, W, near, pixel_center, perspective_correct,
                     static_cast<const unsigned long long*>(workspace), attr, attr_off, C, colors, with_color ? 3 : 0, shade,
                     empty_depth, out_attr, out_zbuf, out_vdepth);
  return rp::check_launch(fn);
}

}  // extern "C"
