// SURVEY.md section 8(f2,f3) "next" rows: the evaluator's per-sample metrics on device.
//   f3  brute-force nearest neighbour for ADD-S      thirdparty/nn/src/nearest_neighborhood.cu:48-163 (the only CUDA
//       kernel of the reference) + its cffi C ABI findNearestPointIdxLauncher (thirdparty/nn/src/ext.h:1-10)
//   f2  ADD / ADD-S / 2-D projection / 5cm-5deg       utils/eval_metric.py:28-37,102-192
// The reference kernel runs one thread per query with a serial loop over ALL reference points straight from global
// memory and re-mallocs / re-uploads every call.  Here reference points are staged through LDS in tiles shared by a
// 256-query workgroup (each reference point is read from HBM once per workgroup instead of once per thread), points
// stay device-resident, and the metric reductions happen in the same call.
// Semantics kept bit-exact for the index: strict '<' over ascending reference index (first minimum wins), fp32
// distance with the reference's operation order (no fma contraction).
#include "common.hpp"

#include <cfloat>
#include <cstdio>
#include <cstring>

#pragma clang fp contract(off)

namespace {

constexpr int NN_TILE = 1024;

template <int DIM>
__global__ __launch_bounds__(256) void nn_search_kernel(const float* __restrict__ ref, const float* __restrict__ que,
                                                        int* __restrict__ idxs, int pn1, int pn2, int exclude_self) {
  __shared__ float tile[NN_TILE * DIM];
  const int bi = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool live = q < pn2;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (live) {
    const float* p = que + (static_cast<long long>(bi) * pn2 + q) * DIM;
    qx = p[0];
    qy = p[1];
    if (DIM == 3) qz = p[2];
  }
  float best = FLT_MAX;
  int best_i = 0;
  const float* rb = ref + static_cast<long long>(bi) * pn1 * DIM;
  for (int t0 = 0; t0 < pn1; t0 += NN_TILE) {
    const int n = min(NN_TILE, pn1 - t0);
    __syncthreads();
    for (int e = threadIdx.x; e < n * DIM; e += 256) tile[e] = rb[static_cast<long long>(t0) * DIM + e];
    __syncthreads();
    if (live) {
      for (int j = 0; j < n; ++j) {
        const int p1i = t0 + j;
        if (exclude_self && p1i == q) continue;
        const float dx = tile[j * DIM + 0] - qx, dy = tile[j * DIM + 1] - qy;
        float d = dx * dx + dy * dy;
        if (DIM == 3) {
          const float dz = tile[j * DIM + 2] - qz;
          d = d + dz * dz;
        }
        if (d < best) {
          best = d;
          best_i = p1i;
        }
      }
    }
  }
  if (live) idxs[static_cast<long long>(bi) * pn2 + q] = best_i;
}

// model (P,3) fp32, pose (B,3,4) fp32 -> pts (B,P,3) fp32 = float(R x + t computed in fp64)   (eval_metric.py:117-119)
__global__ __launch_bounds__(256) void transform_points_kernel(const float* __restrict__ model, const float* __restrict__ pose,
                                                               float* __restrict__ pts, int P) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float* T = pose + 12 * b;
  const double x = model[3 * i], y = model[3 * i + 1], z = model[3 * i + 2];
  float* o = pts + (static_cast<long long>(b) * P + i) * 3;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    o[r] = static_cast<float>(static_cast<double>(T[4 * r]) * x + static_cast<double>(T[4 * r + 1]) * y +
                              static_cast<double>(T[4 * r + 2]) * z + static_cast<double>(T[4 * r + 3]));
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int d = 32; d >= 1; d >>= 1) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_down(lo, d);
    hi = __shfl_down(hi, d);
    v += __hiloint2double(hi, lo);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < 4; ++w) s += red[w];
  return s;
}

// one workgroup per sample -> out (B,5) fp64: [ADD mean distance, ADD-S mean distance (or -1), mean 2-D projection
// error (px), translation error (cm), rotation error (deg)]
__global__ __launch_bounds__(256) void pose_metrics_kernel(const float* __restrict__ model, int P,
                                                           const float* __restrict__ pose_pred,
                                                           const float* __restrict__ pose_gt, const float* __restrict__ K,
                                                           const float* __restrict__ pts_pred, const float* __restrict__ pts_gt,
                                                           const int* __restrict__ nn_idx, double* __restrict__ out) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  const float* Tp = pose_pred + 12 * b;
  const float* Tg = pose_gt + 12 * b;
  double add = 0.0, adds = 0.0, proj = 0.0;
  for (int i = threadIdx.x; i < P; i += 256) {
    const double x = model[3 * i], y = model[3 * i + 1], z = model[3 * i + 2];
    double pp[3], pg[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      pp[r] = static_cast<double>(Tp[4 * r]) * x + static_cast<double>(Tp[4 * r + 1]) * y + static_cast<double>(Tp[4 * r + 2]) * z + static_cast<double>(Tp[4 * r + 3]);
      pg[r] = static_cast<double>(Tg[4 * r]) * x + static_cast<double>(Tg[4 * r + 1]) * y + static_cast<double>(Tg[4 * r + 2]) * z + static_cast<double>(Tg[4 * r + 3]);
    }
    add += sqrt((pp[0] - pg[0]) * (pp[0] - pg[0]) + (pp[1] - pg[1]) * (pp[1] - pg[1]) + (pp[2] - pg[2]) * (pp[2] - pg[2]));
    // project(): xyz K^T, divide by z (eval_metric.py:34-36)
    double up[3], ug[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      up[r] = static_cast<double>(K[3 * r]) * pp[0] + static_cast<double>(K[3 * r + 1]) * pp[1] + static_cast<double>(K[3 * r + 2]) * pp[2];
      ug[r] = static_cast<double>(K[3 * r]) * pg[0] + static_cast<double>(K[3 * r + 1]) * pg[1] + static_cast<double>(K[3 * r + 2]) * pg[2];
    }
    const double ex = up[0] / up[2] - ug[0] / ug[2], ey = up[1] / up[2] - ug[1] / ug[2];
    proj += sqrt(ex * ex + ey * ey);
    if (nn_idx) {   // ADD-S: nearest PREDICTED point of every TARGET point (eval_metric.py:121-125)
      const float* a = pts_pred + (static_cast<long long>(b) * P + nn_idx[static_cast<long long>(b) * P + i]) * 3;
      const float* g = pts_gt + (static_cast<long long>(b) * P + i) * 3;
      const double dx = static_cast<double>(a[0]) - g[0], dy = static_cast<double>(a[1]) - g[1], dz = static_cast<double>(a[2]) - g[2];
      adds += sqrt(dx * dx + dy * dy + dz * dz);
    }
  }
  add = block_sum(add, red);
  proj = block_sum(proj, red);
  adds = block_sum(adds, red);
  if (threadIdx.x == 0) {
    const double tx = static_cast<double>(Tp[3]) - Tg[3], ty = static_cast<double>(Tp[7]) - Tg[7], tz = static_cast<double>(Tp[11]) - Tg[11];
    double tr = 0.0;                                             // trace(R_pred R_gt^T)   (eval_metric.py:183-186)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) tr += static_cast<double>(Tp[4 * r + c]) * Tg[4 * r + c];
    if (tr > 3.0) tr = 3.0;
    out[5 * b + 0] = add / P;
    out[5 * b + 1] = nn_idx ? adds / P : -1.0;
    out[5 * b + 2] = proj / P;
    out[5 * b + 3] = sqrt(tx * tx + ty * ty + tz * tz) * 100.0;
    out[5 * b + 4] = acos((tr - 1.0) / 2.0) * (180.0 / 3.14159265358979323846);
  }
}

int launch_nn(const float* ref, const float* que, int* idxs, int b, int pn1, int pn2, int dim, int exclude_self,
              hipStream_t st) {
  dim3 grid(rp::cdiv(pn2, 256), b), block(256);
  if (dim == 3)
    hipLaunchKernelGGL(nn_search_kernel<3>, grid, block, 0, st, ref, que, idxs, pn1, pn2, exclude_self);
  else
    hipLaunchKernelGGL(nn_search_kernel<2>, grid, block, 0, st, ref, que, idxs, pn1, pn2, exclude_self);
  return 0;
}

}  // namespace

extern "C" {

int rnnpose_nn_search_f32(const float* ref_pts, const float* que_pts, int* idxs, int b, int pn1, int pn2, int dim,
                          int exclude_self, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_nn_search_f32";
  RP_REQUIRE(ref_pts && que_pts && idxs, fn, "null pointer");
  RP_REQUIRE(b > 0 && b < 65536 && pn1 > 0 && pn2 > 0 && (dim == 2 || dim == 3), fn, "bad size (dim must be 2 or 3)");
  launch_nn(ref_pts, que_pts, idxs, b, pn1, pn2, dim, exclude_self, rp::as_stream(stream));
  return rp::check_launch(fn);
}

size_t rnnpose_pose_metrics_workspace_bytes(int B, int P) {
  if (B <= 0 || P <= 0) return 0;
  return static_cast<size_t>(B) * P * (2 * 3 * sizeof(float) + sizeof(int));
}

int rnnpose_pose_metrics_f64(const float* model, int P, const float* pose_pred, const float* pose_gt, const float* K, int B,
                             int symmetric, void* workspace, size_t workspace_bytes, double* out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_pose_metrics_f64";
  RP_REQUIRE(model && pose_pred && pose_gt && K && out, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && P > 0, fn, "bad size");
  hipStream_t st = rp::as_stream(stream);
  float *pp = nullptr, *pg = nullptr;
  int* idx = nullptr;
  if (symmetric) {
    RP_REQUIRE(workspace && workspace_bytes >= rnnpose_pose_metrics_workspace_bytes(B, P), fn, "workspace too small");
    pp = static_cast<float*>(workspace);
    pg = pp + static_cast<size_t>(B) * P * 3;
    idx = reinterpret_cast<int*>(pg + static_cast<size_t>(B) * P * 3);
    hipLaunchKernelGGL(transform_points_kernel, dim3(rp::cdiv(P, 256), B), dim3(256), 0, st, model, pose_pred, pp, P);
    hipLaunchKernelGGL(transform_points_kernel, dim3(rp::cdiv(P, 256), B), dim3(256), 0, st, model, pose_gt, pg, P);
    launch_nn(pp, pg, idx, B, P, P, 3, 0, st);      // ref = predicted points, queries = target points
  }
  hipLaunchKernelGGL(pose_metrics_kernel, dim3(B), dim3(256), 0, st, model, P, pose_pred, pose_gt, K, pp, pg, idx, out);
  return rp::check_launch(fn);
}

// Drop-in for the reference's cffi symbol (thirdparty/nn/src/ext.h:1-10): HOST buffers, synchronous.  The reference
// exit()s on any CUDA error; this one reports on stderr and fills idxs with -1.
void findNearestPointIdxLauncher(float* ref_pts, float* que_pts, int* idxs, int b, int pn1, int pn2, int dim, int exclude_self) {
  const size_t nr = static_cast<size_t>(b) * pn1 * dim * sizeof(float), nq = static_cast<size_t>(b) * pn2 * dim * sizeof(float);
  const size_t ni = static_cast<size_t>(b) * pn2 * sizeof(int);
  float *dr = nullptr, *dq = nullptr;
  int* di = nullptr;
  hipError_t e = hipSuccess;
  bool ok = b > 0 && pn1 > 0 && pn2 > 0 && (dim == 2 || dim == 3) && ref_pts && que_pts && idxs;
  if (ok) ok = (e = hipMalloc(reinterpret_cast<void**>(&dr), nr)) == hipSuccess;
  if (ok) ok = (e = hipMalloc(reinterpret_cast<void**>(&dq), nq)) == hipSuccess;
  if (ok) ok = (e = hipMalloc(reinterpret_cast<void**>(&di), ni)) == hipSuccess;
  if (ok) ok = (e = hipMemcpy(dr, ref_pts, nr, hipMemcpyHostToDevice)) == hipSuccess;
  if (ok) ok = (e = hipMemcpy(dq, que_pts, nq, hipMemcpyHostToDevice)) == hipSuccess;
  if (ok) {
    launch_nn(dr, dq, di, b, pn1, pn2, dim, exclude_self, nullptr);
    ok = (e = hipGetLastError()) == hipSuccess;
  }
  if (ok) ok = (e = hipMemcpy(idxs, di, ni, hipMemcpyDeviceToHost)) == hipSuccess;
  if (!ok) {
    fprintf(stderr, "findNearestPointIdxLauncher: %s\n", e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    if (idxs && b > 0 && pn2 > 0) memset(idxs, 0xff, ni);
  }
  if (dr) (void)hipFree(dr);
  if (dq) (void)hipFree(dq);
  if (di) (void)hipFree(di);
}

}  // extern "C"
