// NHWC companions of the fused update-block engine (a4/a5/a6): layout conversion, flow bookkeeping, the two
// degenerate convolutions of the update block (Cin = 2 is handled by padding in the igemm; Cout = 2 is this
// file's bandwidth kernel) and the convex upsampling that reads the mask in NHWC.
#define RP_CONTRACT_LOCAL 1      // (geometry.cuh / induced.cuh: no fma contraction inside THEIR functions only)
#include "common.hpp"
#include "f16x3.cuh"
#include "induced.cuh"

namespace {

// ---- (B,C,HW) <-> (B,HW,Cs) tiled transposes through LDS ------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int HW, int dst_cs, int dst_co) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r, pp = p0 + tx;
    tile[ty + 8 * r][tx] = (c < C && pp < HW) ? src[(static_cast<long long>(b) * C + c) * HW + pp] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int pp = p0 + ty + 8 * r, c = c0 + tx;
    if (c < C && pp < HW) dst[(static_cast<long long>(b) * HW + pp) * dst_cs + dst_co + c] = tile[tx][ty + 8 * r];
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int HW, int src_cs, int src_co) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int pp = p0 + ty + 8 * r, c = c0 + tx;
    tile[ty + 8 * r][tx] = (c < C && pp < HW) ? src[(static_cast<long long>(b) * HW + pp) * src_cs + src_co + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r, pp = p0 + tx;
    if (c < C && pp < HW) dst[(static_cast<long long>(b) * C + c) * HW + pp] = tile[tx][ty + 8 * r];
  }
}

// ---- flow bookkeeping of one GRU step (model/CFNet.py:147-157, update.py:97) -----------------------------
// coords1 (B,2,h,w) planar -> flow = coords1 - grid (subtract_grid != 0; or coords1 already IS the flow: the
// BasicUpdateBlock facade) written (a) as a 4-channel zero-padded NHWC tensor (input of the 7x7 flow convolution) and
// (b) into channels [co, co+2) of the NHWC motion-feature tensor.
__global__ __launch_bounds__(256) void flow_prep_kernel(const float* __restrict__ coords1, float* __restrict__ flow4,
                                                        float* __restrict__ motion, int motion_cs, int motion_co, int h,
                                                        int w, long long total, int subtract_grid) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int n = h * w;
  const int b = static_cast<int>(t / n), pix = static_cast<int>(t - static_cast<long long>(b) * n);
  const int X = pix % w, Y = pix / w;
  const float fx = coords1[(static_cast<long long>(b) * 2 + 0) * n + pix] - (subtract_grid ? static_cast<float>(X) : 0.f);
  const float fy = coords1[(static_cast<long long>(b) * 2 + 1) * n + pix] - (subtract_grid ? static_cast<float>(Y) : 0.f);
  *reinterpret_cast<float4*>(flow4 + t * 4) = make_float4(fx, fy, 0.f, 0.f);
  *reinterpret_cast<float2*>(motion + t * motion_cs + motion_co) = make_float2(fx, fy);
}

// ---- FlowHead.conv2: 3x3, Cin -> 2 (update.py:10,14) fused with coords1 += delta (CFNet.py:157) -------------
// Bandwidth kernel: one wave per output pixel, lanes split the Cin channels (float4 each), 9 taps x 2 outputs
// accumulated per lane, wave64 shuffle reduction.  x: NHWC (B,h,w,cs) using channels [co, co+Cin), Cin = 256.
// w: (2, Cin, 3, 3) PyTorch layout.  Outputs: delta (B,h,w,2) NHWC, coords1_out (B,2,h,w) planar = coords1 + delta,
// flow_lr (B,h,w,2) NHWC = coords1_out - grid.
__global__ __launch_bounds__(256) void conv3x3_cout2_kernel(const float* __restrict__ x, int cs, int co, int Cin,
                                                            const float* __restrict__ wgt, const float* __restrict__ bias,
                                                            const float* __restrict__ coords1, float* __restrict__ delta,
                                                            float* __restrict__ coords1_out, float* __restrict__ flow_lr,
                                                            int B, int h, int w) {
  extern __shared__ float wl[];                 // [o][tap][Cin]
  const int tid = threadIdx.x;
  for (int e = tid; e < 2 * 9 * Cin; e += 256) {
    const int c = e % Cin, tap = (e / Cin) % 9, o = e / (9 * Cin);
    wl[e] = wgt[(static_cast<long long>(o) * Cin + c) * 9 + tap];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int n = h * w;
  const long long total = static_cast<long long>(B) * n;
  for (long long p = static_cast<long long>(blockIdx.x) * 4 + wave; p < total; p += static_cast<long long>(gridDim.x) * 4) {
    const int b = static_cast<int>(p / n), pix = static_cast<int>(p - static_cast<long long>(b) * n);
    const int X = pix % w, Y = pix / w;
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane * 4; c < Cin; c += 256) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int yy = Y + tap / 3 - 1, xx = X + tap % 3 - 1;
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (static_cast<long long>(b) * n + yy * w + xx) * cs + co + c);
        const float4 w0 = *reinterpret_cast<const float4*>(wl + (0 * 9 + tap) * Cin + c);
        const float4 w1 = *reinterpret_cast<const float4*>(wl + (1 * 9 + tap) * Cin + c);
        a0 += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
        a1 += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      a0 += __shfl_down(a0, d);
      a1 += __shfl_down(a1, d);
    }
    if (lane == 0) {
      const float dx = a0 + bias[0], dy = a1 + bias[1];
      *reinterpret_cast<float2*>(delta + p * 2) = make_float2(dx, dy);
      const float cx = coords1[(static_cast<long long>(b) * 2 + 0) * n + pix] + dx;
      const float cy = coords1[(static_cast<long long>(b) * 2 + 1) * n + pix] + dy;
      if (coords1_out) {
        coords1_out[(static_cast<long long>(b) * 2 + 0) * n + pix] = cx;
        coords1_out[(static_cast<long long>(b) * 2 + 1) * n + pix] = cy;
      }
      *reinterpret_cast<float2*>(flow_lr + p * 2) = make_float2(cx - static_cast<float>(X), cy - static_cast<float>(Y));
    }
  }
}

// Fast path of the same operation for Cin <= 256: a wave owns a run of 4 horizontally adjacent pixels; lane l carries
// channels 4l..4l+3 with its 72 weights (2 outputs x 9 taps x 4 channels) in registers for the whole kernel, loads the
// 3 x 6 input quads the run needs (4.5 loads per pixel instead of 9, no weight traffic at all) and the 8 partial sums
// (4 pixels x 2 outputs) are reduced over the 64 lanes with a halving butterfly (10 shuffles instead of 48).
__global__ __launch_bounds__(256) void conv3x3_cout2_run4_kernel(const float* __restrict__ x, int cs, int co, int Cin,
                                                                 const float* __restrict__ wgt, const float* __restrict__ bias,
                                                                 const float* __restrict__ coords1, float* __restrict__ delta,
                                                                 float* __restrict__ coords1_out, float* __restrict__ flow_lr,
                                                                 int B, int h, int w) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane * 4;
  const bool cl = c < Cin;
  float4 w0[9], w1[9];                       // [tap] x 4 channels, outputs 0 / 1
  {
    float t0[36], t1[36];                 // wgt (2, Cin, 3, 3): 36 consecutive floats per (output, 4 channels) = 9 aligned quads
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 q0 = cl ? *reinterpret_cast<const float4*>(wgt + (static_cast<long long>(0) * Cin + c) * 9 + e * 4) : z;
      const float4 q1 = cl ? *reinterpret_cast<const float4*>(wgt + (static_cast<long long>(1) * Cin + c) * 9 + e * 4) : z;
      t0[e * 4 + 0] = q0.x; t0[e * 4 + 1] = q0.y; t0[e * 4 + 2] = q0.z; t0[e * 4 + 3] = q0.w;
      t1[e * 4 + 0] = q1.x; t1[e * 4 + 1] = q1.y; t1[e * 4 + 2] = q1.z; t1[e * 4 + 3] = q1.w;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      w0[tap] = make_float4(t0[tap], t0[9 + tap], t0[18 + tap], t0[27 + tap]);
      w1[tap] = make_float4(t1[tap], t1[9 + tap], t1[18 + tap], t1[27 + tap]);
    }
  }
  const int n = h * w;
  const int runs_per_row = (w + 3) >> 2;
  const long long total_runs = static_cast<long long>(B) * h * runs_per_row;
  for (long long r = static_cast<long long>(blockIdx.x) * 4 + wave; r < total_runs; r += static_cast<long long>(gridDim.x) * 4) {
    const unsigned ru = static_cast<unsigned>(r);                       // total_runs < 2^31 (host check): 32-bit divisions
    const unsigned by = ru / static_cast<unsigned>(runs_per_row);
    const int xr = static_cast<int>(ru - by * runs_per_row);
    const int b = static_cast<int>(by / static_cast<unsigned>(h)), Y = static_cast<int>(by - static_cast<unsigned>(b) * h);
    const int X0 = xr * 4;
    // the pixel this lane will write (if any) is known from its lane bits: its old coordinates are requested together with
    // the input quads instead of in a second, dependent round trip after the reduction
    const int idx = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    const int px = idx >> 1, o = idx & 1;
    const int Xw = X0 + px < w ? X0 + px : w - 1;
    const int pixw = Y * w + Xw;
    const float c1x = coords1[(static_cast<long long>(b) * 2 + 0) * n + pixw];
    const float c1y = coords1[(static_cast<long long>(b) * 2 + 1) * n + pixw];
    float4 v[3][6];
#pragma unroll
    for (int ry = 0; ry < 3; ++ry)
#pragma unroll
      for (int cx = 0; cx < 6; ++cx) {
        const int yy = Y + ry - 1, xx = X0 + cx - 1;
        const bool ok = cl && yy >= 0 && yy < h && xx >= 0 && xx < w;
        v[ry][cx] = ok ? *reinterpret_cast<const float4*>(x + (static_cast<long long>(b) * n + yy * w + xx) * cs + co + c)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    float acc[8];                            // [pixel][output]
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int ry = 0; ry < 3; ++ry)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 q = v[ry][px + k], u0 = w0[ry * 3 + k], u1 = w1[ry * 3 + k];
          a0 += q.x * u0.x + q.y * u0.y + q.z * u0.z + q.w * u0.w;
          a1 += q.x * u1.x + q.y * u1.y + q.z * u1.z + q.w * u1.w;
        }
      acc[px * 2 + 0] = a0;
      acc[px * 2 + 1] = a1;
    }
    // halving butterfly: after the xor-32 / 16 / 8 steps a lane keeps ONE of the 8 sums (index = lane bits 5,4,3)
    float s4[4], s2[2], s1;
    {
      const bool up = lane & 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = up ? acc[4 + i] : acc[i], give = up ? acc[i] : acc[4 + i];
        s4[i] = keep + __shfl_xor(give, 32);
      }
      const bool up2 = lane & 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = up2 ? s4[2 + i] : s4[i], give = up2 ? s4[i] : s4[2 + i];
        s2[i] = keep + __shfl_xor(give, 16);
      }
      const bool up3 = lane & 8;
      const float keep = up3 ? s2[1] : s2[0], give = up3 ? s2[0] : s2[1];
      s1 = keep + __shfl_xor(give, 8);
      s1 += __shfl_xor(s1, 4);
      s1 += __shfl_xor(s1, 2);
      s1 += __shfl_xor(s1, 1);
    }
    // lane bits (5,4,3) = (i2, i1, i0): value index = 4*i2 + 2*i1 + i0 = pixel*2 + output (idx, px, o above)
    const float other = __shfl_xor(s1, 8);          // the pixel's other output component lives 8 lanes away
    if ((lane & 7) == 0 && o == 0 && X0 + px < w) {
      const int X = X0 + px, pix = Y * w + X;
      const long long p = static_cast<long long>(b) * n + pix;
      const float dx = s1 + bias[0], dy = other + bias[1];
      *reinterpret_cast<float2*>(delta + p * 2) = make_float2(dx, dy);
      const float cxv = c1x + dx;
      const float cyv = c1y + dy;
      if (coords1_out) {
        coords1_out[(static_cast<long long>(b) * 2 + 0) * n + pix] = cxv;
        coords1_out[(static_cast<long long>(b) * 2 + 1) * n + pix] = cyv;
      }
      *reinterpret_cast<float2*>(flow_lr + p * 2) = make_float2(cxv - static_cast<float>(X), cyv - static_cast<float>(Y));
    }
  }
}

// ---- a6 with NHWC inputs: mask (B,h,w,576) [channel k*64 + i*8 + j], flow (B,h,w,2) -> flow_up (B,2,8h,8w) planar
// Workgroup = 4 consecutive low-res pixels x 64 sub-pixels: every mask read is a coalesced 256-byte run, every
// output row segment is 4 x 8 contiguous floats.
__global__ __launch_bounds__(256) void convex_upsample_nhwc_kernel(const float* __restrict__ flow,
                                                                   const float* __restrict__ mask,
                                                                   float* __restrict__ up, int h, int w) {
  const int b = blockIdx.z, Y = blockIdx.y;
  const int X = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = threadIdx.x & 63;     // sub-pixel i*8 + j
  if (X >= w) return;
  const int n = h * w;
  const float* m = mask + (static_cast<long long>(b) * n + Y * w + X) * 576 + t;
  // all 18 loads first, unconditional (neighbours outside the map read the centre pixel and are zeroed afterwards): a load
  // under its bounds test is followed by a vmcnt(0) wait -- nine dependent round trips per thread before this change
  float mv[9], mx = -INFINITY;
  float2 fv[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    mv[k] = m[k * 64];
    const int yy = Y + k / 3 - 1, xx = X + k % 3 - 1;
    const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
    fv[k] = *reinterpret_cast<const float2*>(flow + (static_cast<long long>(b) * n + (ok ? yy * w + xx : Y * w + X)) * 2);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 9; ++k) mx = fmaxf(mx, mv[k]);
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    mv[k] = expf(mv[k] - mx);
    den += mv[k];
  }
  float ax = 0.f, ay = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = Y + k / 3 - 1, xx = X + k % 3 - 1;
    const float2 f = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? fv[k] : make_float2(0.f, 0.f);
    const float wk = mv[k] / den;
    ax += wk * (8.f * f.x);
    ay += wk * (8.f * f.y);
  }
  const int i = t >> 3, j = t & 7;
  const long long Wf = 8LL * w, Pf = 64LL * n;
  const long long o = (8LL * Y + i) * Wf + 8 * X + j;
  up[(static_cast<long long>(b) * 2 + 0) * Pf + o] = ax;
  up[(static_cast<long long>(b) * 2 + 1) * Pf + o] = ay;
}


// ---- instance norm (no affine, eps inside the sqrt) on NHWC tensors: thirdparty/raft/extractor.py:28-31,129-130 ------
// pass 1: per (image, row chunk) partial sums in fp64, fixed order; pass 2: mean / rstd; pass 3: apply (+ReLU,
// + residual add + ReLU), float4 wide.  x viewed as (B, HW, C), C % 4 == 0, C <= 256.
constexpr int IN_ROWS = 512;     // rows per partial
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                               int HW, int C) {
  __shared__ double s1[256], s2[256];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int tid = threadIdx.x;
  const int lanes_per_row = C >> 2;                 // float4 lanes covering one row
  const int rows_per_iter = 256 / lanes_per_row;    // C in {32..256} -> 32..4 rows per sweep (C divides 1024)
  const int c4 = tid % lanes_per_row, r0 = tid / lanes_per_row;
  const int row_beg = chunk * IN_ROWS, row_end = min(HW, row_beg + IN_ROWS);
  double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (r0 < rows_per_iter) {
    for (int r = row_beg + r0; r < row_end; r += rows_per_iter) {
      const float4 v = *reinterpret_cast<const float4*>(x + (static_cast<long long>(b) * HW + r) * C + c4 * 4);
      a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
      q[0] += static_cast<double>(v.x) * v.x; q[1] += static_cast<double>(v.y) * v.y;
      q[2] += static_cast<double>(v.z) * v.z; q[3] += static_cast<double>(v.w) * v.w;
    }
  }
  // reduce the rows_per_iter row groups of each channel quad through LDS (fixed order)
  for (int k = 0; k < 4; ++k) {
    s1[tid] = a[k];
    s2[tid] = q[k];
    __syncthreads();
    if (tid < lanes_per_row) {
      double t1 = 0, t2 = 0;
      for (int g = 0; g < rows_per_iter; ++g) {
        t1 += s1[g * lanes_per_row + tid];
        t2 += s2[g * lanes_per_row + tid];
      }
      const int c = tid * 4 + k;
      part[((static_cast<long long>(b) * nchunk + chunk) * C + c) * 2 + 0] = t1;
      part[((static_cast<long long>(b) * nchunk + chunk) * C + c) * 2 + 1] = t2;
    }
    __syncthreads();
  }
}

__global__ void instnorm_finalize_kernel(const double* __restrict__ part, float* __restrict__ mean_rstd, int nchunk, int C,
                                         int HW, float eps) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double t1 = 0, t2 = 0;
    for (int k = 0; k < nchunk; ++k) {
      t1 += part[((static_cast<long long>(b) * nchunk + k) * C + c) * 2 + 0];
      t2 += part[((static_cast<long long>(b) * nchunk + k) * C + c) * 2 + 1];
    }
    const double m = t1 / HW;
    double var = t2 / HW - m * m;                   // biased variance, as F.instance_norm
    if (var < 0) var = 0;
    mean_rstd[(static_cast<long long>(b) * C + c) * 2 + 0] = static_cast<float>(m);
    mean_rstd[(static_cast<long long>(b) * C + c) * 2 + 1] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  }
}

// statistics from the producing convolution's per-tile (sum, sum of squares) pairs: tile_stats (B * tiles_per_image, C, 2)
__global__ __launch_bounds__(256) void instnorm_finalize_tiles_kernel(const double* __restrict__ tile_stats,
                                                                      float* __restrict__ mean_rstd, int tiles_per_image,
                                                                      int C, int HW, float eps) {
  // grid (B, ceil(C/8)): 8 channels x 32 tile groups per workgroup; group g sums tiles g, g+32, ... (fixed order), the
  // 32 group sums are combined in order through LDS: deterministic, and the dependent chain is tiles/32 loads long
  // (this kernel is pure latency: 600 tiles per image at 240x320)
  __shared__ double s1[32][8], s2[32][8];
  const int b = blockIdx.x, cl = threadIdx.x & 7, c = blockIdx.y * 8 + cl, g = threadIdx.x >> 3;
  double t1 = 0, t2 = 0;
  if (c < C) {
    const double* p = tile_stats + (static_cast<long long>(b) * tiles_per_image * C + c) * 2;
#pragma unroll 4
    for (int k = g; k < tiles_per_image; k += 32) {
      const double2 v = *reinterpret_cast<const double2*>(p + static_cast<long long>(k) * C * 2);
      t1 += v.x;
      t2 += v.y;
    }
  }
  s1[g][cl] = t1;
  s2[g][cl] = t2;
  __syncthreads();
  if (g == 0 && c < C) {
    double a1 = 0, a2 = 0;
    for (int k = 0; k < 32; ++k) { a1 += s1[k][cl]; a2 += s2[k][cl]; }
    const double m = a1 / HW;
    double var = a2 / HW - m * m;
    if (var < 0) var = 0;
    mean_rstd[(static_cast<long long>(b) * C + c) * 2 + 0] = static_cast<float>(m);
    mean_rstd[(static_cast<long long>(b) * C + c) * 2 + 1] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  }
}

// out = act1((x - mean) * rstd); if residual: out = relu(residual' + out), residual' = the residual itself or, with res_mr,
// its own instance norm [+ ReLU] applied on the fly (the RAW stem / down-sampling convolution output: extractor.py:54-58)
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                             const float* __restrict__ residual, float* __restrict__ out,
                                                             int HW, int C, int relu, long long total4,
                                                             const float* __restrict__ res_mr, int res_relu) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  // total4 < 2^31 (host check): 32-bit divisions (three 64-bit ones per float4 cost more than the memory traffic)
  const unsigned iu = static_cast<unsigned>(i), cq = static_cast<unsigned>(C >> 2);
  const unsigned row = iu / cq;
  const int c4 = static_cast<int>(iu - row * cq);
  const int b = static_cast<int>(row / static_cast<unsigned>(HW));
  // all six loads are issued up front (absent operands alias present ones): a load under `if (residual)` costs a second,
  // dependent memory round trip per thread
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float4 m01 = *reinterpret_cast<const float4*>(mean_rstd + (static_cast<long long>(b) * C + c4 * 4) * 2);
  const float4 m23 = *reinterpret_cast<const float4*>(mean_rstd + (static_cast<long long>(b) * C + c4 * 4 + 2) * 2);
  const float* rsrc = residual ? residual : x;
  const float* qsrc = res_mr ? res_mr : mean_rstd;
  float4 r = reinterpret_cast<const float4*>(rsrc)[i];
  const float4 q01 = *reinterpret_cast<const float4*>(qsrc + (static_cast<long long>(b) * C + c4 * 4) * 2);
  const float4 q23 = *reinterpret_cast<const float4*>(qsrc + (static_cast<long long>(b) * C + c4 * 4 + 2) * 2);
  __builtin_amdgcn_sched_barrier(0);
  float4 y = make_float4((v.x - m01.x) * m01.y, (v.y - m01.z) * m01.w, (v.z - m23.x) * m23.y, (v.w - m23.z) * m23.w);
  if (relu) y = make_float4(fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f));
  if (residual) {
    if (res_mr) {
      r = make_float4((r.x - q01.x) * q01.y, (r.y - q01.z) * q01.w, (r.z - q23.x) * q23.y, (r.w - q23.z) * q23.w);
      if (res_relu) r = make_float4(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f));
    }
    y = make_float4(fmaxf(r.x + y.x, 0.f), fmaxf(r.y + y.y, 0.f), fmaxf(r.z + y.z, 0.f), fmaxf(r.w + y.w, 0.f));
  }
  reinterpret_cast<float4*>(out)[i] = y;
}

// ---- BasicMotionEncoder.convf1: 7x7, 2 -> Cout (=128), ReLU (update.py:84,91) ------------------------------------
// K = 98: far too thin for the matrix cores (the implicit-GEMM kernel pads the 2 channels to 32: 16x wasted MFMA work).
// Direct form: thread = output channel with its 98 weights in registers; a workgroup walks a 20-pixel row segment
// whose 7 x 26 x 2 flow patch sits in LDS (wave-uniform broadcast reads); each pixel's Cout outputs leave as one
// contiguous NHWC row.  flow4: (B,h,w,4) [fx,fy,0,0]; wt: weights transposed to (98, Cout) = [(ci*7+ky)*7+kx][c].
// PLANAR: the input is coords1 / flow (B,2,h,w) planar (the pixel grid is subtracted here when subtract_grid != 0) and the
// flow is also written into channels [motion_co, +2) of the motion-feature tensor: flow_prep + convf1 in one launch.
constexpr int F1_TX = 20;
constexpr int F1_ROW = 28;     // F1_TX + 6 halo, rounded up to a multiple of 4
template <bool PLANAR>
__global__ __launch_bounds__(128, 3) void conv7x7_cin2_kernel(const float* __restrict__ flow4, const float* __restrict__ wt,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int out_cs, int out_co, int Cout, int h, int w,
                                                           int subtract_grid, float* __restrict__ motion, int motion_cs,
                                                           int motion_co, int out_hl, int motion_hl, float a_scale,
                                                           unsigned long long* __restrict__ sat, const rp::InducedSrc isrc) {
  // rows padded to 28 floats (112 bytes): every row is seven aligned 16-byte LDS reads
  __shared__ __attribute__((aligned(16))) float patch[2][7][F1_ROW];
  const int b = blockIdx.z, Y = blockIdx.y, X0 = blockIdx.x * F1_TX;
  const int c = threadIdx.x;
  const int n = h * w;
  for (int e = threadIdx.x; e < 7 * (F1_TX + 6); e += 128) {
    const int ky = e / (F1_TX + 6), xx = e % (F1_TX + 6);
    const int yy = Y + ky - 3, x = X0 + xx - 3;
    float2 v = make_float2(0.f, 0.f);
    if (yy >= 0 && yy < h && x >= 0 && x < w) {
      if (PLANAR) {
        if (isrc.depth) {      // r06: coords1 formed here (induced.cuh: bit-identical to induced_coords_lowres_kernel's), then flow = coords1 - grid as below
          const float2 cc = rp::induced_coords_at(isrc.depth + static_cast<long long>(b) * isrc.H * isrc.W, x, yy, isrc.H, isrc.W, h, w, isrc.eps,
                                                  rp::load_intr(isrc.K, b), rp::load_pose(isrc.G, b));
          v.x = cc.x - (subtract_grid ? static_cast<float>(x) : 0.f);
          v.y = cc.y - (subtract_grid ? static_cast<float>(yy) : 0.f);
        } else {
          v.x = flow4[(static_cast<long long>(b) * 2 + 0) * n + yy * w + x] - (subtract_grid ? static_cast<float>(x) : 0.f);
          v.y = flow4[(static_cast<long long>(b) * 2 + 1) * n + yy * w + x] - (subtract_grid ? static_cast<float>(yy) : 0.f);
        }
        if (ky == 3 && xx >= 3 && xx < 3 + F1_TX) {     // this row segment's own pixels: motion[..., co:co+2] = flow (update.py:97)
          float* mrow = motion + (static_cast<long long>(b) * n + yy * w + x) * motion_cs;
          if (motion_hl) {        // split tensor: the two channels are 4 bytes of the hi plane + 4 bytes of the lo plane of their group
            rp::h4 hi, lo;
            const float4 q = make_float4(v.x, v.y, 0.f, 0.f);
            rp::split4(q, a_scale, hi, lo);
            if (sat && rp::quad_saturates(q, a_scale)) atomicAdd(sat, 1ull);
            _Float16* ph = reinterpret_cast<_Float16*>(mrow + (motion_co & ~7)) + (motion_co & 7);
            *reinterpret_cast<rp::h2*>(ph) = rp::h2{hi.x, hi.y};
            *reinterpret_cast<rp::h2*>(ph + 8) = rp::h2{lo.x, lo.y};
          } else {
            *reinterpret_cast<float2*>(mrow + motion_co) = v;
          }
        }
      } else {
        const float4 f = *reinterpret_cast<const float4*>(flow4 + (static_cast<long long>(b) * n + yy * w + x) * 4);
        v = make_float2(f.x, f.y);
      }
    }
    patch[0][ky][xx] = v.x;
    patch[1][ky][xx] = v.y;
  }
  const float bs = c < Cout ? bias[c] : 0.f;
  __syncthreads();
  if (c >= Cout && !out_hl) return;      // (split output: lane pairs exchange their halves below, Cout is even)
  // Row-outer accumulation: one patch row (7 wave-uniform 16-byte LDS reads) feeds 7 taps x 20 pixels, so the LDS pipe sees
  // 98 reads per thread instead of one 4-byte read per multiply-add (1960): the kernel was LDS-issue-bound (r02: 23 us per
  // launch).  The weights are held one input channel (49) at a time: with all 98 (+ their 64-bit addresses) the 20
  // accumulators did not fit.  Per pixel the products are still added in (ci, ky, kx) order.
  float acc[F1_TX];
#pragma unroll
  for (int x = 0; x < F1_TX; ++x) acc[x] = bs;
#pragma unroll 1
  for (int ci = 0; ci < 2; ++ci) {
    float wreg[49];
    const float* wc = wt + static_cast<long long>(ci) * 49 * Cout + (c < Cout ? c : 0);
#pragma unroll
    for (int k = 0; k < 49; ++k) wreg[k] = wc[k * Cout];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      float row[F1_ROW];
#pragma unroll
      for (int q = 0; q < F1_ROW / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(&patch[ci][ky][4 * q]);
        row[4 * q + 0] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int x = 0; x < F1_TX; ++x)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) acc[x] += wreg[ky * 7 + kx] * row[x + kx];
      __builtin_amdgcn_sched_barrier(0);     // one row in registers at a time
    }
  }
  if (out_hl) {
    // split tensor: channel ch = out_co + c sits at fp16 index (ch & 7) of the hi plane of its 32-byte group, lo plane 8 fp16
    // further.  Even lanes store the (hi, hi) pair of channels (ch, ch + 1), odd lanes the (lo, lo) pair: one 4-byte store each.
    const int ch = out_co + c;
    const bool odd = c & 1;
    const long long pbase = (static_cast<long long>(b) * n + Y * w + X0) * out_cs + (ch & ~7);
    bool sflag = false;
#pragma unroll
    for (int x = 0; x < F1_TX; ++x) {
      const float y = fmaxf(acc[x], 0.f);
      rp::h4 hi, lo;
      const float4 q = make_float4(y, 0.f, 0.f, 0.f);
      rp::split4(q, a_scale, hi, lo);
      sflag |= rp::quad_saturates(q, a_scale);
      const unsigned mine = odd ? __builtin_bit_cast(unsigned short, lo.x) : __builtin_bit_cast(unsigned short, hi.x);
      const unsigned send = odd ? __builtin_bit_cast(unsigned short, hi.x) : __builtin_bit_cast(unsigned short, lo.x);
      const unsigned got = __shfl_xor(send, 1);          // even lane receives the odd lane's hi, odd lane the even lane's lo
      const unsigned word = odd ? (got | (mine << 16)) : (mine | (got << 16));
      if (X0 + x < w && c < Cout)
        reinterpret_cast<unsigned*>(out + pbase + static_cast<long long>(x) * out_cs)[(odd ? 4 : 0) + ((ch & 7) >> 1)] = word;
    }
    if (sat && sflag && c < Cout) atomicAdd(sat, 1ull);
    return;
  }
#pragma unroll
  for (int x = 0; x < F1_TX; ++x)
    if (X0 + x < w) out[(static_cast<long long>(b) * n + Y * w + X0 + x) * out_cs + out_co + c] = fmaxf(acc[x], 0.f);
}

// fp32 NHWC channels [co, co + C) of `src` -> the same channels of the split tensor `dst` (C % 8 == 0): thread = one 8-channel group
__global__ __launch_bounds__(256) void split_hl_kernel(const float* __restrict__ src, int scs, int sco, float* __restrict__ dst, int dcs,
                                                       int dco, int C8, long long total, float a_scale,
                                                       unsigned long long* __restrict__ sat) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= total) return;
  const long long m = i / C8;
  const int g = static_cast<int>(i - m * C8);
  const float4 a = *reinterpret_cast<const float4*>(src + m * scs + sco + g * 8);
  const float4 bq = *reinterpret_cast<const float4*>(src + m * scs + sco + g * 8 + 4);
  rp::h4 h0, l0, h1, l1;
  rp::split4(a, a_scale, h0, l0);
  rp::split4(bq, a_scale, h1, l1);
  if (sat && (rp::quad_saturates(a, a_scale) || rp::quad_saturates(bq, a_scale))) atomicAdd(sat, 1ull);
  float* o = dst + m * dcs + dco + g * 8;
  *reinterpret_cast<rp::h8*>(o) = rp::h8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
  *reinterpret_cast<rp::h8*>(o + 4) = rp::h8{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
}

}  // namespace

extern "C" {

int rnnpose_nchw_to_nhwc_f32(const float* src, int B, int C, int HW, float* dst, int dst_c_stride, int dst_c_offset,
                             rnnpose_stream_t stream) {
  const char* fn = "rnnpose_nchw_to_nhwc_f32";
  RP_REQUIRE(src && dst, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && C > 0 && HW > 0 && dst_c_offset >= 0 && dst_c_offset + C <= dst_c_stride, fn, "bad size");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(rp::cdiv(HW, 32), rp::cdiv(C, 32), B), dim3(256), 0, rp::as_stream(stream),
                     src, dst, C, HW, dst_c_stride, dst_c_offset);
  return rp::check_launch(fn);
}

int rnnpose_nhwc_to_nchw_f32(const float* src, int B, int C, int HW, int src_c_stride, int src_c_offset, float* dst,
                             rnnpose_stream_t stream) {
  const char* fn = "rnnpose_nhwc_to_nchw_f32";
  RP_REQUIRE(src && dst, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && C > 0 && HW > 0 && src_c_offset >= 0 && src_c_offset + C <= src_c_stride, fn, "bad size");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(rp::cdiv(HW, 32), rp::cdiv(C, 32), B), dim3(256), 0, rp::as_stream(stream),
                     src, dst, C, HW, src_c_stride, src_c_offset);
  return rp::check_launch(fn);
}

int rnnpose_flow_prep_f32(const float* coords1, int subtract_grid, int B, int h, int w, float* flow4, float* motion,
                          int motion_c_stride, int motion_c_offset, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_flow_prep_f32";
  RP_REQUIRE(coords1 && flow4 && motion, fn, "null pointer");
  RP_REQUIRE(B > 0 && h > 0 && w > 0 && motion_c_offset % 2 == 0 && motion_c_stride % 2 == 0 &&
                 motion_c_offset + 2 <= motion_c_stride, fn, "bad size");
  const long long total = static_cast<long long>(B) * h * w;
  hipLaunchKernelGGL(flow_prep_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, rp::as_stream(stream), coords1, flow4,
                     motion, motion_c_stride, motion_c_offset, h, w, total, subtract_grid);
  return rp::check_launch(fn);
}

int rnnpose_flow_head_out_f32(const float* x, int x_c_stride, int x_c_offset, int c_in, const float* w_oihw,
                              const float* bias, const float* coords1, int B, int h, int w, float* delta,
                              float* coords1_out, float* flow_lr, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_flow_head_out_f32";
  RP_REQUIRE(x && w_oihw && bias && coords1 && delta && flow_lr, fn, "null pointer");
  RP_REQUIRE(B > 0 && h > 0 && w > 0 && c_in > 0 && c_in % 4 == 0 && c_in <= 1024, fn, "bad size");
  RP_REQUIRE(x_c_stride % 4 == 0 && x_c_offset % 4 == 0 && x_c_offset + c_in <= x_c_stride, fn, "bad channel window");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(w_oihw) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0, fn, "x and weights must be 16-byte aligned");
  const long long total = static_cast<long long>(B) * h * w;
  if (c_in <= 256) {
    const long long runs = static_cast<long long>(B) * h * ((w + 3) / 4);
    RP_REQUIRE(runs < (1LL << 31), fn, "too many pixel runs for 32-bit indices");
#ifndef RP_COUT2_MAXWG
#define RP_COUT2_MAXWG 512
#endif
    const int nb = static_cast<int>(runs / 4 + 1 < RP_COUT2_MAXWG ? runs / 4 + 1 : RP_COUT2_MAXWG);      // 2 workgroups per CU: a wave keeps its weights for ~5 runs
    hipLaunchKernelGGL(conv3x3_cout2_run4_kernel, dim3(nb), dim3(256), 0, rp::as_stream(stream), x, x_c_stride, x_c_offset, c_in,
                       w_oihw, bias, coords1, delta, coords1_out, flow_lr, B, h, w);
    return rp::check_launch(fn);
  }
  const int blocks = static_cast<int>(total / 4 + 1 < 4096 ? total / 4 + 1 : 4096);
  const size_t lds = static_cast<size_t>(2) * 9 * c_in * sizeof(float);
  hipLaunchKernelGGL(conv3x3_cout2_kernel, dim3(blocks), dim3(256), lds, rp::as_stream(stream), x, x_c_stride, x_c_offset,
                     c_in, w_oihw, bias, coords1, delta, coords1_out, flow_lr, B, h, w);
  return rp::check_launch(fn);
}

int rnnpose_convex_upsample_nhwc_f32(const float* flow_lr, const float* mask, int B, int h, int w, float* flow_up,
                                     rnnpose_stream_t stream) {
  const char* fn = "rnnpose_convex_upsample_nhwc_f32";
  RP_REQUIRE(flow_lr && mask && flow_up, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && h > 0 && h < 65536 && w > 0, fn, "bad size");
  hipLaunchKernelGGL(convex_upsample_nhwc_kernel, dim3(rp::cdiv(w, 4), h, B), dim3(256), 0, rp::as_stream(stream), flow_lr,
                     mask, flow_up, h, w);
  return rp::check_launch(fn);
}


size_t rnnpose_instnorm_workspace_bytes(int B, int HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0) return 0;
  return static_cast<size_t>(B) * rp::cdiv(HW, IN_ROWS) * C * 2 * sizeof(double);
}

int rnnpose_instnorm_nhwc_f32(const float* x, int B, int HW, int C, float eps, int relu, const float* residual,
                              void* workspace, size_t workspace_bytes, float* mean_rstd, float* out,
                              rnnpose_stream_t stream) {
  const char* fn = "rnnpose_instnorm_nhwc_f32";
  RP_REQUIRE(x && workspace && mean_rstd && out, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && HW > 0 && C >= 32 && C <= 256 && 1024 % C == 0 || C == 96 || C == 192, fn,
             "C must be 32, 64, 96, 128, 192 or 256");
  RP_REQUIRE(workspace_bytes >= rnnpose_instnorm_workspace_bytes(B, HW, C), fn, "workspace too small");
  const int nchunk = rp::cdiv(HW, IN_ROWS);
  hipStream_t st = rp::as_stream(stream);
  hipLaunchKernelGGL(instnorm_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, x, static_cast<double*>(workspace), HW, C);
  hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(B), dim3(256), 0, st, static_cast<const double*>(workspace), mean_rstd,
                     nchunk, C, HW, eps);
  const long long total4 = static_cast<long long>(B) * HW * (C >> 2);
  RP_REQUIRE(total4 < (1LL << 31), fn, "tensor too large for 32-bit float4 indices");
  hipLaunchKernelGGL(instnorm_apply_kernel, dim3(rp::cdiv(total4, 256)), dim3(256), 0, st, x, mean_rstd, residual, out, HW,
                     C, relu, total4, static_cast<const float*>(nullptr), 0);
  return rp::check_launch(fn);
}


int rnnpose_instnorm_tiles_nhwc_f32(const float* x, int B, int HW, int C, float eps, int relu, const float* residual,
                                    const float* residual_mean_rstd, int residual_relu, const double* tile_stats,
                                    int tiles_per_image, float* mean_rstd, float* out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_instnorm_tiles_nhwc_f32";
  RP_REQUIRE(tile_stats && mean_rstd && (x || !out), fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && HW > 0 && C > 0 && C % 4 == 0, fn, "bad size (C % 4 == 0)");
  RP_REQUIRE(tiles_per_image > 0, fn, "tiles_per_image must be positive");
  hipStream_t st = rp::as_stream(stream);
  hipLaunchKernelGGL(instnorm_finalize_tiles_kernel, dim3(B, rp::cdiv(C, 8)), dim3(256), 0, st, tile_stats, mean_rstd, tiles_per_image, C, HW, eps);
  if (out) {
    const long long total4 = static_cast<long long>(B) * HW * (C >> 2);
    RP_REQUIRE(total4 < (1LL << 31), fn, "tensor too large for 32-bit float4 indices");
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3(rp::cdiv(total4, 256)), dim3(256), 0, st, x, mean_rstd, residual, out, HW,
                       C, relu, total4, residual ? residual_mean_rstd : nullptr, residual_relu);
  }
  return rp::check_launch(fn);
}

int rnnpose_flow_conv7x7_relu_f32(const float* flow4, const float* w_t, const float* bias, int B, int h, int w, int c_out,
                                  float* out, int out_c_stride, int out_c_offset, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_flow_conv7x7_relu_f32";
  RP_REQUIRE(flow4 && w_t && bias && out, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && h > 0 && h < 65536 && w > 0 && c_out > 0 && c_out <= 128, fn, "bad size (c_out <= 128)");
  RP_REQUIRE(out_c_offset >= 0 && out_c_offset + c_out <= out_c_stride && reinterpret_cast<uintptr_t>(flow4) % 16 == 0, fn,
             "bad output window / flow4 alignment");
  hipLaunchKernelGGL(conv7x7_cin2_kernel<false>, dim3(rp::cdiv(w, F1_TX), h, B), dim3(128), 0, rp::as_stream(stream), flow4, w_t,
                     bias, out, out_c_stride, out_c_offset, c_out, h, w, 0, static_cast<float*>(nullptr), 0, 0, 0, 0, 1.f,
                     static_cast<unsigned long long*>(nullptr), rp::InducedSrc{});
  return rp::check_launch(fn);
}

static int launch_flow_features(const char* fn, const float* coords1, const rp::InducedSrc isrc, int subtract_grid, const float* w_t, const float* bias,
                                int B, int h, int w, int c_out, float* out, int out_c_stride, int out_c_offset, float* motion, int motion_c_stride,
                                int motion_c_offset, int out_split, int motion_split, float a_scale, rnnpose_stream_t stream) {
  if (out_split) RP_REQUIRE(c_out % 2 == 0 && out_c_offset % 8 == 0 && out_c_stride % 8 == 0 && reinterpret_cast<uintptr_t>(out) % 32 == 0 && a_scale > 0.f, fn,
                            "split-form out: even c_out, channel offset/stride multiples of 8, 32-byte aligned");
  if (motion_split) RP_REQUIRE(motion_c_stride % 8 == 0 && reinterpret_cast<uintptr_t>(motion) % 32 == 0 && a_scale > 0.f, fn,
                               "split-form motion: channel stride multiple of 8, 32-byte aligned");
  RP_REQUIRE((coords1 || isrc.depth) && w_t && bias && out && motion, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && h > 0 && h < 65536 && w > 0 && c_out > 0 && c_out <= 128, fn, "bad size (c_out <= 128)");
  RP_REQUIRE(out_c_offset >= 0 && out_c_offset + c_out <= out_c_stride && motion_c_offset % 2 == 0 && motion_c_stride % 2 == 0 &&
                 motion_c_offset + 2 <= motion_c_stride, fn, "bad output windows");
  hipLaunchKernelGGL(conv7x7_cin2_kernel<true>, dim3(rp::cdiv(w, F1_TX), h, B), dim3(128), 0, rp::as_stream(stream), coords1, w_t,
                     bias, out, out_c_stride, out_c_offset, c_out, h, w, subtract_grid, motion, motion_c_stride, motion_c_offset,
                     out_split, motion_split, a_scale, rp::sat_counter(), isrc);
  return rp::check_launch(fn);
}

int rnnpose_flow_features_f32(const float* coords1, int subtract_grid, const float* w_t, const float* bias, int B, int h, int w,
                              int c_out, float* out, int out_c_stride, int out_c_offset, float* motion, int motion_c_stride,
                              int motion_c_offset, int out_split, int motion_split, float a_scale, rnnpose_stream_t stream) {
  return launch_flow_features("rnnpose_flow_features_f32", coords1, rp::InducedSrc{}, subtract_grid, w_t, bias, B, h, w, c_out, out, out_c_stride,
                              out_c_offset, motion, motion_c_stride, motion_c_offset, out_split, motion_split, a_scale, stream);
}

// r06: the same launch forming coords1 itself from depth (B,1,H,W), K (B,3,3), G (B,4,4) -- rnnpose_induced_coords_lowres_f32's arithmetic, bit for
// bit (csrc/induced.cuh) -- then flow = coords1 - grid as above (model/PoseRefiner.py:324-328 + thirdparty/raft/update.py:84,91,97)
int rnnpose_flow_features_induced_f32(const float* depth, const float* K, const float* G, int H, int W, float eps, const float* w_t,
                                      const float* bias, int B, int h, int w, int c_out, float* out, int out_c_stride, int out_c_offset,
                                      float* motion, int motion_c_stride, int motion_c_offset, int out_split, int motion_split, float a_scale,
                                      rnnpose_stream_t stream) {
  const char* fn = "rnnpose_flow_features_induced_f32";
  RP_REQUIRE(depth && K && G && H >= h && W >= w && W / (w > 0 ? w : 1) >= 1, fn, "null pointer / the depth map must be at least as large as the low-resolution map");
  return launch_flow_features(fn, nullptr, rp::InducedSrc{depth, K, G, H, W, eps}, 1, w_t, bias, B, h, w, c_out, out, out_c_stride, out_c_offset, motion,
                              motion_c_stride, motion_c_offset, out_split, motion_split, a_scale, stream);
}

int rnnpose_split_hl_f32(const float* src, int src_c_stride, int src_c_offset, long long n_pixels, int c_count, float a_scale,
                         float* dst, int dst_c_stride, int dst_c_offset, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_split_hl_f32";
  RP_REQUIRE(src && dst && n_pixels > 0 && c_count > 0 && a_scale > 0.f, fn, "bad argument");
  RP_REQUIRE(c_count % 8 == 0 && src_c_offset % 4 == 0 && src_c_stride % 4 == 0 && dst_c_offset % 8 == 0 && dst_c_stride % 8 == 0 &&
                 src_c_offset + c_count <= src_c_stride && dst_c_offset + c_count <= dst_c_stride &&
                 reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 32 == 0, fn,
             "whole 8-channel groups, aligned channel windows");
  const long long total = n_pixels * (c_count / 8);
  hipLaunchKernelGGL(split_hl_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, rp::as_stream(stream), src, src_c_stride, src_c_offset,
                     dst, dst_c_stride, dst_c_offset, c_count / 8, total, a_scale, rp::sat_counter());
  return rp::check_launch(fn);
}

}  // extern "C"
