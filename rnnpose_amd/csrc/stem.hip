// f1: the RAFT encoder stem as ONE kernel -- input normalisation 2*(x/255)-1 (model/CFNet.py:42-43) fused into the
// 7x7 stride-2 convolution 3 -> 64 (thirdparty/raft/extractor.py:131,197), NCHW image in, NHWC feature map out, plus the
// per-tile column statistics the following instance norm needs (extractor.py:129,198).
//
// K = 3*49 = 147 is too thin for the implicit-GEMM kernel's 32-channel blocks (it would multiply 8x zeros), so this is an
// im2col-in-LDS GEMM: one workgroup = 8 x 16 output pixels (128 GEMM rows) x 64 output channels, K padded to 160.
//   * the 21 x 37 x 3 input patch of the tile is staged ONCE in LDS (normalised fp32; zero padding applied AFTER the
//     normalisation, as F.conv2d pads the normalised tensor);
//   * every lane gathers its MFMA A operand (8 consecutive k = (c, ky, kx) values of one output pixel) from the patch
//     with compile-time tap offsets, splits it into fp16 hi/lo (f16x3.cuh) and issues 3 x 2 v_mfma_f32_32x32x16_f16;
//   * weights are pre-packed in B-fragment order [k step][32-col tile][lane][8] fp16 hi/lo (20 KB each, L1/L2 resident);
//   * epilogue: accumulators -> wave-private LDS tile -> 16-byte stores (256 B per pixel) + bias, tile statistics in
//     fixed order (no atomics).
// HBM-bound by its output (315 MB per 16 x 480 x 640 images vs 59 MB in); the MFMA work is 75 GFLOP executed.
//
// r06, STEM_V2 = 1 (default): the same tile and the same GEMM, restructured around what the r05 form spent its time on
// (82-115 us per 8 x 480 x 640 launch for 25 us of output stores: every WAVE streamed the 40 KB of packed weights from L2 for its 60
// MFMAs, and every lane re-split the fp32 values it gathered -- a texel was split ~12 times):
//   * PERSISTENT workgroups (two per CU) walk the tile list of their XCD; the packed weights sit in LDS (45 KB, loaded once per
//     workgroup) and arrive as ds_read_b128 B fragments;
//   * the patch is split into fp16 hi / lo ONCE while it is staged (two 6-KB planes, 48 halfs per row: lanes 16 apart in a wave =
//     two patch rows apart = 16 banks apart); K is ordered (c, ky, kx padded to 8), 22 rows of 8 = 11 k-steps, so a lane's A fragment
//     is 8 consecutive halfs of ONE patch row: four 4-byte LDS reads per plane and k-step, no VALU between the MFMAs;
//   * the next tile's patch is requested before the current tile's MFMAs and waited for after its stores.
// STEM_V2 = 0 restores the r02-r05 kernel (one tile per workgroup; its packed weight layout differs: pack and launch are compiled
// together).
#include <algorithm>

#include "common.hpp"
#include "f16x3.cuh"

#ifndef STEM_V2
#define STEM_V2 1
#endif
#ifndef STEM_ABL
#define STEM_ABL 0      // diagnostics builds (results WRONG): 1 no MFMAs, 2 no output stores, 4 no fp64 statistics, 8 no patch requests, 16 no staging split
#endif

namespace {

using rp::f32x16;
using rp::h4;
using rp::h8;

constexpr int TH = 8, TW = 16;                     // output tile
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;    // 21 x 37 input patch (stride 2, 7 x 7 taps)
constexpr int PLANE = PH * PW;                     // 777
constexpr int PATCH = 3 * PLANE;                   // 2331 floats
#if STEM_V2
constexpr int KREAL = 147, KSTEPS = 11;            // K = 3 x 7 rows of 8 (kx padded) + one zero row = 11 MFMA k-steps of 16
constexpr int ROWH = 48;                           // halfs per staged patch row (37 + zero padding; kx = 7 reads column <= 37)
constexpr int PLH = 3 * PH * ROWH;                 // halfs per plane (hi or lo): 3024
__host__ __device__ constexpr int rowoff(int r) { return r < 21 ? ((r / 7) * PH + (r % 7)) * ROWH : 0; }    // (c, ky) row r of the patch
#else
constexpr int KREAL = 147, KSTEPS = 10;            // K padded to 160 = 10 MFMA k-steps of 16
#endif
constexpr int CO = 64;
constexpr int ES = CO + 4;                         // staging row stride (floats)

#if STEM_V2
// one element of the split of f16x3.cuh (split4): x already scaled; hi = fp16(x) rounded to nearest and clamped to +-65504, lo = fp16(x - hi)
__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
  const rp::h2 cap = {static_cast<_Float16>(65504.f), static_cast<_Float16>(65504.f)};
  rp::h2 h = __builtin_convertvector(rp::f32x2{x, x}, rp::h2);
  h = __builtin_elementwise_max(__builtin_elementwise_min(h, cap), -cap);
  const float l = x - static_cast<float>(h.x);                     // exact in fp32 (in range)
  rp::h2 q = __builtin_convertvector(rp::f32x2{l, l}, rp::h2);
  q = __builtin_elementwise_max(__builtin_elementwise_min(q, cap), -cap);
  hi = h.x;
  lo = q.x;
}

// x / 255 correctly rounded without the division's ~10-instruction expansion: q = RN(x * y), r = x - 255 q (exact, one FMA),
// q' = RN(q + r y) with y = RN(1 / 255) -- Markstein's correction step.  Checked EXHAUSTIVELY against x / 255.f for all 2 139 095 040
// finite non-negative floats (tools/probes/div255_markstein.c: 0 differences; negative x by symmetry); a CPU test samples it.
__device__ __forceinline__ float div255(float x) {
  constexpr float y = 1.0f / 255.0f;
  const float q = x * y;
  const float r = __builtin_fmaf(-255.0f, q, x);
  return __builtin_fmaf(r, y, q);
}

// The kernel is bound by instructions per tile, not by any pipe (r06 ablation, profiles/r06_stem.txt: 85 us with everything, 32 us with
// neither MFMAs, stores, statistics, patch requests nor split -- and every item on its own removes 4-22 us): everything that does not
// depend on the tile is computed once per workgroup (patch element -> relative image offset, LDS slot), addresses are 32-bit byte
// offsets from scalar bases, branches are per tile instead of per element.
__global__ __launch_bounds__(256, 2) void stem_conv7x7_s2_kernel(const float* __restrict__ img, int normalize,
                                                                 const uint4* __restrict__ whi, const uint4* __restrict__ wlo,
                                                                 const float* __restrict__ bias, float a_scale, float out_scale,
                                                                 float* __restrict__ out, double* __restrict__ tstats, int H,
                                                                 int W, int Ho, int Wo, int tiles_x, int tiles_y, int ntiles,
                                                                 unsigned long long* sat) {
#pragma clang fp contract(off)
  // 45 KB of weights + 12 KB of patch planes + 4 KB of statistics: 61 KB -> two workgroups (8 waves) per CU
  __shared__ __attribute__((aligned(16))) uint4 sW[2][KSTEPS * 128];         // [hi, lo][k step][32-column tile][lane]
  __shared__ __attribute__((aligned(16))) _Float16 sP[2][PLH];               // [hi, lo][c][patch row][48]
  __shared__ double tsum[4][CO][2];                                          // [wave][column][sum, sum of squares] of the wave's 32 pixels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- tile walk: workgroup (xcd = blockIdx & 7, slot) takes tiles slot, slot + per, ... of its XCD's contiguous chunk of the
  //      tile list: the workgroups an XCD runs together cover neighbouring tiles (their patches overlap by 5 of 21 / 37 texels).
  //      (n, ty, tx) of the current and of the next tile are carried along: a step of `per` tiles = (dn, dy, dx) with carries.
  const int per = static_cast<int>(gridDim.x) >> 3, chunk = (ntiles + 7) >> 3;
  const int xcd = static_cast<int>(blockIdx.x) & 7;
  const int t_end = min((xcd + 1) * chunk, ntiles);
  int t = xcd * chunk + (static_cast<int>(blockIdx.x) >> 3);
  if (t >= t_end) return;
  const int tpi = tiles_x * tiles_y;
  const int dn = per / tpi, dy = (per - dn * tpi) / tiles_x, dx = per - dn * tpi - dy * tiles_x;
  int cn = t / tpi, cy = (t - cn * tpi) / tiles_x, cx = t - cn * tpi - cy * tiles_x;      // current tile
  int nn_ = cn, ny = cy, nx = cx;                                                          // the tile whose patch is in flight

  // ---- per thread, once: its NLD patch elements i = tid + 256 j -> (row, column) in the patch, byte offset in the image relative to
  //      the patch origin, half index in the LDS planes
  constexpr int NLD = (PATCH + 255) / 256;
  int e_py[NLD], e_px[NLD], e_rel[NLD], e_lds[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int i = tid + 256 * j;
    const int c = i / PLANE, r = i - c * PLANE;
    const int py = r / PW, px = r - py * PW;
    const bool valid = i < PATCH;
    e_py[j] = valid ? py : -100000;                     // (never inside the image)
    e_px[j] = px;
    e_rel[j] = ((c * H + py) * W + px) * 4;
    e_lds[j] = valid ? (c * PH + py) * ROWH + px : PLH - 1;      // (the last half of a plane is padding: column 47 of the last row)
  }
  float pv[NLD];
  unsigned inside = 0u;
  const char* const img8 = reinterpret_cast<const char*>(img);
  // all loads of a thread are issued before the first one is used: unconditional, texels outside the image read element 0 of
  // the image and become zero when they are staged (a load under its bounds test is followed by vmcnt(0))
  auto request = [&](int n, int ty, int tx) {
    const int iy0 = 2 * (ty * TH) - 3, ix0 = 2 * (tx * TW) - 3;
    const int base = ((n * 3 * H + iy0) * W + ix0) * 4;              // byte offset of the patch origin (may lie outside the image: only `ok` elements use it)
    inside = 0u;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const bool ok = static_cast<unsigned>(iy0 + e_py[j]) < static_cast<unsigned>(H) && static_cast<unsigned>(ix0 + e_px[j]) < static_cast<unsigned>(W);
      const unsigned off = ok ? static_cast<unsigned>(base + e_rel[j]) : 0u;
      pv[j] = *reinterpret_cast<const float*>(img8 + off);
      inside |= (ok ? 1u : 0u) << j;
    }
  };
  request(cn, cy, cx);

  // ---- once per workgroup: weights -> LDS, patch planes zeroed (the padding columns 37..47 of every row stay zero)
  for (int i = tid; i < KSTEPS * 128; i += 256) {
    sW[0][i] = whi[i];
    sW[1][i] = wlo[i];
  }
  {
    unsigned* z = reinterpret_cast<unsigned*>(&sP[0][0]);
    for (int i = tid; i < PLH; i += 256) z[i] = 0u;                         // 2 planes x PLH halfs = PLH dwords
  }
  const float bias0 = bias[l31], bias1 = bias[32 + l31];                    // this lane's two output columns (C layout: column = lane & 31 of each 32-column tile)

  const int oyl = 2 * wave + (l31 >> 4), oxl = l31 & 15;
  const int pbase = (2 * oyl) * ROWH + 2 * oxl;                             // halfs (even): this lane's output pixel, tap (0, 0)
  const unsigned* const ph32 = reinterpret_cast<const unsigned*>(&sP[0][0]);
  const unsigned* const pl32 = reinterpret_cast<const unsigned*>(&sP[1][0]);
  char* const out8 = reinterpret_cast<char*>(out);
  int t_prev = -1;

  auto write_record = [&](int tt) {                                         // threads 0..63: the tile's (sum, sum of squares) of one column each
    double* o = tstats + (static_cast<long long>(tt) * CO + tid) * 2;
    o[0] = ((tsum[0][tid][0] + tsum[1][tid][0]) + tsum[2][tid][0]) + tsum[3][tid][0];
    o[1] = ((tsum[0][tid][1] + tsum[1][tid][1]) + tsum[2][tid][1]) + tsum[3][tid][1];
  };

  while (t < t_end) {
    __syncthreads();                                       // (A) every wave is done with the previous tile's patch and statistics
    if (tstats && t_prev >= 0 && tid < CO) write_record(t_prev);
    // ---- patch: registers -> LDS, normalised, split once ----
    {
      int nsat = 0;
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        float v = pv[j];
        if (normalize) v = 2.f * div255(v) - 1.f;            // model/CFNet.py:42 (div255: v / 255.f, correctly rounded)
        v = ((inside >> j) & 1u) ? v : 0.f;                  // zero padding AFTER the normalisation, as F.conv2d pads the normalised tensor
        nsat += (fabsf(v) * a_scale <= 65504.f) ? 0 : 1;     // range guard (f16x3.cuh); NaN counts
        _Float16 hi, lo;
#if STEM_ABL & 16
        hi = static_cast<_Float16>(v); lo = hi;
#else
        split1(v * a_scale, hi, lo);
#endif
        sP[0][e_lds[j]] = hi;                                // (elements past the patch: the padding half at the end of the plane gets a zero)
        sP[1][e_lds[j]] = lo;
      }
      if (sat && nsat) atomicAdd(sat, static_cast<unsigned long long>(nsat));
    }
    __syncthreads();                                       // (B)
    const int n = cn, oy0 = cy * TH, ox0 = cx * TW;
    const int t_next = t + per;
    {                                                      // (n, ty, tx) of tile t + per
      nx += dx; ny += dy; nn_ += dn;
      if (nx >= tiles_x) { nx -= tiles_x; ++ny; }
      if (ny >= tiles_y) { ny -= tiles_y; ++nn_; }
    }
    if (t_next < t_end && !(STEM_ABL & 8)) request(nn_, ny, nx);               // the next tile's patch travels under this tile's MFMAs and stores

    // ---- main loop: 11 k-steps; this wave = 32 output pixels (2 tile rows) x 64 channels ----
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      const int d = (pbase + (lh ? rowoff(2 * kk + 1) : rowoff(2 * kk))) >> 1;
      const uint4 ahq = make_uint4(ph32[d], ph32[d + 1], ph32[d + 2], ph32[d + 3]);
      const uint4 alq = make_uint4(pl32[d], pl32[d + 1], pl32[d + 2], pl32[d + 3]);
      const h8 ah = __builtin_bit_cast(h8, ahq), al = __builtin_bit_cast(h8, alq);
      const h8 B0h = __builtin_bit_cast(h8, sW[0][kk * 128 + lane]), B1h = __builtin_bit_cast(h8, sW[0][kk * 128 + 64 + lane]);
      const h8 B0l = __builtin_bit_cast(h8, sW[1][kk * 128 + lane]), B1l = __builtin_bit_cast(h8, sW[1][kk * 128 + 64 + lane]);
#if STEM_ABL & 1
      acc0[0] += static_cast<float>(al[0] + ah[1] + B0h[2] + B0l[3]); acc1[0] += static_cast<float>(B1h[0] + B1l[1]);
      continue;
#endif
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, B0h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, B1h, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B0l, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B1l, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B0h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B1h, acc1, 0, 0, 0);
    }

    // ---- epilogue, straight from the accumulators (C layout of the 32x32 MFMA: a lane holds column lane & 31 of each 32-column tile
    //      for the 16 rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5), row = 16 (tile row) + output pixel): lanes 0..31 of a 4-byte store write
    //      one contiguous 128-byte half of a pixel's 256-byte NHWC row, lanes 32..63 that of the pixel 4 further -- whole cache lines,
    //      immediate offsets, no staging tile (r06: the LDS round trip + float4 stores cost ~80 instructions per tile more).
    //      Statistics: a lane's 16 pixels of a column in fp32 groups of four, the groups in fp64 (as the strip kernels' tile statistics:
    //      conv_strip_kernel.cuh, EV = 1); pixels of a ragged tile outside the image contribute nothing.
    const bool full = oy0 + TH <= Ho && ox0 + TW <= Wo;     // (uniform) no pixel of the tile lies outside the image
    const unsigned obase = static_cast<unsigned>(((n * Ho + oy0 + 2 * wave) * Wo + ox0 + 4 * lh) * (CO * 4) + l31 * 4);
    float y0[16], y1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      y0[r] = __builtin_fmaf(acc0[r], out_scale, bias0);    // (out_scale = 1 / (a_scale w_scale), powers of two: the product is exact, one rounding either way)
      y1[r] = __builtin_fmaf(acc1[r], out_scale, bias1);
    }
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned off = static_cast<unsigned>(((r >> 3) * Wo + (r & 3) + 8 * ((r >> 2) & 1)) * (CO * 4));
#if STEM_ABL & 2
        if (y0[r] == 1.2345e30f)
#endif
        {
          *reinterpret_cast<float*>(out8 + (obase + off)) = y0[r];
          *reinterpret_cast<float*>(out8 + (obase + off + 128u)) = y1[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oy = oy0 + 2 * wave + (r >> 3), ox = ox0 + 4 * lh + (r & 3) + 8 * ((r >> 2) & 1);
        const unsigned off = static_cast<unsigned>(((r >> 3) * Wo + (r & 3) + 8 * ((r >> 2) & 1)) * (CO * 4));
        if (oy < Ho && ox < Wo) {
          *reinterpret_cast<float*>(out8 + (obase + off)) = y0[r];
          *reinterpret_cast<float*>(out8 + (obase + off + 128u)) = y1[r];
        } else {
          y0[r] = 0.f;
          y1[r] = 0.f;
        }
      }
    }
    if (tstats && !(STEM_ABL & 4)) {
      double s0 = 0., s1 = 0., q0 = 0., q1 = 0.;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s0 += static_cast<double>((y0[4 * g] + y0[4 * g + 1]) + (y0[4 * g + 2] + y0[4 * g + 3]));
        s1 += static_cast<double>((y1[4 * g] + y1[4 * g + 1]) + (y1[4 * g + 2] + y1[4 * g + 3]));
        q0 += static_cast<double>(fmaf(y0[4 * g], y0[4 * g], y0[4 * g + 1] * y0[4 * g + 1]) + fmaf(y0[4 * g + 2], y0[4 * g + 2], y0[4 * g + 3] * y0[4 * g + 3]));
        q1 += static_cast<double>(fmaf(y1[4 * g], y1[4 * g], y1[4 * g + 1] * y1[4 * g + 1]) + fmaf(y1[4 * g + 2], y1[4 * g + 2], y1[4 * g + 3] * y1[4 * g + 3]));
      }
      // the two lanes of a column (l and l + 32: the other 16 pixels), then the 4 waves through LDS in fixed order (at (A))
      s0 += rp::shfl_xor_f64(s0, 32); s1 += rp::shfl_xor_f64(s1, 32);
      q0 += rp::shfl_xor_f64(q0, 32); q1 += rp::shfl_xor_f64(q1, 32);
      if (lane < 32) {
        tsum[wave][l31][0] = s0; tsum[wave][l31][1] = q0;
        tsum[wave][32 + l31][0] = s1; tsum[wave][32 + l31][1] = q1;
      }
    }
    t_prev = t;
    t = t_next;
    cn = nn_; cy = ny; cx = nx;
  }
  __syncthreads();
  if (tstats && tid < CO) write_record(t_prev);
}

// (64,3,7,7) fp32 -> B fragments [k step][32-col tile][lane][8] fp16 hi / lo: lane l carries column 32*tile + (l & 31) and the 8 taps
// kx = j of patch row r = 2*kstep + (l >> 5) = (c, ky) (kx = 7 and row 21: zero)
__global__ void stem_pack_kernel(const float* __restrict__ w, float w_scale, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KSTEPS * 2 * 64 * 8) return;
  const int j = i & 7, ln = (i >> 3) & 63, ni = (i >> 9) & 1, kk = i >> 10;
  const int r = 2 * kk + (ln >> 5), nn = ni * 32 + (ln & 31);
  const float v = (r < 21 && j < 7) ? w[nn * KREAL + (r / 7) * 49 + (r % 7) * 7 + j] * w_scale : 0.f;
  const _Float16 h = static_cast<_Float16>(v);
  hi[i] = h;
  lo[i] = static_cast<_Float16>(v - static_cast<float>(h));
}
#else   // ---- STEM_V2 == 0: the r02-r05 kernel ----
__host__ __device__ constexpr int koff(int k) {    // patch offset of GEMM-k index k = c*49 + ky*7 + kx (padding: tap 0, weight 0)
  return k < KREAL ? (k / 49) * PLANE + ((k % 49) / 7) * PW + (k % 7) : 0;
}

__global__ __launch_bounds__(256) void stem_conv7x7_s2_kernel(const float* __restrict__ img, int normalize,
                                                              const uint4* __restrict__ whi, const uint4* __restrict__ wlo,
                                                              const float* __restrict__ bias, float a_scale, float out_scale,
                                                              float* __restrict__ out, double* __restrict__ tstats, int H,
                                                              int W, int Ho, int Wo, int tiles_x, int tiles_y,
                                                              unsigned long long* sat) {
#pragma clang fp contract(off)
  // the input patch (9.3 KB) and the epilogue's staging tiles (4 waves x 16 rows x 68 floats = 17 KB) share one buffer:
  // 19.4 KB per workgroup -> 8 workgroups (32 waves) per CU.  The kernel is a chain of latencies per workgroup (patch from
  // HBM -> LDS -> gather -> MFMA -> LDS -> stores); with the original 46 KB only 3 were resident and it ran at a third
  // of its HBM bound
  __shared__ __attribute__((aligned(16))) float smem[4 * 16 * ES];
  __shared__ double tsum[4][16][8];
  static_assert(4 * 16 * ES >= PATCH + 5, "staging buffer must hold the patch");
  float* const patch = smem;
  float* const stage = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, n = t / tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

  // ---- patch: global (NCHW) -> LDS, normalised ----
  const float* src = img + static_cast<long long>(n) * 3 * H * W;
  // all (PATCH + 255) / 256 loads of a thread are issued before the first one is used: unconditional, texels outside the
  // image read element 0 of the image and become zero below (a load under its bounds test is followed by vmcnt(0):
  // ten dependent HBM round trips per workgroup before this change)
  constexpr int NLD = (PATCH + 255) / 256;
  float pv[NLD];
  unsigned inside = 0u;
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int i = tid + 256 * j;
    const int c = i / PLANE, r = i - c * PLANE;
    const int py = r / PW, px = r - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool ok = i < PATCH && iy >= 0 && iy < H && ix >= 0 && ix < W;
    pv[j] = src[ok ? (static_cast<long long>(c) * H + iy) * W + ix : 0];
    inside |= (ok ? 1u : 0u) << j;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int i = tid + 256 * j;
    float v = 0.f;
    if ((inside >> j) & 1u) {
      v = pv[j];
      if (normalize) v = 2.f * (v / 255.f) - 1.f;          // literal operation order of model/CFNet.py:42
    }
    if (sat && !(fabsf(v) * a_scale <= 65504.f)) atomicAdd(sat, 1ull);     // range guard (f16x3.cuh)
    if (i < PATCH) patch[i] = v;
  }
  __syncthreads();

  // ---- main loop: 10 k-steps; this wave = 32 output pixels (2 tile rows) x 64 channels ----
  const int oyl = 2 * wave + (l31 >> 4), oxl = l31 & 15;
  const float* pl = patch + (2 * oyl) * PW + 2 * oxl;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  uint4 bh0 = whi[lane], bh1 = whi[64 + lane], bl0 = wlo[lane], bl1 = wlo[64 + lane];
#pragma unroll
  for (int kk = 0; kk < KSTEPS; ++kk) {
    uint4 nh0 = bh0, nh1 = bh1, nl0 = bl0, nl1 = bl1;
    if (kk + 1 < KSTEPS) {
      nh0 = whi[(kk + 1) * 128 + lane]; nh1 = whi[(kk + 1) * 128 + 64 + lane];
      nl0 = wlo[(kk + 1) * 128 + lane]; nl1 = wlo[(kk + 1) * 128 + 64 + lane];
    }
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = pl[lh ? koff(kk * 16 + 8 + j) : koff(kk * 16 + j)];
    h4 h0, l0, h1, l1;
    rp::split4(make_float4(x[0], x[1], x[2], x[3]), a_scale, h0, l0);
    rp::split4(make_float4(x[4], x[5], x[6], x[7]), a_scale, h1, l1);
    const h8 ah = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    const h8 al = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    const h8 B0h = __builtin_bit_cast(h8, bh0), B1h = __builtin_bit_cast(h8, bh1);
    const h8 B0l = __builtin_bit_cast(h8, bl0), B1l = __builtin_bit_cast(h8, bl1);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, B0h, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, B1h, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B0l, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B1l, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B0h, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B1h, acc1, 0, 0, 0);
    bh0 = nh0; bh1 = nh1; bl0 = nl0; bl1 = nl1;
  }

  // ---- epilogue: two half-tiles of 16 rows per wave through LDS -> 16-byte row-contiguous stores ----
  __syncthreads();                                         // every wave is done gathering from the patch (aliased below)
  float* S = stage + wave * (16 * ES);
  const int cq = (lane & 15) * 4;                          // this lane's column quad, the same for every row it stores
  const float4 b4 = *reinterpret_cast<const float4*>(bias + cq);
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0., q0 = 0., q1 = 0., q2 = 0., q3 = 0.;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {                          // accumulator rows 16 hf .. 16 hf + 15 (C layout of the 32x32 MFMA)
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      S[row * ES + l31] = acc0[8 * hf + r] * out_scale;
      S[row * ES + 32 + l31] = acc1[8 * hf + r] * out_scale;
    }
    __builtin_amdgcn_wave_barrier();                       // wave-private tile: LDS ops of one wave execute in order
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rl = (lane >> 4) + 4 * k;                  // row within the half-tile (0..15) = output pixel ox0 + rl
      const int oy = oy0 + 2 * wave + hf, ox = ox0 + rl;
      if (oy >= Ho || ox >= Wo) continue;
      float4 y = *reinterpret_cast<const float4*>(S + rl * ES + cq);
      y.x += b4.x; y.y += b4.y; y.z += b4.z; y.w += b4.w;
      const double d0 = y.x, d1 = y.y, d2 = y.z, d3 = y.w;         // fp64 statistics (csrc/conv_igemm.hip: the stem sees an almost constant image)
      s0 += d0; s1 += d1; s2 += d2; s3 += d3;
      q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
      *reinterpret_cast<float4*>(out + ((static_cast<long long>(n) * Ho + oy) * Wo + ox) * CO + cq) = y;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (tstats) {
    // lanes sharing a column quad (same lane & 15): fixed-order butterfly, then the 4 waves through LDS in order
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      s0 += rp::shfl_xor_f64(s0, o); s1 += rp::shfl_xor_f64(s1, o); s2 += rp::shfl_xor_f64(s2, o); s3 += rp::shfl_xor_f64(s3, o);
      q0 += rp::shfl_xor_f64(q0, o); q1 += rp::shfl_xor_f64(q1, o); q2 += rp::shfl_xor_f64(q2, o); q3 += rp::shfl_xor_f64(q3, o);
    }
    if (lane < 16) {
      double* p = tsum[wave][lane];
      p[0] = s0; p[1] = s1; p[2] = s2; p[3] = s3; p[4] = q0; p[5] = q1; p[6] = q2; p[7] = q3;
    }
    __syncthreads();
    if (tid < 16) {
      double* o = tstats + (static_cast<long long>(blockIdx.x) * CO + tid * 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e + 0] = ((tsum[0][tid][e] + tsum[1][tid][e]) + tsum[2][tid][e]) + tsum[3][tid][e];
        o[2 * e + 1] = ((tsum[0][tid][4 + e] + tsum[1][tid][4 + e]) + tsum[2][tid][4 + e]) + tsum[3][tid][4 + e];
      }
    }
  }
}

// (64,3,7,7) fp32 -> B fragments [k step][32-col tile][lane][8] fp16 hi / lo: lane l carries column 32*tile + (l & 31)
// and k = 16*kstep + 8*(l >> 5) + j
__global__ void stem_pack_kernel(const float* __restrict__ w, float w_scale, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KSTEPS * 2 * 64 * 8) return;
  const int j = i & 7, ln = (i >> 3) & 63, ni = (i >> 9) & 1, kk = i >> 10;
  const int k = kk * 16 + (ln >> 5) * 8 + j, nn = ni * 32 + (ln & 31);
  const float v = k < KREAL ? w[nn * KREAL + k] * w_scale : 0.f;
  const _Float16 h = static_cast<_Float16>(v);
  hi[i] = h;
  lo[i] = static_cast<_Float16>(v - static_cast<float>(h));
}
#endif  // STEM_V2

}  // namespace

static int g_stem_wg_cap = 0;       // rnnpose_stem_workgroups: 0 = two persistent workgroups per CU

extern "C" {

int rnnpose_stem_workgroups(int max_workgroups) {       // test / measurement switch (0 = default)
  g_stem_wg_cap = max_workgroups > 0 ? max_workgroups : 0;
  return 0;
}

long long rnnpose_stem_packed_halfs(void) { return KSTEPS * 2 * 64 * 8; }

int rnnpose_stem_pack_weights_f16x3(const float* w_oihw, float w_scale, void* w_hi, void* w_lo, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_stem_pack_weights_f16x3";
  RP_REQUIRE(w_oihw && w_hi && w_lo && w_scale > 0.f, fn, "null pointer / non-positive scale");
  const int total = KSTEPS * 2 * 64 * 8;
  hipLaunchKernelGGL(stem_pack_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, rp::as_stream(stream), w_oihw, w_scale,
                     static_cast<_Float16*>(w_hi), static_cast<_Float16*>(w_lo));
  return rp::check_launch(fn);
}

int rnnpose_stem_tiles(int H, int W, int* tiles_per_image, int* exact) {
  if (H <= 0 || W <= 0 || !tiles_per_image || !exact) return rp::fail_arg("rnnpose_stem_tiles", "bad argument");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  *tiles_per_image = rp::cdiv(Ho, TH) * rp::cdiv(Wo, TW);
  // r03: always 1.  The epilogue drops the pixels of a ragged tile that lie outside the image BEFORE they enter the sums, and
  // instnorm_finalize_tiles divides by the true H*W -- the statistics of ragged tilings (every 240 x 240 crop: 120 x 120
  // outputs are 15 x 7.5 tiles) were always right; r01-r02 sent those shapes through a full instance-norm pass instead.
  (void)Ho; (void)Wo;
  *exact = 1;
  return 0;
}

int rnnpose_stem_conv7x7_s2_f16x3(const float* img_nchw, int N, int H, int W, int normalize, const void* w_hi,
                                  const void* w_lo, const float* bias, float a_scale, float w_scale, float* out_nhwc,
                                  double* tile_stats, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_stem_conv7x7_s2_f16x3";
  RP_REQUIRE(img_nchw && w_hi && w_lo && bias && out_nhwc, fn, "null pointer");
  RP_REQUIRE(N > 0 && H > 0 && W > 0 && a_scale > 0.f && w_scale > 0.f, fn, "bad size / scale");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(out_nhwc) % 16 == 0 && reinterpret_cast<uintptr_t>(bias) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(w_hi) % 16 == 0 && reinterpret_cast<uintptr_t>(w_lo) % 16 == 0,
             fn, "out / bias / packed weights must be 16-byte aligned");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int tiles_x = rp::cdiv(Wo, TW), tiles_y = rp::cdiv(Ho, TH);
  // (tile_stats: ragged tilings included -- pixels outside the image are dropped before they enter the sums)
  const long long blocks = static_cast<long long>(N) * tiles_x * tiles_y;
  RP_REQUIRE(blocks < (1LL << 31), fn, "too many tiles");
#if STEM_V2
  // persistent: two workgroups per CU (or fewer: one per tile of the XCD chunks), a multiple of 8
  const int ntiles = static_cast<int>(blocks);
  int per = std::min((2 * rp::cu_count()) >> 3, (ntiles + 7) >> 3);
  if (g_stem_wg_cap > 0) per = std::min(per, std::max(g_stem_wg_cap >> 3, 1));
  hipLaunchKernelGGL(stem_conv7x7_s2_kernel, dim3(static_cast<unsigned>(8 * std::max(per, 1))), dim3(256), 0, rp::as_stream(stream), img_nchw,
                     normalize, static_cast<const uint4*>(w_hi), static_cast<const uint4*>(w_lo), bias, a_scale,
                     1.0f / (a_scale * w_scale), out_nhwc, tile_stats, H, W, Ho, Wo, tiles_x, tiles_y, ntiles, rp::sat_counter());
#else
  hipLaunchKernelGGL(stem_conv7x7_s2_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, rp::as_stream(stream), img_nchw,
                     normalize, static_cast<const uint4*>(w_hi), static_cast<const uint4*>(w_lo), bias, a_scale,
                     1.0f / (a_scale * w_scale), out_nhwc, tile_stats, H, W, Ho, Wo, tiles_x, tiles_y, rp::sat_counter());
#endif
  return rp::check_launch(fn);
}

}  // extern "C"
