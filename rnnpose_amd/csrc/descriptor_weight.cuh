// a8: the descriptor reliability weight of ONE pixel (model/PoseRefiner.py:342-345), shared by the stand-alone weight kernel
// (pointwise.hip) and the fused weight + normal-equation kernel (lm.hip):
//   taps  normalize_coords_grid (align_corners=True formula) then grid_sample's align_corners=False unnormalise, zero padding
//   dot   s = sum_c g1[c, pixel] * bilinear(g2[c], taps), channels in batches whose 5 loads each are in flight together
#pragma once
#include <hip/hip_runtime.h>
#ifndef RP_CW_WAIT0
#define RP_CW_WAIT0 0
#endif
#ifndef RP_CW_TAPS_SC1
#define RP_CW_TAPS_SC1 0
#endif

namespace rp {

struct DescTaps {
  float w00, w10, w01, w11;
  long long o00, o10, o01, o11;
};

__device__ __forceinline__ DescTaps descriptor_taps(float tx, float ty, int H, int W) {
  const float gx = 2.f * tx / static_cast<float>(W - 1) - 1.f;
  const float gy = 2.f * ty / static_cast<float>(H - 1) - 1.f;
  const float px = ((gx + 1.f) * static_cast<float>(W) - 1.f) / 2.f;
  const float py = ((gy + 1.f) * static_cast<float>(H) - 1.f) / 2.f;
  const bool sane = (px > -1.0e6f) && (px < 1.0e6f) && (py > -1.0e6f) && (py < 1.0e6f);
  const float fx0 = floorf(px), fy0 = floorf(py);
  const int x0 = sane ? static_cast<int>(fx0) : -10, y0 = sane ? static_cast<int>(fy0) : -10;
  const float ax = px - fx0, ay = py - fy0;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  DescTaps t;
  t.w00 = (vx0 && vy0) ? (1.f - ax) * (1.f - ay) : 0.f;
  t.w10 = (vx1 && vy0) ? ax * (1.f - ay) : 0.f;
  t.w01 = (vx0 && vy1) ? (1.f - ax) * ay : 0.f;
  t.w11 = (vx1 && vy1) ? ax * ay : 0.f;
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
  t.o00 = static_cast<long long>(cy0) * W + cx0;
  t.o10 = static_cast<long long>(cy0) * W + cx1;
  t.o01 = static_cast<long long>(cy1) * W + cx0;
  t.o11 = static_cast<long long>(cy1) * W + cx1;
  return t;
}

// r06: the two taps of an image row as ONE 8-byte load.  corr_weight issues 5 four-byte load instructions per channel and pixel (160 per
// pixel at D = 32) for 250 MB per half batch: it runs at 4.0-4.6 TB/s, bound by the load path's instruction rate rather than by HBM.  The
// taps (x0, x0 + 1) of a row are neighbours in memory: the pair is loaded from column clamp(x0, 0, W - 2) (always inside the row; 4-byte
// aligned global_load_dwordx2) and the tap weights are re-assigned ONCE per pixel to whichever element of the pair carries each tap -- in
// the interior (a, b) = (v00, v10) with (wa, wb) = (w00, w10): the per-channel arithmetic is the same expression in the same order.  At
// the left / right border one tap is outside (weight 0) and the other one's weight moves to the element that holds its texel.
struct DescPairs {
  float wa0, wb0, wa1, wb1;     // weights of the pair elements of rows y0 / y1
  long long o0, o1;             // offsets of the pairs (row y0 / y1, column clamp(x0, 0, W - 2))
};

__device__ __forceinline__ DescPairs descriptor_pairs(float tx, float ty, int H, int W) {      // W >= 2
  const float gx = 2.f * tx / static_cast<float>(W - 1) - 1.f;
  const float gy = 2.f * ty / static_cast<float>(H - 1) - 1.f;
  const float px = ((gx + 1.f) * static_cast<float>(W) - 1.f) / 2.f;
  const float py = ((gy + 1.f) * static_cast<float>(H) - 1.f) / 2.f;
  const bool sane = (px > -1.0e6f) && (px < 1.0e6f) && (py > -1.0e6f) && (py < 1.0e6f);
  const float fx0 = floorf(px), fy0 = floorf(py);
  const int x0 = sane ? static_cast<int>(fx0) : -10, y0 = sane ? static_cast<int>(fy0) : -10;
  const float ax = px - fx0, ay = py - fy0;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  const float w00 = (vx0 && vy0) ? (1.f - ax) * (1.f - ay) : 0.f;
  const float w10 = (vx1 && vy0) ? ax * (1.f - ay) : 0.f;
  const float w01 = (vx0 && vy1) ? (1.f - ax) * ay : 0.f;
  const float w11 = (vx1 && vy1) ? ax * ay : 0.f;
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x0 + 1, 0), W - 1);
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y0 + 1, 0), H - 1);
  const int pc = min(max(x0, 0), W - 2);                   // pair column: pc and pc + 1 are inside the row
  const bool t0a = cx0 == pc, t1a = cx1 == pc;            // which pair element holds tap column x0 / x0 + 1 (the clamped texel the 4-tap form reads)
  DescPairs t;
  t.wa0 = (t0a ? w00 : 0.f) + (t1a ? w10 : 0.f);          // (interior: w00 + 0 and 0 + w10 -- exact)
  t.wb0 = (t0a ? 0.f : w00) + (t1a ? 0.f : w10);
  t.wa1 = (t0a ? w01 : 0.f) + (t1a ? w11 : 0.f);
  t.wb1 = (t0a ? 0.f : w01) + (t1a ? 0.f : w11);
  t.o0 = static_cast<long long>(cy0) * W + pc;
  t.o1 = static_cast<long long>(cy1) * W + pc;
  return t;
}

struct __attribute__((packed, aligned(4))) DescF2 { float x, y; };

template <int NB>
__device__ __forceinline__ float descriptor_dot_pairs(const float* __restrict__ a, const float* __restrict__ q, long long P, int D,
                                                      const DescPairs& t) {
  float s = 0.f;
  int c = 0;
  for (; c + NB <= D; c += NB) {
    float av[NB];
    DescF2 r0[NB], r1[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float* qc = q + (c + j) * P;
      av[j] = a[(c + j) * P];
      r0[j] = *reinterpret_cast<const DescF2*>(qc + t.o0);
      r1[j] = *reinterpret_cast<const DescF2*>(qc + t.o1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float wv = ((r0[j].x * t.wa0 + r0[j].y * t.wb0) + r1[j].x * t.wa1) + r1[j].y * t.wb1;
      s += av[j] * wv;
    }
  }
  for (; c < D; ++c) {
    const float* qc = q + c * P;
    const DescF2 r0 = *reinterpret_cast<const DescF2*>(qc + t.o0), r1 = *reinterpret_cast<const DescF2*>(qc + t.o1);
    const float wv = ((r0.x * t.wa0 + r0.y * t.wb0) + r1.x * t.wa1) + r1.y * t.wb1;
    s += a[c * P] * wv;
  }
  return s;
}

// a = g1 + (b * D) * P + pixel, q = g2 + (b * D) * P.  NB channels per batch: all their loads are issued before the first use (the
// fence keeps the compiler from serialising them behind vmcnt(0) waits, which it does as soon as the surrounding control flow
// changes: 104 vs 259 us per launch, r02)
template <int NB>
__device__ __forceinline__ float descriptor_dot(const float* __restrict__ a, const float* __restrict__ q, long long P, int D,
                                                const DescTaps& t) {
  float s = 0.f;
  int c = 0;
  for (; c + NB <= D; c += NB) {
    float av[NB], v00[NB], v10[NB], v01[NB], v11[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float* qc = q + (c + j) * P;
      av[j] = a[(c + j) * P];
#if RP_CW_TAPS_SC1   // diagnostics build: the taps through sc1 loads (never served by this CU's L1)
      v00[j] = __hip_atomic_load(qc + t.o00, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v10[j] = __hip_atomic_load(qc + t.o10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v01[j] = __hip_atomic_load(qc + t.o01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v11[j] = __hip_atomic_load(qc + t.o11, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      av[j] = __hip_atomic_load(a + (c + j) * P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      v00[j] = qc[t.o00];
      v10[j] = qc[t.o10];
      v01[j] = qc[t.o01];
      v11[j] = qc[t.o11];
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
#if RP_CW_WAIT0      // diagnostics build: every load of the batch has returned before the first value is used (no counted waits)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float wv = ((v00[j] * t.w00 + v10[j] * t.w10) + v01[j] * t.w01) + v11[j] * t.w11;
      s += av[j] * wv;
    }
  }
  for (; c < D; ++c) {
    const float* qc = q + c * P;
    const float wv = ((qc[t.o00] * t.w00 + qc[t.o10] * t.w10) + qc[t.o01] * t.w01) + qc[t.o11] * t.w11;
    s += a[c * P] * wv;
  }
  return s;
}

}  // namespace rp
