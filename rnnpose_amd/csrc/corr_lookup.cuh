// Shared by corr_lookup.hip (a3: the window lookup as a kernel of its own) and conv1x1_resident.hip (r06: the lookup as the STAGING
// phase of BasicMotionEncoder.convc1 -- the 324 window features of a pixel tile go straight into the convolution's LDS activation
// tile): the pyramid layout record and the cooperative footprint fetch of 16 pixels of one level by one wave.
// Reference: CorrBlock.__call__ + bilinear_sampler (thirdparty/raft/corr.py:36-57, thirdparty/raft/utils/utils.py:57-71).
#pragma once
#include "common.hpp"

#ifndef RPL_LB
#define RPL_LB 8          // pixels whose footprint loads are in flight together (measurement: 16)
#endif

namespace rplookup {

constexpr int R = 4;
constexpr int WIN = 2 * R + 1;      // 9
constexpr int FP = WIN + 1;         // 10: footprint side
constexpr int FS = FP * FP + 1;     // 101: per-pixel LDS stride (odd -> conflict-free lane-per-pixel reads)
constexpr int PIX = 16;             // pixels per wave and round

struct LookupInfo {
  long long off[RNNPOSE_MAX_LEVELS];
  int hl[RNNPOSE_MAX_LEVELS];
  int wl[RNNPOSE_MAX_LEVELS];
  int n_px, n_patch;          // level 0 is stored j-patch-major: [image][8 x 16 patch][i][8][16] (csrc/corr_pyramid.hip)
};

// Integer base and fractional offsets of the 10 x 10 footprint of a pixel whose window centre is (cx, cy) at this level's scale.
// Non-finite / far-away coordinates sample only padding -> zeros.
__device__ __forceinline__ void footprint_base(float cx, float cy, int& bx, int& by, float& ax, float& ay) {
  const bool sane = (cx > -1.0e6f) && (cx < 1.0e6f) && (cy > -1.0e6f) && (cy < 1.0e6f);
  const float fx0 = floorf(cx), fy0 = floorf(cy);
  bx = sane ? static_cast<int>(fx0) - R : -1000000;
  by = sane ? static_cast<int>(fy0) - R : -1000000;
  ax = sane ? cx - fx0 : 0.f;
  ay = sane ? cy - fy0 : 0.f;
}

// One wave fetches the footprints of `npix` (<= NP) pixels of level `lvl` into foot[q * FS + t] (t = 10 ty + tx), zeros outside the
// level.  Lane q < NP holds its pixel's footprint base (bx, by), its image in the pyramid (bg) and its pixel index inside it (pixg);
// `first_row` = pyramid row (= p_off + flat pixel index) of pixel 0 of the NP (NP = 16: corr_lookup.hip; 8: corr_convc1.hip).  Every load is UNCONDITIONAL (texels outside the level
// read element 0 of the pixel's map and are zeroed on the way into LDS; pixel slots past npix repeat the last pixel) and the loads of
// LB pixels are issued before the first LDS store: with the bounds test around the load the compiler waited vmcnt(0) after every
// pixel -- 16 dependent memory round trips per wave (r02: 46 us per half-batch launch, latency-bound).
template <int NP>
__device__ __forceinline__ void gather_px(const float* __restrict__ pyr, const LookupInfo& info, int lvl, int N, int lane, int npix,
                                         int bx, int by, int bg, int pixg, long long first_row, float* foot) {
  const int hl = info.hl[lvl], wl = info.wl[lvl];
  const float* lvl_base = pyr + info.off[lvl];
  const long long img = static_cast<long long>(hl) * wl;
  const int t0 = lane, t1 = lane + 64;
  const int ty0 = t0 / FP, tx0 = t0 - ty0 * FP;
  const int ty1 = t1 / FP, tx1 = t1 - ty1 * FP;
  constexpr int LB = RPL_LB < NP ? RPL_LB : NP;
  const bool has1 = t1 < FP * FP;
#pragma unroll
  for (int qb = 0; qb < NP; qb += LB) {
    float v0[LB], v1[LB];
    unsigned ok = 0u;
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int q = qb + j;
      const int qq = q < npix ? q : npix - 1;
      const int qbx = __shfl(bx, qq), qby = __shfl(by, qq);
      const int xa = qbx + tx0, ya = qby + ty0, xb = qbx + tx1, yb = qby + ty1;
      const bool oka = xa >= 0 && xa < wl && ya >= 0 && ya < hl;
      const bool okb = has1 && xb >= 0 && xb < wl && yb >= 0 && yb < hl;
      // level 0 is j-patch-major: texel (y, x) of pixel i = patch ((y >> 3) n_px + (x >> 4)), row i, cell (y & 7, x & 15); the other
      // levels are row-major maps per pixel.  Both forms as SELECTS on the (wave-uniform) level, not as a branch: a branch around the
      // loads brought the one-wait-per-load form back (tests/test_isa_guard.py: 24 vmcnt(0) waits for 50 loads)
      const int qbg = __shfl(bg, qq), qpix = __shfl(pixg, qq);
      const long long pstride = static_cast<long long>(N) * 128;
      const bool l0 = lvl == 0;
      const float* src = lvl_base + (l0 ? (static_cast<long long>(qbg) * info.n_patch * N + qpix) * 128 : (first_row + qq) * img);
      const long long ea = l0 ? ((ya >> 3) * info.n_px + (xa >> 4)) * pstride + (ya & 7) * 16 + (xa & 15) : static_cast<long long>(ya * wl + xa);
      const long long eb = l0 ? ((yb >> 3) * info.n_px + (xb >> 4)) * pstride + (yb & 7) * 16 + (xb & 15) : static_cast<long long>(yb * wl + xb);
      v0[j] = src[oka ? ea : 0];
      v1[j] = src[okb ? eb : 0];
      ok |= (oka ? 1u : 0u) << (2 * j) | (okb ? 2u : 0u) << (2 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int q = qb + j;
      foot[q * FS + t0] = (ok >> (2 * j)) & 1u ? v0[j] : 0.f;
      if (has1) foot[q * FS + t1] = (ok >> (2 * j)) & 2u ? v1[j] : 0.f;
    }
  }
}

// r06: the same fetch with a third of the instructions (the lookup is bound by INSTRUCTIONS per wave, not by a pipe: ~1600 per 16 pixels of one
// level, ~1000 of them here -- per load a 64-bit patch-stride multiply, both address forms evaluated and selected, four bpermutes per pixel):
//   * L0 (level 0, j-patch-major) or not is a TEMPLATE parameter: one address form, chosen by one wave-uniform branch around the whole body;
//   * the pixel's footprint base / image / index come from v_readlane (the pixel slot is wave-uniform): scalars, and the pixel's map base is a
//     scalar 64-bit address; a lane's texel is an unsigned 32-bit BYTE offset from it (a level-0 image is N^2 floats <= 1.5 GB at 120 x 160);
//   * bounds tests as unsigned compares.
// Same texels into the same LDS slots: the results do not change by a bit.
template <int NP, bool L0>
__device__ __forceinline__ void gather_px_fast(const float* __restrict__ pyr, const LookupInfo& info, int lvl, int N, int lane, int npix,
                                              int bx, int by, int bg, int pixg, long long first_row, float* foot) {
  const int hl = info.hl[lvl], wl = info.wl[lvl];
  const char* const lvl_base = reinterpret_cast<const char*>(pyr + info.off[lvl]);
  const unsigned img_b = static_cast<unsigned>(hl * wl) * 4u;               // bytes of one pixel's map (levels >= 1)
  const unsigned pstride = static_cast<unsigned>(N) * 128u;                 // level 0: floats from one j patch to the next
  const int t0 = lane, t1 = lane + 64;
  const int ty0 = t0 / FP, tx0 = t0 - ty0 * FP;
  const int ty1 = t1 / FP, tx1 = t1 - ty1 * FP;
  constexpr int LB = RPL_LB < NP ? RPL_LB : NP;
  const bool has1 = t1 < FP * FP;
  const unsigned uw = static_cast<unsigned>(wl), uh = static_cast<unsigned>(hl);
#pragma unroll
  for (int qb = 0; qb < NP; qb += LB) {
    float v0[LB], v1[LB];
    unsigned ok = 0u;
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int q = qb + j;
      const int qq = q < npix ? q : npix - 1;                               // (wave-uniform)
      const int qbx = __builtin_amdgcn_readlane(bx, qq), qby = __builtin_amdgcn_readlane(by, qq);
      const char* src;
      if constexpr (L0) {
        const int qbg = __builtin_amdgcn_readlane(bg, qq), qpix = __builtin_amdgcn_readlane(pixg, qq);
        src = lvl_base + (static_cast<long long>(qbg) * info.n_patch * N + qpix) * 512;
      } else {
        src = lvl_base + (first_row + qq) * static_cast<long long>(img_b);
      }
      const int xa = qbx + tx0, ya = qby + ty0, xb = qbx + tx1, yb = qby + ty1;
      const bool oka = static_cast<unsigned>(xa) < uw && static_cast<unsigned>(ya) < uh;
      const bool okb = has1 && static_cast<unsigned>(xb) < uw && static_cast<unsigned>(yb) < uh;
      unsigned ea, eb;
      if constexpr (L0) {     // texel (y, x) of pixel i = patch ((y >> 3) n_px + (x >> 4)), row i, cell (y & 7, x & 15)
        ea = (static_cast<unsigned>((ya >> 3) * info.n_px + (xa >> 4)) * pstride + static_cast<unsigned>(((ya & 7) << 4) | (xa & 15))) * 4u;
        eb = (static_cast<unsigned>((yb >> 3) * info.n_px + (xb >> 4)) * pstride + static_cast<unsigned>(((yb & 7) << 4) | (xb & 15))) * 4u;
      } else {
        ea = static_cast<unsigned>(ya * wl + xa) * 4u;
        eb = static_cast<unsigned>(yb * wl + xb) * 4u;
      }
      v0[j] = *reinterpret_cast<const float*>(src + (oka ? ea : 0u));
      v1[j] = *reinterpret_cast<const float*>(src + (okb ? eb : 0u));
      ok |= (oka ? 1u : 0u) << (2 * j) | (okb ? 2u : 0u) << (2 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int q = qb + j;
      foot[q * FS + t0] = (ok >> (2 * j)) & 1u ? v0[j] : 0.f;
      if (has1) foot[q * FS + t1] = (ok >> (2 * j)) & 2u ? v1[j] : 0.f;
    }
  }
}

__device__ __forceinline__ void gather16(const float* __restrict__ pyr, const LookupInfo& info, int lvl, int N, int lane, int npix,
                                         int bx, int by, int bg, int pixg, long long first_row, float* foot) {
  gather_px<PIX>(pyr, info, lvl, N, lane, npix, bx, by, bg, pixg, first_row, foot);
}

}  // namespace rplookup
