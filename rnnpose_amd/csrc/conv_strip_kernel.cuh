// The strip convolution kernel template (see conv_strip.hip for the design notes); included by conv_strip.hip (160-row strips),
// and conv_strip_r32.hip (32-row strips): one translation unit per strip height so that they compile in parallel.
#pragma once
#include "common.hpp"
#include "f16x3.cuh"
#include "conv_common.cuh"

namespace {

using namespace rpconv;

// Strip height = 32 SMI rows (SMI = 32-row MFMA tiles per wave): 160 (SMI = 5: the form the launches of the headline come out even
// with) or 32 (SMI = 1) for launches that would not fill the chip with 160-row strips (r04: strip_rows()).
constexpr int SHALO = 4;                // linear strips: halo rows on each side (taps up to +-2 along the fast axis)
constexpr int SPW = 16;                 // 3x3: patch of 2 SMI x 16 pixels ...
constexpr int SHW = SPW + 2;            // ... staged with a one-pixel halo: (2 SMI + 2) x 18 rows

__device__ __attribute__((aligned(64))) const unsigned char g_zero_page[64] = {0};

// LDS-DMA requests.  Inline asm on purpose: hipcc (ROCm 7.2) models the builtin as a FLAT access to both global memory and LDS,
// and while one is outstanding every wait it inserts becomes lgkmcnt(0) / vmcnt(0) -- the fragment prefetch of the next step
// would be waited for in front of every MFMA.  The compiler neither counts nor waits for these requests: every wait is an
// explicit counted s_waitcnt vmcnt below.  M0 (the DMA's LDS base) is saved and restored inside the statement.
// One piece: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [dst, dst + 1024) (wave-uniform dst).
__device__ __forceinline__ void glds16(const void* g, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(dst)
      : "memory");
}
// One weight record of NI x 2 KB: lanes read base + voff (+ 1024 k) into [dst, dst + 2048 NI).  The instruction offset applies to
// the global address AND to the LDS address; scalar base: no per-lane address arithmetic (s_nop 4: the base may come from a
// v_readfirstlane)
template <int NI>
__device__ __forceinline__ void glds_rec(const void* base, unsigned voff, unsigned dst) {
  unsigned keep;
  if constexpr (NI == 1) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(dst)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(dst)
        : "memory");
  }
}
// gates of the fast epilogue: v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the library's expf / tanhf / IEEE division --
// |error| <= ~2e-7 absolute on values in [0, 1] / [-1, 1]; saturates correctly (exp2 -> inf -> rcp -> 0)
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ float fast_tanh(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * 2.8853900817779268f)); }

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifndef RS_ABL
#define RS_ABL 0
#endif
#if RS_ABL & 512          // diagnostics build: wave 0 of every workgroup stamps the 100-MHz clock at four points (tools/strip_timeline.py)
__device__ unsigned long long g_strip_clk[8 * 65536];
#define RS_CLK(I_) { if (tid == 0 && blockIdx.x < 65536) g_strip_clk[blockIdx.x * 8 + (I_)] = wall_clock64(); }
#else
#define RS_CLK(I_)
#endif
#ifndef RS_SLICE_BITS
#define RS_SLICE_BITS 9    // (100-MHz clock: 2^9 ticks = 5.12 us)
#endif
#ifndef RS_S2_SKIP
#define RS_S2_SKIP 0      // 1: the stride-2 form skips the MFMAs of its seven empty (plane, tap) slots instead of multiplying by their zero records.
#endif                    // Built, right, and SLOWER (profiles/r05_ab_s2_skip.txt: l2.0.c1 139 vs 116 us, step 733 vs 745 iters/s): the branch takes the
                          // fragment reads out from between the MFMAs and costs 21-41 spilled registers; the kernel is not MFMA-bound there
#ifndef RS_RING3
#define RS_RING3 1        // 0: the two-wave 3x3 workgroups keep the 6-slot weight ring (r04; same-box A/B)
#endif
#ifndef RS_VAR
#define RS_VAR 0          // schedule variants (measurement): 1 s_setprio(1) around a step's MFMAs, 2 fragment reads in front of the
#endif                    // MFMAs instead of between them, 4 the first read behind the third MFMA
#ifndef RS_ABL
#define RS_ABL 0          // diagnostics builds (tools/strip_ablate.sh; results WRONG): 1 no weight requests in the loop, 2 no activation
#endif                    // requests, 4 no fragment reads, 8 no MFMAs, 32 no barrier, 64 no vmcnt waits, 128 no activation fragment
                          // reads, 256 no weight fragment reads
// ---- fast epilogue (r04; tools/strip_timeline.py: the general form in the kernel costs 6-13 us of a workgroup's 30-60 us -- ~7
// cycles per instruction of a code path that tests every option and every row's validity per row group, multiplies 64-bit
// addresses and waits for each staging write on the spot).  For a wave whose 160 x 32 tile lies INSIDE the problem and one of the
// five option sets the engines use (EV: 0 linear / ReLU -> fp32; 1 the same + tile statistics; 2 linear / ReLU -> split;
// 3 GRU z | r*h with an additive map; 4 GRU h' with an additive map and a split copy): no validity masks, no option tests, the five
// 32-row blocks unrolled over two staging tiles (block mi+1 is written while block mi is read back), the operands of block
// mi+1 requested BEFORE the stores of block mi (vmcnt is in order: a load behind stores waits for them), pixel indices by a walk,
// 32-bit element offsets (strip_launch checks the extents), gates by v_exp / v_rcp.
struct StripEpi {        // the members of the parameter block the fast epilogue reads, by value (a reference to the whole block
  float* dst;            // made the compiler keep its segment table in scratch memory)
  int dst_cs, dst_co;
  float* dst2;
  int dst2_cs, dst2_co;
  float* dsth;
  int dsth_cs, dsth_co;
  const float* addm;
  int addm_cs, addm_co;
  const float* aux0;
  int aux0_cs, aux0_co;
  const float* aux1;
  int aux1_cs, aux1_co;
  float out_scale, a_scale;
  int epi, gru_c, U, V, su, sv;
};
template <int EV, bool SPATIAL, int NBLK, int NI>
__device__ __forceinline__ void strip_epilogue_fast(const StripEpi p, f32x16 (&acc)[NBLK][NI], float* S_, int lane, int colq, int c2,
                                                    const float (&bq)[4], int pixb, int lv, int lu, int UV, double& ts0, double& ts1,
                                                    double& ts2, double& ts3, double& tq0, double& tq1, double& tq2, double& tq3,
                                                    int& sat_n) {
  constexpr int ES = 36, STILE = 32 * ES;
  const int l31 = lane & 31, lh = lane >> 5;
  float* const Sw = S_ + (4 * lh) * ES + l31;                   // staging write base: row (r & 3) + 8 (r >> 2) + 4 lh, column l31
  const float* const Sr = S_ + (lane >> 3) * ES + (lane & 7) * 4;      // read base: row 8 k + (lane >> 3), the lane's quad
  const bool zside = colq < p.gru_c;                            // (EV 3: uniform in the wave, gru_c is a multiple of 32)
  const float lo_ = p.epi == 1 ? 0.f : -__builtin_inff();       // ReLU as a clamp
  const int wrap_v = p.su - p.V * p.sv, wrap_u = UV - p.U * p.su;
  // pixel of row group k of block mi: 3x3 patches: line 2 mi + (k >> 1), column 8 (k & 1) + (lane >> 3) of the patch; linear
  // strips: row 32 mi + 8 k + (lane >> 3) of the strip, walked 8 rows at a time through (fast coordinate, slow coordinate, image)
  int lpix = pixb;
  auto pixel = [&](int mi, int k) {
    if constexpr (SPATIAL) {
      return pixb + (k & 1) * 8 + (2 * mi + (k >> 1)) * p.V;
    } else {
      const int r = lpix;
      lv += 8;
      lpix += 8 * p.sv;
      if (lv >= p.V) {
        lv -= p.V;
        lpix += wrap_v;
        if (++lu >= p.U) { lu = 0; lpix += wrap_u; }
      }
      return r;
    }
  };
  auto eoff = [](int pix, int cs, int co) { return static_cast<size_t>(static_cast<unsigned>(pix) * static_cast<unsigned>(cs) + static_cast<unsigned>(co)); };
  auto stage = [&](int mi, const f32x16& a) {
    float* w = Sw + (mi & 1) * STILE;
#pragma unroll
    for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2)) * ES] = a[r] * p.out_scale;
  };
  int pix[4], pixn[4];
  float4 am[4], hv[4], zv[4];
  auto request = [&](const int (&px)[4]) {       // operands of a block: additive map, h, z
    if constexpr (EV >= 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) am[k] = *reinterpret_cast<const float4*>(p.addm + eoff(px[k], p.addm_cs, p.addm_co + colq));
    }
    if constexpr (EV == 3) {
      if (!zside) {
#pragma unroll
        for (int k = 0; k < 4; ++k) hv[k] = *reinterpret_cast<const float4*>(p.aux0 + eoff(px[k], p.aux0_cs, p.aux0_co + c2));
      }
    }
    if constexpr (EV == 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        zv[k] = *reinterpret_cast<const float4*>(p.aux1 + eoff(px[k], p.aux1_cs, p.aux1_co + colq));
        hv[k] = *reinterpret_cast<const float4*>(p.aux0 + eoff(px[k], p.aux0_cs, p.aux0_co + colq));
      }
    }
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) pix[k] = pixel(0, k);
  request(pix);
  stage(0, acc[0][0]);
#pragma unroll
  for (int mi = 0; mi < NBLK; ++mi) {
    if (mi + 1 < NBLK) stage(mi + 1, acc[mi + 1 < NBLK ? mi + 1 : mi][0]);
    float4 y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(Sr + (mi & 1) * STILE + k * 8 * ES);
      y[k] = make_float4(a.x + bq[0], a.y + bq[1], a.z + bq[2], a.w + bq[3]);
    }
    if constexpr (EV >= 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { y[k].x += am[k].x; y[k].y += am[k].y; y[k].z += am[k].z; y[k].w += am[k].w; }
    }
    if constexpr (EV == 1) {        // tile statistics: four rows per column and block in fp32, the block sums in fp64
      ts0 += static_cast<double>((y[0].x + y[1].x) + (y[2].x + y[3].x));
      ts1 += static_cast<double>((y[0].y + y[1].y) + (y[2].y + y[3].y));
      ts2 += static_cast<double>((y[0].z + y[1].z) + (y[2].z + y[3].z));
      ts3 += static_cast<double>((y[0].w + y[1].w) + (y[2].w + y[3].w));
      tq0 += static_cast<double>(fmaf(y[0].x, y[0].x, y[1].x * y[1].x) + fmaf(y[2].x, y[2].x, y[3].x * y[3].x));
      tq1 += static_cast<double>(fmaf(y[0].y, y[0].y, y[1].y * y[1].y) + fmaf(y[2].y, y[2].y, y[3].y * y[3].y));
      tq2 += static_cast<double>(fmaf(y[0].z, y[0].z, y[1].z * y[1].z) + fmaf(y[2].z, y[2].z, y[3].z * y[3].z));
      tq3 += static_cast<double>(fmaf(y[0].w, y[0].w, y[1].w * y[1].w) + fmaf(y[2].w, y[2].w, y[3].w * y[3].w));
    }
    if constexpr (EV <= 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = make_float4(fmaxf(y[k].x, lo_), fmaxf(y[k].y, lo_), fmaxf(y[k].z, lo_), fmaxf(y[k].w, lo_));
    } else if constexpr (EV == 3) {
      if (zside) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = make_float4(fast_sigmoid(y[k].x), fast_sigmoid(y[k].y), fast_sigmoid(y[k].z), fast_sigmoid(y[k].w));    // z
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)                                                                                                               // r * h
          y[k] = make_float4(fast_sigmoid(y[k].x) * hv[k].x, fast_sigmoid(y[k].y) * hv[k].y, fast_sigmoid(y[k].z) * hv[k].z, fast_sigmoid(y[k].w) * hv[k].w);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {                                                                                                               // h' = (1-z) h + z q
        const float4 z = zv[k], h = hv[k];
        y[k] = make_float4((1.f - z.x) * h.x + z.x * fast_tanh(y[k].x), (1.f - z.y) * h.y + z.y * fast_tanh(y[k].y),
                           (1.f - z.z) * h.z + z.z * fast_tanh(y[k].z), (1.f - z.w) * h.w + z.w * fast_tanh(y[k].w));
      }
    }
    if (mi + 1 < NBLK) {            // (the next block's operands go out in front of this block's stores)
#pragma unroll
      for (int k = 0; k < 4; ++k) pixn[k] = pixel(mi + 1, k);
      request(pixn);
    }
    if constexpr (EV == 0 || EV == 1 || EV == 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(p.dst + eoff(pix[k], p.dst_cs, p.dst_co + colq)) = y[k];
    }
    if constexpr (EV == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) store_quad_hl(p.dst + eoff(pix[k], p.dst_cs, 0), p.dst_co + colq, y[k].x, y[k].y, y[k].z, y[k].w, 4, p.a_scale, sat_n);
    }
    if constexpr (EV == 3) {
      if (zside) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(p.dst + eoff(pix[k], p.dst_cs, p.dst_co + colq)) = y[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) store_quad_hl(p.dst2 + eoff(pix[k], p.dst2_cs, 0), p.dst2_co + c2, y[k].x, y[k].y, y[k].z, y[k].w, 4, p.a_scale, sat_n);
      }
    }
    if constexpr (EV == 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) store_quad_hl(p.dsth + eoff(pix[k], p.dsth_cs, 0), p.dsth_co + colq, y[k].x, y[k].y, y[k].z, y[k].w, 4, p.a_scale, sat_n);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) pix[k] = pixn[k];
  }
}

// TT = taps per half block: 5 (1x5, 5x1: linear strips) or 9 (3x3: 10 x 16 patches with a halo).
// MODE 0: split-tensor sources by LDS-DMA; 1: fp32 sources through registers; 2: fp32 source + fused instance norm / ReLU (p.in_mr)
// NI = 32-column tiles per wave: 1 (wave tile 160 x 32, two workgroups per CU = two waves per SIMD, <= 256 registers) or
// 2 (160 x 64, ONE wave per SIMD with up to 512 registers).  Round 4 ablation of the NI = 1 form (profiles/r04_strip_ablation.txt):
// without MFMAs its LDS / DMA stream alone takes 36 of 78 us -- every wave reads the WHOLE activation tile from LDS for its 32
// columns, 12 ds_read_b128 per 15 MFMAs, and the LDS pipe is a co-bottleneck of the matrix pipe.  With two column tiles per wave
// an activation fragment feeds twice the MFMAs: 14 reads per 30.
// P1 (r04; cfg.raft.mixed_precision, the reference's GPU arithmetic: model/CFNet.py:47,126,152 run the encoder and the update block under
// fp16 autocast): ONE product per multiply-add -- a_hi * b_hi on the same MFMA, fp32 accumulation -- instead of three.  The lo planes
// still arrive (the operands in memory are the same split tensors; activations between layers stay fp32-class, which autocast's are
// not); their fragments are neither read nor multiplied.  Not the headline arithmetic: narrower than the CPU oracle's fp32.
#ifndef RS_MINW3
#define RS_MINW3 0        // measurement (r06): 1 = the two-wave 96-row 3x3 workgroups are compiled for THREE waves per SIMD (<= 168 registers)
#endif
template <int NW, int TT, int MODE, int NI, int SMI, bool P1 = false, bool PERSIST = false>
__global__ __launch_bounds__(NW * 64, NI == 2 ? 1 : ((RS_MINW3 && NW == 2 && SMI == 3 && TT == 9) ? 3 : 2)) void conv_strip_f16x3_kernel(const KParams p) {
  constexpr bool SPATIAL = TT == 9 || TT == 4;             // TT = 4 (r05): a STRIDE-2 3x3 layer as a 2x2-tap layer over the four parity planes
  constexpr bool S2 = TT == 4;                             // of its input (strided views of the NHWC source: conv_strip.hip, strip_launch)
  constexpr int SM = 32 * SMI;                             // strip rows
  constexpr int SPH = 2 * SMI;                             // 3x3: patch lines (10 x 16, 4 x 16, 2 x 16 pixels)
  constexpr int SHR = (SPH + 2) * SHW;                     // staged rows of a patch with its halo (216 / 108 / 72)
  constexpr int BREC = 2048 * NI;                          // weight record of one wave and step
  // weight ring slots per wave: 2 TT is a multiple (slots are compile-time).  r05: the TWO-wave 3x3 workgroups (64-column layers: the
  // encoder's 240 x 320 residual blocks) take a 3-slot ring -- 40 KB of LDS per workgroup instead of 52: FOUR workgroups per CU = two
  // waves per SIMD instead of three = 1.5.  Their records are requested two steps ahead (L2-resident weights: 36 K values per layer).
  constexpr bool RING3 = RS_RING3 && TT == 9 && NW == 2 && NI == 1 && (SMI == 5 || (RS_MINW3 && SMI == 3));
  constexpr int NBST = SPATIAL ? (RING3 ? 3 : (S2 ? 4 : 6)) : 5;
  static_assert((2 * TT) % NBST == 0 && NBST - 2 < TT - 1, "ring period");
  constexpr int NT_ = NW * 64;
  constexpr int ARV = SPATIAL ? SHR : SM + 2 * SHALO;      // staged rows that carry data (216 / 168)
  constexpr int AR = (ARV + 31) / 32 * 32;                 // rows per plane: whole 32-row DMA pieces (224 / 192); rows >= ARV stay zero
  constexpr int PLANE = AR * 32;                           // hi plane, then lo plane: 32 bytes per row
  constexpr int ASLOT = 2 * PLANE;
  constexpr int NPP = AR / 32, NPIECE = 2 * NPP;           // DMA pieces per plane / per half block
  constexpr int PA = (NPIECE + NW - 1) / NW;               // DMA pieces per wave and half block
  constexpr int NQ = (ARV * 4 + NT_ - 1) / NT_;            // register path: float4 per thread and half block
  constexpr int PAW = MODE == 0 ? PA : (MODE == 1 ? NQ : NQ + 2);      // vector-memory requests of one activation half block per wave
  constexpr int ZROW = ARV * 32;                           // an all-zero row of either plane (taps outside the image line read it)
  constexpr int BOFF = 2 * ASLOT;
  constexpr int LDSB = BOFF + NW * NBST * BREC;
  static_assert(PAW + 2 * NI * (NBST - 3) <= 63, "vmcnt is a 6-bit counter");
  static_assert(ZROW + 32 <= PLANE, "a zero row behind the data rows");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDSB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  RS_CLK(0)

  // ---- tile walk: XCD-contiguous chunks of the tile list, column tiles of one strip next to each other (they stage the same
  // activations).  One tile per workgroup (gridDim.x = tiles), or -- r06, p.n_tiles > 0: a PERSISTENT launch of gridDim.x (a multiple of 8)
  // resident workgroups -- workgroup w walks tiles idx, idx + gridDim.x / 8, ... of the chunk of its XCD (w & 7) and requests the first
  // two half blocks of tile t + 1 BEFORE the epilogue of tile t (fp32-source forms; see the loop at the end of the prologue).
  const int t_ntl = PERSIST ? p.n_tiles : static_cast<int>(gridDim.x);
  const int t_per = t_ntl >> 3, t_rem = t_ntl & 7, t_xcd = static_cast<int>(blockIdx.x) & 7;
  const int t_base = t_xcd * t_per + (t_xcd < t_rem ? t_xcd : t_rem), t_cnt = t_per + (t_xcd < t_rem ? 1 : 0);
  const int t_step = PERSIST ? static_cast<int>(gridDim.x) >> 3 : 0x40000000;
  int t_idx = static_cast<int>(blockIdx.x) >> 3;
  if constexpr (PERSIST) {           // (measurement, r06: stagger the co-resident workgroups of a CU -- workgroup i of an XCD sits in slot i / 32 of CU i % 32)
    if (p.stagger > 0) {
      const unsigned long long t_end = wall_clock64() + static_cast<unsigned long long>(((static_cast<int>(blockIdx.x) >> 3) >> 5) * p.stagger);
      while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(32);
    }
  }
  const int UV = p.U * p.V;
  const int Mtot = p.B * UV;
  // geometry of the tile whose activations are being requested ...
  int nt_i, mt_i, img_, py0_, px0_, m0, mend, ct32;
  // ... and of the tile the epilogue writes (the same one until the next tile's first requests go out in front of the epilogue)
  int e_mt_i, e_img, e_py0, e_px0, e_m0, e_mend, e_ct32;
#define RS_TILE_GEOM(BID_)                                                                                   \
  {                                                                                                          \
    const int bid_ = (BID_);                                                                                 \
    nt_i = bid_ % p.n_nt; mt_i = bid_ / p.n_nt;                                                              \
    img_ = p.tpi > 0 ? mt_i / p.tpi : 0;                                                                     \
    const int pt_ = p.tpi > 0 ? mt_i - img_ * p.tpi : 0;                                                     \
    py0_ = SPATIAL ? (pt_ / p.sp_tx) * SPH : 0; px0_ = SPATIAL ? (pt_ % p.sp_tx) * SPW : 0;                  \
    m0 = p.tpi > 0 ? img_ * UV + pt_ * SM : mt_i * SM;                                                       \
    mend = p.tpi > 0 ? (img_ + 1) * UV : Mtot;                                                               \
    ct32 = (nt_i * NW + wave) * NI;        /* this wave's first 32-column tile */                            \
  }
  RS_TILE_GEOM(t_base + t_idx)
  const int ntiles32 = p.Npad >> 5;
  unsigned char* const sB = lds + BOFF + wave * (NBST * BREC);      // this wave's weight ring
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)lds));   // LDS byte address
  const unsigned sB0 = lds0 + BOFF + wave * (NBST * BREC);

  // staged row j of the strip -> pixel index (or -1: outside the image / the problem -> zeros)
#define RS_ROW_PIXEL(OUT_, J_)                                                                               \
  {                                                                                                          \
    const int j_ = (J_);                                                                                     \
    if (SPATIAL) {                                                                                           \
      const int hy_ = j_ / SHW, y_ = py0_ - 1 + hy_, x_ = px0_ - 1 + (j_ - hy_ * SHW);                       \
      const bool ok_ = j_ < SHR && static_cast<unsigned>(y_) < static_cast<unsigned>(p.U) &&                 \
                       static_cast<unsigned>(x_) < static_cast<unsigned>(p.V);                               \
      OUT_ = ok_ ? img_ * (p.Uin * p.Vin) + y_ * p.su + x_ * p.sv : -1;    /* (stride 1: Uin x Vin = U x V, su = V, sv = 1; stride 2: a parity plane) */ \
    } else {                                                                                                 \
      const int m_ = m0 - SHALO + j_;                                                                        \
      const bool ok_ = j_ < ARV && m_ >= 0 && m_ < Mtot;                                                     \
      const int mm_ = ok_ ? m_ : 0;                                                                          \
      const int q_ = mm_ / p.V, v_ = mm_ - q_ * p.V;                                                         \
      const int b_ = q_ / p.U, u_ = q_ - b_ * p.U;                                                           \
      OUT_ = ok_ ? b_ * UV + u_ * p.su + v_ * p.sv : -1;                                                     \
    }                                                                                                        \
  }
  // swizzle bit of staged row j: bit 3 of the row (linear strips), parity of the halo-tile line (3x3)
#define RS_SWZ(J_) (SPATIAL ? (((J_) / SHW) & 1) : (((J_) >> 3) & 1))
  // MODE 0: DMA piece q = wave + NW i of a half block = 32 rows of one plane; lane -> row lane >> 1, group position lane & 1,
  // which holds group (lane & 1) ^ swizzle(row): apix = the row's pixel, asrc = byte offset of the lane's 16 bytes inside the
  // pixel's 64-byte half block ([g0 hi | g0 lo | g1 hi | g1 lo])
  int apix[PA > NQ ? PA : NQ];
  int asrc[MODE == 0 ? PA : 1];
#define RS_TILE_ROWS()                                                                                       \
  {                                                                                                          \
    if constexpr (MODE == 0) {                                                                               \
      _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                       \
        const int q = wave + NW * i < NPIECE ? wave + NW * i : NPIECE - 1;      /* (surplus requests repeat the last piece: identical bytes) */ \
        const int plane = q / NPP, j = (q - plane * NPP) * 32 + (lane >> 1);                                 \
        RS_ROW_PIXEL(apix[i], j)                                                                             \
        asrc[i] = (((lane & 1) ^ RS_SWZ(j)) << 5) + (plane << 4);                                            \
      }                                                                                                      \
    } else {        /* register path: quad idx = tid + NT_ * i -> row idx >> 2, channels 4 (idx & 3) .. +3 of the half block */ \
      _Pragma("unroll") for (int i = 0; i < NQ; ++i) {                                                       \
        const int idx = tid + NT_ * i;                                                                       \
        RS_ROW_PIXEL(apix[i], idx >> 2)                                                                      \
      }                                                                                                      \
    }                                                                                                        \
  }
  RS_TILE_ROWS()
  // fragment addresses of this lane inside an activation slot (hi plane; lo: + PLANE), tile row r = 32 mi + l31:
  //   linear: one per (row tile, tap), the tap mask folded in (taps beyond the image line read the zero row)
  //   3x3:    per row tile the top-left tap of an even-offset line (dy = 0, 2) and of the centre line (dy = 1); tap (dy, dx) adds
  //           the compile-time offset (dy = 2 ? 2 * 18 * 32 : 0) + 32 dx
  int aad[SMI][SPATIAL ? 2 : TT];
#define RS_TILE_FRAG_ADDR()     /* (3x3: the same for every tile; linear strips: the tap masks depend on where the strip starts) */ \
  {                                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < SMI; ++mi) {                                                     \
      const int r = mi * 32 + l31;                                                                           \
      if constexpr (SPATIAL) {                                                                               \
        const int hyc = (r >> 4) + 1, xc = (r & 15) + 1;                                                     \
        aad[mi][0] = ((hyc - 1) * SHW + xc - 1) * 32 + ((lh ^ ((hyc - 1) & 1)) << 4);                        \
        aad[mi][1] = (hyc * SHW + xc - 1) * 32 + ((lh ^ (hyc & 1)) << 4);                                    \
      } else {                                                                                               \
        const int m = m0 + r;                                                                                \
        const int fv = m < mend ? m % p.V : -1000;          /* fast-axis coordinate */                       \
        _Pragma("unroll") for (int t = 0; t < TT; ++t) {                                                     \
          const int dv = p.dv0 + t, row = r + SHALO + dv;                                                    \
          const bool ok = static_cast<unsigned>(fv + dv) < static_cast<unsigned>(p.V);                       \
          aad[mi][t] = ok ? row * 32 + ((lh ^ ((row >> 3) & 1)) << 4) : ZROW + (lh << 4);                    \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  }
  RS_TILE_FRAG_ADDR()
  const int boff = l31 * 32 + ((lh ^ ((l31 >> 3) & 1)) << 4);      // this lane's hi fragment inside a weight record (lo: + 1024)
  const unsigned lane16 = lane * 16;

  f32x16 acc[SMI][NI];

  const int NHB = 2 * p.ncb;                 // half blocks (even)
  const int S = NHB * TT;                    // steps
  float4 areg[MODE == 0 ? 1 : NQ];           // register path: the half block in flight
  float4 nrm01 = make_float4(0.f, 1.f, 0.f, 1.f), nrm23 = nrm01;
  int sat_n = 0;
  bool sat_here = MODE != 0 && p.sat != nullptr && nt_i == 0;

  // ---- activation half blocks are requested in order 0, 1, 2, ...: the per-lane source pointers of the NEXT one to request are
  // kept and advanced by 64 bytes (16 channels) per half block inside a source segment; only at a segment change (virtual concat:
  // at most three times per launch) are they rebuilt from the parameter block.  Rows outside the image / the problem point at the
  // zero page and do not advance.  Past the last half block the pointers stay: the surplus requests of the last steps re-read it.
  int a_next = 0;                            // half block the pointers stand at
  int a_seg_end = 0;                         // first half block of the next segment
  const unsigned char* asp[MODE == 0 ? PA : NQ];       // MODE 0: this lane's 16 bytes of piece i; else: its float4 of quad i
  unsigned ainc[MODE == 0 ? PA : NQ];
  const float* nsp = p.in_mr;                // MODE 2: mean / rstd of the lane's four channels
#define RS_A_REBUILD()                                                                                       \
  {                                                                                                          \
    Seg sg_ = p.seg0;                                                                                        \
    int cb0_ = 0, cbe_ = p.cb1;                                                                              \
    const int blk_ = a_next >> 1;                                                                            \
    if (blk_ >= p.cb1) { sg_ = p.seg1; cb0_ = p.cb1; cbe_ = p.cb2; }                                         \
    if (blk_ >= p.cb2) { sg_ = p.seg2; cb0_ = p.cb2; cbe_ = p.cb3; }                                         \
    if (blk_ >= p.cb3) { sg_ = p.seg3; cb0_ = p.cb3; cbe_ = p.ncb; }                                         \
    a_seg_end = 2 * (cbe_ < p.ncb ? cbe_ : p.ncb);                                                           \
    const int cl_ = (blk_ - cb0_) * BK + (a_next & 1) * 16 + (MODE == 0 ? 0 : (tid & 3) * 4);                \
    const unsigned char* sb_ = reinterpret_cast<const unsigned char*>(sg_.ptr + sg_.coff + cl_);            \
    _Pragma("unroll") for (int i = 0; i < (MODE == 0 ? PA : NQ); ++i) {                                      \
      const bool ok_ = apix[i] >= 0;                                                                         \
      const unsigned eo_ = static_cast<unsigned>(apix[i]) * static_cast<unsigned>(sg_.cstride);              \
      if (MODE == 0) asp[i] = ok_ ? sb_ + (static_cast<size_t>(eo_) * 4 + asrc[i]) : g_zero_page + ((lane & 3) << 4); \
      else asp[i] = ok_ ? sb_ + static_cast<size_t>(eo_) * 4 : g_zero_page;                                  \
      ainc[i] = ok_ ? 64u : 0u;                                                                              \
    }                                                                                                        \
    if (MODE == 2) nsp = p.in_mr + (static_cast<long long>(img_) * sg_.cstride + sg_.coff + cl_) * 2;        \
  }
#define RS_A_ADVANCE()                                                                                       \
  {                                                                                                          \
    if (a_next + 1 < NHB) {                                                                                  \
      ++a_next;                                                                                              \
      if (a_next == a_seg_end) {                                                                             \
        RS_A_REBUILD()                                                                                       \
      } else {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < (MODE == 0 ? PA : NQ); ++i) asp[i] += ainc[i];                 \
        if (MODE == 2) nsp += 32;                                                                            \
      }                                                                                                      \
    }                                                                                                        \
  }
  // MODE 0: request the next half block into activation slot SLOT_ (PA DMA pieces per wave)
#define RS_ISSUE_A(SLOT_)                                                                                    \
  {                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                         \
      const int q_ = wave + NW * i < NPIECE ? wave + NW * i : NPIECE - 1;                                    \
      glds16(asp[i], __builtin_amdgcn_readfirstlane(lds0 + (SLOT_) * ASLOT + q_ * 1024));                    \
    }                                                                                                        \
    RS_A_ADVANCE()                                                                                           \
  }
  // MODE 1 / 2: request the next half block into registers ...
#define RS_LOAD_A()                                                                                          \
  {                                                                                                          \
    if (MODE == 2) {                                                                                         \
      nrm01 = *reinterpret_cast<const float4*>(nsp);                                                         \
      nrm23 = *reinterpret_cast<const float4*>(nsp + 4);                                                     \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NQ; ++i) areg[i] = *reinterpret_cast<const float4*>(asp[i]);       \
    RS_A_ADVANCE()                                                                                           \
  }
  // ... and split + store it into slot SLOT_ (rows outside the image: zeros, AFTER the normalisation)
#define RS_STORE_A(SLOT_)                                                                                    \
  {                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NQ; ++i) {                                                         \
      const int idx_ = tid + NT_ * i, row_ = idx_ >> 2, quad_ = idx_ & 3;                                    \
      if (row_ < ARV) {                                                                                      \
        float4 xv_ = areg[i];                                                                                \
        if (MODE == 2) xv_ = make_float4(fmaxf((xv_.x - nrm01.x) * nrm01.y, 0.f), fmaxf((xv_.y - nrm01.z) * nrm01.w, 0.f), \
                                         fmaxf((xv_.z - nrm23.x) * nrm23.y, 0.f), fmaxf((xv_.w - nrm23.z) * nrm23.w, 0.f)); \
        const bool in_ = apix[i] >= 0;                                                                       \
        if (sat_here) sat_n += (in_ && rp::quad_saturates(xv_, p.a_scale)) ? 1 : 0;                          \
        h4 hi_, lo_;                                                                                         \
        split4(in_ ? xv_ : make_float4(0.f, 0.f, 0.f, 0.f), p.a_scale, hi_, lo_);                            \
        const int ad_ = (SLOT_) * ASLOT + row_ * 32 + (((quad_ >> 1) ^ RS_SWZ(row_)) << 4) + (quad_ & 1) * 8; \
        *reinterpret_cast<h4*>(lds + ad_) = hi_;                                                             \
        *reinterpret_cast<h4*>(lds + ad_ + PLANE) = lo_;                                                     \
      }                                                                                                      \
    }                                                                                                        \
  }
  // weight records are requested in step order from a scalar pointer that advances by one record row per step; past the last
  // step it stays (the surplus requests re-read the last record into slots nobody reads any more: the request counts are the
  // same in every step)
  const unsigned char* wptr;
  const size_t wstep = static_cast<size_t>(ntiles32) * 2048;
  int b_left;                                // records behind the one wptr stands at
#define RS_ISSUE_B(SL_)                                                                                      \
  {                                                                                                          \
    glds_rec<NI>(wptr, lane16, __builtin_amdgcn_readfirstlane(sB0 + (SL_) * BREC));                          \
    if (b_left > 0) { wptr += wstep; --b_left; }                                                             \
  }
  h8 fa_h[2][SMI], fa_l[2][SMI], fb_h[2][NI], fb_l[2][NI];
  constexpr int NRD = P1 ? NI + SMI : 2 * NI + 2 * SMI;      // fragment reads per step
  constexpr int NMM = (P1 ? 1 : 3) * SMI * NI;               // MFMAs per step
  // fragment K_ (weights hi x NI, activations lo x 5, weights lo x NI, activations hi x 5 -- the order the MFMAs want them) of
  // tap TN_ of activation slot AS_ / weight slot BS_ -> register set SET_
#define RS_READ1(K_, SET_, TN_, AS_, BS_)                                                                    \
  {                                                                                                          \
    if ((K_) < NI) fb_h[SET_][(K_)] = *reinterpret_cast<const h8*>(sB + (BS_) * BREC + (K_) * 2048 + boff);  \
    else if (P1) {                 /* (single product: weights hi x NI, activations hi x SMI) */                 \
      const int mi_ = (K_) - NI;                                                                             \
      const int tn9_ = S2 ? ((TN_) >> 1) * 3 + ((TN_) & 1) : (TN_);     /* 2x2 taps (dy, dx) in {0, 1}^2 of the 3x3 halo geometry */ \
      const int dyq_ = tn9_ / 3;                                                                             \
      const int ad_ = SPATIAL ? aad[mi_][dyq_ == 1 ? 1 : 0] + (dyq_ == 2 ? 2 * SHW * 32 : 0) + (tn9_ - 3 * dyq_) * 32 \
                              : aad[mi_][SPATIAL ? 0 : (TN_)];                                               \
      fa_h[SET_][mi_] = *reinterpret_cast<const h8*>(lds + (AS_) * ASLOT + ad_);                             \
    }                                                                                                        \
    else if ((K_) >= NI + SMI && (K_) < 2 * NI + SMI)                                                        \
      fb_l[SET_][(K_) - NI - SMI] = *reinterpret_cast<const h8*>(sB + (BS_) * BREC + ((K_) - NI - SMI) * 2048 + 1024 + boff); \
    else {                                                                                                   \
      const bool lo_ = (K_) < NI + SMI;                                                                      \
      const int mi_ = lo_ ? (K_) - NI : (K_) - 2 * NI - SMI;                                                 \
      const int tn9_ = S2 ? ((TN_) >> 1) * 3 + ((TN_) & 1) : (TN_);                                          \
      const int dyq_ = tn9_ / 3;                                                                             \
      const int ad_ = SPATIAL ? aad[mi_][dyq_ == 1 ? 1 : 0] + (dyq_ == 2 ? 2 * SHW * 32 : 0) + (tn9_ - 3 * dyq_) * 32 \
                              : aad[mi_][SPATIAL ? 0 : (TN_)];                                               \
      if (lo_) fa_l[SET_][mi_] = *reinterpret_cast<const h8*>(lds + (AS_) * ASLOT + PLANE + ad_);            \
      else fa_h[SET_][mi_] = *reinterpret_cast<const h8*>(lds + (AS_) * ASLOT + ad_);                        \
    }                                                                                                        \
  }
  // MFMA K_ of a step in term-major order (consecutive MFMAs go to different accumulator tiles): lo x hi, hi x lo, hi x hi
#define RS_MMA1(K_, SET_)                                                                                    \
  {                                                                                                          \
    const int term_ = P1 ? 2 : (K_) / (SMI * NI), idx_ = (K_) % (SMI * NI), mi_ = idx_ % SMI, ni_ = idx_ / SMI; \
    if (term_ == 0) acc[mi_][ni_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_l[SET_][mi_], fb_h[SET_][ni_], acc[mi_][ni_], 0, 0, 0);       \
    else if (term_ == 1) acc[mi_][ni_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[SET_][mi_], fb_l[SET_][ni_], acc[mi_][ni_], 0, 0, 0);  \
    else acc[mi_][ni_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[SET_][mi_], fb_h[SET_][ni_], acc[mi_][ni_], 0, 0, 0);  \
  }

  // ---- prologue: half blocks 0 and 1, weight records 0 .. NBST-2 ----
  // fp32-source forms: both half blocks are requested before the first is split (one memory latency instead of two: r04 timeline) --
  // half block 0 parked in areg0 / n01 / n23, half block 1 in areg / nrm.  RS_PRO_LOADS runs in front of the prologue's weight requests
  // for a workgroup's first tile and IN FRONT OF THE PREVIOUS TILE'S EPILOGUE for every later tile of a persistent launch (r06).
  float4 areg0[MODE == 0 ? 1 : NQ];
  float4 n01 = nrm01, n23 = nrm23;
#define RS_PRO_LOADS()                                                                                       \
  {                                                                                                          \
    RS_LOAD_A()                                                                                              \
    n01 = nrm01; n23 = nrm23;                                                                                \
    _Pragma("unroll") for (int i = 0; i < NQ; ++i) areg0[i] = areg[i];                                       \
    RS_LOAD_A()                                                                                              \
  }
  bool first_tile = true;
  for (;;) {
#pragma unroll
  for (int i = 0; i < SMI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  wptr = reinterpret_cast<const unsigned char*>(p.wpk_strip) + static_cast<size_t>(ct32) * 2048;
  b_left = S - 1;
  e_mt_i = mt_i; e_img = img_; e_py0 = py0_; e_px0 = px0_; e_m0 = m0; e_mend = mend; e_ct32 = ct32;
  RS_CLK(5)
  if (!first_tile) wg_barrier();     // every wave has left the last tile's main loop and epilogue staging: the activation slots may be rewritten
  if constexpr (MODE != 0) {       // the zero row of both planes of both slots (the DMA path refills it with every half block; an epilogue
    if (tid < 8) *reinterpret_cast<uint4*>(lds + (tid >> 2) * ASLOT + ((tid >> 1) & 1) * PLANE + ZROW + (tid & 1) * 16) = make_uint4(0u, 0u, 0u, 0u);      // staged over the slots may have overwritten it)
  }
  a_next = 0;
  RS_A_REBUILD()
  if constexpr (MODE == 0) {
    RS_ISSUE_A(0)
    RS_ISSUE_A(1)
  }
#pragma unroll
  for (int i = 0; i < NBST - 1; ++i) RS_ISSUE_B(i)
  if constexpr (MODE != 0) {
    RS_PRO_LOADS()
    {
      float4 areg1[NQ];
      const float4 m01 = nrm01, m23 = nrm23;
#pragma unroll
      for (int i = 0; i < NQ; ++i) { areg1[i] = areg[i]; areg[i] = areg0[i]; }
      nrm01 = n01; nrm23 = n23;
      RS_STORE_A(0)
#pragma unroll
      for (int i = 0; i < NQ; ++i) areg[i] = areg1[i];
      nrm01 = m01; nrm23 = m23;
      RS_STORE_A(1)
    }
  }
  RS_CLK(6)
  wait_vm<0>();
  wg_barrier();
  RS_CLK(1)
#pragma unroll
  for (int k = 0; k < NRD; ++k) RS_READ1(k, 0, 0, 0, 0)

  // ---- main loop: pairs of half blocks (activation slots 0, 1), taps unrolled.  Step s = tap T_ of half block hb = hbp + H_:
  //   (a) wait for weight record s+1: the NBST-3 newest records stay in flight -- and the activation half block requested at
  //       the last boundary while it sits between them in the queue (the first NBST-3 taps of a half block)
  //   (b) last tap: barrier -- every wave's share of half block hb+1 has landed (requested a half block ago) and nobody reads
  //       slot H_ any more (the fragments of this step are in registers) -> request half block hb+2 into it
  //   (c) request weight record s+NBST-1
  //   (d) 15 MFMAs of step s, the 12 fragment reads of step s+1 between them
#define RS_STEP(H_, T_)                                                                                      \
  {                                                                                                          \
    constexpr int q_ = (H_) * TT + (T_);                       /* step inside the pair: set q_ & 1, weight slot q_ % NBST */ \
    constexpr int set_ = q_ & 1, sl_ = q_ % NBST;                                                            \
    constexpr bool last_ = (T_) == TT - 1;                                                                   \
    constexpr int tn_ = last_ ? 0 : (T_) + 1, asn_ = last_ ? 1 - (H_) : (H_);                                \
    /* (3-slot ring: the record of step s+1 is the newest request but for the activation half block that follows it in the queue) */ \
    if (!(RS_ABL & 64)) wait_vm<RING3 ? ((T_) == 0 ? PAW : 0) : 2 * NI * (NBST - 3) + ((T_) <= NBST - 4 ? PAW : 0)>();  \
    if constexpr (MODE != 0 && (T_) == NBST - 2) { RS_STORE_A(1 - (H_)) }    /* (requested at the last boundary; landed: (a)) */ \
    if constexpr (last_ && !(RS_VAR & 8)) {                                                                  \
      wait_lds();                                                                                            \
      if (!(RS_ABL & 32)) wg_barrier();                                                                      \
      if (RS_VAR & 16) {          /* variant: time-sliced issue priority (the two waves of a SIMD take turns of RS_SLICE_BITS clock bits) */ \
        unsigned long long t_;                                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
        if (((static_cast<unsigned>(t_) >> RS_SLICE_BITS) ^ slotpar_) & 1u) __builtin_amdgcn_s_setprio(2);   \
        else __builtin_amdgcn_s_setprio(0);                                                                  \
      }                                                                                                      \
      if constexpr (RING3) { RS_ISSUE_B((sl_ + NBST - 1) % NBST) }     /* (3-slot ring: the weight record FIRST, so that the next step can wait for it and leave the half block in flight) */ \
      if (!(RS_ABL & 2)) { if constexpr (MODE == 0) { RS_ISSUE_A(H_) } else { RS_LOAD_A() } }   /* (half block hb+2; past the end: the last one again) */ \
    }                                                                                                        \
    if (!(RS_ABL & 1) && !(RS_VAR & 8) && !(RING3 && last_)) RS_ISSUE_B((sl_ + NBST - 1) % NBST)             \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (RS_VAR & 2) {              /* variant: all fragment reads of the next step in front of the MFMAs */    \
      _Pragma("unroll") for (int k = 0; k < NRD; ++k) RS_READ1(k, 1 - set_, tn_, asn_, (sl_ + 1) % NBST)     \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    if (RS_VAR & 1) __builtin_amdgcn_s_setprio(1);                                                           \
    /* stride-2 form: a (plane, tap) slot that carries no weight (its packed record is zero) runs no MFMAs -- one wave-uniform branch \
       around the step's MFMA block; the next step's fragment reads then stand in front of it instead of between the MFMAs */ \
    if (S2 && RS_S2_SKIP) {                                                                                  \
      _Pragma("unroll") for (int k = 0; k < NRD; ++k) RS_READ1(k, 1 - set_, tn_, asn_, (sl_ + 1) % NBST)     \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    if (!(S2 && RS_S2_SKIP) || ((s2_taps_ >> (T_)) & 1u)) {                                                  \
    _Pragma("unroll") for (int k = 0; k < NMM; ++k) {                                                        \
      if (!(RS_ABL & 8)) RS_MMA1(k, set_)                                                                    \
      if (RS_VAR & 8) {            /* variant: the step's bookkeeping between its last MFMAs instead of in front of the first */ \
        if (last_ && k == 2 * SMI * NI) { if (!(RS_ABL & 32)) wg_barrier(); }       /* (every fragment of this step has fed an MFMA: the reads of slot H_ are complete) */ \
        if (last_ && k == 2 * SMI * NI + 1) { if (!(RS_ABL & 2)) { if constexpr (MODE == 0) { RS_ISSUE_A(H_) } else { RS_LOAD_A() } } } \
        if (k == 2 * SMI * NI + 2) { if (!(RS_ABL & 1)) RS_ISSUE_B((sl_ + NBST - 1) % NBST) }                \
      }                                                                                                      \
      if (!(RS_VAR & 2) && !(S2 && RS_S2_SKIP)) {                                                           \
        constexpr int kr0_ = (RS_VAR & 4) ? 2 : 0;        /* variant: the first reads behind the third MFMA */ \
        const bool isb_ = (k - kr0_ < NI) || (k - kr0_ >= NI + SMI && k - kr0_ < 2 * NI + SMI);              \
        if (k >= kr0_ && k - kr0_ < NRD && !(RS_ABL & 4) && !((RS_ABL & 128) && !isb_) && !((RS_ABL & 256) && isb_)) \
          RS_READ1(k - kr0_, 1 - set_, tn_, asn_, (sl_ + 1) % NBST)                                          \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    }                                                                                                        \
    if constexpr (NRD > NMM && !(RS_VAR & 2) && !(S2 && RS_S2_SKIP)) {      /* (32-row strips: four fragment reads for three MFMAs) */   \
      _Pragma("unroll") for (int k = NMM; k < NRD; ++k) { if (!(RS_ABL & 4)) RS_READ1(k, 1 - set_, tn_, asn_, (sl_ + 1) % NBST) } \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    if (RS_VAR & 1) __builtin_amdgcn_s_setprio(0);                                                           \
  }
  const unsigned slotpar_ = __builtin_amdgcn_s_getreg(6148) & 1u;       // HW_ID.wave_id (bits 3:0): the wave slot on this SIMD
  // stride-2 form: which of the four 2x2 taps of the current parity plane carry a weight (bit t = 2 ty + tx).  Plane (py, px): row
  // tap ty = 0 (offset -1) only for py = 1, column tap tx = 0 only for px = 1; the 1x1 form (p.gkw == 1): the centre tap of plane (0, 0).
  // Both half blocks of a pair lie in one 32-channel block, i.e. in one plane.
  unsigned s2_taps_ = 15u;
  const int s2_nb_ = S2 ? (p.cb1 < p.ncb ? p.cb1 : p.ncb) : 1;          // 32-channel blocks per plane
  for (int hbp = 0; hbp < NHB; hbp += 2) {
    if constexpr (S2) {
      const int plane_ = (hbp >> 1) / s2_nb_;
      const unsigned rows_ = (plane_ >> 1) ? 3u : 2u, cols_ = (plane_ & 1) ? 3u : 2u;     // valid ty / tx bits
      s2_taps_ = p.gkw == 1 ? 8u : ((((rows_ & 1u) && (cols_ & 1u)) ? 1u : 0u) | ((rows_ & 1u) ? 2u : 0u) | ((cols_ & 1u) ? 4u : 0u) | 8u);
    }
    if constexpr (TT == 5) {
      RS_STEP(0, 0) RS_STEP(0, 1) RS_STEP(0, 2) RS_STEP(0, 3) RS_STEP(0, 4)
      RS_STEP(1, 0) RS_STEP(1, 1) RS_STEP(1, 2) RS_STEP(1, 3) RS_STEP(1, 4)
    } else if constexpr (TT == 4) {
      RS_STEP(0, 0) RS_STEP(0, 1) RS_STEP(0, 2) RS_STEP(0, 3)
      RS_STEP(1, 0) RS_STEP(1, 1) RS_STEP(1, 2) RS_STEP(1, 3)
    } else {
      RS_STEP(0, 0) RS_STEP(0, 1) RS_STEP(0, 2) RS_STEP(0, 3) RS_STEP(0, 4) RS_STEP(0, 5) RS_STEP(0, 6) RS_STEP(0, 7) RS_STEP(0, 8)
      RS_STEP(1, 0) RS_STEP(1, 1) RS_STEP(1, 2) RS_STEP(1, 3) RS_STEP(1, 4) RS_STEP(1, 5) RS_STEP(1, 6) RS_STEP(1, 7) RS_STEP(1, 8)
    }
  }
  wait_vm<0>();        // (the surplus requests of the last steps land in slots the epilogue is about to reuse)
  wait_lds();
  RS_CLK(2)
  // ------------------------------------------- epilogue -------------------------------------------
  // accumulators -> wave-private LDS tile (the wave's own weight ring: every request into it has been waited for, all its
  // fragment reads are done) -> 16-byte row-contiguous stores; the operands of all four row groups of a 32-row block (additive
  // map, h, z) are requested before the block goes through LDS (conv_igemm.hip has the history of this order).
  constexpr int ES = 32 * NI + 4, F4 = 8 * NI, KG = 4 * NI;       // row stride of the staging tile (floats), float4 per row, row groups per block
  // (persistent form: the epilogue's lane-derived addresses are loop invariants; hoisted out of the tile loop they live through the main
  //  loop and cost 70-400 spilled registers -- the thread index goes through an opaque move so that they are formed here, per tile)
  int tid_e_ = tid;
  if constexpr (PERSIST) asm volatile("" : "+v"(tid_e_));
  bool more_ = false;
  {
  const int tid = tid_e_, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  (void)l31; (void)lh;
  float* S_ = reinterpret_cast<float*>(sB);              // 32 x ES floats = 4.5 / 8.5 KB <= the ring's 10 / 24 KB
  constexpr bool EPI_IN_ASLOTS = NBST * BREC < 2 * 32 * ES * 4;      // a 3- / 4-slot ring (6 / 8 KB) does not hold the fast epilogue's two staging
  if constexpr (EPI_IN_ASLOTS) {                                      // tiles (9 KB per wave): they are laid over the whole LDS block from its start
    wg_barrier();                                                     // (activation slots, then rings) once EVERY wave has left the main loop
    S_ = reinterpret_cast<float*>(lds + wave * (2 * 32 * ES * 4));
    static_assert(!EPI_IN_ASLOTS || NW * 2 * 32 * (32 * NI + 4) * 4 <= LDSB, "epilogue staging inside the workgroup's LDS block");
  }
  const int colw = e_ct32 * 32;
  const int colq = colw + (lane % F4) * 4;
  const bool colok = colq < p.Cout;
  const int colc = colok ? colq : 0;
  const int nv = colok ? (p.Cout - colq < 4 ? p.Cout - colq : 4) : 0;
  float bq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bq[e] = p.bias[colc + (e < nv ? e : 0)];
  const int c2 = colc >= p.gru_c ? colc - p.gru_c : 0;
  double ts0 = 0., ts1 = 0., ts2 = 0., ts3 = 0., tq0 = 0., tq1 = 0., tq2 = 0., tq3 = 0.;
  // ---- persistent launches (r06): the NEXT tile of this workgroup.  Its geometry and source pointers are formed HERE, in front of this
  // tile's epilogue, and -- fp32-source forms -- the 128-byte lines of its first two half blocks are PULLED TOWARDS THE CU by LDS-DMA
  // requests whose LDS destination is a 1-KB sink nobody reads (a free part of the wave's own weight ring): no registers, no wait --
  // the 3-6 us of first-byte latency that every tile's prologue used to sit out (profiles/r04_strip_timeline.txt: prologue 4.7-6.7 us
  // of a 30-37 us workgroup in the encoder's layers) run under the epilogue's staging and stores, and the prologue's real loads then
  // hit L2.  (Loading the half blocks into registers across the epilogue was built first: 165-405 spilled registers.)  The bias
  // values are passed through a v_mov first so that the compiler's wait for THEIR loads stands in front of the DMA requests: vmcnt is
  // in order, and a wait behind them would sit out the very latency this hides.
  t_idx += t_step;
  more_ = PERSIST && t_idx < t_cnt;                  // (PERSIST is a template parameter: the one-tile forms keep their straight-line code)
  if (more_ && MODE != 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("v_mov_b32 %0, %1" : "=v"(bq[e]) : "v"(bq[e]));
    // (SHADOW copies of the tile variables: the macros fill these, the requests go out, and they die here -- kept alive across the
    //  epilogue they cost 100-400 spilled registers; the geometry is formed again for real behind the epilogue)
    int nt_i, mt_i, img_, py0_, px0_, m0, mend, ct32;
    int apix[PA > NQ ? PA : NQ];
    int asrc[MODE == 0 ? PA : 1];
    int a_next = 0, a_seg_end = 0;
    const unsigned char* asp[MODE == 0 ? PA : NQ];
    unsigned ainc[MODE == 0 ? PA : NQ];
    const float* nsp = p.in_mr;
    RS_TILE_GEOM(t_base + t_idx)
    RS_TILE_ROWS()
    RS_A_REBUILD()
    (void)nt_i; (void)mt_i; (void)mend; (void)ct32; (void)asrc; (void)a_seg_end; (void)ainc; (void)nsp;
    constexpr int SINK = EPI_IN_ASLOTS ? 0 : 2 * 32 * ES * 4;       // the ring is free (staging in the activation slots) / its tail behind the two staging tiles
    static_assert(!PERSIST || MODE == 0 || SINK + 1024 <= NBST * BREC, "a 1-KB sink inside the wave's weight ring");
#pragma unroll
    for (int i = 0; i < NQ; ++i) glds16(asp[i], __builtin_amdgcn_readfirstlane(sB0 + SINK));
  }
  // ---- fast path (strip_epilogue_fast above): a wave whose 160 x 32 tile lies inside the problem, with one of the option sets the
  // engines use -- every strip of the update block and the encoder at the headline shapes
  int ev_ = -1;
  if (NI == 1 && p.off32 && colw + 32 <= p.Cout && (SPATIAL ? (e_py0 + SPH <= p.U && e_px0 + SPW <= p.V) : (e_m0 + SM <= e_mend && p.V >= 8))) {
    if (p.epi <= 1 && !p.addm && !p.dsth) ev_ = p.dst_hl ? (p.tstats ? -1 : 2) : (p.tstats ? 1 : 0);
    else if (p.epi == 2 && p.addm && !p.dst_hl && p.dst2_hl && !p.dsth && (p.gru_c & 31) == 0 && !p.tstats) ev_ = 3;
    else if (p.epi == 3 && p.addm && !p.dst_hl && p.dsth && !p.tstats) ev_ = 4;
  }
  if (NI == 1 && ev_ >= 0) {
    int pixb, lv_ = 0, lu_ = 0;
    if constexpr (SPATIAL) {
      pixb = e_img * UV + e_py0 * p.V + e_px0 + (lane >> 3);
    } else {
      const int m = e_m0 + (lane >> 3);
      const int q = m / p.V;
      lv_ = m - q * p.V;
      const int b = q / p.U;
      lu_ = q - b * p.U;
      pixb = b * UV + lu_ * p.su + lv_ * p.sv;
    }
    const StripEpi pe_{p.dst, p.dst_cs, p.dst_co, p.dst2, p.dst2_cs, p.dst2_co, p.dsth, p.dsth_cs, p.dsth_co, p.addm, p.addm_cs, p.addm_co,
                       p.aux0, p.aux0_cs, p.aux0_co, p.aux1, p.aux1_cs, p.aux1_co, p.out_scale, p.a_scale, p.epi, p.gru_c, p.U, p.V, p.su, p.sv};
    RS_CLK(4)
#define RS_FAST(EV_) strip_epilogue_fast<EV_, SPATIAL, SMI, NI>(pe_, acc, S_, lane, colq, c2, bq, pixb, lv_, lu_, UV, ts0, ts1, ts2, ts3, tq0, tq1, tq2, tq3, sat_n);
    switch (ev_) {
      case 0: RS_FAST(0) break;
      case 1: RS_FAST(1) break;
      case 2: RS_FAST(2) break;
      case 3: RS_FAST(3) break;
      default: RS_FAST(4) break;
    }
#undef RS_FAST
    RS_CLK(7)
  } else {
  // ---- general form: any tile (ragged rows / columns), any option.  (The row-tile loop stays ROLLED: five copies of this body pass
  //  the compiler's unroll budget at NI = 2, a partly unrolled loop indexes the accumulators dynamically, and they then live in
  //  scratch memory for the whole main loop.  The tile of the iteration is selected by a uniform switch over constant indices.)
#pragma unroll 1
  for (int mi = 0; mi < SMI; ++mi) {
    f32x16 at[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      switch (mi) {
        case 0: at[ni] = acc[0][ni]; break;
        case 1: at[ni] = acc[SMI > 1 ? 1 : 0][ni]; break;
        case 2: at[ni] = acc[SMI > 2 ? 2 : 0][ni]; break;
        case 3: at[ni] = acc[SMI > 3 ? 3 : 0][ni]; break;
        default: at[ni] = acc[SMI > 4 ? 4 : 0][ni]; break;
      }
    }
    long long pixk[KG];
    float4 am[KG], hv[KG], zv[KG];
    unsigned rowok = 0u;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int rl = (lane + 64 * k) / F4;
      const int r = mi * 32 + rl;
      long long pix;
      if (SPATIAL) {
        const int y = e_py0 + (r >> 4), x = e_px0 + (r & 15);
        rowok |= ((y < p.U && x < p.V) ? 1u : 0u) << k;
        pix = static_cast<long long>(e_img) * UV + (y < p.U ? y : p.U - 1) * p.V + (x < p.V ? x : p.V - 1);
      } else {
        const int m = e_m0 + r;
        const int mc = m < e_mend ? m : e_mend - 1;
        rowok |= (m < e_mend ? 1u : 0u) << k;
        pix = mc;
        if (p.sv != 1) {
          const int q = mc / p.V, v = mc - q * p.V;
          const int b = q / p.U, u = q - b * p.U;
          pix = static_cast<long long>(b) * UV + u * p.su + v * p.sv;
        }
      }
      pixk[k] = pix;
    }
    if (p.addm) {
#pragma unroll
      for (int k = 0; k < KG; ++k) am[k] = *reinterpret_cast<const float4*>(p.addm + pixk[k] * p.addm_cs + p.addm_co + colc);
    }
    if (p.epi == 2) {
#pragma unroll
      for (int k = 0; k < KG; ++k) hv[k] = *reinterpret_cast<const float4*>(p.aux0 + pixk[k] * p.aux0_cs + p.aux0_co + c2);
    } else if (p.epi == 3) {
#pragma unroll
      for (int k = 0; k < KG; ++k) {
        zv[k] = *reinterpret_cast<const float4*>(p.aux1 + pixk[k] * p.aux1_cs + p.aux1_co + colc);
        hv[k] = *reinterpret_cast<const float4*>(p.aux0 + pixk[k] * p.aux0_cs + p.aux0_co + colc);
      }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) S_[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + ni * 32 + l31] = at[ni][r] * p.out_scale;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int idx = lane + 64 * k;
      const int rl = idx / F4, c = (idx % F4) * 4;
      if (!((rowok >> k) & 1u) || !colok) continue;
      const long long pix = pixk[k];
      const float4 a4 = *reinterpret_cast<const float4*>(S_ + rl * ES + c);
      float y[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] += (e < nv) ? bq[e] : 0.f;
      if (p.addm) { y[0] += am[k].x; y[1] += am[k].y; y[2] += am[k].z; y[3] += am[k].w; }
      float* drow = p.dst + pix * p.dst_cs;
      int dch = p.dst_co + colq;
      int dhl = p.dst_hl;
      if (p.tstats) {
        const double y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
        if (nv > 0) { ts0 += y0; tq0 += y0 * y0; }
        if (nv > 1) { ts1 += y1; tq1 += y1 * y1; }
        if (nv > 2) { ts2 += y2; tq2 += y2 * y2; }
        if (nv > 3) { ts3 += y3; tq3 += y3 * y3; }
      }
      if (p.epi == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
      } else if (p.epi == 2) {
        if (colq < p.gru_c) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = sigmoidf_(y[e]);                             // z
        } else {
          y[0] = sigmoidf_(y[0]) * hv[k].x; y[1] = sigmoidf_(y[1]) * hv[k].y;             // r * h
          y[2] = sigmoidf_(y[2]) * hv[k].z; y[3] = sigmoidf_(y[3]) * hv[k].w;
          drow = p.dst2 + pix * p.dst2_cs;
          dch = p.dst2_co + c2;
          dhl = p.dst2_hl;
        }
      } else if (p.epi == 3) {
        const float4 z = zv[k], h4_ = hv[k];
        y[0] = (1.f - z.x) * h4_.x + z.x * tanhf(y[0]); y[1] = (1.f - z.y) * h4_.y + z.y * tanhf(y[1]);   // h' = (1-z)h + z q
        y[2] = (1.f - z.z) * h4_.z + z.z * tanhf(y[2]); y[3] = (1.f - z.w) * h4_.w + z.w * tanhf(y[3]);
      }
      if (dhl) {
        store_quad_hl(drow, dch, y[0], y[1], y[2], y[3], nv, p.a_scale, sat_n);
      } else if (nv == 4) {
        *reinterpret_cast<float4*>(drow + dch) = make_float4(y[0], y[1], y[2], y[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (e < nv) drow[dch + e] = y[e];
      }
      if (p.dsth) store_quad_hl(p.dsth + pix * p.dsth_cs, p.dsth_co + colq, y[0], y[1], y[2], y[3], nv, p.a_scale, sat_n);
    }
  }
  }
  if (p.tstats) {
    // a wave owns all 160 rows of its 32 columns: lanes sharing a column quad (same lane % 8) -> lanes 0..7, fixed order; one
    // (sum, sum of squares) pair per tile and column, no atomics, nothing to combine across waves
#pragma unroll
    for (int o = F4; o < 64; o <<= 1) {
      ts0 += rp::shfl_xor_f64(ts0, o); ts1 += rp::shfl_xor_f64(ts1, o); ts2 += rp::shfl_xor_f64(ts2, o); ts3 += rp::shfl_xor_f64(ts3, o);
      tq0 += rp::shfl_xor_f64(tq0, o); tq1 += rp::shfl_xor_f64(tq1, o); tq2 += rp::shfl_xor_f64(tq2, o); tq3 += rp::shfl_xor_f64(tq3, o);
    }
    if (lane < F4) {
      const int col = colw + lane * 4;
      const double ts[4] = {ts0, ts1, ts2, ts3}, tq[4] = {tq0, tq1, tq2, tq3};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < p.Cout) {
          double* o = p.tstats + (static_cast<long long>(e_mt_i) * p.Cout + col + e) * 2;
          o[0] = ts[e];
          o[1] = tq[e];
        }
    }
  }
  RS_CLK(3)
  }      // (epilogue scope)
  first_tile = false;
  if constexpr (!PERSIST) break;
  if (!more_) break;
  RS_TILE_GEOM(t_base + t_idx)
  RS_TILE_ROWS()
  if constexpr (!SPATIAL) { RS_TILE_FRAG_ADDR() }
  sat_here = MODE != 0 && p.sat != nullptr && nt_i == 0;
  }      // tile loop
  if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));
}

// every kernel of one strip height (SMI_ 32-row tiles per wave): nw waves of ni 32-column tiles, 3x3 (spatial) or 1x5 / 5x1, sources
// split (hlin) / fp32 / fp32 with the fused normalisation (norm).  Two tiles per wave exist for 160-row strips only.
template <int SMI_, bool P1_ = false>
int strip_launch_height(const KParams& p, int nw, int ni, bool spatial, bool hlin, bool norm, unsigned nwg, hipStream_t st) {
  const dim3 grid(nwg), block(nw * 64);
  if (p.stride == 2) {          // the 2x2-tap form of a stride-2 3x3 / 1x1 layer: fp32 source through registers, one tile per wave, either strip height
    if constexpr (!P1_) {
      if (ni != 1 || hlin || norm || !spatial) return 1;
      if (nw == 2) hipLaunchKernelGGL((conv_strip_f16x3_kernel<2, 4, 1, 1, SMI_, false>), grid, block, 0, st, p);
      else if (nw == 3) hipLaunchKernelGGL((conv_strip_f16x3_kernel<3, 4, 1, 1, SMI_, false>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((conv_strip_f16x3_kernel<4, 4, 1, 1, SMI_, false>), grid, block, 0, st, p);
      return 0;
    } else {
      return 1;
    }
  }
#define RS_LAUNCH(NW_, NI_)                                                                                               \
  if (spatial) {                                                                                                          \
    if (norm) {                                                                                                           \
      if constexpr (NI_ == 1 && SMI_ == 5 && !P1_) {                                                                      \
        if (p.n_tiles > 0) hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 9, 2, NI_, SMI_, P1_, true>), grid, block, 0, st, p);   \
        else hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 9, 2, NI_, SMI_, P1_>), grid, block, 0, st, p);             \
      } else hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 9, 2, NI_, SMI_, P1_>), grid, block, 0, st, p);             \
    }                                                                                                                     \
    else if (hlin) hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 9, 0, NI_, SMI_, P1_>), grid, block, 0, st, p);            \
    else {                                                                                                                \
      if constexpr (NI_ == 1 && SMI_ == 5 && !P1_) {                                                                      \
        if (p.n_tiles > 0) hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 9, 1, NI_, SMI_, P1_, true>), grid, block, 0, st, p);   \
        else hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 9, 1, NI_, SMI_, P1_>), grid, block, 0, st, p);             \
      } else hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 9, 1, NI_, SMI_, P1_>), grid, block, 0, st, p);             \
    }                                                                                                                     \
  } else {                                                                                                                \
    if (hlin) hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 5, 0, NI_, SMI_, P1_>), grid, block, 0, st, p);                 \
    else hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, 5, 1, NI_, SMI_, P1_>), grid, block, 0, st, p);                      \
  }
  if (ni == 2) {
    if constexpr (SMI_ == 5 && !P1_) {
      if (nw == 4) { RS_LAUNCH(4, 2) } else if (nw == 3) { RS_LAUNCH(3, 2) } else { RS_LAUNCH(2, 2) }
    } else {
      return 1;
    }
  } else {
    if (nw == 2) { RS_LAUNCH(2, 1) } else if (nw == 3) { RS_LAUNCH(3, 1) } else { RS_LAUNCH(4, 1) }
  }
#undef RS_LAUNCH
  return 0;
}

}  // namespace
