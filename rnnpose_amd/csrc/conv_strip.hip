// a4 / f1: the dense stride-1 convolutions (1x5, 5x1, 3x3) as STRIP kernels for CDNA4 -- the second implicit-GEMM family of
// this library (thirdparty/raft/update.py:33-60,79-97,172-188; thirdparty/raft/extractor.py:41-58).  Same arithmetic as
// conv_igemm.hip (fp16x3 split: a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_f16, fp32 accumulation), same
// parameter block and epilogues; a different launch geometry and operand path, built for what round 3's ablations left standing
// (DESIGN.md section 5.1): tile quantisation over 256 CUs, the per-wave weight stream through the vector-memory path, and the
// staging / barrier chain at every 32-channel block.
//
//   * One workgroup = a STRIP of 160 output pixels x 32*NW output channels, NW = 3 or 4 waves; a wave owns ALL 160 rows of its
//     32 columns (five 32x32 accumulator tiles).  160 rows because the launches of this workload then come out even: 60x80
//     feature maps are 30 strips per image (3x3 layers: 10 x 16 pixel patches), 240 strips x column tiles per batch of 8 on
//     256 CUs, two workgroups per CU (<= 62 KB of LDS, <= 256 registers).
//   * Every operand arrives by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction, no registers, no vector ALU):
//       - weights: WAVE-PRIVATE.  A wave only ever multiplies its own 32 columns, so its weight fragments go through its own
//         4-slot ring of 2-KB records (one record = 16 channels x 32 columns x (hi, lo) of one tap), three steps ahead, waited for
//         with a counted s_waitcnt vmcnt -- no workgroup barrier is involved in the weight stream at all;
//       - activations: from SPLIT TENSORS (rnnpose_hip.h: fp16 hi|lo per 8-channel group, written by the producer's epilogue),
//         one HALF block (16 channels = one MFMA K) of the strip with its halo at a time, two slots; the next half block is
//         requested a whole half block (5 or 9 taps) ahead.  ONE barrier per half block, placed in front of its last tap.
//       LDS rows are 64 bytes ([group 0 hi | group 0 lo | group 1 hi | group 1 lo] x 8 fp16); the 16-byte chunk position is
//       XOR-swizzled with (row >> 2) & 3 -- on the SOURCE address of the DMA (the LDS image of a DMA is lane-linear), in the
//       packed weight order, and on every fragment read: ds_read_b128 of 16 different rows hits 16 different bank groups.
//     fp32 sources (and the encoder's fused instance norm + ReLU, which needs the vector ALU anyway) take the same kernel with the
//     activation half block staged through registers instead (MODE 1 / 2).
//   * A step = one tap of one half block: 15 MFMAs per wave (5 row tiles x 3 products), 12 ds_read_b128, 2 DMA requests.  The
//     fragments of step s+1 are read while the MFMAs of step s run (two register sets).
//   * Epilogue: the parameter block's fused forms (bias, ReLU, GRU gates, additive map, split outputs, fp64 tile statistics),
//     through a wave-private LDS tile (the wave's own weight ring) with 16-byte row-contiguous stores.
#include "common.hpp"
#include "f16x3.cuh"
#include "conv_common.cuh"

namespace {

using namespace rpconv;

constexpr int SM = 160, SMI = 5;        // strip rows, 32-row MFMA tiles per wave
constexpr int SHALO = 4;                // linear strips: halo rows on each side (taps up to +-2 along the fast axis; 4 keeps 16-row pieces)
constexpr int SPH = 10, SPW = 16;       // 3x3: patch of 10 x 16 pixels ...
constexpr int SHW = SPW + 2, SHR = (SPH + 2) * SHW;       // ... staged with a one-pixel halo: 12 x 18 = 216 rows
constexpr int NBST = 4;                 // weight ring slots per wave (2 KB each)

__device__ __attribute__((aligned(64))) const unsigned char g_zero_page[64] = {0};

// One LDS-DMA request: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [dst, dst + 1024) (wave-uniform dst).
// Inline asm on purpose: hipcc (ROCm 7.2) models the builtin as a FLAT access to both global memory and LDS, and while one is
// outstanding every wait it inserts becomes lgkmcnt(0) / vmcnt(0) -- the fragment prefetch of the next step would be waited for
// in front of every MFMA group.  The compiler neither counts nor waits for these requests: every wait is an explicit counted
// s_waitcnt vmcnt below.  M0 (the DMA's LDS base) is saved and restored inside the statement.
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

// MODE 0: split-tensor sources by LDS-DMA; 1: fp32 sources through registers; 2: fp32 source + fused instance norm / ReLU (p.in_mr)
template <int NW, bool SPATIAL, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void conv_strip_f16x3_kernel(const KParams p) {
  constexpr int NT_ = NW * 64;
  constexpr int ARV = SPATIAL ? SHR : SM + 2 * SHALO;      // staged rows that carry data (216 / 168)
  constexpr int AR = (ARV + 15) / 16 * 16;                 // rows per slot: whole 16-row DMA pieces (224 / 176)
  constexpr int ASLOT = AR * 64;
  constexpr int NPIECE = AR / 16;
  constexpr int PA = (NPIECE + NW - 1) / NW;               // DMA pieces per wave and half block
  constexpr int NQ = (ARV * 4 + NT_ - 1) / NT_;            // register path: float4 per thread and half block
  constexpr int PAW = MODE == 0 ? PA : (MODE == 1 ? NQ : NQ + 2);      // vector-memory requests of one activation half block per wave
  constexpr int ZOFF = 2 * ASLOT;                          // one all-zero row (taps outside the image line read it)
  constexpr int BOFF = ZOFF + 64;
  constexpr int LDSB = BOFF + NW * NBST * 2048;
  static_assert(PAW + 2 <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDSB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int TT = p.T;

  // ---- tile id: XCD-contiguous chunks, column tiles of one strip next to each other (they stage the same activations) ----
  int bid = static_cast<int>(blockIdx.x);
  {
    const int ntl = static_cast<int>(gridDim.x);
    const int per = ntl >> 3, rem = ntl & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int nt_i = bid % p.n_nt, mt_i = bid / p.n_nt;
  const int UV = p.U * p.V;
  const int Mtot = p.B * UV;
  const int img_ = p.tpi > 0 ? mt_i / p.tpi : 0;
  const int pt_ = p.tpi > 0 ? mt_i - img_ * p.tpi : 0;
  const int py0_ = SPATIAL ? (pt_ / p.sp_tx) * SPH : 0, px0_ = SPATIAL ? (pt_ % p.sp_tx) * SPW : 0;
  const int m0 = p.tpi > 0 ? img_ * UV + pt_ * SM : mt_i * SM;
  const int mend = p.tpi > 0 ? (img_ + 1) * UV : Mtot;
  const int ct32 = nt_i * NW + wave;               // this wave's 32-column tile
  const int ntiles32 = p.Npad >> 5;
  unsigned char* const sB = lds + BOFF + wave * (NBST * 2048);      // this wave's weight ring
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)lds));   // LDS byte address
  const unsigned sB0 = lds0 + BOFF + wave * (NBST * 2048);

  // staged row j of the strip -> pixel index (or -1: outside the image / the problem -> zeros)
#define RS_ROW_PIXEL(OUT_, J_)                                                                               \
  {                                                                                                          \
    const int j_ = (J_);                                                                                     \
    if (SPATIAL) {                                                                                           \
      const int hy_ = j_ / SHW, y_ = py0_ - 1 + hy_, x_ = px0_ - 1 + (j_ - hy_ * SHW);                       \
      const bool ok_ = j_ < SHR && static_cast<unsigned>(y_) < static_cast<unsigned>(p.U) &&                 \
                       static_cast<unsigned>(x_) < static_cast<unsigned>(p.V);                               \
      OUT_ = ok_ ? img_ * UV + y_ * p.V + x_ : -1;                                                           \
    } else {                                                                                                 \
      const int m_ = m0 - SHALO + j_;                                                                        \
      const bool ok_ = j_ < ARV && m_ >= 0 && m_ < Mtot;                                                     \
      const int mm_ = ok_ ? m_ : 0;                                                                          \
      const int q_ = mm_ / p.V, v_ = mm_ - q_ * p.V;                                                         \
      const int b_ = q_ / p.U, u_ = q_ - b_ * p.U;                                                           \
      OUT_ = ok_ ? b_ * UV + u_ * p.su + v_ * p.sv : -1;                                                     \
    }                                                                                                        \
  }
  // MODE 0: this lane's row of each of the wave's PA pieces (piece = 16 rows x 64 bytes; lane -> row lane >> 2, chunk position
  // lane & 3, which holds the logical chunk (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3))
  int apix[PA > NQ ? PA : NQ];
  const int acho = MODE == 0 ? (((lane & 3) ^ ((lane >> 4) & 3)) << 4) : 0;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int pc = wave + NW * i < NPIECE ? wave + NW * i : NPIECE - 1;      // (surplus requests repeat the last piece: identical bytes)
      RS_ROW_PIXEL(apix[i], pc * 16 + (lane >> 2))
    }
  } else {        // register path: quad idx = tid + NT_ * i -> row idx >> 2, channels 4 (idx & 3) .. +3 of the half block
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int idx = tid + NT_ * i;
      RS_ROW_PIXEL(apix[i], idx >> 2)
    }
  }
  // fragment rows of this lane: tile row r = 32 mi + l31
  int rowc[SMI], fv[SMI];
#pragma unroll
  for (int mi = 0; mi < SMI; ++mi) {
    const int r = mi * 32 + l31;
    if (SPATIAL) {
      rowc[mi] = ((r >> 4) + 1) * SHW + (r & 15) + 1;     // halo-tile row of the centre tap
      fv[mi] = 0;
    } else {
      const int m = m0 + r;
      rowc[mi] = r + SHALO;
      fv[mi] = m < mend ? m % p.V : -1000;                // fast-axis coordinate: taps beyond the image line are masked
    }
  }
  const int boff = l31 * 64 + (((lh * 2) ^ ((l31 >> 2) & 3)) << 4);      // this lane's hi chunk inside a weight record (lo: ^ 16)

  f32x16 acc[SMI];
#pragma unroll
  for (int i = 0; i < SMI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int NHB = 2 * p.ncb;                 // half blocks
  const int S = NHB * TT;                    // steps (even)
  float4 areg[MODE == 0 ? 1 : NQ];           // register path: the half block in flight
  float4 nrm01 = make_float4(0.f, 1.f, 0.f, 1.f), nrm23 = nrm01;
  unsigned amask = 0u;
  int sat_n = 0;
  const bool sat_here = MODE != 0 && p.sat != nullptr && nt_i == 0;

  // half block HB_ -> segment, first channel inside it, in range?
#define RS_SEG(HB_)                                                                                          \
  Seg sg_ = p.seg0;                                                                                          \
  int cb0_ = 0;                                                                                              \
  {                                                                                                          \
    const int blk_ = (HB_) >> 1;                                                                             \
    if (blk_ >= p.cb1) { sg_ = p.seg1; cb0_ = p.cb1; }                                                       \
    if (blk_ >= p.cb2) { sg_ = p.seg2; cb0_ = p.cb2; }                                                       \
    if (blk_ >= p.cb3) { sg_ = p.seg3; cb0_ = p.cb3; }                                                       \
  }                                                                                                          \
  const int cl_ = (((HB_) >> 1) - cb0_) * BK + ((HB_) & 1) * 16;
  // MODE 0: request half block HB_ into activation slot SLOT_ (PA DMA pieces per wave)
#define RS_ISSUE_A(HB_, SLOT_)                                                                               \
  {                                                                                                          \
    RS_SEG(HB_)                                                                                              \
    const bool cok_ = cl_ < sg_.ccount;                                                                      \
    const unsigned char* sb_ = reinterpret_cast<const unsigned char*>(sg_.ptr + sg_.coff + cl_) + acho;     \
    _Pragma("unroll") for (int i = 0; i < PA; ++i) {                                                         \
      const int pc_ = wave + NW * i < NPIECE ? wave + NW * i : NPIECE - 1;                                   \
      const bool ok_ = cok_ && apix[i] >= 0;                                                                 \
      const unsigned eo_ = static_cast<unsigned>(apix[i]) * static_cast<unsigned>(sg_.cstride);              \
      const unsigned char* src_ = ok_ ? sb_ + static_cast<size_t>(eo_) * 4 : g_zero_page + ((lane & 3) << 4); \
      glds16(src_, __builtin_amdgcn_readfirstlane(lds0 + (SLOT_) * ASLOT + pc_ * 1024));                     \
    }                                                                                                        \
  }
  // MODE 1 / 2: request half block HB_ into registers ...
#define RS_LOAD_A(HB_)                                                                                       \
  {                                                                                                          \
    RS_SEG(HB_)                                                                                              \
    const int cq_ = cl_ + (tid & 3) * 4;                                                                     \
    const bool cok_ = cq_ < sg_.ccount;                                                                      \
    const float* sb_ = sg_.ptr + sg_.coff + cq_;                                                             \
    if (MODE == 2) {                                                                                         \
      const float* mr_ = p.in_mr + (static_cast<long long>(img_) * sg_.cstride + sg_.coff + (cok_ ? cq_ : 0)) * 2; \
      nrm01 = *reinterpret_cast<const float4*>(mr_);                                                         \
      nrm23 = *reinterpret_cast<const float4*>(mr_ + 4);                                                     \
    }                                                                                                        \
    amask = 0u;                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NQ; ++i) {                                                         \
      const bool ok_ = cok_ && apix[i] >= 0;                                                                 \
      const unsigned eo_ = ok_ ? static_cast<unsigned>(apix[i]) * static_cast<unsigned>(sg_.cstride) : 0u;   \
      areg[i] = *reinterpret_cast<const float4*>((ok_ ? sb_ : sg_.ptr) + eo_);                               \
      amask |= ok_ ? (1u << i) : 0u;                                                                         \
    }                                                                                                        \
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
        const bool in_ = (amask >> i) & 1u;                                                                  \
        if (sat_here) sat_n += (in_ && rp::quad_saturates(xv_, p.a_scale)) ? 1 : 0;                          \
        h4 hi_, lo_;                                                                                         \
        split4(in_ ? xv_ : make_float4(0.f, 0.f, 0.f, 0.f), p.a_scale, hi_, lo_);                            \
        const int ad_ = (SLOT_) * ASLOT + row_ * 64 + ((((quad_ >> 1) * 2) ^ ((row_ >> 2) & 3)) << 4) + (quad_ & 1) * 8; \
        *reinterpret_cast<h4*>(lds + ad_) = hi_;                                                             \
        *reinterpret_cast<h4*>(lds + (ad_ ^ 16)) = lo_;                                                      \
      }                                                                                                      \
    }                                                                                                        \
  }
  // weight record of step S_ -> this wave's ring slot S_ & 3 (2 DMA pieces)
#define RS_ISSUE_B(S_)                                                                                       \
  {                                                                                                          \
    const unsigned char* src_ = reinterpret_cast<const unsigned char*>(p.wpk_strip) +                        \
                                (static_cast<size_t>(S_) * ntiles32 + ct32) * 2048 + lane * 16;             \
    const unsigned dst_ = __builtin_amdgcn_readfirstlane(sB0 + ((S_) & (NBST - 1)) * 2048);                  \
    glds16(src_, dst_);                                                                                      \
    glds16(src_ + 1024, dst_ + 1024);                                                                        \
  }
  h8 fa_h[2][SMI], fa_l[2][SMI], fb_h[2], fb_l[2];
  // fragments of the step (tap TAP_, activation slot ASL_, weight slot BSL_) -> register set SET_
#define RS_READ(SET_, TAP_, ASL_, BSL_)                                                                      \
  {                                                                                                          \
    const int tq_ = ((TAP_) * 11) >> 5;                          /* TAP_ / 3 for TAP_ < 9 */                 \
    const int sh_ = SPATIAL ? (tq_ - 1) * SHW + ((TAP_) - 3 * tq_) - 1 : p.dv0 + (TAP_);                      \
    _Pragma("unroll") for (int mi = 0; mi < SMI; ++mi) {                                                     \
      const int row_ = rowc[mi] + sh_;                                                                       \
      int ad_ = (ASL_) * ASLOT + row_ * 64 + (((lh * 2) ^ ((row_ >> 2) & 3)) << 4);                          \
      if (!SPATIAL) {        /* tap beyond the image line: the all-zero row (mask arithmetic: a select here is compiled into a   \
                                branch per row tile, and a branch between the reads costs the counted lgkmcnt waits) */    \
        const int ok_ = -static_cast<int>(static_cast<unsigned>(fv[mi] + sh_) < static_cast<unsigned>(p.V));  \
        ad_ = (ZOFF + lh * 32) + ((ad_ - (ZOFF + lh * 32)) & ok_);                                             \
      }                                                                                                      \
      fa_h[SET_][mi] = *reinterpret_cast<const h8*>(lds + ad_);                                              \
      fa_l[SET_][mi] = *reinterpret_cast<const h8*>(lds + (ad_ ^ 16));                                       \
    }                                                                                                        \
    fb_h[SET_] = *reinterpret_cast<const h8*>(sB + (BSL_) * 2048 + boff);                                    \
    fb_l[SET_] = *reinterpret_cast<const h8*>(sB + (BSL_) * 2048 + (boff ^ 16));                             \
  }
  // term-major: consecutive MFMAs go to different accumulator tiles
#define RS_MMA(SET_)                                                                                         \
  {                                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < SMI; ++mi)                                                       \
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_l[SET_][mi], fb_h[SET_], acc[mi], 0, 0, 0);     \
    _Pragma("unroll") for (int mi = 0; mi < SMI; ++mi)                                                       \
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[SET_][mi], fb_l[SET_], acc[mi], 0, 0, 0);     \
    _Pragma("unroll") for (int mi = 0; mi < SMI; ++mi)                                                       \
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[SET_][mi], fb_h[SET_], acc[mi], 0, 0, 0);     \
  }

  // ---- prologue: zero row, half blocks 0 and 1, weight records 0..2 ----
  if (tid < 4) *reinterpret_cast<uint4*>(lds + ZOFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  if constexpr (MODE == 0) {
    RS_ISSUE_A(0, 0)
    if (NHB > 1) RS_ISSUE_A(1, 1)
  }
  RS_ISSUE_B(0)
  if (1 < S) RS_ISSUE_B(1)
  if (2 < S) RS_ISSUE_B(2)
  if constexpr (MODE != 0) {
    RS_LOAD_A(0)
    RS_STORE_A(0)
    if (NHB > 1) {
      RS_LOAD_A(1)
      RS_STORE_A(1)
    }
  }
  wait_vm<0>();
  wg_barrier();
  RS_READ(0, 0, 0, 0)

  // ---- main loop.  Step s = tap `tap` of half block `hb`:
  //   (a) wait for weight record s+1 (requested two steps ago; the newest requests stay in flight)
  //   (b) last tap of a half block: barrier -- every wave's share of half block hb+1 has landed (requested a half block ago) and
  //       nobody reads slot hb & 1 any more (the fragments of this step are in registers) -> request half block hb+2 into it
  //   (c) request weight record s+3   (d) read the fragments of step s+1   (e) 15 MFMAs of step s
  int tap = 0, hb = 0;
  bool a_pending = false;          // the previous step requested an activation half block (it sits between two weight records in the queue)
  bool a_in_regs = false;          // register path: a half block is waiting in registers for its LDS store
#define RS_STEP(CUR_, NXT_)                                                                                  \
  {                                                                                                          \
    if (s + 2 < S) {                                                                                         \
      if (a_pending) wait_vm<PAW + 2>(); else wait_vm<2>();                                                  \
    } else {                                                                                                 \
      wait_vm<0>();                                                                                          \
    }                                                                                                        \
    a_pending = false;                                                                                       \
    if constexpr (MODE != 0) {                                                                               \
      if (a_in_regs && tap == 1) {                       /* (half block hb+1, requested at the last boundary) */ \
        RS_STORE_A((hb + 1) & 1)                                                                             \
        a_in_regs = false;                                                                                   \
      }                                                                                                      \
    }                                                                                                        \
    if (tap == TT - 1 && hb + 1 < NHB) {                                                                     \
      wait_lds();                                                                                            \
      wg_barrier();                                                                                          \
      if (hb + 2 < NHB) {                                                                                    \
        if constexpr (MODE == 0) { RS_ISSUE_A(hb + 2, hb & 1) } else { RS_LOAD_A(hb + 2) a_in_regs = true; } \
        a_pending = true;                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
    if (s + 3 < S) RS_ISSUE_B(s + 3)                                                                         \
    {   /* (unconditional: a read under a branch makes the compiler wait lgkmcnt(0) in front of every MFMA group; behind the \
           last step it fetches a stale slot, harmlessly) */                                                 \
      const bool wrap_ = tap == TT - 1;                                                                      \
      const int nt_ = wrap_ ? 0 : tap + 1, nh_ = wrap_ ? hb + 1 : hb;                                        \
      RS_READ(NXT_, nt_, nh_ & 1, (s + 1) & (NBST - 1))                                                      \
    }                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    RS_MMA(CUR_)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    ++s;                                                                                                     \
    if (++tap == TT) { tap = 0; ++hb; }                                                                      \
  }
  for (int s = 0; s < S;) {
    RS_STEP(0, 1)
    RS_STEP(1, 0)
  }
  wait_lds();

  // ------------------------------------------- epilogue -------------------------------------------
  // accumulators -> wave-private LDS tile (the wave's own weight ring: every request into it has been waited for, all its
  // fragment reads are done) -> 16-byte row-contiguous stores; the operands of all four row groups of a 32-row block (additive
  // map, h, z) are requested before the block goes through LDS (conv_igemm.hip has the history of this order).
  constexpr int ES = 36, F4 = 8, KG = 4;
  float* S_ = reinterpret_cast<float*>(sB);              // 32 x 36 floats = 4.5 KB <= 8 KB
  const int colw = ct32 * 32;
  const int colq = colw + (lane % F4) * 4;
  const bool colok = colq < p.Cout;
  const int colc = colok ? colq : 0;
  const int nv = colok ? (p.Cout - colq < 4 ? p.Cout - colq : 4) : 0;
  float bq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bq[e] = p.bias[colc + (e < nv ? e : 0)];
  const int c2 = colc >= p.gru_c ? colc - p.gru_c : 0;
  double ts0 = 0., ts1 = 0., ts2 = 0., ts3 = 0., tq0 = 0., tq1 = 0., tq2 = 0., tq3 = 0.;
#pragma unroll
  for (int mi = 0; mi < SMI; ++mi) {
    long long pixk[KG];
    float4 am[KG], hv[KG], zv[KG];
    unsigned rowok = 0u;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int rl = (lane + 64 * k) / F4;
      const int r = mi * 32 + rl;
      long long pix;
      if (SPATIAL) {
        const int y = py0_ + (r >> 4), x = px0_ + (r & 15);
        rowok |= ((y < p.U && x < p.V) ? 1u : 0u) << k;
        pix = static_cast<long long>(img_) * UV + (y < p.U ? y : p.U - 1) * p.V + (x < p.V ? x : p.V - 1);
      } else {
        const int m = m0 + r;
        const int mc = m < mend ? m : mend - 1;
        rowok |= (m < mend ? 1u : 0u) << k;
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
    for (int r = 0; r < 16; ++r) S_[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + l31] = acc[mi][r] * p.out_scale;
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
  if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));
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
          double* o = p.tstats + (static_cast<long long>(mt_i) * p.Cout + col + e) * 2;
          o[0] = ts[e];
          o[1] = tq[e];
        }
    }
  }
}

// packed order of the strip kernels: [half block hb = 2 cb + kk][tap][32-column tile][column n][chunk position][8] fp16 -- one
// step's record of one column tile is 2 KB contiguous = the LDS image itself (the DMA copies it linearly): column n is a 64-byte
// row [group 0 hi | group 0 lo | group 1 hi | group 1 lo] of channels 16 kk + 8 g + j, chunk position = logical chunk ^ ((n >> 2) & 3)
__global__ void pack_strip_kernel(const float* __restrict__ w, _Float16* __restrict__ pk, const PackParams q, int TT, int spatial) {
  const int nt32 = q.Npad / 32;
  const long long total = static_cast<long long>(q.ncb) * 2 * TT * q.Npad * 32;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j8 = static_cast<int>(i & 7);
  const int cp = static_cast<int>((i >> 3) & 3);
  const int n32 = static_cast<int>((i >> 5) & 31);
  const int ctile = static_cast<int>((i >> 10) % nt32);
  const long long st = (i >> 10) / nt32;
  const int tap = static_cast<int>(st % TT);
  const int hbk = static_cast<int>(st / TT);
  const int blk = hbk >> 1, kk = hbk & 1;
  const int c = cp ^ ((n32 >> 2) & 3);
  const int g = c >> 1, part = c & 1;
  const int k = kk * 16 + g * 8 + j8;
  const int n = ctile * 32 + n32;
  const int s = q.cb_seg[blk];
  const int cl = q.cb_c0[blk] + k;
  float v = 0.f;
  if (n < q.Cout && cl < q.seg_count[s]) {
    const int ci = q.seg_start[s] + cl;
    const int ky = spatial ? tap / 3 : (q.vertical ? tap : 0), kx = spatial ? tap % 3 : (q.vertical ? 0 : tap);
    v = w[((static_cast<long long>(n) * q.Cin + ci) * q.kh + ky) * q.kw + kx] * q.w_scale;
  }
  const _Float16 h = static_cast<_Float16>(v);
  pk[i] = part == 0 ? h : static_cast<_Float16>(v - static_cast<float>(h));
}

bool strip_kernel_shape(int kh, int kw) { return (kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1); }

}  // namespace

namespace rpconv {

int strip_waves(int c_out) {
  if (c_out <= 64) return 0;                          // (two-wave workgroups are not built: such layers stay on the 128-row kernel)
  if (c_out % 96 == 0 && c_out % 128 != 0) return 3;  // 96, 192, 288: whole 96-column tiles
  return 4;
}

int strip_tiles_per_image(int H, int W, int kh, int kw) {
  if (kh == 3 && kw == 3) return rp::cdiv(W, SPW) * rp::cdiv(H, SPH);
  return rp::cdiv(static_cast<long long>(H) * W, SM);
}

bool strip_auto(int H, int W, int kh, int kw, int stride, int c_out) {
  if (stride != 1 || !strip_kernel_shape(kh, kw) || strip_waves(c_out) == 0) return false;
  const int nw = strip_waves(c_out);
  const long long tiles = static_cast<long long>(strip_tiles_per_image(H, W, kh, kw)) * rp::cdiv(c_out, 32 * nw);
  if (tiles < 24) return false;                       // (a few images of this size do not fill the chip with strips)
  if (kh == 3) {                                      // ragged patches: at most 15 % of the rows wasted
    const long long covered = static_cast<long long>(rp::cdiv(W, SPW)) * SPW * rp::cdiv(H, SPH) * SPH;
    if (covered * 100 > static_cast<long long>(H) * W * 115) return false;
  }
  return true;
}

void strip_pack(const float* w, _Float16* pk, const PackParams& q, hipStream_t st) {
  if (!strip_kernel_shape(q.kh, q.kw)) return;        // (the second copy stays unwritten: never read for other shapes)
  const int TT = q.kh * q.kw;
  const long long total = static_cast<long long>(q.ncb) * 2 * TT * q.Npad * 32;
  hipLaunchKernelGGL(pack_strip_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, st, w, pk, q, TT, (q.kh == 3 && q.kw == 3) ? 1 : 0);
}

int strip_launch(KParams& p, int H, int W, int kh, int kw, bool hlin, bool per_image, hipStream_t st) {
  const char* fn = "rnnpose_conv2d_nhwc_f16x3";
  RP_REQUIRE(strip_kernel_shape(kh, kw) && p.stride == 1, fn, "strip kernel: 3x3, 1x5 or 5x1, stride 1");
  const int nw = strip_waves(p.Cout);
  RP_REQUIRE(nw != 0, fn, "strip kernel: c_out must exceed 64");
  const bool spatial = kh == 3;
  const bool norm = p.in_mr != nullptr;
  RP_REQUIRE(!norm || (spatial && !hlin), fn, "strip kernel: the fused normalisation is the 3x3 fp32-source form");
  if (hlin) {        // LDS-DMA: whole 64-byte half blocks
    const Seg* sg[4] = {&p.seg0, &p.seg1, &p.seg2, &p.seg3};
    const int nseg = p.cb3 < p.ncb ? 4 : (p.cb2 < p.ncb ? 3 : (p.cb1 < p.ncb ? 2 : 1));
    for (int s = 0; s < nseg; ++s)
      RP_REQUIRE(sg[s]->ccount % 16 == 0 && sg[s]->coff % 8 == 0 && sg[s]->cstride % 8 == 0, fn,
                 "strip kernel, split-tensor sources: channel counts in multiples of 16 (whole half blocks), offsets / strides of 8");
  }
  p.ksplit = 1;
  p.n_nt = rp::cdiv(p.Cout, 32 * nw);
  RP_REQUIRE(p.n_nt * nw * 32 <= p.Npad, fn, "strip kernel: column tiles exceed the packed width");
  if (spatial) {
    p.T = 9; p.dv0 = 0;
    p.sp_tx = rp::cdiv(W, SPW); p.sp_ty = rp::cdiv(H, SPH);
    p.tpi = p.sp_tx * p.sp_ty;
    p.n_mt = p.B * p.tpi;
  } else if (per_image) {
    p.tpi = rp::cdiv(static_cast<long long>(H) * W, SM);
    p.n_mt = p.B * p.tpi;
  } else {
    p.tpi = 0;
    p.n_mt = rp::cdiv(static_cast<long long>(p.B) * H * W, SM);
  }
  const dim3 grid(static_cast<unsigned>(p.n_mt) * p.n_nt), block(nw * 64);
#define RS_LAUNCH(NW_)                                                                                                    \
  if (spatial) {                                                                                                          \
    if (norm) hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, true, 2>), grid, block, 0, st, p);                         \
    else if (hlin) hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, true, 0>), grid, block, 0, st, p);                    \
    else hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, true, 1>), grid, block, 0, st, p);                              \
  } else {                                                                                                                \
    if (hlin) hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, false, 0>), grid, block, 0, st, p);                        \
    else hipLaunchKernelGGL((conv_strip_f16x3_kernel<NW_, false, 1>), grid, block, 0, st, p);                             \
  }
  if (nw == 3) { RS_LAUNCH(3) } else { RS_LAUNCH(4) }
#undef RS_LAUNCH
  return rp::check_launch(fn);
}

}  // namespace rpconv
