// a4: the update block's dense convolutions as ONE hand-written implicit-GEMM kernel family for CDNA4
// (thirdparty/raft/update.py:6-14,33-60,79-97,164-188).
//
// Numerics -- "fp16x3 split" fp32 emulation on the fp16 matrix cores.  gfx950 has no TF32/xf32, and its exact
// fp32 MFMA runs at 1/16 of the fp16 rate.  Every fp32 operand x is split as x*S = hi + lo (S a power of two keeping
// lo out of the subnormal range; hi = fp16(x*S) rounded to nearest, lo = fp16(x*S - hi)) and
//     a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (3 MFMAs, fp32 accumulation)
// The neglected terms are O(2^-22) relative per product -- fp32 round-off class (measured against an fp64
// reference in tests/test_gpu_conv.py: within 3x of MIOpen's fp32 error), at 16/3 = 5.3x the fp32-MFMA rate.
// Weights are split once (rnnpose_conv_pack_weights_f16x3); activations are split on the fly while staging to LDS.
//
// Structure (per 128x128 or 128x64 output tile, 4 waves; see the COLS4 note at the kernel for the two wave layouts):
//   * activations are NHWC, so the GEMM K axis (channels) is contiguous: one 128-byte row per pixel per 32-ch
//     block.  Up to 4 source tensors are read as a VIRTUAL CONCAT (hidden state | context | motion features
//     never get copied into one buffer).
//   * taps that differ along the tile's fast axis share ONE staged activation tile: the tile is loaded with a
//     halo (BM + 8 rows) and tap dv just reads LDS rows shifted by dv; lanes whose neighbour falls outside the
//     image line read an all-zero LDS row (2x2 layout) or are masked in registers (4-column layout).  A 1x5 conv
//     therefore reads its input once, not 5 times.  (5x1 convs tile the image column-major so the same trick
//     applies; 3x3 = 3 groups of 3 taps; stride 2: every tap is its own group.)
//   * weights: pre-packed in MFMA B-fragment order [group][tap][32-ch block][32-col tile][k half][lane][8] fp16 hi/lo;
//     every wave loads its own fragments straight into registers two taps ahead (no weight LDS, one barrier per
//     32-channel block).  All main-loop loads are unconditional so that the waits stay counted (vmcnt(8..15)).
//   * the activation tile is double-buffered in LDS, rows padded to 80 bytes: ds_read_b128 of 32 consecutive rows is
//     bank-conflict free.
//   * fused epilogues (accumulators -> wave-private LDS tile -> 16-byte row-contiguous stores): bias + {linear,
//     ReLU, GRU z|r gate (sigmoid, r*h), GRU state update (tanh, (1-z)h+zq)}, optional per-tile column statistics
//     for the encoder's instance norm, writing straight into a channel slice of the destination NHWC tensor.
#include "common.hpp"
#include "f16x3.cuh"
#include "conv_common.cuh"

#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, NT = 256;
constexpr int HALO = 4;                 // rows of halo on each side (taps up to +-3 along the fast axis)
constexpr int AROWS = BM + 2 * HALO;    // 136
constexpr int PROWS = AROWS + 1;        // rows per LDS plane: + one all-zero row that masked-out fragment lanes read
constexpr int RS = 40;                  // LDS row stride in halfs: 32 + 8 pad = 80 bytes

using namespace rpconv;

// Every global load of the main loop is UNCONDITIONAL (out-of-range activation rows read element 0 of their tensor and
// are zeroed when they are split into LDS, weight requests past the end re-read the last tile): with loads under
// branches the compiler cannot count how many younger loads are in flight and falls back to s_waitcnt vmcnt(0) -- r01
// in-kernel timestamps showed two full memory-latency stalls per 32-channel block (the weight prefetch drained
// before the activation loads and again before the LDS store).
// fp32 -> fp16 hi + lo with as few vector-ALU instructions as possible (the split runs once per activation element and
// per 32-channel block; for 1x1 convolutions it used more VALU-port time than the MFMAs of the block):
//   hi = x with the mantissa TRUNCATED to fp16's 11 significant bits (one v_and; exactly representable, so the packed
//        convert is exact), lo = fp16(x - hi) (x - hi is exact in fp32; 11 more bits) -> 21-22 significant bits.
// Packed fp32 math (v_pk_mul_f32 / v_pk_add_f32) and v_cvt_pk_f16_f32 halve the rest.  |x*s| > 65504: hi converts to
// inf and is clamped to +-65504 (lo stays tiny): saturation, never NaN from finite inputs.

// NI = MFMA column tiles per wave: output tile = 128 x (64*NI).  NI=2 (128x128) for wide layers, NI=1 (128x64) when
// the 128-wide tiling would leave the 256 CUs short of workgroups (Cout <= 128 at 300 row tiles) or pad Cout.
// STRIDED = stride-2 mode (every tap is its own group, general input addressing); a template flag so that the
// stride-1 instantiations keep their register budget (the 64-wide variant must stay <= 128 VGPRs for 4 waves/SIMD).
// COLS4 = wave layout of the 128-wide tile: false = 2 (rows) x 2 (columns) waves of 64 x 32*NI; true = 4 waves side by
// side, each ALL 128 rows x 32 columns (NI must be 1).  Same MFMA count per wave, but a wave then requests 2 weight
// fragments per k-slab from L2/L1 instead of 4 (the fragment loads were stalling in the vector-memory queue when two
// workgroups share a CU: r01 timestamps) and reads 8 activation fragments from LDS instead of 4.
// TT = taps per group as a compile-time constant (1, 3, 5, 7: the stride-1 convolutions) or 0 (generic stage machine: the
// stride-2 convolutions, where every tap is its own group).  With TT > 0 the main loop is a plain nest
//     for (group, 32-channel block): [request next activation tile] ; T taps unrolled ; [split + store, barrier]
// whose LDS buffer index, weight-register stage and tap shift are compile-time constants, and the weight fragments of
// consecutive stages are consecutive 4-KB records: r02 counters showed ~40 vector + ~50 scalar bookkeeping instructions
// per 12 MFMAs in the generic loop, on a SIMD that can issue ~8 instructions per MFMA slot for all its waves together.
// NORM: the (single) source is the RAW output of a convolution whose instance norm + ReLU (extractor.py:48-58: relu(norm1(conv1 x)))
// is applied here, while the tile is split into LDS -- the normalised tensor never exists in HBM.  Needs per-image tiling
// (one image per workgroup: the statistics of the staged channel quad are two 16-byte loads per 32-channel block).
// HLIN: the sources are SPLIT tensors (fp16 hi|lo per 8-channel group, written by a producer's epilogue: rnnpose_hip.h): the
// staging is a 16-byte copy per thread and row -- no conversion, no range test in the loop.  r02 counters: the on-the-fly
// split was ~3 of the 4.8-6 vector instructions per MFMA of this kernel, on a SIMD whose issue slots (about 8 per MFMA
// period for all its waves) were the limiter; a 3x3 layer with 256 outputs re-split every activation 12 times.
// DEEP (split-tensor sources, 2x2 layout): software pipeline one whole channel block deep instead of occupancy.  r03 ablation
// builds (profiles/r03_conv_ablation.txt): without the weight loads a GRU convolution runs 20 % faster, without the activation
// staging 15 %, without BOTH LDS traffic and loads 30 % -- the waves wait on the (in-order) vector-memory queue, not on issue
// slots.  Here the weight fragments of ALL TT taps of a block sit in registers (slot = tap) and are re-requested for the NEXT
// block right after their MFMAs (a full block = TT taps of latency cover instead of 2 taps), and the activation tile is
// requested TWO blocks ahead into a second register set.  Costs ~60 registers: 2 waves per SIMD instead of 3.
// SPATIAL (r03, 3x3 stride 1): the 128 output rows of a workgroup are an 8 x 16 PATCH of one image instead of 128 consecutive
// pixels, and the staged activation tile is the patch with a one-pixel halo (10 x 18 = 180 rows, zeros outside the image).  All
// NINE taps then read one staged tile at a constant row offset ((dy-1) * 18 + (dx-1)): one staging + one barrier per 32-channel
// block instead of three (the row-major tiling staged a separate tile per kernel row), a third of the activation traffic, and
// no tap masks at all.  r03 ablation: the staging is 20-25 % of a 3x3 layer (17 % of a 1x5 one, which already stages once).
constexpr int SPR = 180, SPW = 18;      // halo tile rows, halo tile width
template <int NI, bool STRIDED, bool COLS4 = false, int TT = 0, bool NORM = false, bool HLIN = false, bool DEEP = false,
          bool SPATIAL = false>
#ifndef RP_COLS4_WAVES
#define RP_COLS4_WAVES 2      // waves per SIMD the 4-column layout is compiled for (3: 168 VGPRs with 60-90 spilled once the statistics are fp64; 2: none; measured equal)
#endif
#ifndef RP_SP_SINGLE
#define RP_SP_SINGLE 0        // 1: patch-tiled 3x3 layers (2x2 waves) keep ONE halo tile in LDS (29 KB, three workgroups per CU, two barriers
#endif                        // per channel block) instead of a double-buffered one (58 KB, two per CU).  r03 same-box: 669.8 vs 672.4
                              // iters/s, single image 5.20 vs 5.08 ms -- the kernel does not respond to occupancy either; off
__global__ __launch_bounds__(NT, ((NI == 2 || DEEP || (SPATIAL && (COLS4 || !RP_SP_SINGLE))) ? 2 : (SPATIAL ? 3 : (COLS4 ? (TT == 3 ? 3 : RP_COLS4_WAVES) : (TT > 0 ? 3 : 2))))) void conv_igemm_f16x3_kernel(const KParams p) {
  static_assert(!DEEP || (HLIN && !COLS4 && TT >= 3), "the deep pipeline is for split sources, the 2x2 wave layout, 3 or 5 taps");
  static_assert(!SPATIAL || (TT == 9 && !STRIDED && !DEEP), "patch tiling is the 3x3 stride-1 form: nine taps on one staged tile");
  constexpr int ARW = SPATIAL ? SPR : AROWS;            // staged rows
  constexpr int PRW = ARW + 1;                          // + the all-zero row
  static_assert(TT == 0 || (!STRIDED && (TT & 1)), "the unrolled loop is for stride 1 and odd tap counts");
  static_assert(!HLIN || (!NORM && !STRIDED && TT > 0), "split-tensor sources: stride 1, no fused normalisation");
  static_assert(!COLS4 || NI == 1, "the 4-column layout has one 32-column MFMA tile per wave");
  constexpr int MI = COLS4 ? 4 : 2;                     // 32-row MFMA tiles per wave
  constexpr int BNT = COLS4 ? 128 : 64 * NI;
  // activation tile only, double-buffered: [buffer][hi, lo][row * RS + k]  (43.5 KB).  Weights never touch LDS: each
  // wave loads its own MFMA B fragments straight from the fragment-ordered packed array (two waves of a workgroup read
  // the same lines; the second hits L1).  r01 ablation: staging weights through LDS cost 16 % in ds_write alone, made
  // the LDS pipe a co-bottleneck with the matrix pipe, and needed a barrier per tap (now: one per 32-channel block).
  // SPATIAL + RP_SP_SINGLE (2x2 waves): one buffer; the next block's tile is stored between two barriers once every wave is done
  // with the current one (108 MFMAs per wave between barrier pairs)
  constexpr int NBUF = (SPATIAL && RP_SP_SINGLE && !COLS4) ? 1 : 2;
  constexpr int EPI_H = (4 * 32 * (32 * NI + 4) * 4 + 4 * (8 * NI) * 8 * 8) / 2;      // epilogue staging + statistics, in halfs
  constexpr int LDS_H = NBUF * 2 * PRW * RS > EPI_H ? NBUF * 2 * PRW * RS : EPI_H;
  __shared__ __attribute__((aligned(16))) _Float16 sA[LDS_H];
  _Float16* const sAf = &sA[0];
#define RP_BUF(X_) (NBUF == 1 ? 0 : (X_))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = COLS4 ? 0 : wave >> 1, wn = COLS4 ? wave : wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- tile id: XCD-contiguous chunks, n fastest (the n tiles of one m tile share the activation tile) ----
  // K split (launches of a few dozen tiles: B = 1 crops): the `ks` workgroups of a tile each multiply a contiguous range of the
  // channel blocks, leave their accumulators in a workspace, and the one that arrives last adds them up IN SPLIT ORDER (the
  // result does not depend on who is last) and runs the epilogue.  A 3x3 256-channel layer of a 30 x 30 map is 8-24 tiles whose
  // 72-stage serial chain is what the launch takes (25 us); four splits of 18 stages take half.
  const int ks = p.ksplit;
  const int ntiles = ks > 1 ? static_cast<int>(gridDim.x) / ks : static_cast<int>(gridDim.x);
  const int split = ks > 1 ? static_cast<int>(blockIdx.x) / ntiles : 0;
  int bid = ks > 1 ? static_cast<int>(blockIdx.x) - split * ntiles : static_cast<int>(blockIdx.x);
  {
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int nt_i = bid % p.n_nt, mt_i = bid / p.n_nt;
  const int Mtot = p.B * p.U * p.V;          // < 2^31 - 256 (checked on the host): 32-bit index math throughout
  // global tiling: tile t = rows [128 t, 128 t + 128) of the B*U*V pixel rows; per-image tiling (tpi > 0): image t / tpi,
  // rows [128 (t % tpi), ...) of that image, output rows end with the image (halo rows of a neighbour image are never used:
  // they could only serve taps across an image line boundary, which the fast-axis masks already zero)
  const int img_ = p.tpi > 0 ? mt_i / p.tpi : 0;
  // SPATIAL: patch (py_, px_) of image img_: output row r of the tile = pixel (8 py_ + (r >> 4), 16 px_ + (r & 15))
  const int pt_ = SPATIAL ? mt_i - img_ * p.tpi : 0;
  const int py0_ = SPATIAL ? (pt_ / p.sp_tx) * 8 : 0, px0_ = SPATIAL ? (pt_ % p.sp_tx) * 16 : 0;
  const int m0 = p.tpi > 0 ? img_ * (p.U * p.V) + (mt_i - img_ * p.tpi) * BM : mt_i * BM;
  const int mend = p.tpi > 0 ? (img_ + 1) * (p.U * p.V) : Mtot;
  const int n0 = nt_i * BNT;
  const int UV = p.U * p.V;

  // ---- per-thread activation rows (global -> LDS staging): rows j = (tid>>3) + 32 r, 16-byte column c4 ----
  // (named scalars + macros on purpose: arrays / structs captured by lambdas end up in scratch or LDS here)
  const int c4 = tid & 7;
#define RP_ROW_INIT(R_)                                                                                     \
  int a_pix##R_, a_u##R_, a_v##R_;   /* rows outside the problem carry a_u = -2^20: every range test fails */ \
  {                                                                                                         \
    const int j_ = (tid >> 3) + 32 * R_;                                                                    \
    if (SPATIAL) {                   /* halo row j_ = pixel (py0 - 1 + j_ / 18, px0 - 1 + j_ % 18) of image img_ */ \
      const int hy_ = j_ / SPW, y_ = py0_ - 1 + hy_, x_ = px0_ - 1 + (j_ - hy_ * SPW);                      \
      const bool okr_ = (j_ < SPR) && static_cast<unsigned>(y_) < static_cast<unsigned>(p.U) &&             \
                        static_cast<unsigned>(x_) < static_cast<unsigned>(p.V);                             \
      a_u##R_ = okr_ ? 0 : -(1 << 20);                                                                      \
      a_v##R_ = 0;                                                                                          \
      a_pix##R_ = okr_ ? img_ * UV + y_ * p.V + x_ : 0;                                                     \
    } else {                                                                                                \
    const int m_ = m0 - HALO + j_;                                                                          \
    const bool okr_ = (j_ < AROWS) && m_ >= 0 && m_ < Mtot;                                                 \
    const int mm_ = okr_ ? m_ : 0;                                                                          \
    const int q_ = mm_ / p.V, v_ = mm_ - q_ * p.V;                                                          \
    const int b_ = q_ / p.U, u_ = q_ - b_ * p.U;                                                            \
    a_u##R_ = okr_ ? (STRIDED ? u_ * 2 : u_) : -(1 << 20);                                                  \
    a_v##R_ = STRIDED ? v_ * 2 : 0;                                                                         \
    a_pix##R_ = STRIDED ? b_ * (p.Uin * p.Vin) : b_ * UV + u_ * p.su + v_ * p.sv;                           \
    }                                                                                                       \
  }
  RP_ROW_INIT(0) RP_ROW_INIT(1) RP_ROW_INIT(2) RP_ROW_INIT(3) RP_ROW_INIT(4) RP_ROW_INIT(5)
  // ---- per-lane fragment rows: fast-axis coordinate for the tap masks ----
  int fv[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * 64 + mi * 32 + l31;
    fv[mi] = (m < mend) ? m % p.V : -1000;
    if (SPATIAL) {                   // (re-used as the lane's halo-tile row of the CENTRE tap: ((r >> 4) + 1) * 18 + (r & 15) + 1)
      const int r = wm * 64 + mi * 32 + l31;
      fv[mi] = ((r >> 4) + 1) * SPW + (r & 15) + 1;
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 av0_0, av0_1, av0_2, av0_3, av0_4, av0_5;   // staged activation rows in flight: register set 0 (row 5: SPATIAL) ...
  float4 av1_0, av1_1, av1_2, av1_3, av1_4, av1_5;   // ... and set 1 (DEEP only: the tile two blocks ahead)
  float4 nrm01 = make_float4(0.f, 1.f, 0.f, 1.f), nrm23 = nrm01;       // NORM: (mean, rstd) x 4 channels of the tile in flight
  int sat_n = 0;                    // range guard (p.sat != NULL): staged quads this thread had to clamp
  // every column tile of a row tile stages the SAME activation rows: the first one checks them for all (r03: the in-loop check
  // of all tiles cost 1.3 % of the step; the count is per staged quad of column tile 0 now)
  const bool sat_here = p.sat != nullptr && nt_i == 0;
  unsigned amask0 = 0u, amask1 = 0u;   // bit r: staged row r of the tile in flight (set 0 / 1) is inside the image (else: zeros)
#define RP_LOAD_A_ROW(R_, S_)                                                                                 \
  {                                                                                                         \
    const int uu_ = a_u##R_ + du_, vv_ = a_v##R_ + dvg_;                                                    \
    const bool in_ = cok_ && (STRIDED ? (static_cast<unsigned>(uu_) < static_cast<unsigned>(p.Uin) &&       \
                                         static_cast<unsigned>(vv_) < static_cast<unsigned>(p.Vin))         \
                                      : static_cast<unsigned>(uu_) < static_cast<unsigned>(p.U));           \
    const unsigned px_ = STRIDED ? static_cast<unsigned>(a_pix##R_) + static_cast<unsigned>(uu_) * p.su + static_cast<unsigned>(vv_) * p.sv \
                                 : static_cast<unsigned>(a_pix##R_ + dpix_);   /* (unsigned: rows outside wrap harmlessly) */ \
    const unsigned off_ = in_ ? px_ * static_cast<unsigned>(sg_.cstride) + cc_ : 0u;   /* < 2^31 (host check) */ \
    av##S_##_##R_ = *reinterpret_cast<const float4*>(sg_.ptr + off_);                                       \
    amask##S_ |= in_ ? (1u << R_) : 0u;                                                                     \
  }
#define RP_LOAD_A(G_, CB_, S_)                                                                              \
  do {                                                                                                      \
    Seg sg_ = p.seg0;                                                                                       \
    int cb0_ = 0;                                                                                           \
    if ((CB_) >= p.cb1) { sg_ = p.seg1; cb0_ = p.cb1; }                                                     \
    if ((CB_) >= p.cb2) { sg_ = p.seg2; cb0_ = p.cb2; }                                                     \
    if ((CB_) >= p.cb3) { sg_ = p.seg3; cb0_ = p.cb3; }                                                     \
    const int c_ = ((CB_) - cb0_) * BK + c4 * 4;                                                            \
    /* split sources: this thread's 16 bytes are one plane (c4 & 1) of the 8-channel group c4 >> 1 of the block */ \
    const bool cok_ = (HLIN ? ((CB_) - cb0_) * BK + (c4 >> 1) * 8 : c_) < sg_.ccount;                       \
    const int cc_ = sg_.coff + c_;                                                                          \
    const int du_ = p.du0 + (STRIDED ? (G_) / p.gkw : (G_));                                                \
    const int dvg_ = STRIDED ? p.dvg0 + (G_) % p.gkw : 0;                                                   \
    const int dpix_ = du_ * p.su;                                                                           \
    if (NORM) {                      /* mean, rstd of the 4 channels this thread stages (image img_, source 0) */ \
      const float* mr_ = p.in_mr + (static_cast<long long>(img_) * sg_.cstride + cc_) * 2;                  \
      nrm01 = *reinterpret_cast<const float4*>(mr_);                                                        \
      nrm23 = *reinterpret_cast<const float4*>(mr_ + 4);                                                    \
    }                                                                                                       \
    amask##S_ = 0u;                                                                                         \
    RP_LOAD_A_ROW(0, S_) RP_LOAD_A_ROW(1, S_) RP_LOAD_A_ROW(2, S_) RP_LOAD_A_ROW(3, S_) RP_LOAD_A_ROW(4, S_) \
    if (SPATIAL) RP_LOAD_A_ROW(5, S_)                                                                       \
  } while (0)
#define RP_STORE_A_ROW(R_, AB_, S_)                                                                         \
  {                                                                                                         \
    const int j_ = (tid >> 3) + 32 * R_;                                                                    \
    if (HLIN) {                                                                                             \
      if (j_ < ARW) {                /* rows outside the image: zeros (bitwise AND: a select of two float4 went through scratch) */ \
        const unsigned mk_ = 0u - ((amask##S_ >> R_) & 1u);                                                 \
        uint4 u_ = __builtin_bit_cast(uint4, av##S_##_##R_);                                                \
        u_.x &= mk_; u_.y &= mk_; u_.z &= mk_; u_.w &= mk_;                                                 \
        *reinterpret_cast<uint4*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + (c4 & 1) * (PRW * RS) + j_ * RS + (c4 >> 1) * 8) = u_; \
      }                                                                                                     \
    } else if (j_ < ARW) {                                                                                  \
      h4 hi_, lo_;                                                                                          \
      const float4 z4_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
      float4 xv_ = av##S_##_##R_;                                                                           \
      if (NORM) xv_ = make_float4(fmaxf((xv_.x - nrm01.x) * nrm01.y, 0.f), fmaxf((xv_.y - nrm01.z) * nrm01.w, 0.f), \
                                  fmaxf((xv_.z - nrm23.x) * nrm23.y, 0.f), fmaxf((xv_.w - nrm23.z) * nrm23.w, 0.f)); \
      split4((amask##S_ >> R_) & 1u ? xv_ : z4_, p.a_scale, hi_, lo_);   /* padding / out-of-range rows: zeros (AFTER the norm) */ \
      *reinterpret_cast<h4*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + j_ * RS + c4 * 4) = hi_;                        \
      *reinterpret_cast<h4*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + PRW * RS + j_ * RS + c4 * 4) = lo_;             \
    }                                                                                                       \
  }
#define RP_SAT_ROW(R_, S_) sat_n += (((amask##S_ >> R_) & 1u) && rp::quad_saturates(av##S_##_##R_, p.a_scale)) ? 1 : 0;
#define RP_STORE_A(AB_, S_)                                                                                 \
  do {                                                                                                      \
    if (!HLIN && sat_here) { RP_SAT_ROW(0, S_) RP_SAT_ROW(1, S_) RP_SAT_ROW(2, S_) RP_SAT_ROW(3, S_) RP_SAT_ROW(4, S_) if (SPATIAL) RP_SAT_ROW(5, S_) }   /* uniform branch, VALU only */ \
    RP_STORE_A_ROW(0, AB_, S_) RP_STORE_A_ROW(1, AB_, S_) RP_STORE_A_ROW(2, AB_, S_) RP_STORE_A_ROW(3, AB_, S_) RP_STORE_A_ROW(4, AB_, S_) \
    if (SPATIAL) RP_STORE_A_ROW(5, AB_, S_)                                                                 \
  } while (0)
  // weight fragment registers: two stages (named locals + macros: structs/arrays handed to lambdas end up in LDS or
  // scratch with this compiler).  Stage S holds, for the tap being consumed, this wave's B fragments
  // [ni][kk] x (hi, lo): 16 bytes per lane each, already in MFMA operand order.
  uint4 b0h00, b0h01, b0h10, b0h11, b0l00, b0l01, b0l10, b0l11;
  uint4 b1h00, b1h01, b1h10, b1h11, b1l00, b1l01, b1l10, b1l11;
  const int ntile0 = (n0 >> 5) + wn * NI;             // first 32-column tile of this wave
  const int ntiles32 = p.Npad >> 5;
#define RP_LOAD_B(S_, G_, T_, CB_)                                                                          \
  do {                                                                                                      \
    /* record index in the packed order [ky][cb][kx]; strided mode: group G_ = ky * kw + kx, one tap each */ \
    const long long st_ = STRIDED ? (static_cast<long long>((G_) / p.gkw) * p.ncb + (CB_)) * p.gkw + (G_) % p.gkw \
                                  : (static_cast<long long>(G_) * p.ncb + (CB_)) * p.T + (T_);             \
    const long long f_ = (st_ * ntiles32 + ntile0) * 256 + lane;                                            \
    const uint4* ws_ = p.wpk + f_;                                                                          \
    b##S_##h00 = ws_[0];                                                                                    \
    b##S_##h01 = ws_[64];                                                                                   \
    b##S_##l00 = ws_[128];                                                                                  \
    b##S_##l01 = ws_[192];                                                                                  \
    if (NI == 2) {                                                                                          \
      b##S_##h10 = ws_[256];                                                                                \
      b##S_##h11 = ws_[320];                                                                                \
      b##S_##l10 = ws_[384];                                                                                \
      b##S_##l11 = ws_[448];                                                                                \
    }                                                                                                       \
  } while (0)
#define RP_LOAD_B_CLAMPED(S_)                                                                               \
  do {                                                                                                      \
    const bool v_ = pg < p.G;                                                                               \
    RP_LOAD_B(S_, v_ ? pg : p.G - 1, v_ ? pt : p.T - 1, v_ ? pcb : p.ncb - 1);                              \
  } while (0)
#define RP_MMA_KK(S_, KK_, AB_)                                                                             \
  {                                                                                                         \
    const int ko = KK_ * 16 + lh * 8;                                                                       \
    h8 ah[MI], al[MI], bh[2], bl[2];                                                                        \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                     \
      /* tap outside the image row: 2x2 layout -> the lane reads the all-zero LDS row (1 select per fragment row instead \
         of 8 v_cndmask per fragment; -4 % on the 64-wide kernels); the 4-column layout sits at its 168-VGPR cap, where \
         the extra address registers spilled (measured slower), so it masks the loaded fragments instead */ \
      const int row = (COLS4 || okm_[mi]) ? wm * 64 + mi * 32 + l31 + HALO + dv_ : AROWS;                   \
      ah[mi] = *reinterpret_cast<const h8*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + row * RS + ko);                 \
      al[mi] = *reinterpret_cast<const h8*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + PRW * RS + row * RS + ko);      \
      if (COLS4 && !okm_[mi]) { ah[mi] = zero_; al[mi] = zero_; }                                           \
    }                                                                                                       \
    bh[0] = __builtin_bit_cast(h8, b##S_##h0##KK_);                                                         \
    bl[0] = __builtin_bit_cast(h8, b##S_##l0##KK_);                                                         \
    if (NI == 2) {                                                                                          \
      bh[1] = __builtin_bit_cast(h8, b##S_##h1##KK_);                                                       \
      bl[1] = __builtin_bit_cast(h8, b##S_##l1##KK_);                                                       \
    }                                                                                                       \
    /* term-major: the accumulator tiles are independent, consecutive MFMAs never wait on each other */    \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)     \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);        \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)     \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);        \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)     \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);        \
  }
#define RP_MMA(S_, DV_, AB_)                                                                                \
  do {                                                                                                      \
    const int dv_ = (DV_);                                                                                  \
    const h8 zero_ = {0, 0, 0, 0, 0, 0, 0, 0};                                                              \
    bool okm_[MI];                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                       \
        okm_[mi] = static_cast<unsigned>(fv[mi] + dv_) < static_cast<unsigned>(p.V);                        \
    RP_MMA_KK(S_, 0, AB_)                                                                                   \
    RP_MMA_KK(S_, 1, AB_)                                                                                   \
  } while (0)
  // one pipeline stage = one tap of one (group, 32-channel block):
  //   first tap: request the NEXT block's activation rows (consumed T taps later)
  //   MFMAs of this tap (activations from LDS buffer `ab`, weights from stage-S registers)
  //   refill stage S with the weights of tap i+2
  //   last tap: split + store the next activation tile into the other LDS buffer, ONE barrier, swap buffers
#define RP_STAGE(S_)                                                                                        \
  do {                                                                                                      \
    int ncb_ = ccb + 1, ng_ = cg;                                                                           \
    if (ncb_ == p.ncb) { ncb_ = 0; ++ng_; }                                                                 \
    if (ct == 0) RP_LOAD_A(ng_ < p.G ? ng_ : cg, ng_ < p.G ? ncb_ : ccb, 0);                                \
    RP_MMA(S_, p.dv0 + ct, ab);                                                                             \
    RP_LOAD_B_CLAMPED(S_);                                                                                  \
    if (++pt == p.T) { pt = 0; if (++pcb == p.ncb) { pcb = 0; ++pg; } }                                     \
    if (++ct == p.T) {                                                                                      \
      ct = 0;                                                                                               \
      ccb = ncb_; cg = ng_;                                                                                 \
      RP_STORE_A(ab ^ 1, 0);                                                                                \
        __syncthreads();                                                                                      \
      ab ^= 1;                                                                                              \
    }                                                                                                       \
  } while (0)

  if (tid < NBUF * 2 * (RS / 8)) {       // the zero row of each (buffer, hi/lo) plane: 80 bytes = 5 x 16, never overwritten
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(sAf + (tid / (RS / 8)) * (PRW * RS) + ARW * RS + (tid % (RS / 8)) * 8) = z;
  }
  if constexpr (TT > 0) {
    // ---------------- unrolled main loop (stride 1) ----------------
    // weight fragments: two stages of [ni][hi kk0, hi kk1, lo kk0, lo kk1]; stage s of this wave = 4 (8 for NI = 2) wave
    // loads at fixed offsets from  wp + s * sstride  (packed in consumption order)
    constexpr int NSTG = DEEP ? TT : 2;      // weight-fragment register stages (DEEP: one per tap of a block)
    uint4 bf[NSTG][NI][4];
    const uint4* wp = p.wpk + static_cast<long long>(ntile0) * 256 + lane;
    const int sstride = ntiles32 * 256;
    const int nit = SPATIAL ? p.ncb : p.G * p.ncb, nst = nit * TT;     // (SPATIAL: all nine taps belong to one channel block)
#define RP_LOADB2(S_, ST_)                                                                                  \
    {                                                                                                       \
      int st_ = (ST_) < nst ? (ST_) : nst - 1;                /* unconditional: past the end re-reads the last record */ \
      if (SPATIAL) {                 /* stage = 9 cb + (3 ky + kx); the records are packed [ky][cb][kx] */  \
        const int cbq_ = st_ / 9, tq_ = st_ - cbq_ * 9, kyq_ = tq_ / 3;                                     \
        st_ = (kyq_ * p.ncb + cbq_) * 3 + (tq_ - kyq_ * 3);                                                 \
      }                                                                                                     \
      const uint4* q_ = wp + static_cast<long long>(st_) * sstride;                                         \
      /* ablation bit 32: only the upper row half of the workgroup (wm == 0) requests weights -- HALF the L1 / TA traffic of  \
         the weight stream at unchanged everything else (results are wrong): is the stream's bandwidth what costs 20 %? */  \
      if (!((RP_ABL & 32) && wm == 1)) {                                                                    \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                   \
            _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) bf[S_][ni][qq] = q_[ni * 256 + qq * 64];       \
      }                                                                                                     \
    }
#define RP_MMA_KK2(S_, KK_, AB_)                                                                            \
    {                                                                                                       \
      const int ko = KK_ * 16 + lh * 8;                                                                     \
      h8 ah[MI], al[MI], bh[NI], bl[NI];                                                                    \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                   \
        /* TT == 1 (1x1 kernels): no tap shift, nothing to mask -- rows outside the problem only feed accumulator rows \
           that are never stored (the masks were 75 v_cndmask per 24 MFMAs in the 4-column layout) */       \
        /* SPATIAL: the lane's halo-tile row of the centre tap + the tap's constant offset; nothing to mask */ \
        const int row = SPATIAL ? fv[mi] + dv_                                                              \
                                : ((TT == 1 || COLS4 || okm_[mi]) ? wm * 64 + mi * 32 + l31 + HALO + dv_ : AROWS); \
        ah[mi] = *reinterpret_cast<const h8*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + row * RS + ko);               \
        al[mi] = *reinterpret_cast<const h8*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + PRW * RS + row * RS + ko);    \
        if (!SPATIAL && TT != 1 && COLS4 && !okm_[mi]) { ah[mi] = zero_; al[mi] = zero_; }                  \
      }                                                                                                     \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                   \
        bh[ni] = __builtin_bit_cast(h8, bf[S_][ni][KK_]);                                                   \
        bl[ni] = __builtin_bit_cast(h8, bf[S_][ni][2 + KK_]);                                               \
      }                                                                                                     \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)   \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);      \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)   \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);      \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)   \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);      \
    }
    // Software pipeline of one tap (2x2 wave layout; the compiler left alone sinks every LDS read and weight load to just
    // before the MFMA that needs it -- r02 ISA: ds_read_b128 / s_waitcnt lgkmcnt(0) / v_mfma, over and over):
    //     read A fragments of k-half 1 | 6 MFMAs of k-half 0 | read A fragments of the NEXT tap's k-half 0 | 6 MFMAs of k-half 1
    //     | request the weight record two stages ahead
    // with full scheduling fences between the groups, so every LDS read has six MFMAs (>= 192 cycles) and every weight load
    // a whole tap to land.  Fragment registers are double-buffered (fa_[set]).  Across the barrier at the end of a
    // (group, channel block) nothing can be prefetched (the other LDS buffer is still being written): one exposed LDS
    // latency per T taps.  The 4-column layout (MI = 4) has no registers left for this and keeps the simple form.
    h8 fa_h[2][MI], fa_l[2][MI];
#ifndef RP_ABL
#define RP_ABL 0          // ablation build (tools/conv_ablate.sh): bit 0 no weight loads, 1 no LDS fragment reads, 2 no
#endif                    // activation staging, 3 no barrier, 4 no epilogue stores -- in the main loop of the 2x2 layout
#define RP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define RP_READ_A(SET_, T_, KK_, AB_)                                                                       \
    {                                                                                                       \
      /* SPATIAL: tap T_ = 3 dy + dx reads the halo tile at the constant offset (dy - 1) * 18 + (dx - 1) */  \
      const int dvr_ = SPATIAL ? (((T_) / 3) - 1) * SPW + ((T_) % 3) - 1 : p.dv0 + (T_);                    \
      const int ko = (KK_) * 16 + lh * 8;                                                                   \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                   \
        const bool ok_ = SPATIAL || TT == 1 || static_cast<unsigned>(fv[mi] + dvr_) < static_cast<unsigned>(p.V); \
        const int row = SPATIAL ? fv[mi] + dvr_ : (ok_ ? wm * 64 + mi * 32 + l31 + HALO + dvr_ : AROWS);    /* masked tap: the all-zero row */ \
        fa_h[SET_][mi] = *reinterpret_cast<const h8*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + row * RS + ko);       \
        fa_l[SET_][mi] = *reinterpret_cast<const h8*>(sAf + RP_BUF(AB_) * (2 * PRW * RS) + PRW * RS + row * RS + ko); \
      }                                                                                                     \
    }
#define RP_MFMA6(SET_, S_, KK_)                                                                             \
    {                                                                                                       \
      h8 bh[NI], bl[NI];                                                                                    \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                   \
        bh[ni] = __builtin_bit_cast(h8, bf[S_][ni][KK_]);                                                   \
        bl[ni] = __builtin_bit_cast(h8, bf[S_][ni][2 + KK_]);                                               \
      }                                                                                                     \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)   \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_l[SET_][mi], bh[ni], acc[mi][ni], 0, 0, 0); \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)   \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[SET_][mi], bl[ni], acc[mi][ni], 0, 0, 0); \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)   \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_h[SET_][mi], bh[ni], acc[mi][ni], 0, 0, 0); \
    }
    // one (group, channel block): PAR_ = parity of its index = LDS buffer it reads = weight stage of its first tap
    // Block boundary (r03): the next tile is written to the other LDS buffer BEFORE the last tap (its loads have had TT - 1 taps
    // to land), and the barrier sits in the MIDDLE of the last tap: k-half 0 of the last tap | barrier | first fragment reads of
    // the NEXT block | k-half 1 of the last tap (its fragments are already in registers).  r02 had store -> barrier -> read ->
    // wait -> MFMA at every boundary: an LDS write + barrier + LDS read latency chain with an idle matrix pipe per block and wave
    // (the r03 ablation builds put the staging at 15 % of a GRU convolution although it is 10 instructions per 60 MFMAs).
#ifndef RP_OVERLAP
#define RP_OVERLAP 0      // measured neutral on the step (r03: 664.8 vs 666.3 iters/s same-box): off, kept for the record
#endif
    constexpr bool OVL = RP_OVERLAP && !COLS4 && TT >= 3;
#define RP_BODY(PAR_, OTH_)                                                                                 \
    {                                                                                                       \
      int ncb_ = ccb + 1, ng_ = cg;                                                                         \
      if (ncb_ == p.ncb) { ncb_ = 0; ++ng_; }                                                               \
      if (!OVL && !COLS4 && !(RP_ABL & 2)) { RP_READ_A(0, 0, 0, PAR_) }   /* (OVL: read at the previous boundary / in the prologue) */ \
      if (DEEP) {                    /* the tile TWO blocks ahead -> register set PAR_ (stored to LDS at the end of the previous block) */ \
        int ncb2_ = ncb_ + 1, ng2_ = ng_;                                                                   \
        if (ncb2_ == p.ncb) { ncb2_ = 0; ++ng2_; }                                                          \
        const bool v2_ = ng2_ < p.G, v1_ = ng_ < p.G;                                                       \
        RP_LOAD_A(v2_ ? ng2_ : (v1_ ? ng_ : cg), v2_ ? ncb2_ : (v1_ ? ncb_ : ccb), PAR_);                   \
      } else if (!(RP_ABL & 4)) {                                                                           \
        RP_LOAD_A(ng_ < p.G ? ng_ : cg, ng_ < p.G ? ncb_ : ccb, 0);   /* next tile (the last one re-requests itself) */ \
      }                                                                                                     \
      if (!COLS4) RP_SCHED_FENCE();                                                                         \
      _Pragma("unroll") for (int t = 0; t < TT; ++t) {                                                      \
        if (COLS4) {                                                                                        \
          const int dv_ = SPATIAL ? ((t / 3) - 1) * SPW + (t % 3) - 1 : p.dv0 + t;                          \
          const h8 zero_ = {0, 0, 0, 0, 0, 0, 0, 0};                                                        \
          bool okm_[MI];                                                                                    \
          _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                 \
              okm_[mi] = SPATIAL || static_cast<unsigned>(fv[mi] + dv_) < static_cast<unsigned>(p.V);       \
          RP_MMA_KK2(((PAR_) + t) & 1, 0, PAR_)                                                             \
          RP_MMA_KK2(((PAR_) + t) & 1, 1, PAR_)                                                             \
          RP_LOADB2(((PAR_) + t) & 1, s0 + t + 2)                                                           \
        } else {                                                                                            \
          const int slot_ = DEEP ? t : (((PAR_) + t) & 1);                                                  \
          const bool last_ = OVL && t == TT - 1;                                                            \
          if (last_) {               /* the next block's tile -> the other LDS buffer, before the last tap */ \
            if (DEEP) { RP_STORE_A(OTH_, OTH_); } else if (!(RP_ABL & 4)) { RP_STORE_A(OTH_, 0); }          \
            RP_SCHED_FENCE();                                                                               \
          }                                                                                                 \
          if (!(RP_ABL & 2)) { RP_READ_A(1, t, 1, PAR_) }                                                   \
          RP_SCHED_FENCE();                                                                                 \
          RP_MFMA6(0, slot_, 0)                                                                             \
          RP_SCHED_FENCE();                                                                                 \
          if (last_) {               /* boundary: everybody's stores are in, nobody reads buffer PAR_ any more */ \
            if (!(RP_ABL & 8)) __syncthreads();                                                             \
            if (!(RP_ABL & 2)) { RP_READ_A(0, 0, 0, OTH_) }                                                 \
          } else if (t + 1 < TT && !(RP_ABL & 2)) { RP_READ_A(0, t + 1, 0, PAR_) }                          \
          RP_SCHED_FENCE();                                                                                 \
          RP_MFMA6(1, slot_, 1)                                                                             \
          RP_SCHED_FENCE();                                                                                 \
          if (!(RP_ABL & 1)) { RP_LOADB2(slot_, s0 + t + (DEEP ? TT : 2)) }   /* DEEP: this tap of the NEXT block */ \
          RP_SCHED_FENCE();                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
      s0 += TT;                                                                                             \
      ccb = ncb_; cg = ng_;                                                                                 \
      if (!OVL) {                                                                                           \
        if (NBUF == 1 && !(RP_ABL & 8)) __syncthreads();   /* single buffer: every wave is done reading the current tile */ \
        if (DEEP) { RP_STORE_A(OTH_, OTH_); }            /* the NEXT block's tile, requested one block ago */  \
        else if (!(RP_ABL & 4)) { RP_STORE_A(OTH_, 0); }                                                    \
        if (!(RP_ABL & 8)) __syncthreads();                                                                 \
      }                                                                                                     \
    }
    // this workgroup's range of (group, channel block) iterations (all of them unless the launch splits K)
    const int it0 = ks > 1 ? nit * split / ks : 0, it1 = ks > 1 ? nit * (split + 1) / ks : nit;
    const int ksg0 = SPATIAL ? 0 : it0 / p.ncb, kscb0 = SPATIAL ? it0 : it0 - ksg0 * p.ncb;
    // (weight requests first: the activation tile is waited for right away and its wait then covers both; measured neutral)
    if constexpr (DEEP) {
#pragma unroll
      for (int t = 0; t < TT; ++t) RP_LOADB2(t, t)
    } else {
      RP_LOADB2(0, it0 * TT)
      RP_LOADB2(1, it0 * TT + 1)
    }
    RP_LOAD_A(ksg0, kscb0, 0);
    RP_SCHED_FENCE();
    RP_STORE_A(0, 0);
    if constexpr (DEEP) {            // second tile in flight (stored at the end of the first block)
      const bool v1_ = p.ncb > 1 || p.G > 1;
      RP_LOAD_A(v1_ ? (p.ncb > 1 ? 0 : 1) : 0, v1_ ? (p.ncb > 1 ? 1 : 0) : 0, 1);
    }
    __syncthreads();
    if (RP_ABL & 2) { RP_READ_A(0, 0, 0, 0) RP_READ_A(1, 0, 1, 0) }
    int cg = ksg0, ccb = kscb0, s0 = it0 * TT, it = it0;
    if (OVL && !(RP_ABL & 2)) { RP_READ_A(0, 0, 0, 0) }      // first fragments of the first block (later blocks: at the boundary)
    for (; it + 1 < it1; it += 2) {
      RP_BODY(0, 1)
      RP_BODY(1, 0)
    }
    if (it < it1) RP_BODY(0, 1)
  } else {
    // ---------------- generic stage machine (stride 2) ----------------
 RP_LOAD_A(0, 0, 0);
  RP_STORE_A(0, 0);
  int ab = 0;                      // LDS buffer holding the activation tile being consumed
  int cg = 0, ccb = 0, ct = 0;     // coordinates of the tap being consumed
  int pg = 0, pcb = 0, pt = 0;     // coordinates of the next weight tile to request
  const int total = p.G * p.ncb * p.T;
  RP_LOAD_B(0, pg, pt, pcb);
  if (++pt == p.T) { pt = 0; if (++pcb == p.ncb) { pcb = 0; ++pg; } }
  RP_LOAD_B_CLAMPED(1);
  if (++pt == p.T) { pt = 0; if (++pcb == p.ncb) { pcb = 0; ++pg; } }
  __syncthreads();
  // (stage pairs in a branch-free loop body + an odd tail: with "if (i + 1 < total) stage 1" inside the loop the compiler
  //  sees a path stage 0 -> latch -> stage 0 on which the stage-0 weight loads are the youngest in flight, and drains
  //  the stage-1 prefetch with vmcnt(0) in every stage 0)
  const int total_even = total & ~1;
  if (total_even) {
    int i = 0;
    do {
      RP_STAGE(0);
      RP_STAGE(1);
      i += 2;
    } while (i < total_even);
  }
  if (total & 1) RP_STAGE(0);

  }
  if (ks > 1) {                      // K split: partial accumulators -> workspace; the last arrival of the tile sums them in split order
    constexpr int ACCN = MI * NI * 16;
    const int tile_lin = mt_i * p.n_nt + nt_i;
    float* wsb = p.ks_ws + (static_cast<long long>(tile_lin) * ks + split) * (NT * ACCN) + tid;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) wsb[((i * NI + j) * 16 + r) * NT] = acc[i][j][r];
    __shared__ int ks_last;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(p.ks_cnt + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == static_cast<unsigned>(ks - 1);
      if (last) __hip_atomic_store(p.ks_cnt + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      ks_last = last;
    }
    __syncthreads();
    // range guard: the quads this workgroup clamped while staging ITS share of the channel blocks are counted here -- every
    // workgroup of a tile but the last leaves on the next line (r03 counted only behind the epilogue: with ksplit = 4 three
    // quarters of the clamp events were lost, and which quarter survived depended on the arrival order)
    if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));
    sat_n = 0;
    if (!ks_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float* wsr = p.ks_ws + static_cast<long long>(tile_lin) * ks * (NT * ACCN) + tid;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = wsr[((i * NI + j) * 16 + r) * NT];
    for (int sp = 1; sp < ks; ++sp) {
      const float* wq = wsr + static_cast<long long>(sp) * (NT * ACCN);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += wq[((i * NI + j) * 16 + r) * NT];
    }
  }
  // ------------------------------------------- epilogue -------------------------------------------
  // accumulators -> wave-private LDS tile (aliasing the weight staging buffers) -> 16-byte row-contiguous stores:
  // 16 lanes cover 256 contiguous bytes of one output pixel (the MFMA C layout would give 4-byte stores spread
  // over 2 rows per instruction: measured 0.86 TB/s, the epilogue was 35-45 % of the kernel).
  __syncthreads();
  constexpr int ES = 32 * NI + 4;                       // row stride (floats) of the staging tile
  constexpr int F4 = 8 * NI;                            // float4 per tile row
  // tile statistics of this lane's column quad, in fp64 (r03): the instance norm takes var = E[x^2] - mean^2, and the encoder's
  // first layers see an almost constant image (2 (x / 255) - 1 of a [0,1] image: model/CFNet.py:42-43), mean^2 / var up to 2e3 --
  // fp32 sums lost 4-5 digits of the variance there and were the largest single source of the GPU's distance to the oracle
  double ts0 = 0., ts1 = 0., ts2 = 0., ts3 = 0., tq0 = 0., tq1 = 0., tq2 = 0., tq3 = 0.;
  float* S = reinterpret_cast<float*>(sAf) + wave * (32 * ES);    // 4 x 32 x 68 floats = 34.8 KB <= 43.5 KB
  const int colw = n0 + wn * (32 * NI);
  // Every lane works on ONE column quad (64 % F4 == 0): its bias is loaded once.  Per 32-row block (mi) the epilogue operands of
  // all 4*NI row groups -- the additive map, h and z of the GRU forms -- are requested up front, unconditionally (rows /
  // columns outside the problem read row mend-1 / column 0 and are dropped at the store), and only then are the accumulators
  // staged through LDS: with the loads under the per-row `continue` the compiler waited vmcnt(0) in every row group, 4*NI*MI
  // dependent memory round trips per workgroup -- a large part of the ~10 us launch floor of this kernel (r02: +4.6 % on the
  // step).  Requesting BOTH row blocks of the 2x2 layout at once (96 operand registers) spilled and was 2 % slower.
  constexpr int KG = 4 * NI;
  const int colq = colw + (lane % F4) * 4;
  const bool colok = colq < p.Cout;
  const int colc = colok ? colq : 0;
  const int nv = colok ? (p.Cout - colq < 4 ? p.Cout - colq : 4) : 0;     // valid columns of this quad (Cout = 126 -> tail of 2)
  float bq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bq[e] = p.bias[colc + (e < nv ? e : 0)];
  const int c2 = colc >= p.gru_c ? colc - p.gru_c : 0;                    // epi 2: column inside the r half
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    long long pixk[KG];
    float4 am[KG], hv[KG], zv[KG];
    unsigned rowok = 0u;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int rl = (lane + 64 * k) / F4;
      const int m = m0 + wm * 64 + mi * 32 + rl;
      const int mc = m < mend ? m : mend - 1;
      long long pix = mc;
      if (SPATIAL) {                 // tile row r -> pixel (py0 + (r >> 4), px0 + (r & 15)) of image img_; partial patches at the border
        const int r = wm * 64 + mi * 32 + rl;
        const int y = py0_ + (r >> 4), x = px0_ + (r & 15);
        rowok |= ((y < p.U && x < p.V) ? 1u : 0u) << k;
        pix = static_cast<long long>(img_) * UV + (y < p.U ? y : p.U - 1) * p.V + (x < p.V ? x : p.V - 1);
      } else {
        rowok |= (m < mend ? 1u : 0u) << k;
      }
      if (!SPATIAL && p.sv != 1) {
        const int q = mc / p.V, v = mc - q * p.V;
        const int b = q / p.U, u = q - b * p.U;
        pix = static_cast<long long>(b) * UV + u * p.su + v * p.sv;
      }
      pixk[k] = pix;
    }
    if (p.addm) {                   // (c_out % 4 == 0 checked on the host)
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
      for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + ni * 32 + l31] = acc[mi][ni][r] * p.out_scale;
#pragma unroll
    for (int k = 0; k < KG; ++k) {
      const int idx = lane + 64 * k;
      const int rl = idx / F4, c = (idx % F4) * 4;
      if (!((rowok >> k) & 1u) || !colok) continue;
      if ((RP_ABL & 16) && acc[0][0][0] != 12345.678f) continue;
      const long long pix = pixk[k];
      const int col = colq;
      const float4 a4 = *reinterpret_cast<const float4*>(S + rl * ES + c);
      float y[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] += (e < nv) ? bq[e] : 0.f;
      if (p.addm) { y[0] += am[k].x; y[1] += am[k].y; y[2] += am[k].z; y[3] += am[k].w; }
      float* drow = p.dst + pix * p.dst_cs;      // destination pixel row, channel index inside it, fp32 or split form
      int dch = p.dst_co + col;
      int dhl = p.dst_hl;
      if (p.tstats) {              // (column quad of a lane is the same for every k and mi: 64 % F4 == 0)
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
        if (col < p.gru_c) {
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
        const float4 z = zv[k], h4 = hv[k];
        y[0] = (1.f - z.x) * h4.x + z.x * tanhf(y[0]); y[1] = (1.f - z.y) * h4.y + z.y * tanhf(y[1]);   // h' = (1-z)h + z q
        y[2] = (1.f - z.z) * h4.z + z.z * tanhf(y[2]); y[3] = (1.f - z.w) * h4.w + z.w * tanhf(y[3]);
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
      if (p.dsth) store_quad_hl(p.dsth + pix * p.dsth_cs, p.dsth_co + col, y[0], y[1], y[2], y[3], nv, p.a_scale, sat_n);
    }
  }
  if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));
  if (p.tstats) {
    // lanes sharing a column quad (same lane % F4) -> lanes 0..F4-1, fixed order; then rows 0-63 (wm = 0) + rows 64-127
    // (wm = 1) through LDS; one (sum, sum of squares) pair per tile and column: deterministic, no atomics
#pragma unroll
    for (int o = F4; o < 64; o <<= 1) {
      ts0 += rp::shfl_xor_f64(ts0, o); ts1 += rp::shfl_xor_f64(ts1, o); ts2 += rp::shfl_xor_f64(ts2, o); ts3 += rp::shfl_xor_f64(ts3, o);
      tq0 += rp::shfl_xor_f64(tq0, o); tq1 += rp::shfl_xor_f64(tq1, o); tq2 += rp::shfl_xor_f64(tq2, o); tq3 += rp::shfl_xor_f64(tq3, o);
    }
    if (COLS4) {                       // a wave owns all 128 rows of its 32 columns: nothing to combine
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
    } else {
      double* TS = reinterpret_cast<double*>(reinterpret_cast<float*>(sAf) + 4 * (32 * ES));      // behind the four staging tiles (8-byte aligned: 32 * ES is even)
      __syncthreads();
      if (lane < F4) {
        double* t = TS + (wave * F4 + lane) * 8;
        t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = ts3; t[4] = tq0; t[5] = tq1; t[6] = tq2; t[7] = tq3;
      }
      __syncthreads();
      if (wm == 0 && lane < F4) {
        const double* t0 = TS + (wave * F4 + lane) * 8;
        const double* t1 = TS + ((wave + 2) * F4 + lane) * 8;
        const int col = colw + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < p.Cout) {
            double* o = p.tstats + (static_cast<long long>(mt_i) * p.Cout + col + e) * 2;
            o[0] = t0[e] + t1[e];
            o[1] = t0[4 + e] + t1[4 + e];
          }
      }
    }
  }
}

// ---- weight packing: (Cout, Cin, kh, kw) fp32 -> [g][t][cb][Npad][32] fp16 hi / lo -------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ pk, const PackParams q) {
  const long long total = static_cast<long long>(q.G) * q.T * q.ncb * q.Npad * BK * 2;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  // consumption order of the kernels: [g][cb][t][32-col tile][part: hi kk0, hi kk1, lo kk0, lo kk1][lane][8] -- one stage
  // (group, channel block, tap) of one 32-column tile is 4 KB contiguous = the four 1-KB wave loads of that stage at fixed
  // offsets from ONE address; lane l carries column n = 32*tile + (l & 31) and the 8 channels k = 16*kk + 8*(l >> 5) + j
  // -- exactly one MFMA B operand (16 bytes per lane)
  const int j8 = static_cast<int>(i % 8);
  const int ln = static_cast<int>((i / 8) % 64);
  const int part = static_cast<int>((i / 512) % 4);
  const int kk = part & 1;
  const int nt32 = q.Npad / 32;
  const int ntl = static_cast<int>((i / 2048) % nt32);
  const long long st = i / (2048LL * nt32);                 // stage index (g * ncb + cb) * T + t
  const int t = static_cast<int>(st % q.T);
  const int cb = static_cast<int>((st / q.T) % q.ncb);
  const int g = static_cast<int>(st / (static_cast<long long>(q.T) * q.ncb));
  const int k = kk * 16 + (ln >> 5) * 8 + j8;
  const int n = ntl * 32 + (ln & 31);
  const int s = q.cb_seg[cb];
  const int cl = q.cb_c0[cb] + k;
  float v = 0.f;
  if (n < q.Cout && cl < q.seg_count[s]) {
    const int ci = q.seg_start[s] + cl;
    // vertical (kw == 1, kh > 1): one group, taps along y.  otherwise groups = ky, taps = kx.
    const int ky = q.vertical ? t : g, kx = q.vertical ? 0 : t;
    v = w[((static_cast<long long>(n) * q.Cin + ci) * q.kh + ky) * q.kw + kx] * q.w_scale;
  }
  const _Float16 h = static_cast<_Float16>(v);
  pk[i] = part < 2 ? h : static_cast<_Float16>(v - static_cast<float>(h));
}

int fill_cb_tables(const int* counts, int n, unsigned char* cb_seg, short* cb_c0) {
  int ncb = 0;
  for (int s = 0; s < n; ++s)
    for (int c = 0; c < counts[s]; c += BK) {
      if (ncb >= MAX_CB) return -1;
      cb_seg[ncb] = static_cast<unsigned char>(s);
      cb_c0[ncb] = static_cast<short>(c);
      ++ncb;
    }
  return ncb;
}

}  // namespace

using namespace rpconv;

static int g_conv_strip = 1;            // 0: never the strip kernels (conv_strip.hip); 1: automatic (rnnpose_conv_strip)
static bool g_conv_spatial = true;      // 3x3 stride-1 convolutions on 8 x 16 image patches (rnnpose_conv_spatial_tiles)
static bool g_conv_ksplit = true;       // K split of launches with few tiles when the caller provides a workspace (rnnpose_conv_ksplit)
static int g_ks_max_tiles = 24, g_ks_max_splits = 4;      // (rnnpose_conv_ksplit_limits: measurement)
constexpr int KS_MAX_WG = 192;          // tiles x splits of a split launch
constexpr size_t KS_CNT_BYTES = 1024;   // arrival counters (<= 96 tiles) at the head of the workspace

extern "C" {

int rnnpose_conv_spatial_tiles(int enable) {     // measurement switch: 0 = the r02 row-major tiling for 3x3 layers too
  g_conv_spatial = enable != 0;
  return 0;
}

int rnnpose_conv_strip(int mode) {               // 0 = never the strip kernels, 1 = automatic (default); measurement: 2 = automatic without
  if (mode < 0 || mode > 7) {                    // the two-wave workgroups of the 64-channel layers, 3 = two 32-column tiles per wave,
    rp::set_error("rnnpose_conv_strip: mode 0..7");      // 4 = 160-row strips only (by the image shape: r04's first rule),
    return 1;                                    // 5 = automatic without the stride-2 form (r05), 6 = automatic without r06's 32-row strips for
  }                                              //     launches of <= 256 waves
  g_conv_strip = mode != 0;
  strip_allow_small(mode != 4);
  strip_allow_s2(mode != 5);
  strip_allow_small32(mode != 6);
  strip_allow_persist(mode == 7);                // 7 = automatic WITH r06's persistent launches of the fp32-source strip forms (measured slower: off by default)
  strip_allow_two_wave(mode != 2);
  strip_force_ni(mode == 3 ? 2 : 1);
  return 0;
}

int rnnpose_conv_ksplit(int enable) {            // measurement switch: 0 = never split K
  g_conv_ksplit = enable != 0;
  return 0;
}

int rnnpose_conv_ksplit_limits(int max_tiles, int max_splits) {      // measurement: which launches split (default 24 tiles, 4 splits)
  if (max_tiles < 1 || max_tiles > 96 || max_splits < 2 || max_splits > 8) {
    rp::set_error("rnnpose_conv_ksplit_limits: max_tiles 1..96, max_splits 2..8");
    return 1;
  }
  g_ks_max_tiles = max_tiles; g_ks_max_splits = max_splits;
  return 0;
}

size_t rnnpose_conv_ksplit_workspace_bytes(void) {
  return KS_CNT_BYTES + static_cast<size_t>(KS_MAX_WG) * (NT * 32) * sizeof(float);
}

int rnnpose_conv_tiles_per_image(int H, int W, int kh, int kw, int stride) {
  if (H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return -1;
  const int Ho = stride == 2 ? (H + 1) / 2 : H, Wo = stride == 2 ? (W + 1) / 2 : W;
  if (g_conv_spatial && kh == 3 && kw == 3 && stride == 1) return rp::cdiv(W, 16) * rp::cdiv(H, 8);
  return rp::cdiv(static_cast<long long>(Ho) * Wo, BM);
}

// Strip height of the launch `d` describes (0: the 128-row kernels).  tile 5 / 6 force a height (errors surface in the launch); the
// automatic choice takes strips only when the sources qualify -- whole 32-channel blocks, the fused normalisation only for 3x3 --
// and otherwise falls back, with or without tile statistics (r04 raised an error there: ADVICE).  The ONE place this is decided.
static int strip_tile_rows(int tile) { return tile == 5 ? 160 : (tile == 6 ? 32 : 96); }      // tile codes of the strip heights (7: r06)
static int desc_strip_rows(const rnnpose_conv_desc_t* d) {
  if (d->stride == 2) {      // the parity-plane form: ONE fp32 source of whole 32-channel blocks, no fused normalisation, <= MAX_CB blocks over the four planes
    if (d->n_src != 1 || d->src[0].c_count % 32 != 0 || d->src_hl || d->src0_mean_rstd || 4 * (d->src[0].c_count / 32) > MAX_CB) return 0;
    if (!((d->kh == 3 && d->kw == 3) || (d->kh == 1 && d->kw == 1))) return 0;
  }
  if (d->tile >= 5) return strip_rows(d->H, d->W, d->kh, d->kw, d->stride, d->c_out, d->B, strip_tile_rows(d->tile));
  if (d->tile != 0 || !g_conv_strip) return 0;
  const int rows = strip_rows(d->H, d->W, d->kh, d->kw, d->stride, d->c_out, d->B, 0);
  if (!rows) return 0;
  for (int s = 0; s < d->n_src; ++s)
    if (d->src[s].c_count % 32 != 0) return 0;
  if (d->src0_mean_rstd && !(d->kh == 3 && d->kw == 3)) return 0;
  return rows;
}

int rnnpose_conv_tiles_per_image_desc(const rnnpose_conv_desc_t* d) {
  if (!d || d->H <= 0 || d->W <= 0 || (d->stride != 1 && d->stride != 2) || d->c_out <= 0 || d->tile < 0 || d->tile > 7 || d->B < 1 ||
      d->n_src < 1 || d->n_src > 4)
    return -1;
  const int rows = desc_strip_rows(d);
  if (d->tile >= 5 && rows == 0) return -1;
  if (rows) return d->stride == 2 ? strip_tiles_per_image(d->H / 2, d->W / 2, 3, 3, rows)      // (tiles of the OUTPUT grid, 10 x 16 patches; even sizes only)
                                  : strip_tiles_per_image(d->H, d->W, d->kh, d->kw, rows);
  return rnnpose_conv_tiles_per_image(d->H, d->W, d->kh, d->kw, d->stride);
}

int rnnpose_conv_products_desc(const rnnpose_conv_desc_t* d) {
  if (!d || rnnpose_conv_tiles_per_image_desc(d) < 0) return -1;
  const int rows = desc_strip_rows(d);
  return (d->single_product && rows == 160 && d->stride == 1 && (strip_waves(d->c_out) >> 4) == 1) ? 1 : 3;      // (the rule of strip_launch: P1 = 160-row strips, one column tile per wave)
}

int rnnpose_conv_tiles_per_image_ex(int H, int W, int kh, int kw, int stride, int c_out, int tile, int batch) {
  if (H <= 0 || W <= 0 || (stride != 1 && stride != 2) || c_out <= 0 || tile < 0 || tile > 7 || batch < 1) return -1;
  int rows = 0;
  if (tile >= 5) rows = strip_rows(H, W, kh, kw, stride, c_out, batch, strip_tile_rows(tile));
  else if (tile == 0 && g_conv_strip) rows = strip_rows(H, W, kh, kw, stride, c_out, batch, 0);
  if (tile >= 5 && rows == 0) return -1;
  // stride 2: whether the launch takes the strip form also depends on its sources (one fp32 source of whole 32-channel blocks,
  // no fused normalisation) -- a shape-only answer could size the statistics for a tiling the launch then refuses: ask _desc
  if (rows && stride == 2) return -1;
  if (rows) return strip_tiles_per_image(H, W, kh, kw, rows);
  return rnnpose_conv_tiles_per_image(H, W, kh, kw, stride);
}

long long rnnpose_conv_packed_halfs(int c_out, int kh, int kw, const int* h_seg_counts, int n_seg) {
  if (c_out <= 0 || kh <= 0 || kw <= 0 || !h_seg_counts || n_seg < 1 || n_seg > 4) return -1;
  unsigned char cs[MAX_CB];
  short c0[MAX_CB];
  const int ncb = fill_cb_tables(h_seg_counts, n_seg, cs, c0);
  if (ncb < 0) return -1;
  const long long Npad = static_cast<long long>(rp::cdiv(c_out, BN)) * BN;
  // hi and lo parts interleaved; TWO copies: the 128-row kernels' fragment order, then the strip kernels' record order -- and for a 3x3 / 1x1
  // layer with one source of whole 32-channel blocks a THIRD one: the stride-2 form over parity planes (conv_strip.hip, r05: +32 units next to
  // 36 for a 3x3 layer, 4x the stride-1 size with three quarters zeros for a 1x1 one; packed whether or not the layer ever runs at stride 2 --
  // the pack call has no stride argument: ADVICE r05, accepted: weights are packed once per parameter version, 1.6 MB for the whole encoder)
  return 2 * static_cast<long long>(kh) * kw * ncb * Npad * BK * 2 + strip_s2_halfs(h_seg_counts[0], kh, kw, ncb, static_cast<int>(Npad), n_seg);
}

int rnnpose_conv_pack_weights_f16x3(const float* w_oihw, int c_out, int c_in, int kh, int kw, const int* h_seg_counts,
                                    int n_seg, float w_scale, void* w_packed, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_conv_pack_weights_f16x3";
  RP_REQUIRE(w_oihw && w_packed && h_seg_counts, fn, "null pointer");
  RP_REQUIRE(n_seg >= 1 && n_seg <= 4, fn, "1..4 source segments");
  RP_REQUIRE((kh & 1) && (kw & 1) && kh <= 7 && kw <= 7, fn, "odd kernel sizes up to 7");
  PackParams q{};
  int tot = 0;
  for (int s = 0; s < n_seg; ++s) {
    RP_REQUIRE(h_seg_counts[s] > 0 && h_seg_counts[s] % 4 == 0, fn, "segment channel counts must be positive multiples of 4");
    q.seg_start[s] = static_cast<short>(tot);
    q.seg_count[s] = static_cast<short>(h_seg_counts[s]);
    tot += h_seg_counts[s];
  }
  RP_REQUIRE(tot == c_in, fn, "segment channel counts must sum to c_in");
  q.ncb = fill_cb_tables(h_seg_counts, n_seg, q.cb_seg, q.cb_c0);
  RP_REQUIRE(q.ncb > 0, fn, "too many channel blocks");
  q.Cout = c_out; q.Cin = c_in; q.kh = kh; q.kw = kw;
  q.vertical = (kw == 1 && kh > 1) ? 1 : 0;
  q.G = q.vertical ? 1 : kh;
  q.T = q.vertical ? kh : kw;
  q.Npad = rp::cdiv(c_out, BN) * BN;
  q.w_scale = w_scale;
  const long long total = static_cast<long long>(q.G) * q.T * q.ncb * q.Npad * BK * 2;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, rp::as_stream(stream), w_oihw,
                     static_cast<_Float16*>(w_packed), q);
  strip_pack(w_oihw, static_cast<_Float16*>(w_packed) + total, q, rp::as_stream(stream));      // (3x3, 1x5, 5x1 only)
  return rp::check_launch(fn);
}

int rnnpose_conv2d_nhwc_f16x3(const rnnpose_conv_desc_t* d, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_conv2d_nhwc_f16x3";
  RP_REQUIRE(d, fn, "null descriptor");
  RP_REQUIRE(d->n_src >= 1 && d->n_src <= 4, fn, "1..4 sources");
  RP_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->c_out > 0, fn, "bad size");
  RP_REQUIRE((d->kh & 1) && (d->kw & 1) && d->kh <= 7 && d->kw <= 7, fn, "odd kernel sizes up to 7");
  RP_REQUIRE(d->w_packed && d->bias && d->dst, fn, "null pointer");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(d->w_packed) % 16 == 0, fn, "packed weights must be 16-byte aligned");
  RP_REQUIRE(d->epilogue >= 0 && d->epilogue <= 3, fn, "epilogue must be 0..3");
  RP_REQUIRE(d->stride == 1 || d->stride == 2, fn, "stride must be 1 or 2");
  RP_REQUIRE(d->a_scale > 0.f && d->w_scale > 0.f, fn, "scales must be positive");
  if (d->epilogue == 2) RP_REQUIRE(d->aux0 && d->dst2 && d->gru_c > 0 && d->c_out == 2 * d->gru_c, fn, "gru_zr needs aux0 (h), dst2 (r*h), c_out == 2*gru_c");
  if (d->epilogue == 3) RP_REQUIRE(d->aux0 && d->aux1, fn, "gru_q needs aux0 (h) and aux1 (z)");
  RP_REQUIRE(d->dst_c_stride % 4 == 0 && d->dst_c_offset % 4 == 0 && reinterpret_cast<uintptr_t>(d->dst) % 16 == 0, fn,
             "dst: 16-byte aligned, channel stride/offset multiples of 4");
  if (d->epilogue >= 2) {
    RP_REQUIRE(d->c_out % 4 == 0 && d->gru_c % 4 == 0, fn, "GRU epilogues need c_out % 4 == 0");
    RP_REQUIRE(d->aux0_c_stride % 4 == 0 && d->aux0_c_offset % 4 == 0 && reinterpret_cast<uintptr_t>(d->aux0) % 16 == 0, fn,
               "aux0: 16-byte aligned, channel stride/offset multiples of 4");
  }
  if (d->epilogue == 2) RP_REQUIRE(d->dst2_c_stride % 4 == 0 && d->dst2_c_offset % 4 == 0 && reinterpret_cast<uintptr_t>(d->dst2) % 16 == 0, fn, "dst2 alignment");
  if (d->epilogue == 3) RP_REQUIRE(d->aux1_c_stride % 4 == 0 && d->aux1_c_offset % 4 == 0 && reinterpret_cast<uintptr_t>(d->aux1) % 16 == 0, fn, "aux1 alignment");
  KParams p{};
  int counts[4];
  for (int s = 0; s < d->n_src; ++s) {
    const rnnpose_conv_src_t& sr = d->src[s];
    RP_REQUIRE(sr.ptr && sr.c_count > 0 && sr.c_count % 4 == 0 && sr.c_stride % 4 == 0 && sr.c_offset % 4 == 0 &&
                   sr.c_offset + sr.c_count <= sr.c_stride && reinterpret_cast<uintptr_t>(sr.ptr) % 16 == 0,
               fn, "source: 16-byte aligned pointer, channel stride/offset/count multiples of 4");
    RP_REQUIRE(static_cast<long long>(d->B) * d->H * d->W * sr.c_stride < (1LL << 31), fn,
               "source tensor too large for 32-bit element offsets (B*H*W*c_stride must be < 2^31)");
    const Seg sg{sr.ptr, sr.c_stride, sr.c_offset, sr.c_count};
    (s == 0 ? p.seg0 : s == 1 ? p.seg1 : s == 2 ? p.seg2 : p.seg3) = sg;
    counts[s] = sr.c_count;
  }
  {
    unsigned char cs[MAX_CB];
    short c0[MAX_CB];
    p.ncb = fill_cb_tables(counts, d->n_src, cs, c0);
    RP_REQUIRE(p.ncb > 0, fn, "too many channel blocks");
    int st[5] = {0, p.ncb, p.ncb, p.ncb, p.ncb};
    for (int s = 1; s < d->n_src; ++s) st[s] = st[s - 1] + rp::cdiv(counts[s - 1], BK);
    p.cb1 = st[1]; p.cb2 = st[2]; p.cb3 = st[3];
  }
  const bool vertical = (d->kw == 1 && d->kh > 1);
  p.B = d->B;
  if (vertical) {           // column-major tiling: fast axis = y
    p.U = d->W; p.V = d->H; p.su = 1; p.sv = d->W;
    p.G = 1; p.T = d->kh; p.du0 = 0; p.dv0 = -(d->kh / 2);
  } else {
    p.U = d->H; p.V = d->W; p.su = d->W; p.sv = 1;
    p.G = d->kh; p.T = d->kw; p.du0 = -(d->kh / 2); p.dv0 = -(d->kw / 2);
  }
  const bool spatial = g_conv_spatial && d->kh == 3 && d->kw == 3 && d->stride == 1;      // 8 x 16 patches, nine taps on one staged tile
  if (spatial) { p.G = 1; p.T = 9; p.du0 = 0; p.dv0 = 0; }
  p.stride = 1; p.Uin = p.U; p.Vin = p.V; p.gkw = 0; p.dvg0 = 0;
  int Ho = d->H, Wo = d->W;
  if (d->stride == 2) {      // strided: every tap is its own group (no tap sharing), row-major tiling over the OUTPUT
    Ho = (d->H + 1) / 2; Wo = (d->W + 1) / 2;
    p.stride = 2; p.Uin = d->H; p.Vin = d->W; p.su = d->W; p.sv = 1;
    p.U = Ho; p.V = Wo;
    p.G = d->kh * d->kw; p.T = 1; p.gkw = d->kw; p.du0 = -(d->kh / 2); p.dvg0 = -(d->kw / 2); p.dv0 = 0;
  }
  RP_REQUIRE(spatial || (p.T / 2 <= HALO && (p.stride == 2 || p.T == 1 || p.T == 3 || p.T == 5 || p.T == 7)), fn, "kernel too wide for the staged halo");
  p.wpk = static_cast<const uint4*>(d->w_packed);
  p.Npad = rp::cdiv(d->c_out, BN) * BN;
  p.bias = d->bias; p.Cout = d->c_out;
  p.a_scale = d->a_scale;
  p.out_scale = 1.0f / (d->a_scale * d->w_scale);
  p.epi = d->epilogue;
  p.dst = d->dst; p.dst_cs = d->dst_c_stride; p.dst_co = d->dst_c_offset;
  p.aux0 = d->aux0; p.aux0_cs = d->aux0_c_stride; p.aux0_co = d->aux0_c_offset;
  p.aux1 = d->aux1; p.aux1_cs = d->aux1_c_stride; p.aux1_co = d->aux1_c_offset;
  p.dst2 = d->dst2; p.dst2_cs = d->dst2_c_stride; p.dst2_co = d->dst2_c_offset;
  p.gru_c = d->gru_c;
  p.tstats = d->tile_stats;
  p.in_mr = d->src0_mean_rstd;
  p.addm = d->add_map; p.addm_cs = d->add_c_stride; p.addm_co = d->add_c_offset;
  if (d->add_map) RP_REQUIRE(d->c_out % 4 == 0 && d->add_c_stride % 4 == 0 && d->add_c_offset % 4 == 0 &&
                                 reinterpret_cast<uintptr_t>(d->add_map) % 16 == 0, fn, "add_map: c_out % 4 == 0, 16-byte aligned, stride/offset multiples of 4");
  p.sat = d->src_bounded ? nullptr : rp::sat_counter();        // (split-form outputs of such a launch are not range-checked either)
  p.ksplit = 1; p.ks_ws = nullptr; p.ks_cnt = nullptr;
  // split-tensor sources / destinations (see the header): whole 8-channel groups, 32-byte aligned rows
  const bool hlin = d->src_hl != 0;
  if (hlin) {
    RP_REQUIRE(d->stride == 1 && !d->src0_mean_rstd && (p.T == 1 || p.T == 3 || p.T == 5 || spatial), fn,
               "split-tensor sources: stride 1, 1/3/5 taps per group, no fused normalisation");
    for (int s = 0; s < d->n_src; ++s)
      RP_REQUIRE(d->src[s].c_count % 8 == 0 && d->src[s].c_offset % 8 == 0 && d->src[s].c_stride % 8 == 0 &&
                     reinterpret_cast<uintptr_t>(d->src[s].ptr) % 32 == 0, fn,
                 "split-tensor source: channel stride/offset/count multiples of 8, 32-byte aligned");
  }
  p.dst_hl = d->dst_hl; p.dst2_hl = d->dst2_hl;
  p.dsth = d->dst_split; p.dsth_cs = d->dst_split_c_stride; p.dsth_co = d->dst_split_c_offset;
  if (d->dst_hl) RP_REQUIRE(d->dst_c_stride % 8 == 0 && reinterpret_cast<uintptr_t>(d->dst) % 32 == 0 && !d->tile_stats, fn,
                            "split-form dst: channel stride multiple of 8, 32-byte aligned, no tile statistics");
  if (d->dst2_hl) RP_REQUIRE(d->epilogue == 2 && d->dst2_c_stride % 8 == 0 && reinterpret_cast<uintptr_t>(d->dst2) % 32 == 0, fn,
                             "split-form dst2: GRU z|r epilogue, channel stride multiple of 8, 32-byte aligned");
  if (d->dst_split) RP_REQUIRE(d->epilogue != 2 && d->dst_split_c_stride % 8 == 0 && d->dst_split_c_offset % 4 == 0 &&
                                   reinterpret_cast<uintptr_t>(d->dst_split) % 32 == 0, fn,
                               "dst_split: not with the GRU z|r epilogue; channel stride multiple of 8, offset of 4, 32-byte aligned");
  RP_REQUIRE(d->tile >= 0 && d->tile <= 7, fn, "tile must be 0 (auto), 1 (128x64), 2 (128x128, 4 column waves), 3 (128x128, 2x2 waves), 4 (128x64, deep pipeline), 5 / 6 / 7 (strips of 160 / 32 / 96 rows)");
  if (d->tile_stats) RP_REQUIRE(d->epilogue == 0, fn, "tile_stats needs the linear epilogue");
  const long long Mtot = static_cast<long long>(d->B) * Ho * Wo;
  RP_REQUIRE(Mtot < (1LL << 31) - 256 && static_cast<long long>(d->B) * d->H * d->W < (1LL << 31) - 256, fn, "too many pixels");
  {
    // the packed array holds both orders (rnnpose_conv_packed_halfs): the strip kernels' copy sits behind the first one
    const long long first = static_cast<long long>(d->kh) * d->kw * p.ncb * p.Npad * BK * 2;
    p.wpk_strip = reinterpret_cast<const uint4*>(static_cast<const _Float16*>(d->w_packed) + first);
    p.single_product = d->single_product != 0;
  }
  // ---- strip kernels (conv_strip*.hip): 160- / 32-row strips, operands by LDS-DMA.  tile 5 / 6 ask for them; the automatic choice
  // takes them for the stride-1 3x3 / 1x5 / 5x1 layers whose launch fills the chip with strips of one of the heights (strip_rows; a
  // launch with per-image tile records tiles as rnnpose_conv_tiles_per_image_ex says for the same shape and batch)
  if (d->tile_stats) {     // the buffer is checked against the tiling of the kernel THIS launch takes (ABI 3)
    const int tpi = rnnpose_conv_tiles_per_image_desc(d);
    RP_REQUIRE(tpi > 0 && static_cast<long long>(d->tile_stats_records) == static_cast<long long>(d->B) * tpi, fn,
               "tile_stats_records must equal B * rnnpose_conv_tiles_per_image_desc(desc) (a consumer derives the tiling from the record count: "
               "an oversized buffer would make it sum records this launch never writes): size the statistics buffer with that call");
  }
  {
    const bool per_image = d->tile_stats || d->src0_mean_rstd;
    const int rows = desc_strip_rows(d);
    if (d->tile >= 5)
      RP_REQUIRE(rows != 0, fn, "strip kernels: stride 1, 3x3 / 1x5 / 5x1, c_out > 32 (32-row strips: one column tile per wave)");
    if (rows && d->stride == 2) {
      // stride-2 3x3 layer on strips (r05): the four parity planes of the source as four segments of a virtual concatenation -- strided
      // views (pixel stride 2, row stride 2 W) starting at pixels (0,0), (0,1), (1,0), (1,1) -- on the half-resolution output grid,
      // 2 x 2 taps per plane (conv_strip.hip: pack_strip_s2_kernel has the tap table).  p.U / p.V are the output extents already.
      const rnnpose_conv_src_t& sr = d->src[0];
      const int nb = sr.c_count / BK;
      if (d->kh == 3) {
        for (int k = 0; k < 4; ++k) {
          const Seg sg{sr.ptr + (static_cast<long long>(k >> 1) * d->W + (k & 1)) * sr.c_stride, sr.c_stride, sr.c_offset, sr.c_count};
          (k == 0 ? p.seg0 : k == 1 ? p.seg1 : k == 2 ? p.seg2 : p.seg3) = sg;
        }
        p.cb1 = nb; p.cb2 = 2 * nb; p.cb3 = 3 * nb; p.ncb = 4 * nb;
        p.wpk_strip = reinterpret_cast<const uint4*>(reinterpret_cast<const _Float16*>(p.wpk_strip) + static_cast<long long>(nb) * 2 * 9 * p.Npad * 32);   // behind the stride-1 strip copy
      }                        // (1x1: plane (0, 0) = segment 0 as it stands, one tap in four; its packed copy sits right behind the first one)
      p.su = 2 * d->W; p.sv = 2; p.Uin = d->H; p.Vin = d->W;
      p.G = 1; p.du0 = 0; p.dv0 = 0;
      return strip_launch(p, Ho, Wo, d->kh, d->kw, false, per_image, rows, rp::as_stream(stream));
    }
    if (rows) {          // (a forced strip launch whose sources do not fit surfaces its error in strip_launch)
      if (vertical) { p.G = 1; p.T = d->kh; p.du0 = 0; p.dv0 = -(d->kh / 2); }
      else { p.G = d->kh; p.T = d->kw; p.du0 = -(d->kh / 2); p.dv0 = -(d->kw / 2); }
      return strip_launch(p, d->H, d->W, d->kh, d->kw, hlin, per_image, rows, rp::as_stream(stream));
    }
  }
  p.n_mt = rp::cdiv(Mtot, BM);
  p.tpi = 0;
  if (d->tile_stats || d->src0_mean_rstd) {      // statistics / fused normalisation are per image: tile M per image
    p.tpi = rp::cdiv(static_cast<long long>(Ho) * Wo, BM);
    p.n_mt = d->B * p.tpi;
  }
  if (spatial) {                                 // patches never straddle images: always per-image tiling
    p.sp_tx = rp::cdiv(d->W, 16); p.sp_ty = rp::cdiv(d->H, 8);
    p.tpi = p.sp_tx * p.sp_ty;
    p.n_mt = d->B * p.tpi;
  }
  if (d->src0_mean_rstd)
    RP_REQUIRE(d->n_src == 1 && d->stride == 1 && d->kh == 3 && d->kw == 3 && reinterpret_cast<uintptr_t>(d->src0_mean_rstd) % 16 == 0,
               fn, "src0_mean_rstd (fused instance norm + ReLU of the input) needs one source, a 3x3 stride-1 kernel, 16-byte alignment");
  // tile width: 128 when Cout fills it and there are enough workgroups for 2 per CU, else 64.  d->tile overrides (measurement)
  bool wide = !d->src0_mean_rstd && (d->c_out % 128 == 0) && (static_cast<long long>(p.n_mt) * (d->c_out / 128) >= 512);
  bool wide22 = false;            // 128x128 as 2x2 waves of 64x64 (NI = 2): stride-1 split-source layers only
  if (d->tile == 1) wide = false;
  if (d->tile == 2) wide = !d->src0_mean_rstd;
  if (d->tile == 3 && hlin && !spatial) { wide = false; wide22 = true; }
  const bool deep = d->tile == 4 && hlin && !spatial && (p.T == 3 || p.T == 5);     // 128x64, weights of a whole block + two activation tiles in flight
  if (d->tile == 4) wide = false;
  const dim3 block(NT);
  hipStream_t st = rp::as_stream(stream);
  // K split of small launches (128x64 tiles, 2x2 waves, stride 1): only with a caller-provided workspace (two chains on two
  // streams must not share one), tiles * splits <= KS_MAX_WG, at least two channel-block iterations per split
  auto choose_ksplit = [&](int tiles, int nit) {
    // (measured per layer at B = 1, 30 x 30, profiles/r03_conv_ksplit_layers.txt: 8-24 tiles gain 15-35 %; 32-64 tiles -- GRU z|r,
    //  heads, the hoisted inp convolutions -- lose 5-30 % to the reduction pass: not split)
    if (!g_conv_ksplit || !d->ksplit_ws || tiles > g_ks_max_tiles || nit < 4) return 1;
    int ksn = g_ks_max_splits;
    if (ksn > nit / 2) ksn = nit / 2;
    if (ksn * tiles > KS_MAX_WG) ksn = KS_MAX_WG / tiles;
    if (ksn < 2) return 1;
    const size_t need = KS_CNT_BYTES + static_cast<size_t>(tiles) * ksn * (NT * 32) * sizeof(float);     // 2 x 1 MFMA tiles of 16 registers per lane
    if (d->ksplit_ws_bytes < need || reinterpret_cast<uintptr_t>(d->ksplit_ws) % 16 != 0) return 1;
    p.ksplit = ksn;
    p.ks_cnt = static_cast<unsigned*>(d->ksplit_ws);
    p.ks_ws = reinterpret_cast<float*>(static_cast<char*>(d->ksplit_ws) + KS_CNT_BYTES);
    return ksn;
  };
  // stride 1: the main loop unrolled over the taps of a group (T = kw, or kh for vertical kernels); stride 2: generic loop
#define RP_LAUNCH_T(NI_, COLS4_, HL_)                                                                                   \
  switch (p.T) {                                                                                                         \
    case 1: hipLaunchKernelGGL((conv_igemm_f16x3_kernel<NI_, false, COLS4_, 1, false, HL_>), grid, block, 0, st, p); break;    \
    case 3: hipLaunchKernelGGL((conv_igemm_f16x3_kernel<NI_, false, COLS4_, 3, false, HL_>), grid, block, 0, st, p); break;    \
    case 5: hipLaunchKernelGGL((conv_igemm_f16x3_kernel<NI_, false, COLS4_, 5, false, HL_>), grid, block, 0, st, p); break;    \
    default:                                                                                                             \
      if constexpr (!(HL_)) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<NI_, false, COLS4_, 7, false, false>), grid, block, 0, st, p); \
      break;                                                                                                             \
  }
  if (spatial) {                  // 3x3 stride 1: 8 x 16 patches, nine taps per staged tile (tile shapes 3 / 4 do not apply)
    // 128x64 as 2x2 waves unless the 4-column layout is asked for: at 2 workgroups per CU (58 KB of LDS) the 128x128 tile's
    // 600-odd workgroups need a second round (heads, half batch: 100 vs 80 us); at the full batch the two are equal
    if (d->tile == 2 && d->c_out % 128 == 0 && !d->src0_mean_rstd) {
      p.n_nt = p.Npad / 128;
      const dim3 grid(static_cast<unsigned>(p.n_mt) * p.n_nt);
      if (hlin) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, true, 9, false, true, false, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, true, 9, false, false, false, true>), grid, block, 0, st, p);
    } else {
      p.n_nt = rp::cdiv(d->c_out, 64);
      const int ksn = choose_ksplit(p.n_mt * p.n_nt, p.ncb);
      const dim3 grid(static_cast<unsigned>(p.n_mt) * p.n_nt * ksn);
      if (d->src0_mean_rstd) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, false, 9, true, false, false, true>), grid, block, 0, st, p);
      else if (hlin) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, false, 9, false, true, false, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, false, 9, false, false, false, true>), grid, block, 0, st, p);
    }
  } else if (deep) {
    p.n_nt = rp::cdiv(d->c_out, 64);
    const dim3 grid(static_cast<unsigned>(p.n_mt) * p.n_nt);
    if (p.T == 3) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, false, 3, false, true, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, false, 5, false, true, true>), grid, block, 0, st, p);
  } else if (wide22) {
    p.n_nt = p.Npad / 128;
    const dim3 grid(static_cast<unsigned>(p.n_mt) * p.n_nt);
    RP_LAUNCH_T(2, false, true)
  } else if (wide) {
    p.n_nt = p.Npad / 128;
    const dim3 grid(static_cast<unsigned>(p.n_mt) * p.n_nt);
    if (p.stride == 2) {
      hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, true, true>), grid, block, 0, st, p);
    } else if (hlin) {
      RP_LAUNCH_T(1, true, true)
    } else {
      RP_LAUNCH_T(1, true, false)
    }
  } else {
    p.n_nt = rp::cdiv(d->c_out, 64);
    const int ksn = p.stride == 2 ? 1 : choose_ksplit(p.n_mt * p.n_nt, p.G * p.ncb);
    const dim3 grid(static_cast<unsigned>(p.n_mt) * p.n_nt * ksn);
    if (p.stride == 2) {
      hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, true, false>), grid, block, 0, st, p);
    } else if (d->src0_mean_rstd) {
      hipLaunchKernelGGL((conv_igemm_f16x3_kernel<1, false, false, 3, true>), grid, block, 0, st, p);
    } else if (hlin) {
      RP_LAUNCH_T(1, false, true)
    } else {
      RP_LAUNCH_T(1, false, false)
    }
  }
#undef RP_LAUNCH_T
  return rp::check_launch(fn);
}

}  // extern "C"
