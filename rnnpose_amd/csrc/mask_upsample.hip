// a4 (last layer) + a6 in one kernel:  mask = 0.25 * mask.2(x)  (1x1 convolution 256 -> 576, thirdparty/raft/update.py:183-187)
// followed by the convex 8x up-sampling of the low-resolution flow (model/CFNet.py:95-106):
//     up[b, :, 8Y+i, 8X+j] = sum_k softmax_k(mask[b, k*64 + i*8 + j, Y, X]) * 8 * flow[b, :, Y + k/3 - 1, X + k%3 - 1]
// The 576-channel mask tensor (44 MB per half batch and iteration) never exists in HBM, and the activation tile is split
// into fp16 hi / lo ONCE for all 576 output columns: as a stand-alone 1x1 convolution the same layer re-split its input for
// each of its nine 64-column tiles and ran at 15 % of the matrix peak (r02: 42 us + 15 us for the up-sampling kernel).
//
// Workgroup = 32 consecutive low-resolution pixels x all 576 columns, 4 waves: wave ct owns the 16 sub-pixel columns
// [16 ct, 16 ct + 16) of EVERY tap for the 32 pixels (two 16x16 tiles of v_mfma_f32_16x16x32_f16): the softmax over the taps
// never leaves the lane.  32-pixel tiles because of the grid: 4 x 4800 pixels are 600 workgroups, all resident at once at 3
// per CU (168 VGPRs, 35 KB LDS); with 64-pixel tiles and one workgroup per CU the 300 workgroups needed two rounds on 256
// CUs and the kernel lasted two workgroup lifetimes.  LDS: the whole K = 256 activation tile as fp16
// hi / lo (32 KB, 16-byte chunks XOR-swizzled instead of padded) + the 3x3 flow neighbourhood of every pixel (4.6 KB).  A wave computes the logits of one tap at a time (16
// accumulators) and folds them into running softmax sums; weights arrive as MFMA B fragments straight from their own packed
// array (rnnpose_mask_upsample_pack_f16x3: [tap][channel block][column tile][hi, lo][lane] x 16 B), four stages ahead.
// Numerics: fp16x3 split as in csrc/conv_igemm.hip; online softmax (running max / denominator / weighted sums).
// History (r02, 4 x 4800 pixels): 1x1 convolution + up-sampling kernel 42 + 15 us; first fused version (nine 32x32 logit
// tiles per wave in registers, one weight stage ahead) 95 us; tap-outer loop + online softmax + 4-stage weight ring 51 us;
// 16x16x32 tiles with column-split waves 49 us; one exponential per element + the update interleaved with the next tap's
// MFMAs 44 us; 8 waves per workgroup 45 us; 32-pixel tiles (one round of workgroups instead of two) 35 us.
#include "common.hpp"
#include "f16x3.cuh"

namespace {

using rp::f32x16; using rp::h4; using rp::h8; using rp::split4;

constexpr int MQ = 32;          // pixels per workgroup (= threads / 8)
constexpr int KC = 256;         // input channels (mask.0 output)
constexpr int NCB = KC / 32;    // 32-channel blocks
constexpr int NTAP = 9;
#ifndef MU_STORE_WT
#define MU_STORE_WT 0
#endif
#ifndef MU_RELEASE
#define MU_RELEASE 0
#endif
#ifndef MU_TILE32
#define MU_TILE32 0         // 0 (default): the 16 x 16-tile kernel (r02-r05 geometry, on mfma16_c32 = the K = 16 shape since r06); 1: 32 x 32 tiles on
#endif                      // v_mfma_f32_32x32x16_f16, two waves per workgroup (r06: built, parity-tested, NOT faster -- 37.4 vs 36.2 us -- and it disturbs a
                            // packed-fp32 neighbour in 430 of 800 launches where the K = 16 form disturbs none: profiles/r06_pk_neighbour.txt).  The packed
                            // weight layout follows the choice (pack and launch are compiled together).
#ifndef MU_LINE_STORES
#define MU_LINE_STORES 1    // 0: the r02-r04 epilogue (4-byte stores straight from the softmax registers) for A/B measurements
#endif

struct MUParams {
  const float* x;               // (B,h,w,cs): channels [co, co + 256) = relu(mask.0(h))
  int cs, co;
  const uint4* wpk;             // packed mask.2 weights: record ((k * 8 + cb) * 4 + column tile) * 128 + part * 64 + lane  (part: hi, lo)
  const float* bias;            // (576), post-scale folded
  float a_scale, out_scale;
  const float* flow;            // (B,h,w,2) low-resolution flow
  float* up;                    // (B,2,8h,8w)
  int B, h, w;
  unsigned long long* sat;      // fp16x3 range guard counter (NULL = off)
};

// (launch bounds: three 4-wave workgroups per CU = 3 waves per SIMD: <= 168 registers; the compiler settles at 166, no spills)
__global__ __launch_bounds__(8 * MQ, 3) void mask_upsample_kernel(const MUParams p) {
  __shared__ __attribute__((aligned(16))) _Float16 sA[NCB][2][MQ * 32];     // [channel block][hi, lo][row * 32 + swizzled chunk * 8 + e]
  __shared__ float2 sF[MQ][NTAP];                                            // 8 * flow of the 3x3 neighbourhood (0 outside the map)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.h * p.w;
  const long long total = static_cast<long long>(p.B) * n;
  const long long m0 = static_cast<long long>(blockIdx.x) * MQ;

  // ---- stage the activation tile: thread -> rows (tid >> 3) + 32 * pass, channel quad tid & 7 of every 32-channel block.
  //      All 16 loads + the neighbourhood loads are issued before the first use (rows past the end re-read the last pixel).
  const int c4 = tid & 7;
  float4 av[NCB];
  {
    const long long m = m0 + (tid >> 3);
    const long long mc = m < total ? m : total - 1;
    const float* src = p.x + mc * p.cs + p.co + c4 * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) av[cb] = *reinterpret_cast<const float4*>(src + cb * 32);
  }
  float2 fq[2];
  bool fok[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {                     // 64 * 9 = 576 table entries, 2 per thread (the second pass is partial)
    const int e = tid + 8 * MQ * q;
    const int px = e / NTAP, k = e - px * NTAP;
    const long long m = m0 + px;
    const unsigned mc = static_cast<unsigned>((e < MQ * NTAP && m < total) ? m : (m0 < total ? m0 : total - 1));   // total < 2^31 (host check):
    const int b = static_cast<int>(mc / static_cast<unsigned>(n));                                                 // 32-bit divisions
    const int pix = static_cast<int>(mc - static_cast<unsigned>(b) * static_cast<unsigned>(n));
    const int Y = static_cast<int>(static_cast<unsigned>(pix) / static_cast<unsigned>(p.w)), X = pix - Y * p.w;
    const int yy = Y + k / 3 - 1, xx = X + k % 3 - 1;
    fok[q] = e < MQ * NTAP && m < total && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
    fq[q] = *reinterpret_cast<const float2*>(p.flow + (static_cast<long long>(b) * n + (fok[q] ? yy * p.w + xx : pix)) * 2);
  }
  __builtin_amdgcn_sched_barrier(0);
  int sat_n = 0;
  {
    const int r = tid >> 3;
    const int chunk = (c4 >> 1) ^ (((r >> 2) & 1) << 1);            // rows r and r + 4 would share LDS banks: swap chunk pairs
    const int off = r * 32 + chunk * 8 + (c4 & 1) * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      h4 hi, lo;
      split4(av[cb], p.a_scale, hi, lo);
      if (p.sat) sat_n += rp::quad_saturates(av[cb], p.a_scale) ? 1 : 0;
      *reinterpret_cast<h4*>(&sA[cb][0][off]) = hi;
      *reinterpret_cast<h4*>(&sA[cb][1][off]) = lo;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + 8 * MQ * q;
    if (e < MQ * NTAP) sF[e / NTAP][e % NTAP] = fok[q] ? make_float2(8.f * fq[q].x, 8.f * fq[q].y) : make_float2(0.f, 0.f);
  }
  if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));
  __syncthreads();

  // ---- main loop: tap k outermost, 8 channel blocks inside.  Stage (k, cb) = 12 MFMAs (4 pixel tiles x 3 split products)
  //      on 4 independent accumulators; its weight fragment (hi, lo) sits in ring slot cb & 3, which is re-armed with the
  //      stage 4 ahead.  The softmax over the taps is ONLINE: softmax(mask) . neighbours, rounded differently from the
  //      two-pass form (a few 1e-7 relative).
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int l15 = lane & 15, lq = lane >> 4;                          // A/B operand: row / column l15, 8 channels 8 lq ..
  const int ct = wave & 3, rh = wave >> 2;                           // column tile (16 sub-pixels) / 32-pixel block of this wave (MQ = 32: rh = 0)
  const int sub = 16 * ct + l15;
  uint4 bq[4][2];
#define MU_LOADB(SLOT_, K_, CB_)                                                                  \
  {                                                                                               \
    const uint4* rec_ = p.wpk + ((static_cast<long long>(K_) * NCB + (CB_)) * 4 + ct) * 128 + lane; \
    bq[SLOT_][0] = rec_[0]; bq[SLOT_][1] = rec_[64];                                              \
  }
  MU_LOADB(0, 0, 0) MU_LOADB(1, 0, 1) MU_LOADB(2, 0, 2) MU_LOADB(3, 0, 3)
  float rm[8], rden[8], rax[8], ray[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { rm[r] = -INFINITY; rden[r] = 0.f; rax[r] = 0.f; ray[r] = 0.f; }
  // One row of the online update.  Exactly one of the two exponentials of the textbook form is exp(0): with d = l - max,
  // either the maximum moves (d > 0: old sums are scaled by exp(-d), the new term enters with weight 1) or it stays (the new
  // term enters with exp(d)).  (-inf start: d = +inf, scale 0.)
#define MU_SOFT_ROW(ACC_, T_, E_, KP_, BKP_)                                                      \
  {                                                                                               \
    const int r_ = 4 * (T_) + (E_);                                                               \
    const float2 f_ = sF[32 * rh + 16 * (T_) + 4 * lq + (E_)][KP_];                               \
    const float l_ = ACC_[T_][E_] * p.out_scale + (BKP_);                                         \
    const float d_ = l_ - rm[r_];                                                                 \
    const float t_ = expf(-fabsf(d_));       /* (a 6-instruction exp2-based form was measured: no faster) */ \
    const bool up_ = d_ > 0.f;                                                                    \
    const float sc_ = up_ ? t_ : 1.f, ex_ = up_ ? 1.f : t_;                                       \
    rden[r_] = rden[r_] * sc_ + ex_;                                                              \
    rax[r_] = rax[r_] * sc_ + ex_ * f_.x;                                                         \
    ray[r_] = ray[r_] * sc_ + ex_ * f_.y;                                                         \
    rm[r_] = up_ ? l_ : rm[r_];                                                                   \
  }
  // The 8 stages of tap K_ into ACC_; between the MFMA groups of stage cb, rows 2 cb and 2 cb + 1 of the PREVIOUS tap's
  // logits (PACC_, tap KP_) are folded into the running sums: vector-ALU work under the matrix pipe instead of after it.
#define MU_TAP(ACC_, K_, HAVE_PREV_, PACC_, KP_, BKP_)                                            \
  {                                                                                               \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) ACC_[t] = f32x4{0.f, 0.f, 0.f, 0.f};           \
    int aoff = (32 * rh + l15) * 32;   /* opaque per tap: hoisted out of the tap loop the fragments would be 128 registers */ \
    asm volatile("" : "+v"(aoff));                                                                \
    _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) {                                          \
      h8 ah[2], al[2];                                                                            \
      const int phys = lq ^ (((l15 >> 2) & 1) << 1);                                              \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                             \
        ah[t] = *reinterpret_cast<const h8*>(&sA[cb][0][aoff + 16 * t * 32 + phys * 8]);         \
        al[t] = *reinterpret_cast<const h8*>(&sA[cb][1][aoff + 16 * t * 32 + phys * 8]);         \
      }                                                                                           \
      const int sl = cb & 3;                                                                      \
      const h8 bh = __builtin_bit_cast(h8, bq[sl][0]), bl = __builtin_bit_cast(h8, bq[sl][1]);    \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) ACC_[t] = rp::mfma16_c32(al[t], bh, ACC_[t]); \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) ACC_[t] = rp::mfma16_c32(ah[t], bl, ACC_[t]); \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) ACC_[t] = rp::mfma16_c32(ah[t], bh, ACC_[t]); \
      /* re-arm the slot with the stage 4 ahead: (k, cb + 4) or (k + 1, cb - 4); past the end: the last tap again */ \
      const int nk = cb < 4 ? (K_) : ((K_) + 1 < NTAP ? (K_) + 1 : NTAP - 1);                     \
      MU_LOADB(sl, nk, (cb + 4) & 7)                                                              \
      if (HAVE_PREV_) MU_SOFT_ROW(PACC_, cb >> 2, cb & 3, KP_, BKP_)                              \
    }                                                                                             \
  }
  f32x4 accA[2], accB[2];
  float bkA = p.bias[sub], bkB = 0.f;
  MU_TAP(accA, 0, false, accB, 0, 0.f)
  for (int k = 1; k < NTAP; k += 2) {              // taps (1, 2), (3, 4), (5, 6), (7, 8)
    bkB = p.bias[k * 64 + sub];
    MU_TAP(accB, k, true, accA, k - 1, bkA)
    bkA = p.bias[(k + 1) * 64 + sub];
    MU_TAP(accA, k + 1, true, accB, k, bkB)
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) MU_SOFT_ROW(accA, t, e, NTAP - 1, bkA)
#undef MU_TAP
#undef MU_SOFT_ROW
#undef MU_LOADB
  const int si = sub >> 3, sj = sub & 7;
  const long long Wf = 8LL * p.w, Pf = 64LL * n;
#if MU_LINE_STORES
  // Epilogue (r05): the 32 x 64 x 2 up-sampled values of the tile go through LDS (the activation tile's space) and leave as WHOLE
  // 128-byte lines: one wave instruction = one (plane, sub-row) run of 32 pixels x 8 sub-columns = 1 KB of an output row in 16-byte
  // pieces.  Before, a line was completed piecewise by four 4-byte store instructions of one wave (16 dword stores per lane; now 4
  // dwordx4) -- the one producer of the loop that handed partially written lines to the memory system.
  // LDS layout [plane][sub-row si: 336][4-pixel group: 40][pixel & 3: 8][sj] floats: conflict-free for the column-major writes.
  constexpr int SI_ST = 336, G_ST = 40, PL_ST = 8 * SI_ST;
  float* sO = reinterpret_cast<float*>(&sA[0][0][0]);
  static_assert(2 * PL_ST * sizeof(float) <= sizeof(sA), "output staging must fit the activation tile");
  __syncthreads();                                   // every wave is done reading the activation tile
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = 32 * rh + 16 * (r >> 2) + 4 * lq + (r & 3);
    const int a = si * SI_ST + (row >> 2) * G_ST + (row & 3) * 8 + sj;
    sO[a] = rax[r] / rden[r];
    sO[PL_ST + a] = ray[r] / rden[r];
  }
  __syncthreads();
  {
    const int piece = tid & 63, px = piece >> 1, half = piece & 1;
    const long long m = m0 + px;
    if (m < total) {
      const unsigned mu = static_cast<unsigned>(m);
      const int b = static_cast<int>(mu / static_cast<unsigned>(n)), pix = static_cast<int>(mu - static_cast<unsigned>(b) * static_cast<unsigned>(n));
      const int Y = static_cast<int>(static_cast<unsigned>(pix) / static_cast<unsigned>(p.w)), X = pix - Y * p.w;
      float* dst = p.up + static_cast<long long>(b) * 2 * Pf + 8LL * Y * Wf + 8 * X + 4 * half;
      const float* src = sO + (px >> 2) * G_ST + (px & 3) * 8 + 4 * half;
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {                  // run = wave + 4 q: plane run >> 3, sub-row run & 7
        const int run = wave + 4 * q;
        v[q] = *reinterpret_cast<const float4*>(src + (run >> 3) * PL_ST + (run & 7) * SI_ST);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int run = wave + 4 * q;
#if MU_STORE_WT     // diagnostics build (tools/det_variants.sh): write-through stores -- the lines never sit dirty in this XCD's L2
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v vv = {v[q].x, v[q].y, v[q].z, v[q].w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + (run >> 3) * Pf + (run & 7) * Wf), "v"(vv) : "memory");
#else
        *reinterpret_cast<float4*>(dst + (run >> 3) * Pf + (run & 7) * Wf) = v[q];
#endif
      }
    }
  }
#if MU_STORE_WT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if MU_RELEASE      // diagnostics build: agent-scope release (buffer_wbl2 sc1) by one lane per workgroup behind a barrier
  __syncthreads();
  if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
#else
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = 32 * rh + 16 * (r >> 2) + 4 * lq + (r & 3);
    const long long m = m0 + row;
    if (m < total) {
      const unsigned mu = static_cast<unsigned>(m);
      const int b = static_cast<int>(mu / static_cast<unsigned>(n)), pix = static_cast<int>(mu - static_cast<unsigned>(b) * static_cast<unsigned>(n));
      const int Y = static_cast<int>(static_cast<unsigned>(pix) / static_cast<unsigned>(p.w)), X = pix - Y * p.w;
      const long long o = (8LL * Y + si) * Wf + 8 * X + sj;
      p.up[(static_cast<long long>(b) * 2 + 0) * Pf + o] = rax[r] / rden[r];
      p.up[(static_cast<long long>(b) * 2 + 1) * Pf + o] = ray[r] / rden[r];
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// r06: the same layer on v_mfma_f32_32x32x16_f16 (-DMU_TILE32=1; NOT the default, see the macro).  The idea: the 16 x 16 tiles above were built on
// v_mfma_f32_16x16x32_f16, which on MI355X corrupts packed-fp32 arithmetic of OTHER waves on the SIMD (csrc/f16x3.cuh); the
// safe 16 x 16 shape (K = 16) has half the rate (35.7 vs 30.5 us per half-batch launch); 32x32x16 has the full rate and a bare loop of it is harmless.
// Measured: this kernel is no faster than the K = 16 form (the layer is not matrix-bound) and, fed from LDS with real operands, it disturbs the
// neighbour too -- the defect follows the load on the matrix pipe, not the opcode.
// Workgroup = 32 pixels x all 576 columns as before, but TWO waves: wave cw owns the 32 sub-pixel columns [32 cw, 32 cw + 32) of every tap
// for all 32 pixels -- one 32 x 32 tile per tap, the softmax over the taps still never leaves the lane (16 pixels per lane: rows
// (v & 3) + 8 (v >> 2) + 4 (lane >> 5) of the C layout).  Half the waves of the 16 x 16 form issue the same flops at the K = 32 rate: the
// chip's matrix time per launch halves against the K = 16 form, and 600 x 2 waves fit the chip at two per SIMD (<= 256 registers).
// Per tap: 16 k-steps of 16 channels x 3 products; row v of the PREVIOUS tap's logits is folded into the running softmax sums between
// the MFMAs of k-step v (16 rows, 16 steps).  Weights: [tap][k-step][cw][hi, lo][lane] x 16 B, one record pair per step, 4-slot ring.
// LDS: the activation tile as before ([32-channel block][hi, lo][row][4 chunks of 8 channels]) with a 2-bit XOR swizzle (chunk ^ ((row >> 2)
// & 3)): the 16 lanes of a ds_read_b128 group (rows {0-3, 12-15, 20-27}) then hit 16 different 16-byte units.
__global__ __launch_bounds__(4 * MQ, 2) void mask_upsample32_kernel(const MUParams p) {
  __shared__ __attribute__((aligned(16))) _Float16 sA[NCB][2][MQ * 32];
  __shared__ float2 sF[MQ][NTAP];
  constexpr int NTH = 4 * MQ;                                          // 128 threads
  const int tid = threadIdx.x, lane = tid & 63, cw = tid >> 6;
  const int n = p.h * p.w;
  const long long total = static_cast<long long>(p.B) * n;
  const long long m0 = static_cast<long long>(blockIdx.x) * MQ;

  // ---- stage the activation tile: thread -> rows (tid >> 3) + 16 * pass, channel quad tid & 7 of every 32-channel block
  const int c4 = tid & 7;
  float4 av[2][NCB];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const long long m = m0 + (tid >> 3) + 16 * ps;
    const long long mc = m < total ? m : total - 1;
    const float* src = p.x + mc * p.cs + p.co + c4 * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) av[ps][cb] = *reinterpret_cast<const float4*>(src + cb * 32);
  }
  float2 fq[3];
  bool fok[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {                     // 32 * 9 = 288 table entries over 128 threads (the third pass is partial)
    const int e = tid + NTH * q;
    const int px = e / NTAP, k = e - px * NTAP;
    const long long m = m0 + px;
    const unsigned mc = static_cast<unsigned>((e < MQ * NTAP && m < total) ? m : (m0 < total ? m0 : total - 1));
    const int b = static_cast<int>(mc / static_cast<unsigned>(n));
    const int pix = static_cast<int>(mc - static_cast<unsigned>(b) * static_cast<unsigned>(n));
    const int Y = static_cast<int>(static_cast<unsigned>(pix) / static_cast<unsigned>(p.w)), X = pix - Y * p.w;
    const int yy = Y + k / 3 - 1, xx = X + k % 3 - 1;
    fok[q] = e < MQ * NTAP && m < total && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
    fq[q] = *reinterpret_cast<const float2*>(p.flow + (static_cast<long long>(b) * n + (fok[q] ? yy * p.w + xx : pix)) * 2);
  }
  __builtin_amdgcn_sched_barrier(0);
  int sat_n = 0;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = (tid >> 3) + 16 * ps;
    const int chunk = (c4 >> 1) ^ ((r >> 2) & 3);
    const int off = r * 32 + chunk * 8 + (c4 & 1) * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      h4 hi, lo;
      split4(av[ps][cb], p.a_scale, hi, lo);
      if (p.sat) sat_n += rp::quad_saturates(av[ps][cb], p.a_scale) ? 1 : 0;
      *reinterpret_cast<h4*>(&sA[cb][0][off]) = hi;
      *reinterpret_cast<h4*>(&sA[cb][1][off]) = lo;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int e = tid + NTH * q;
    if (e < MQ * NTAP) sF[e / NTAP][e % NTAP] = fok[q] ? make_float2(8.f * fq[q].x, 8.f * fq[q].y) : make_float2(0.f, 0.f);
  }
  if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  const int sub = 32 * cw + l31;                                        // sub-pixel column of this lane (i = sub >> 3, j = sub & 7)
  constexpr int NKS = KC / 16;                                          // 16 k-steps per tap
  uint4 bq[4][2];
#define MU_LOADB(SLOT_, K_, KS_)                                                                        \
  {                                                                                                     \
    const uint4* rec_ = p.wpk + ((static_cast<long long>(K_) * NKS + (KS_)) * 2 + cw) * 128 + lane;     \
    bq[SLOT_][0] = rec_[0]; bq[SLOT_][1] = rec_[64];                                                    \
  }
  MU_LOADB(0, 0, 0) MU_LOADB(1, 0, 1) MU_LOADB(2, 0, 2) MU_LOADB(3, 0, 3)
  float rm[16], rden[16], rax[16], ray[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { rm[r] = -INFINITY; rden[r] = 0.f; rax[r] = 0.f; ray[r] = 0.f; }
#define MU_SOFT_ROW(ACC_, V_, KP_, BKP_)                                                                \
  {                                                                                                     \
    const float2 f_ = sF[((V_) & 3) + 8 * ((V_) >> 2) + 4 * lh][KP_];                                   \
    const float l_ = ACC_[V_] * p.out_scale + (BKP_);                                                   \
    const float d_ = l_ - rm[V_];                                                                       \
    const float t_ = expf(-fabsf(d_));                                                                  \
    const bool up_ = d_ > 0.f;                                                                          \
    const float sc_ = up_ ? t_ : 1.f, ex_ = up_ ? 1.f : t_;                                             \
    rden[V_] = rden[V_] * sc_ + ex_;                                                                    \
    rax[V_] = rax[V_] * sc_ + ex_ * f_.x;                                                               \
    ray[V_] = ray[V_] * sc_ + ex_ * f_.y;                                                               \
    rm[V_] = up_ ? l_ : rm[V_];                                                                         \
  }
#define MU_TAP(ACC_, K_, HAVE_PREV_, PACC_, KP_, BKP_)                                                  \
  {                                                                                                     \
    _Pragma("unroll") for (int v = 0; v < 16; ++v) ACC_[v] = 0.f;                                       \
    int aoff = l31 * 32;   /* opaque per tap: hoisted out of the tap loop the fragments of all steps would be live at once */ \
    asm volatile("" : "+v"(aoff));                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {                                                \
      const int phys = ((2 * (ks & 1) + lh) ^ ((l31 >> 2) & 3)) * 8;                                    \
      const h8 ah = *reinterpret_cast<const h8*>(&sA[ks >> 1][0][aoff + phys]);                         \
      const h8 al = *reinterpret_cast<const h8*>(&sA[ks >> 1][1][aoff + phys]);                         \
      const int sl = ks & 3;                                                                            \
      const h8 bh = __builtin_bit_cast(h8, bq[sl][0]), bl = __builtin_bit_cast(h8, bq[sl][1]);          \
      ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ACC_, 0, 0, 0);                             \
      ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ACC_, 0, 0, 0);                             \
      ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, ACC_, 0, 0, 0);                             \
      /* re-arm the slot with the step 4 ahead: (k, ks + 4) or (k + 1, ks - 12); past the end: the last tap again */ \
      const int nk = ks < NKS - 4 ? (K_) : ((K_) + 1 < NTAP ? (K_) + 1 : NTAP - 1);                     \
      MU_LOADB(sl, nk, (ks + 4) & (NKS - 1))                                                            \
      if (HAVE_PREV_) MU_SOFT_ROW(PACC_, ks, KP_, BKP_)                                                 \
    }                                                                                                   \
  }
  f32x16 accA, accB;
  float bkA = p.bias[sub], bkB = 0.f;
  MU_TAP(accA, 0, false, accB, 0, 0.f)
  for (int k = 1; k < NTAP; k += 2) {              // taps (1, 2), (3, 4), (5, 6), (7, 8)
    bkB = p.bias[k * 64 + sub];
    MU_TAP(accB, k, true, accA, k - 1, bkA)
    bkA = p.bias[(k + 1) * 64 + sub];
    MU_TAP(accA, k + 1, true, accB, k, bkB)
  }
#pragma unroll
  for (int v = 0; v < 16; ++v) MU_SOFT_ROW(accA, v, NTAP - 1, bkA)
#undef MU_TAP
#undef MU_SOFT_ROW
#undef MU_LOADB
  // ---- epilogue: as above (whole 128-byte lines through the activation tile's LDS), two waves: 16 (plane, sub-row) runs = 8 per wave
  const int si = sub >> 3, sj = sub & 7;
  const long long Wf = 8LL * p.w, Pf = 64LL * n;
  constexpr int SI_ST = 336, G_ST = 40, PL_ST = 8 * SI_ST;
  float* sO = reinterpret_cast<float*>(&sA[0][0][0]);
  static_assert(2 * PL_ST * sizeof(float) <= sizeof(sA), "output staging must fit the activation tile");
  __syncthreads();                                   // every wave is done reading the activation tile
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int row = (v & 3) + 8 * (v >> 2) + 4 * lh;
    const int a = si * SI_ST + (row >> 2) * G_ST + (row & 3) * 8 + sj;
    sO[a] = rax[v] / rden[v];
    sO[PL_ST + a] = ray[v] / rden[v];
  }
  __syncthreads();
  {
    const int piece = tid & 63, px = piece >> 1, half = piece & 1;
    const long long m = m0 + px;
    if (m < total) {
      const unsigned mu = static_cast<unsigned>(m);
      const int b = static_cast<int>(mu / static_cast<unsigned>(n)), pix = static_cast<int>(mu - static_cast<unsigned>(b) * static_cast<unsigned>(n));
      const int Y = static_cast<int>(static_cast<unsigned>(pix) / static_cast<unsigned>(p.w)), X = pix - Y * p.w;
      float* dst = p.up + static_cast<long long>(b) * 2 * Pf + 8LL * Y * Wf + 8 * X + 4 * half;
      const float* src = sO + (px >> 2) * G_ST + (px & 3) * 8 + 4 * half;
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {                  // run = cw + 2 q: plane run >> 3, sub-row run & 7
        const int run = cw + 2 * q;
        v[q] = *reinterpret_cast<const float4*>(src + (run >> 3) * PL_ST + (run & 7) * SI_ST);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int run = cw + 2 * q;
        *reinterpret_cast<float4*>(dst + (run >> 3) * Pf + (run & 7) * Wf) = v[q];
      }
    }
  }
}

// MU_TILE32 weights: [tap k][k-step ks][column half cw][hi, lo][lane][8] fp16: lane l of the record carries column k*64 + 32*cw + (l & 31) and the 8
// channels ks*16 + 8*(l >> 5) + j -- one B operand of v_mfma_f32_32x32x16_f16.
__global__ void mask_pack32_kernel(const float* __restrict__ w, float scale, float post, _Float16* __restrict__ pk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;                  // 576 * 256 * 2 halfs
  if (i >= 576 * KC * 2) return;
  const int j = i & 7, ln = (i >> 3) & 63, part = (i >> 9) & 1, cw = (i >> 10) & 1, ks = (i >> 11) & 15, k = i >> 15;
  const int col = k * 64 + 32 * cw + (ln & 31), ch = ks * 16 + 8 * (ln >> 5) + j;
  const float v = w[col * KC + ch] * post * scale;
  const _Float16 h = static_cast<_Float16>(v);
  pk[i] = part == 0 ? h : static_cast<_Float16>(v - static_cast<float>(h));
}

// fp32 (576, 256) weights -> [tap k][channel block][column tile ct][hi, lo][lane][8] fp16: lane l of the record carries column
// k*64 + 16*ct + (l & 15) and the 8 channels cb*32 + 8*(l >> 4) + j -- one B operand of v_mfma_f32_16x16x32_f16.
__global__ void mask_pack_kernel(const float* __restrict__ w, float scale, float post, _Float16* __restrict__ pk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;                  // 576 * 256 * 2 halfs
  if (i >= 576 * KC * 2) return;
  const int j = i & 7, ln = (i >> 3) & 63, part = (i >> 9) & 1, ct = (i >> 10) & 3, cb = (i >> 12) & 7, k = i >> 15;
  const int col = k * 64 + 16 * ct + (ln & 15), ch = cb * 32 + 8 * (ln >> 4) + j;
  const float v = w[col * KC + ch] * post * scale;
  const _Float16 h = static_cast<_Float16>(v);
  pk[i] = part == 0 ? h : static_cast<_Float16>(v - static_cast<float>(h));
}

}  // namespace

extern "C" size_t rnnpose_mask_upsample_packed_bytes(void) { return static_cast<size_t>(576) * KC * 2 * sizeof(_Float16); }

extern "C" int rnnpose_mask_upsample_pack_f16x3(const float* weight, float post_scale, float w_scale, void* packed,
                                                rnnpose_stream_t stream) {
  const char* fn = "rnnpose_mask_upsample_pack_f16x3";
  RP_REQUIRE(weight && packed && reinterpret_cast<uintptr_t>(packed) % 16 == 0, fn, "null or misaligned pointer");
  RP_REQUIRE(w_scale > 0.f, fn, "w_scale must be positive");
  const int total = 576 * KC * 2;
#if MU_TILE32
  hipLaunchKernelGGL(mask_pack32_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, rp::as_stream(stream), weight, w_scale, post_scale,
                     static_cast<_Float16*>(packed));
#else
  hipLaunchKernelGGL(mask_pack_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, rp::as_stream(stream), weight, w_scale, post_scale,
                     static_cast<_Float16*>(packed));
#endif
  return rp::check_launch(fn);
}

extern "C" int rnnpose_mask_upsample_f16x3(const float* x, int x_c_stride, int x_c_offset, const void* w_packed, int c_out,
                                           const float* bias, float a_scale, float w_scale, const float* flow_lr, int B, int h,
                                           int w, float* flow_up, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_mask_upsample_f16x3";
  RP_REQUIRE(x && w_packed && bias && flow_lr && flow_up, fn, "null pointer");
  RP_REQUIRE(c_out == 64 * NTAP, fn, "c_out must be 576 (9 taps x 8 x 8 sub-pixels)");
  RP_REQUIRE(B > 0 && B < 65536 && h > 0 && w > 0 && h < 65536 && static_cast<long long>(h) * w < (1LL << 24) &&
                 static_cast<long long>(B) * h * w < (1LL << 31), fn, "bad size");
  RP_REQUIRE(x_c_offset >= 0 && x_c_offset % 4 == 0 && x_c_stride % 4 == 0 && x_c_offset + KC <= x_c_stride, fn,
             "the 256 input channels must lie inside the row, 16-byte aligned");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w_packed) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(flow_lr) % 8 == 0 && reinterpret_cast<uintptr_t>(flow_up) % 16 == 0, fn,
             "x / weights / flow_up must be 16-byte, flow 8-byte aligned");
  RP_REQUIRE(a_scale > 0.f && w_scale > 0.f, fn, "scales must be positive");
  MUParams p{};
  p.x = x; p.cs = x_c_stride; p.co = x_c_offset;
  p.wpk = static_cast<const uint4*>(w_packed);
  p.bias = bias;
  p.a_scale = a_scale; p.out_scale = 1.0f / (a_scale * w_scale);
  p.flow = flow_lr; p.up = flow_up;
  p.B = B; p.h = h; p.w = w;
  p.sat = rp::sat_counter();
  const long long total = static_cast<long long>(B) * h * w;
#if MU_TILE32
  hipLaunchKernelGGL(mask_upsample32_kernel, dim3(static_cast<unsigned>(rp::cdiv(total, MQ))), dim3(4 * MQ), 0, rp::as_stream(stream), p);
#else
  hipLaunchKernelGGL(mask_upsample_kernel, dim3(static_cast<unsigned>(rp::cdiv(total, MQ))), dim3(8 * MQ), 0, rp::as_stream(stream), p);
#endif
  return rp::check_launch(fn);
}
