// a3 + a4 (r06): the window lookup and BasicMotionEncoder.convc1 as ONE kernel --
//     corr = CorrBlock.__call__(coords)      thirdparty/raft/corr.py:36-57 (bilinear_sampler: thirdparty/raft/utils/utils.py:57-71)
//     cor  = relu(convc1(corr))              thirdparty/raft/update.py:80,87          (chained by model/CFNet.py:147-152)
// The two-kernel path (csrc/corr_lookup.hip -> csrc/conv1x1_resident.hip) writes the (B,h,w,324) window features to HBM and reads
// them back (25 MB each way per half-batch launch) across a kernel boundary.  Here a workgroup owns 32 consecutive pixels and ALL 256
// output columns, like the resident 1x1 kernel, but its activation tile is COMPUTED, level by level, and never complete in LDS:
//
//   for level l = 0..3:   (a) the four waves fetch the 10 x 10 footprints of 8 pixels each (cooperative loads as in corr_lookup.hip,
//                             all 16 loads of a lane in flight at once) into a 13-KB LDS buffer;                       barrier
//                         (b) 256 threads = 32 pixels x 8 groups of window rows form the level's 81 bilinear taps in fp32 and split
//                             them (fp16 hi | lo, csrc/f16x3.cuh) into a RING of four 32-channel blocks (16 KB);           barrier
//                         (c) the channel blocks the level COMPLETED (0-1 | 2-4 | 5-6 | 7-10: 81 channels do not end on block
//                             boundaries, the partial block stays in its ring slot) go through v_mfma_f32_16x16x32_f16 with the
//                             weights as B fragments straight from convc1's packed array, two blocks ahead;               barrier
//
// 29 KB of LDS per workgroup (33 KB with the epilogue's staging tile aliasing it) instead of 71 KB for "whole tile resident + footprints":
// FOUR workgroups per CU, so a half-batch launch (600 workgroups) is one round and one workgroup's footprint latency runs under the
// others' MFMAs.  (First form of this fusion, r06: the LOOKUP staging phase in front of the unchanged resident kernel -- 71 KB, two
// workgroups per CU, two rounds: 44.5 us against 23.2 + 18.7 for the two kernels; profiles/r06_lookup_convc1_fusion.txt.)
// Numerics: the operations of the two-kernel path per element (fp32 taps in the same order, one split, the same MFMA sequence per block).
#include "common.hpp"
#include "f16x3.cuh"
#include "corr_lookup.cuh"

namespace {

using namespace rplookup;
using rp::h8;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MQ = 32;          // pixels per workgroup
constexpr int CT = 4;           // 16-column tiles per wave: Cout = 4 waves x 4 x 16 = 256
constexpr int NCB = 11;         // 4 x 81 = 324 channels = 11 blocks of 32 (block 10: channels 320 .. 323, then zeros)
constexpr int NSLOT = 4;        // ring of channel blocks: [hi plane 32 x 32 | lo plane] fp16 = 4 KB each
constexpr int GP = 8;           // pixels per wave in the footprint fetch
constexpr int SLOT_HALFS = 2 * MQ * 32;
constexpr int RING_BYTES = NSLOT * SLOT_HALFS * 2;
constexpr int FOOT_BYTES = MQ * FS * 4;
constexpr int RSF = 260;                                            // epilogue staging row stride in floats (256 + 4: conflict-free)
constexpr int EPI_BYTES = MQ * RSF * 4;
constexpr int LDS_BYTES = RING_BYTES + FOOT_BYTES > EPI_BYTES ? RING_BYTES + FOOT_BYTES : EPI_BYTES;

struct CParams {
  const float* pyr;             // the whole pyramid buffer (csrc/corr_pyramid.hip layout)
  const float* coords;          // (images of this launch, 2, h, w): window centres at level-0 scale
  LookupInfo info;
  int N;                        // h * w
  long long p_off;              // pyramid row of pixel 0 of this launch (= first image x N)
  const uint4* wpk;             // convc1 packed by rnnpose_conv1x1_resident_pack_f16x3 (c_in = 324): record (cb * 16 + column tile) * 128 + part * 64 + lane
  const float* bias;
  float a_scale, out_scale;
  float* dst;                   // (M, dcs): channels [dco, dco + 256)
  int dcs, dco, relu;
  long long M;
  unsigned long long* sat;
  int dst_hl;
};

// x * s = hi + lo for ONE value: the arithmetic of rp::split4 (round to nearest, clamped at +-65504)
__device__ __forceinline__ void split1(float v, float s, _Float16& hi, _Float16& lo) {
  const float x = v * s;
  const _Float16 cap = static_cast<_Float16>(65504.f);
  _Float16 h = static_cast<_Float16>(x);
  h = h > cap ? cap : (h < -cap ? -cap : h);
  _Float16 l = static_cast<_Float16>(x - static_cast<float>(h));
  l = l > cap ? cap : (l < -cap ? -cap : l);
  hi = h;
  lo = l;
}

__global__ __launch_bounds__(8 * MQ, 3) void corr_convc1_kernel(const CParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  _Float16* const ring = reinterpret_cast<_Float16*>(lds);            // slot s: [hi 32 x 32 | lo 32 x 32], row * 32 + swizzled chunk * 8 + e
  float* const foot = reinterpret_cast<float*>(lds + RING_BYTES);     // [pixel of the tile][FS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long m0 = static_cast<long long>(blockIdx.x) * MQ;
  const int N = p.N;

  // ---- per-thread pixels: the one whose footprint base this lane carries in the fetch (pixel GP * wave + (lane & 7) of the tile) and
  //      the one whose taps it forms (pixel tid & 31); window centres at level-0 scale
  const long long gfirst = m0 + GP * wave;
  const long long gleft = p.M - gfirst;
  const int gnpix = gleft >= GP ? GP : (gleft > 0 ? static_cast<int>(gleft) : 1);
  const long long gfrow = p.p_off + (gleft > 0 ? gfirst : p.M - 1);
  float gcx, gcy, tcx, tcy;
  int bg, pixg;
  {
    const long long pp = gfirst + (lane & (GP - 1));
    const long long pc = pp < p.M ? pp : p.M - 1;
    const int b = static_cast<int>(pc / N), pix = static_cast<int>(pc - static_cast<long long>(b) * N);
    gcx = p.coords[(static_cast<long long>(b) * 2 + 0) * N + pix];
    gcy = p.coords[(static_cast<long long>(b) * 2 + 1) * N + pix];
    const long long prow = p.p_off + pc;
    bg = static_cast<int>(prow / N);
    pixg = static_cast<int>(prow - static_cast<long long>(bg) * N);
    const long long tp = m0 + (tid & 31);
    const long long tc = tp < p.M ? tp : p.M - 1;
    const int tb = static_cast<int>(tc / N), tpix = static_cast<int>(tc - static_cast<long long>(tb) * N);
    tcx = p.coords[(static_cast<long long>(tb) * 2 + 0) * N + tpix];
    tcy = p.coords[(static_cast<long long>(tb) * 2 + 1) * N + tpix];
  }
  const int r = tid & 31, g = tid >> 5;                       // tap phase: tile row, group of window rows: g = 0: j = 0, 1; g >= 1: j = g + 1
  const int j0 = g == 0 ? 0 : g + 1, nj = g == 0 ? 2 : 1;
  const int swz = ((r >> 2) & 1) << 1;
  int sat_n = 0;

  // ---- MFMA side (csrc/conv1x1_resident.hip): 4 column tiles x 2 pixel tiles x 3 split products = 24 MFMAs per channel block
  const int l15 = lane & 15, lq = lane >> 4;
  const int phys = lq ^ (((l15 >> 2) & 1) << 1);
  f32x4 acc[2][CT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 bq[2][CT][2];                                                 // ring of 2 stages x column tile x (hi, lo)
#define C_LOADB(SLOT_, CB_)                                                                       \
  {                                                                                               \
    const uint4* rec_ = p.wpk + (static_cast<long long>(CB_) * 16 + wave * CT) * 128 + lane;      \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) { bq[SLOT_][j][0] = rec_[j * 128]; bq[SLOT_][j][1] = rec_[j * 128 + 64]; } \
  }
  C_LOADB(0, 0)
  C_LOADB(1, 1)
  __builtin_amdgcn_sched_barrier(0);

  // one pyramid level: fetch -> taps -> the channel blocks [CB0_, CB1_) it completes
#define C_LEVEL(L_, CB0_, CB1_)                                                                                           \
  {                                                                                                                       \
    constexpr int lvl_ = (L_);                                                                                            \
    constexpr float inv_ = 1.0f / static_cast<float>(1 << lvl_);                                                          \
    {                                                                                                                     \
      int bx_, by_;                                                                                                       \
      float ax_, ay_;                                                                                                     \
      footprint_base(gcx * inv_, gcy * inv_, bx_, by_, ax_, ay_);                                                         \
      gather_px<GP>(p.pyr, p.info, lvl_, N, lane, gnpix, bx_, by_, bg, pixg, gfrow, foot + (GP * wave) * FS);             \
    }                                                                                                                     \
    if (lvl_ == 3) {     /* block 10 = channels 320 .. 351: zeros behind channel 323 (its slot held block 6, consumed before the last barrier) */ \
      reinterpret_cast<uint4*>(ring + ((NCB - 1) % NSLOT) * SLOT_HALFS)[tid] = make_uint4(0u, 0u, 0u, 0u);                \
    }                                                                                                                     \
    __syncthreads();                                                                                                      \
    {                                                                                                                     \
      int bx_, by_;                                                                                                       \
      float ax_, ay_;                                                                                                     \
      footprint_base(tcx * inv_, tcy * inv_, bx_, by_, ax_, ay_);                                                         \
      const float w00 = (1.f - ax_) * (1.f - ay_), w10 = ax_ * (1.f - ay_), w01 = (1.f - ax_) * ay_, w11 = ax_ * ay_;     \
      const float* f_ = foot + r * FS + j0 * FP;                                                                          \
      float prev_[FP], cur_[FP];                                                                                          \
      _Pragma("unroll") for (int x = 0; x < FP; ++x) prev_[x] = f_[x];                                                    \
      _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                                  \
        if (jj < nj) {                                                                                                    \
          _Pragma("unroll") for (int x = 0; x < FP; ++x) cur_[x] = f_[(jj + 1) * FP + x];                                 \
          _Pragma("unroll") for (int i = 0; i < WIN; ++i) {         /* channel lvl * 81 + i * 9 + j (x-major window) */     \
            const float v_ = w00 * prev_[i] + w10 * prev_[i + 1] + w01 * cur_[i] + w11 * cur_[i + 1];                     \
            const int c_ = lvl_ * (WIN * WIN) + i * WIN + j0 + jj;                                                        \
            const int k_ = c_ & 31;                                                                                       \
            const int off_ = ((c_ >> 5) % NSLOT) * SLOT_HALFS + r * 32 + (((k_ >> 3) ^ swz) << 3) + (k_ & 7);             \
            _Float16 hi_, lo_;                                                                                            \
            split1(v_, p.a_scale, hi_, lo_);                                                                              \
            if (p.sat) sat_n += !(fabsf(v_) <= 65504.f / p.a_scale) ? 1 : 0;                                             \
            ring[off_] = hi_;                                                                                             \
            ring[off_ + MQ * 32] = lo_;                                                                                   \
          }                                                                                                               \
          _Pragma("unroll") for (int x = 0; x < FP; ++x) prev_[x] = cur_[x];                                              \
        }                                                                                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
    __syncthreads();                                                                                                      \
    _Pragma("unroll") for (int cb = (CB0_); cb < (CB1_); ++cb) {                                                          \
      const _Float16* sl_ = ring + (cb % NSLOT) * SLOT_HALFS;                                                             \
      h8 ah[2], al[2];                                                                                                    \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                     \
        ah[t] = *reinterpret_cast<const h8*>(sl_ + (16 * t + l15) * 32 + phys * 8);                                       \
        al[t] = *reinterpret_cast<const h8*>(sl_ + MQ * 32 + (16 * t + l15) * 32 + phys * 8);                             \
      }                                                                                                                   \
      const int st_ = cb & 1;                                                                                             \
      _Pragma("unroll") for (int j = 0; j < CT; ++j) {                                                                    \
        const h8 bh = __builtin_bit_cast(h8, bq[st_][j][0]), bl = __builtin_bit_cast(h8, bq[st_][j][1]);                  \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) acc[t][j] = rp::mfma16_c32(al[t], bh, acc[t][j]); \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) acc[t][j] = rp::mfma16_c32(ah[t], bl, acc[t][j]); \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) acc[t][j] = rp::mfma16_c32(ah[t], bh, acc[t][j]); \
      }                                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
      if (cb + 2 < NCB) C_LOADB(st_, cb + 2)                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
    }                                                                                                                     \
    __syncthreads();         /* the next level overwrites the footprints and the consumed ring slots */                    \
  }
  C_LEVEL(0, 0, 2)           // channels   0 ..  80: blocks 0, 1 complete (block 2 up to channel 80)
  C_LEVEL(1, 2, 5)           // channels  81 .. 161: blocks 2, 3, 4
  C_LEVEL(2, 5, 7)           // channels 162 .. 242: blocks 5, 6
  C_LEVEL(3, 7, 11)          // channels 243 .. 323: blocks 7 .. 10
#undef C_LEVEL
#undef C_LOADB
  if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));

  // ---- epilogue (csrc/conv1x1_resident.hip): accumulators -> LDS tile (aliasing ring + footprints; the last barrier above has passed) ->
  //      16-byte stores, a whole 1-KB output row per 64 lanes
  float* S = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = 64 * wave + 16 * j + l15;
    const float b = p.bias[col];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = acc[t][j][e] * p.out_scale + b;
        if (p.relu) y = fmaxf(y, 0.f);
        S[(16 * t + 4 * lq + e) * RSF + col] = y;
      }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = tid + 256 * k;
    const int row = idx >> 6, cq = (idx & 63) * 4;
    const long long m = m0 + row;
    if (m >= p.M) continue;
    const float4 y = *reinterpret_cast<const float4*>(S + row * RSF + cq);
    if (p.dst_hl) {               // quad cq of the row -> 8 bytes of the hi plane + 8 bytes of the lo plane of its 8-channel group
      rp::h4 hi, lo;
      rp::split4(y, p.a_scale, hi, lo);
      if (p.sat && rp::quad_saturates(y, p.a_scale)) atomicAdd(p.sat, 1ull);
      const int ch = p.dco + cq;
      float* ph = p.dst + m * p.dcs + (ch & ~7) + ((ch >> 2) & 1) * 2;
      *reinterpret_cast<rp::h4*>(ph) = hi;
      *reinterpret_cast<rp::h4*>(ph + 4) = lo;
    } else {
      *reinterpret_cast<float4*>(p.dst + m * p.dcs + p.dco + cq) = y;
    }
  }
}

}  // namespace

// Images [b0, b1) of a pyramid built for B_total images; coords (b1 - b0, 2, h, w) and dst (b1 - b0, h, w, dst_c_stride) are the sub-batch tensors.
extern "C" int rnnpose_corr_lookup_convc1_f16x3(const float* pyramid, const float* coords, int B_total, int b0, int b1, int h, int w,
                                                int levels, int radius, const void* w_packed, const float* bias, float a_scale,
                                                float w_scale, int relu, float* dst, int dst_c_stride, int dst_c_offset, int dst_split,
                                                rnnpose_stream_t stream) {
  const char* fn = "rnnpose_corr_lookup_convc1_f16x3";
  RP_REQUIRE(pyramid && coords && w_packed && bias && dst, fn, "null pointer");
  RP_REQUIRE(levels == 4 && radius == R, fn, "4 pyramid levels, radius 4 (324 window features = the c_in the weights were packed for)");
  RP_REQUIRE(b0 >= 0 && b0 < b1 && b1 <= B_total, fn, "image range must satisfy 0 <= b0 < b1 <= B");
  if (dst_split) RP_REQUIRE(dst_c_stride % 8 == 0 && reinterpret_cast<uintptr_t>(dst) % 32 == 0, fn, "split-form dst: channel stride multiple of 8, 32-byte aligned");
  RP_REQUIRE(dst_c_offset >= 0 && dst_c_offset + 256 <= dst_c_stride && dst_c_offset % 4 == 0 && dst_c_stride % 4 == 0 &&
                 reinterpret_cast<uintptr_t>(dst) % 16 == 0 && reinterpret_cast<uintptr_t>(w_packed) % 16 == 0, fn,
             "the 256 output channels must lie inside the row, 16-byte aligned");
  RP_REQUIRE(a_scale > 0.f && w_scale > 0.f, fn, "scales must be positive");
  int64_t offs[RNNPOSE_MAX_LEVELS + 1];
  CParams p{};
  if (int e = rnnpose_corr_pyramid_layout(B_total, h, w, levels, offs, p.info.hl, p.info.wl)) return e;
  p.info.n_px = rp::cdiv(w, 16);
  p.info.n_patch = rp::cdiv(h, 8) * p.info.n_px;
  for (int l = 0; l < levels; ++l) {
    p.info.off[l] = offs[l];
    RP_REQUIRE(p.info.hl[l] >= 2 && p.info.wl[l] >= 2, fn, "every pyramid level must be at least 2x2");
  }
  const long long n_pixels = static_cast<long long>(b1 - b0) * h * w;
  RP_REQUIRE(n_pixels < (1LL << 31), fn, "bad pixel count");
  p.pyr = pyramid; p.coords = coords; p.N = h * w; p.p_off = static_cast<long long>(b0) * h * w;
  p.wpk = static_cast<const uint4*>(w_packed);
  p.bias = bias;
  p.a_scale = a_scale; p.out_scale = 1.0f / (a_scale * w_scale);
  p.dst = dst; p.dcs = dst_c_stride; p.dco = dst_c_offset; p.relu = relu;
  p.M = n_pixels;
  p.sat = rp::sat_counter();
  p.dst_hl = dst_split;
  const dim3 grid(static_cast<unsigned>(rp::cdiv(n_pixels, MQ))), block(8 * MQ);
  hipLaunchKernelGGL(corr_convc1_kernel, grid, block, 0, rp::as_stream(stream), p);
  return rp::check_launch(fn);
}
