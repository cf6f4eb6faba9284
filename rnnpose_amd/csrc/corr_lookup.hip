// a3: correlation-pyramid window lookup.  Replaces CorrBlock.__call__ + bilinear_sampler
// (thirdparty/raft/corr.py:36-57, thirdparty/raft/utils/utils.py:57-71).
//
// out[b, l*81 + i*9 + j, Y, X] = bilinear_zero_pad( pyr_l[(b,Y,X), :, :], x = cx/2^l + (i-4), y = cy/2^l + (j-4) )
// (x-major window; align_corners=True so normalise/unnormalise is the identity up to fp32 rounding.)
//
// HBM-bound gather.  All 81 taps of a level share one fractional offset, so a pixel needs a 10x10 texel
// footprint per level (400 B) and produces 81 outputs.
//   phase 1: the wave walks its 16 pixels; for each one the 64 lanes fetch the 100 footprint texels
//            (row-contiguous 40-byte runs -> ~10-14 cache lines per instruction instead of 64) into LDS;
//   phase 2: lane = pixel; each lane slides a two-row register window over its footprint (100 LDS reads,
//            stride 101 floats -> conflict-free) and writes 81 channels; consecutive lanes are consecutive
//            pixels, so every channel row is one coalesced 256-byte store.
// One wave per workgroup (one pyramid level x 16 pixels), 6.5 KB LDS.  The gather is latency-bound: r01 ablation =
// 0.075 of 0.11 ms in the footprint loads; 64 pixels per wave (25.9 KB LDS, 6 waves per CU) ran at 0.111 ms, 32 at
// 0.067, 16 at 0.057 (16 waves per CU, the VGPR limit) -- phase 2 then only uses 16 lanes, but it is 5 % of the time.
#define RP_CONTRACT_LOCAL 1      // (geometry.cuh / induced.cuh: no fma contraction inside THEIR functions only)
#include "corr_lookup.cuh"
#include "induced.cuh"

namespace {

using namespace rplookup;

__global__ __launch_bounds__(64) void corr_lookup_kernel(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                         float* __restrict__ out, int B, int h, int w, int levels,
                                                         LookupInfo info, int nhwc, long long p_off, const rp::InducedSrc isrc,
                                                         float* __restrict__ coords_out) {
  __shared__ float foot[PIX * FS];
  const int N = h * w;
  const long long total = static_cast<long long>(B) * N;
  const int lane = threadIdx.x;
  const int lvl = blockIdx.y;
  const long long p = static_cast<long long>(blockIdx.x) * PIX + lane;   // flat (b, Y, X)
  const bool live = lane < PIX && p < total;
  const int hl = info.hl[lvl], wl = info.wl[lvl];
  const float inv = 1.0f / static_cast<float>(1 << lvl);

  float cx = 0.f, cy = 0.f;
  int b = 0, pix = 0;
  // row of this lane's pixel in the pyramid (image of the WHOLE batch the pyramid was built for, pixel inside it)
  const long long prow = p_off + (p < total ? p : total - 1);
  const int bg = static_cast<int>(prow / N), pixg = static_cast<int>(prow - static_cast<long long>(bg) * N);
  if (live) {
    b = static_cast<int>(p / N);
    pix = static_cast<int>(p - static_cast<long long>(b) * N);
    if (isrc.depth) {
      // r06: the pose-induced coordinates are formed HERE (induced.cuh: the very function of induced_coords_lowres_kernel) instead of being
      // read from the output of a launch of their own; the level-0 workgroup of a pixel writes them out for the later consumers
      const int Y = pix / w, X = pix - Y * w;
      const float2 c = rp::induced_coords_at(isrc.depth + static_cast<long long>(b) * isrc.H * isrc.W, X, Y, isrc.H, isrc.W, h, w, isrc.eps,
                                             rp::load_intr(isrc.K, b), rp::load_pose(isrc.G, b));
      if (lvl == 0 && coords_out) {
        coords_out[(static_cast<long long>(b) * 2 + 0) * N + pix] = c.x;
        coords_out[(static_cast<long long>(b) * 2 + 1) * N + pix] = c.y;
      }
      cx = c.x * inv;
      cy = c.y * inv;
    } else {
      cx = coords[(static_cast<long long>(b) * 2 + 0) * N + pix] * inv;
      cy = coords[(static_cast<long long>(b) * 2 + 1) * N + pix] * inv;
    }
  }
  int bx, by;
  float ax, ay;
  footprint_base(cx, cy, bx, by, ax, ay);

  // ---- phase 1: cooperative footprint fetch (corr_lookup.cuh) ----
  const long long first = static_cast<long long>(blockIdx.x) * PIX;
  const int npix = static_cast<int>(total - first < PIX ? total - first : PIX);
  gather16(pyr, info, lvl, N, lane, npix, bx, by, bg, pixg, p_off + first, foot);
  __syncthreads();
  // ---- phase 2: lane = pixel ----
  const float w00 = (1.f - ax) * (1.f - ay), w10 = ax * (1.f - ay), w01 = (1.f - ax) * ay, w11 = ax * ay;
  const float* f = foot + (lane < PIX ? lane : 0) * FS;      // (lanes >= PIX idle through phase 2)
  float res[WIN * WIN];
  float prev[FP], cur[FP];
#pragma unroll
  for (int x = 0; x < FP; ++x) prev[x] = f[x];
#pragma unroll
  for (int j = 0; j < WIN; ++j) {          // y offset j-4  -> footprint rows j, j+1
#pragma unroll
    for (int x = 0; x < FP; ++x) cur[x] = f[(j + 1) * FP + x];
#pragma unroll
    for (int i = 0; i < WIN; ++i)          // x offset i-4  -> footprint cols i, i+1 ; channel i*9 + j
      res[i * WIN + j] = w00 * prev[i] + w10 * prev[i + 1] + w01 * cur[i] + w11 * cur[i + 1];
#pragma unroll
    for (int x = 0; x < FP; ++x) prev[x] = cur[x];
  }
  if (!nhwc) {
    // (B, L*81, h, w): consecutive lanes are consecutive pixels -> one coalesced 256-byte row per channel
    if (live) {
      float* o = out + (static_cast<long long>(b) * levels * (WIN * WIN) + static_cast<long long>(lvl) * (WIN * WIN)) * N + pix;
#pragma unroll
      for (int c = 0; c < WIN * WIN; ++c) o[static_cast<long long>(c) * N] = res[c];
    }
  } else {
    // (B, h, w, L*81): transpose through LDS (aliasing the footprints) so that every pixel's 81 values of this
    // level leave as one contiguous 324-byte run
    __syncthreads();
    if (lane < PIX) {
#pragma unroll
      for (int c = 0; c < WIN * WIN; ++c) foot[lane * (WIN * WIN) + c] = res[c];
    }
    __syncthreads();
    const int ctot = levels * WIN * WIN;
    float* o = out + first * ctot + lvl * (WIN * WIN);
    for (int q = 0; q < npix; ++q) {
      o[static_cast<long long>(q) * ctot + lane] = foot[q * (WIN * WIN) + lane];
      if (lane < WIN * WIN - 64) o[static_cast<long long>(q) * ctot + 64 + lane] = foot[q * (WIN * WIN) + 64 + lane];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// r06: the same lookup at half the instructions per wave (RPL_V2 = 1, default; 0 keeps the r01-r05 kernel above for A/B).  The kernel is
// bound by instructions, not by HBM (98 MB per half-batch launch in 26 us) -- ~1600 per wave for 16 pixels of one level:
//   * phase 1 by gather_px_fast (corr_lookup.cuh): one address form per level class, scalar pixel bases, 32-bit byte offsets;
//   * every lane carries the coordinates of pixel (lane & 15), so that phase 2 runs on 48 lanes instead of 16: lane (q, part) slides the
//     two-row window over footprint rows 3 part .. 3 part + 3 and produces the 27 channels i * 9 + j, j in [3 part, 3 part + 3) -- the
//     same expression per channel, bit-identical values -- and writes them into the transposition buffer itself;
//   * the NHWC rows leave as before (a pixel's 81 values of the level = one contiguous 324-byte run).
template <bool L0>
__device__ __forceinline__ void lookup_body_v2(const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out, int B,
                                               int h, int w, int levels, const LookupInfo& info, int nhwc, long long p_off,
                                               const rp::InducedSrc& isrc, float* __restrict__ coords_out, float* foot, int lvl) {
  const int N = h * w;
  const long long total = static_cast<long long>(B) * N;
  const int lane = threadIdx.x;
  const int q16 = lane & 15, part = lane >> 4;
  const long long first = static_cast<long long>(blockIdx.x) * PIX;
  const long long p = first + q16;                                        // flat (b, Y, X) of this lane's pixel
  const bool live = p < total;
  const float inv = 1.0f / static_cast<float>(1 << lvl);

  float cx = 0.f, cy = 0.f;
  int b = 0, pix = 0;
  const long long prow = p_off + (p < total ? p : total - 1);
  const int bg = static_cast<int>(prow / N), pixg = static_cast<int>(prow - static_cast<long long>(bg) * N);
  if (live) {
    b = static_cast<int>(p / N);
    pix = static_cast<int>(p - static_cast<long long>(b) * N);
    if (isrc.depth) {
      const int Y = pix / w, X = pix - Y * w;
      const float2 c = rp::induced_coords_at(isrc.depth + static_cast<long long>(b) * isrc.H * isrc.W, X, Y, isrc.H, isrc.W, h, w, isrc.eps,
                                             rp::load_intr(isrc.K, b), rp::load_pose(isrc.G, b));
      if (lvl == 0 && coords_out && part == 0) {
        coords_out[(static_cast<long long>(b) * 2 + 0) * N + pix] = c.x;
        coords_out[(static_cast<long long>(b) * 2 + 1) * N + pix] = c.y;
      }
      cx = c.x * inv;
      cy = c.y * inv;
    } else {
      cx = coords[(static_cast<long long>(b) * 2 + 0) * N + pix] * inv;
      cy = coords[(static_cast<long long>(b) * 2 + 1) * N + pix] * inv;
    }
  }
  int bx, by;
  float ax, ay;
  footprint_base(cx, cy, bx, by, ax, ay);

  // ---- phase 1: cooperative footprint fetch ----
  const int npix = static_cast<int>(total - first < PIX ? total - first : PIX);
  gather_px_fast<PIX, L0>(pyr, info, lvl, N, lane, npix, bx, by, bg, pixg, p_off + first, foot);
  __syncthreads();
  // ---- phase 2: lane = (pixel, third of the window rows) ----
  const float w00 = (1.f - ax) * (1.f - ay), w10 = ax * (1.f - ay), w01 = (1.f - ax) * ay, w11 = ax * ay;
  const int pr = part < 3 ? part : 2;                                     // (the fourth lane group repeats the third one's work and writes nothing)
  const float* f = foot + q16 * FS + 3 * pr * FP;
  float res[3][WIN];
  {
    float prev[FP], cur[FP];
#pragma unroll
    for (int x = 0; x < FP; ++x) prev[x] = f[x];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {         // y offset 3 part + jj - 4 -> footprint rows 3 part + jj, + 1
#pragma unroll
      for (int x = 0; x < FP; ++x) cur[x] = f[(jj + 1) * FP + x];
#pragma unroll
      for (int i = 0; i < WIN; ++i)          // x offset i - 4 -> footprint cols i, i + 1; channel i * 9 + j
        res[jj][i] = w00 * prev[i] + w10 * prev[i + 1] + w01 * cur[i] + w11 * cur[i + 1];
#pragma unroll
      for (int x = 0; x < FP; ++x) prev[x] = cur[x];
    }
  }
  __syncthreads();                                                        // every lane is done reading the footprints: the buffer is re-used
  if (part < 3) {
#pragma unroll
    for (int jj = 0; jj < 3; ++jj)
#pragma unroll
      for (int i = 0; i < WIN; ++i) foot[q16 * (WIN * WIN) + i * WIN + 3 * part + jj] = res[jj][i];
  }
  __syncthreads();
  if (!nhwc) {
    // (B, L*81, h, w): lanes 0..15 = the 16 pixels; a channel row of 16 consecutive pixels is one 64-byte run
    if (lane < PIX && live) {
      float* o = out + (static_cast<long long>(b) * levels * (WIN * WIN) + static_cast<long long>(lvl) * (WIN * WIN)) * N + pix;
#pragma unroll
      for (int c = 0; c < WIN * WIN; ++c) o[static_cast<long long>(c) * N] = foot[lane * (WIN * WIN) + c];
    }
  } else {
    const int ctot = levels * WIN * WIN;
    float* o = out + first * ctot + lvl * (WIN * WIN);
    for (int q = 0; q < npix; ++q) {
      o[static_cast<long long>(q) * ctot + lane] = foot[q * (WIN * WIN) + lane];
      if (lane < WIN * WIN - 64) o[static_cast<long long>(q) * ctot + 64 + lane] = foot[q * (WIN * WIN) + 64 + lane];
    }
  }
}

__global__ __launch_bounds__(64) void corr_lookup_v2_kernel(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                            float* __restrict__ out, int B, int h, int w, int levels,
                                                            LookupInfo info, int nhwc, long long p_off, const rp::InducedSrc isrc,
                                                            float* __restrict__ coords_out) {
  __shared__ float foot[PIX * FS];
  const int lvl = blockIdx.y;
  if (lvl == 0) lookup_body_v2<true>(pyr, coords, out, B, h, w, levels, info, nhwc, p_off, isrc, coords_out, foot, lvl);
  else lookup_body_v2<false>(pyr, coords, out, B, h, w, levels, info, nhwc, p_off, isrc, coords_out, foot, lvl);
}

#ifndef RPL_V2
#define RPL_V2 1
#endif

}  // namespace

static int g_lookup_v2 = RPL_V2;
extern "C" int rnnpose_corr_lookup_variant(int variant) {      // measurement / test switch: 1 (default) = the r06 kernel, 0 = the r01-r05 kernel; bit-identical results
  g_lookup_v2 = variant ? 1 : 0;
  return 0;
}

// B_total: batch the pyramid was built for (its layout); [b0, b1): the images this launch looks up.  coords / out point at
// image b0 (sub-batch tensors); the pyramid pointer is the whole buffer.
static int launch_lookup(const char* fn, const float* pyramid, const float* coords, int B_total, int b0, int b1, int h, int w,
                         int levels, int radius, float* out, int nhwc, rnnpose_stream_t stream, const rp::InducedSrc isrc = rp::InducedSrc{},
                         float* coords_out = nullptr) {
  RP_REQUIRE(pyramid && (coords || isrc.depth) && out, fn, "null pointer");
  RP_REQUIRE(radius == R, fn, "radius must be 4");
  RP_REQUIRE(b0 >= 0 && b0 < b1 && b1 <= B_total, fn, "image range must satisfy 0 <= b0 < b1 <= B");
  int64_t offs[RNNPOSE_MAX_LEVELS + 1];
  LookupInfo info{};
  if (int e = rnnpose_corr_pyramid_layout(B_total, h, w, levels, offs, info.hl, info.wl)) return e;
  info.n_px = rp::cdiv(w, 16);
  info.n_patch = rp::cdiv(h, 8) * info.n_px;
  for (int l = 0; l < levels; ++l) {
    info.off[l] = offs[l];
    // the reference divides by (W_l - 1): a 1-wide level yields inf/NaN there
    RP_REQUIRE(info.hl[l] >= 2 && info.wl[l] >= 2, fn, "every pyramid level must be at least 2x2");
  }
  const int B = b1 - b0;
  const long long total = static_cast<long long>(B) * h * w;
  dim3 grid(static_cast<unsigned>(rp::cdiv(total, PIX)), static_cast<unsigned>(levels)), block(64);
  // (the r06 kernel addresses a level-0 image by unsigned 32-bit byte offsets: N^2 floats below 4 GB -- true up to 180 x 180 maps; larger ones keep the r05 kernel)
  const bool v2 = g_lookup_v2 && static_cast<long long>(info.n_patch) * h * w * 512 < (1LL << 32);
  if (v2)
    hipLaunchKernelGGL(corr_lookup_v2_kernel, grid, block, 0, rp::as_stream(stream), pyramid, coords, out, B, h, w, levels, info,
                       nhwc, static_cast<long long>(b0) * h * w, isrc, coords_out);
  else
    hipLaunchKernelGGL(corr_lookup_kernel, grid, block, 0, rp::as_stream(stream), pyramid, coords, out, B, h, w, levels, info,
                       nhwc, static_cast<long long>(b0) * h * w, isrc, coords_out);
  return rp::check_launch(fn);
}

extern "C" int rnnpose_corr_lookup_f32(const float* pyramid, const float* coords, int B, int h, int w, int levels,
                                       int radius, float* out, rnnpose_stream_t stream) {
  return launch_lookup("rnnpose_corr_lookup_f32", pyramid, coords, B, 0, B, h, w, levels, radius, out, 0, stream);
}

// same lookup, output laid out (B, h, w, levels*81) for the NHWC update-block engine
extern "C" int rnnpose_corr_lookup_nhwc_f32(const float* pyramid, const float* coords, int B, int h, int w, int levels,
                                            int radius, float* out, rnnpose_stream_t stream) {
  return launch_lookup("rnnpose_corr_lookup_nhwc_f32", pyramid, coords, B, 0, B, h, w, levels, radius, out, 1, stream);
}

// NHWC lookup of the images [b0, b1) of a batch-B pyramid: coords (b1-b0, 2, h, w) and out (b1-b0, h, w, levels*81) are the
// sub-batch tensors.  Lets the two half-batch chains of the update engine each start with their own lookup.
extern "C" int rnnpose_corr_lookup_nhwc_part_f32(const float* pyramid, const float* coords, int B, int b0, int b1, int h, int w,
                                                 int levels, int radius, float* out, rnnpose_stream_t stream) {
  return launch_lookup("rnnpose_corr_lookup_nhwc_part_f32", pyramid, coords, B, b0, b1, h, w, levels, radius, out, 1, stream);
}

// r06: the same launch forming its coordinates itself -- coords1 = grid + down-sampled pose-induced flow (rnnpose_induced_coords_lowres_f32's
// arithmetic, bit for bit: csrc/induced.cuh) from depth (b1-b0,1,H,W), K (b1-b0,3,3), G (b1-b0,4,4) of the launch's images; coords_out
// (b1-b0,2,h,w) receives them for the later consumers of the iteration (flow head).  One launch and one dependent kernel boundary fewer per
// GRU iteration (model/PoseRefiner.py:324-328 + thirdparty/raft/corr.py:36-57).
extern "C" int rnnpose_corr_lookup_induced_nhwc_part_f32(const float* pyramid, const float* depth, const float* K, const float* G, int H, int W,
                                                         float eps, int B, int b0, int b1, int h, int w, int levels, int radius,
                                                         float* coords_out, float* out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_corr_lookup_induced_nhwc_part_f32";
  RP_REQUIRE(depth && K && G && coords_out, fn, "null pointer");
  RP_REQUIRE(H >= h && W >= w && h > 0 && w > 0 && W / w >= 1, fn, "the depth map must be at least as large as the low-resolution map");
  return launch_lookup(fn, pyramid, nullptr, B, b0, b1, h, w, levels, radius, out, 1, stream, rp::InducedSrc{depth, K, G, H, W, eps}, coords_out);
}
