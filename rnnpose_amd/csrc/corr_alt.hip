// a3': volume-free correlation lookup -- the reference's AlternateCorrBlock (thirdparty/raft/corr.py:70-98; its CUDA extension
// alt_cuda_corr is not part of the reference tree and the reference never enables it, model/CFNet.py:63-64).  The 9x9 window of
// level l around coords / 2^l is computed ON THE FLY from fmap1 and the 2^l-pooled fmap2 instead of being read from a
// materialised volume: pooling is linear, so  mean_{2x2}(<f1, f2>) = <f1, mean_{2x2}(f2)>  and the result equals the pyramid
// lookup (csrc/corr_lookup.hip) up to fp32 summation order.
//
//   out[b, Y, X, l*81 + i*9 + j] = bilinear_zero_pad( <f1[b,Y,X,:], f2_l[b, ., ., :]> / sqrt(C),  x = cx/2^l + i - 4,  y = cy/2^l + j - 4 )
//
// One wave per (pixel, level): the 64 lanes hold the pixel's fmap1 vector (4 channels each per 256), walk the 10 x 10 integer
// footprint (one coalesced C*4-byte row of the level's map per tap), reduce the 100 partial dot products with a reduce-scatter
// butterfly (63 lane exchanges per 64 taps instead of 6 per tap), and blend the 81 outputs from the wave's 100 totals in LDS.
//
// MEASUREMENT, not the product path (profiles/r03_corr_alt.txt): per pixel the window position differs, so this is a
// gather of 100 dot products per level -- 100 KB of fmap2 rows through the CU's load path per (pixel, level) -- not a GEMM; the
// materialised volume (build once per outer iteration + 10 x 10 texel gathers per iteration) stays the graded kernel.
#include "common.hpp"

namespace {

constexpr int R = 4;
constexpr int WIN = 2 * R + 1;      // 9
constexpr int FP = WIN + 1;         // 10: footprint side
constexpr int NTAP = FP * FP;       // 100

struct AltInfo {
  long long off[RNNPOSE_MAX_LEVELS];      // float offset of level l inside the pooled buffer (level 0 unused: it is fmap2 itself)
  int hl[RNNPOSE_MAX_LEVELS];
  int wl[RNNPOSE_MAX_LEVELS];
};

// level l (B, h_l, w_l, C) from level l-1: 2x2 mean, floor cropping, the volume pyramid's summation order
__global__ __launch_bounds__(256) void fmap_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int hs,
                                                        int ws, int C4) {
  const int hd = hs >> 1, wd = ws >> 1;
  const long long total = static_cast<long long>(B) * hd * wd * C4;
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = static_cast<int>(i % C4);
  const long long pix = i / C4;
  const int x = static_cast<int>(pix % wd);
  const long long r = pix / wd;
  const int y = static_cast<int>(r % hd), b = static_cast<int>(r / hd);
  const float4* s = reinterpret_cast<const float4*>(src) + ((static_cast<long long>(b) * hs + 2 * y) * ws + 2 * x) * C4 + c;
  const float4 a = s[0], bq = s[C4], cq = s[static_cast<long long>(ws) * C4], d = s[static_cast<long long>(ws) * C4 + C4];
  float4 o;
  o.x = (((a.x + bq.x) + cq.x) + d.x) * 0.25f;
  o.y = (((a.y + bq.y) + cq.y) + d.y) * 0.25f;
  o.z = (((a.z + bq.z) + cq.z) + d.z) * 0.25f;
  o.w = (((a.w + bq.w) + cq.w) + d.w) * 0.25f;
  reinterpret_cast<float4*>(dst)[i] = o;
}

// 64 per-lane partial sums v[0..63] (tap t in v[t]) -> lane l holds the wave total of tap l (63 exchanges)
__device__ __forceinline__ float reduce_scatter64(float (&v)[64], int lane) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned m = (lane & off) ? 0xffffffffu : 0u;       // lanes with the bit set keep the upper half
#pragma unroll
    for (int k = 0; k < off; ++k) {
      // (bitwise blends: a select between two elements of the array becomes a dynamically indexed load -- scratch)
      const unsigned lo = __float_as_uint(v[k]), hi = __float_as_uint(v[k + off]);
      const float send = __uint_as_float((lo & m) | (hi & ~m));
      const float keep = __uint_as_float((hi & m) | (lo & ~m));
      v[k] = keep + __shfl_xor(send, off);
    }
  }
  return v[0];
}

// footprint rows [ROW0, ROW0 + ROWS) of one (pixel, level): per row the 10 row loads are issued together, unconditionally (taps
// outside the level read texel 0 and are zeroed); the wave totals of the ROWS * 10 taps land in tot[ROW0 * 10 + ...]
template <int QN, int ROW0, int ROWS>
__device__ __forceinline__ void rows_batch(const float4 (&a)[QN], const float* __restrict__ map, int bx, int by, int hl, int wl, int C,
                                           int C4, int lane, float scale, float* tot) {
  float v[64];
#pragma unroll
  for (int t = 0; t < 64; ++t) v[t] = 0.f;
#pragma unroll
  for (int tr = 0; tr < ROWS; ++tr) {
    const int ty = ROW0 + tr;
    float4 ld[FP][QN];
    unsigned okm = 0u;
#pragma unroll
    for (int tx = 0; tx < FP; ++tx) {
      const int X = bx + tx, Y = by + ty;
      const bool ok = X >= 0 && X < wl && Y >= 0 && Y < hl;
      const float4* row = reinterpret_cast<const float4*>(map + (ok ? (static_cast<long long>(Y) * wl + X) * C : 0));
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int c4 = lane + 64 * q;
        ld[tx][q] = row[c4 < C4 ? c4 : 0];
      }
      okm |= ok ? (1u << tx) : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tx = 0; tx < FP; ++tx) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int c4 = lane + 64 * q;
        const float4 x = ld[tx][q];
        const float d = ((a[q].x * x.x + a[q].y * x.y) + a[q].z * x.z) + a[q].w * x.w;
        s += c4 < C4 ? d : 0.f;
      }
      v[tr * FP + tx] = (okm >> tx) & 1u ? s : 0.f;
    }
  }
  const float t = reduce_scatter64(v, lane);
  if (lane < ROWS * FP) tot[ROW0 * FP + lane] = t * scale;
}

// QN = float4 per lane (C <= 256 QN): channels [(lane + 64 q) * 4, +4)
template <int QN>
__global__ __launch_bounds__(256) void corr_alt_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                       const float* __restrict__ f2p, const float* __restrict__ coords,
                                                       float* __restrict__ out, int B, int h, int w, int C, int levels, AltInfo info,
                                                       int out_cs, int out_co, float scale) {
  __shared__ float tot_s[4][128];
  const int lane = threadIdx.x & 63, lvl = threadIdx.x >> 6;
  if (lvl >= levels) return;                 // (no workgroup barrier below: the waves are independent)
  const int N = h * w;
  const long long p = blockIdx.x;            // flat (b, Y, X)
  const int b = static_cast<int>(p / N);
  const int pix = static_cast<int>(p - static_cast<long long>(b) * N);
  // (explicit selects: a dynamically indexed by-value array would be copied to scratch)
  const int hl = lvl == 0 ? info.hl[0] : lvl == 1 ? info.hl[1] : lvl == 2 ? info.hl[2] : info.hl[3];
  const int wl = lvl == 0 ? info.wl[0] : lvl == 1 ? info.wl[1] : lvl == 2 ? info.wl[2] : info.wl[3];
  const long long loff = lvl == 1 ? info.off[1] : lvl == 2 ? info.off[2] : info.off[3];
  const float inv = 1.0f / static_cast<float>(1 << lvl);
  const float cx = coords[(static_cast<long long>(b) * 2 + 0) * N + pix] * inv;
  const float cy = coords[(static_cast<long long>(b) * 2 + 1) * N + pix] * inv;
  // integer base of the footprint; non-finite / far-away coordinates sample only padding -> zeros (as csrc/corr_lookup.hip)
  const bool sane = (cx > -1.0e6f) && (cx < 1.0e6f) && (cy > -1.0e6f) && (cy < 1.0e6f);
  const float fx0 = floorf(cx), fy0 = floorf(cy);
  const int bx = sane ? static_cast<int>(fx0) - R : -1000000;
  const int by = sane ? static_cast<int>(fy0) - R : -1000000;
  const float ax = sane ? cx - fx0 : 0.f;
  const float ay = sane ? cy - fy0 : 0.f;

  const int C4 = C >> 2;
  float4 a[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int c4 = lane + 64 * q;
    a[q] = c4 < C4 ? reinterpret_cast<const float4*>(f1 + p * C)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float* map = lvl == 0 ? f2 + static_cast<long long>(b) * N * C
                              : f2p + loff + static_cast<long long>(b) * hl * wl * C;
  float* tot = tot_s[lvl];
  // two batches of footprint rows (rows 0-5 = 60 taps, rows 6-9 = 40 taps, padded to 64 partial sums each)
  rows_batch<QN, 0, 6>(a, map, bx, by, hl, wl, C, C4, lane, scale, tot);
  rows_batch<QN, 6, 4>(a, map, bx, by, hl, wl, C, C4, lane, scale, tot);
  __builtin_amdgcn_wave_barrier();           // tot is wave-private: LDS operations of one wave execute in order
  const float w00 = (1.f - ax) * (1.f - ay), w10 = ax * (1.f - ay), w01 = (1.f - ax) * ay, w11 = ax * ay;
  float* o = out + p * out_cs + out_co + lvl * (WIN * WIN);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int oi = lane + 64 * pass;         // output i * 9 + j: x offset i - 4 (footprint columns i, i + 1), y offset j - 4
    if (oi < WIN * WIN) {
      const int i = oi / WIN, j = oi - i * WIN;
      const float* f = tot + j * FP + i;
      o[oi] = w00 * f[0] + w10 * f[1] + w01 * f[FP] + w11 * f[FP + 1];
    }
  }
}

int alt_layout(const char* fn, int B, int h, int w, int C, int levels, AltInfo& info, long long* total) {
  RP_REQUIRE(B > 0 && h > 0 && w > 0, fn, "B,h,w must be positive");
  RP_REQUIRE(levels >= 1 && levels <= RNNPOSE_MAX_LEVELS, fn, "levels must be 1..4");
  RP_REQUIRE(C > 0 && C % 4 == 0 && C <= 512, fn, "C must be a positive multiple of 4, at most 512");
  for (int l = 0; l < levels; ++l) {
    info.hl[l] = h >> l; info.wl[l] = w >> l;
    RP_REQUIRE(info.hl[l] >= 1 && info.wl[l] >= 1, fn, "feature map too small for the requested number of levels");
  }
  // levels 1.. are laid out back to back from offset 0 of the pooled buffer; level 0 is fmap2 itself
  long long o2 = 0;
  info.off[0] = 0;
  for (int l = 1; l < levels; ++l) { info.off[l] = o2; o2 += static_cast<long long>(B) * info.hl[l] * info.wl[l] * C; }
  *total = o2;
  return 0;
}

}  // namespace

extern "C" {

size_t rnnpose_fmap_pyramid_floats(int B, int h, int w, int C, int levels) {
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || levels < 1 || levels > RNNPOSE_MAX_LEVELS) return 0;
  size_t n = 0;
  for (int l = 1; l < levels; ++l) n += static_cast<size_t>(B) * (h >> l) * (w >> l) * C;
  return n;
}

int rnnpose_fmap_pyramid_f32(const float* fmap2_nhwc, int B, int h, int w, int C, int levels, float* pooled,
                             rnnpose_stream_t stream) {
  const char* fn = "rnnpose_fmap_pyramid_f32";
  RP_REQUIRE(fmap2_nhwc && (pooled || levels == 1), fn, "null pointer");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(fmap2_nhwc) % 16 == 0 && reinterpret_cast<uintptr_t>(pooled) % 16 == 0, fn, "16-byte alignment");
  AltInfo info{};
  long long total = 0;
  if (int e = alt_layout(fn, B, h, w, C, levels, info, &total)) return e;
  hipStream_t st = rp::as_stream(stream);
  const float* src = fmap2_nhwc;
  for (int l = 1; l < levels; ++l) {
    float* dst = pooled + info.off[l];
    const long long n = static_cast<long long>(B) * info.hl[l] * info.wl[l] * (C / 4);
    hipLaunchKernelGGL(fmap_pool_kernel, dim3(rp::cdiv(n, 256)), dim3(256), 0, st, src, dst, B, info.hl[l - 1], info.wl[l - 1], C / 4);
    src = dst;
  }
  return rp::check_launch(fn);
}

int rnnpose_corr_alt_lookup_f32(const float* fmap1_nhwc, const float* fmap2_nhwc, const float* pooled, const float* coords, int B,
                                int h, int w, int C, int levels, int radius, float* out, int out_c_stride, int out_c_offset,
                                rnnpose_stream_t stream) {
  const char* fn = "rnnpose_corr_alt_lookup_f32";
  RP_REQUIRE(fmap1_nhwc && fmap2_nhwc && coords && out && (pooled || levels == 1), fn, "null pointer");
  RP_REQUIRE(radius == R, fn, "radius must be 4");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(fmap1_nhwc) % 16 == 0 && reinterpret_cast<uintptr_t>(fmap2_nhwc) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(pooled) % 16 == 0, fn, "feature maps must be 16-byte aligned");
  AltInfo info{};
  long long total = 0;
  if (int e = alt_layout(fn, B, h, w, C, levels, info, &total)) return e;
  RP_REQUIRE(out_c_stride >= out_c_offset + levels * WIN * WIN && out_c_offset >= 0, fn, "output slice does not fit its channel stride");
  const long long npix = static_cast<long long>(B) * h * w;
  RP_REQUIRE(npix < (1LL << 31), fn, "too many pixels");
  const float scale = 1.0f / sqrtf(static_cast<float>(C));
  const dim3 grid(static_cast<unsigned>(npix)), block(256);
  hipStream_t st = rp::as_stream(stream);
  if (C <= 256) {
    hipLaunchKernelGGL(corr_alt_kernel<1>, grid, block, 0, st, fmap1_nhwc, fmap2_nhwc, pooled, coords, out, B, h, w, C, levels, info,
                       out_c_stride, out_c_offset, scale);
  } else {
    hipLaunchKernelGGL(corr_alt_kernel<2>, grid, block, 0, st, fmap1_nhwc, fmap2_nhwc, pooled, coords, out, B, h, w, C, levels, info,
                       out_c_stride, out_c_offset, scale);
  }
  return rp::check_launch(fn);
}

}  // extern "C"
