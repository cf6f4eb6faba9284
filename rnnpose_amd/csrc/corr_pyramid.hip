// a1+a2: all-pairs correlation volume + 4-level average pyramid in ONE pass.
// Replaces CorrBlock.corr + the avg_pool2d chain (thirdparty/raft/corr.py:13-34,59-67).
//
// corr[b,i,j] = sum_c f1[b,c,i] * f2[b,c,j] / sqrt(C)      (i,j index the h*w positions of the 1/8 maps)
//
// MI355X mapping
//   * fp32-in/fp32-acc MFMA (v_mfma_f32_32x32x2_f32): exact fp32 (the 1e-4 flow tolerance rules out
//     bf16/fp16 MFMA), 157 TFLOP/s peak.  At C=256 the kernel is MFMA-bound (2*N^2*C flop vs 4*1.33*N^2
//     bytes: ~96 flop/B), so the design keeps the matrix pipe busy and makes every output byte leave
//     the chip exactly once.
//   * Workgroup tile: 128 rows (i) x one 8(y) x 16(x) PATCH of the j image (128 columns).  Tiling j by
//     2-D patches (instead of 128 consecutive j) puts every 2x2 / 4x4 / 8x8 pooling window inside one
//     tile, so levels 1..3 are produced in the epilogue from the accumulators -- level 0 is never
//     re-read from HBM (the reference re-reads 92 MB/image three times).
//   * 4 waves, each 32(i) x 128(j): the whole patch of a row lives in ONE wave, pooling needs no
//     cross-wave traffic.  Operands are k-major in memory ((C, N) per image), exactly the layout the
//     32x32x2 MFMA wants from LDS: lane l reads A[k=l>>5][i=l&31] -> conflict-free ds_read_b32.
//   * K loop: BK=16, double-buffered LDS, global->register prefetch of tile t+1 under the MFMAs of t.
//   * Epilogue: accumulators -> wave-private LDS tile (32x32) -> coalesced 64-byte row segments of
//     level 0; pooled levels via LDS.  Epilogue LDS aliases the staging buffers.
//   * Block id -> tile: consecutive ids on one XCD (id % 8) walk the j patches of one i tile, so the A
//     tile and the B panel stay in that XCD's L2.
#include "common.hpp"

namespace {

constexpr int BM = 128;   // i rows per workgroup
constexpr int PY = 8;     // patch rows
constexpr int PX = 16;    // patch cols
constexpr int BN = PY * PX;
constexpr int BK = 16;
constexpr int NT = 256;
constexpr int ST = 8;     // supertile side (tiles)

struct PyrInfo {
  long long off[RNNPOSE_MAX_LEVELS];
  int hl[RNNPOSE_MAX_LEVELS];
  int wl[RNNPOSE_MAX_LEVELS];
  int levels;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TileRegs {
  float4 a[2];
  float4 b[2];
};

// Loads one BK x 128 slab of both operands into registers.  ALIGNED: N % 4 == 0 && w % 4 == 0.
template <bool ALIGNED>
__device__ __forceinline__ void load_tile(TileRegs& t, const float* __restrict__ f1b, const float* __restrict__ f2b,
                                          int k0, int N, int h, int w, int i0, int y0, int x0, int tid) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int f = tid + r * NT;
    const int row = f >> 5;          // 0..15
    const int c = (f & 31) << 2;     // 0..124
    // A: 4 consecutive i
    {
      const float* p = f1b + static_cast<long long>(k0 + row) * N + i0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ALIGNED && i0 + c + 3 < N) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        if (i0 + c + 0 < N) v.x = p[0];
        if (i0 + c + 1 < N) v.y = p[1];
        if (i0 + c + 2 < N) v.z = p[2];
        if (i0 + c + 3 < N) v.w = p[3];
      }
      t.a[r] = v;
    }
    // B: 4 consecutive x of patch row yy
    {
      const int yy = c >> 4, xx = c & 15;
      const int y = y0 + yy, x = x0 + xx;
      const float* p = f2b + static_cast<long long>(k0 + row) * N + static_cast<long long>(y) * w + x;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y < h) {
        if (ALIGNED && x + 3 < w) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (x + 0 < w) v.x = p[0];
          if (x + 1 < w) v.y = p[1];
          if (x + 2 < w) v.z = p[2];
          if (x + 3 < w) v.w = p[3];
        }
      }
      t.b[r] = v;
    }
  }
}

__device__ __forceinline__ void store_tile(const TileRegs& t, float* As, float* Bs, int tid) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int f = tid + r * NT;
    const int row = f >> 5;
    const int c = (f & 31) << 2;
    *reinterpret_cast<float4*>(As + row * BM + c) = t.a[r];
    *reinterpret_cast<float4*>(Bs + row * BN + c) = t.b[r];
  }
}

template <bool ALIGNED>
__global__ __launch_bounds__(NT) void corr_pyramid_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          float* __restrict__ pyr, int B, int C, int h, int w,
                                                          int n_it, int n_py, int n_px, float scale, PyrInfo info) {
  // staging: 2 buffers x (A 16x128 + B 16x128) floats = 32 KiB; epilogue: 4 waves x (1024 + 1024 + 256) floats = 36 KiB
  __shared__ __attribute__((aligned(16))) float smem[4 * 2304];
  const int N = h * w;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- block id -> (b, i tile, patch): XCD-contiguous chunks (blocks id%8 share an XCD / L2) ----
  const int ntiles = gridDim.x;
  int bid = blockIdx.x;
  {
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    // XCD x owns `per` tiles (+1 for the first `rem` XCDs); bijective for any grid size
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  // Within an image, tiles are ordered in 8x8 SUPERTILES (8 i tiles x 8 j patches): the ~96 tiles an XCD has in
  // flight share 8+8 operand panels (2 MB) instead of sweeping the whole f2 map (4.9 MB > the 4 MB L2) once per
  // i tile.  r01 PMC: 1.32 GB fetched per launch for 79 MB of unique operand bytes before this ordering.
  const int n_patch = n_py * n_px;
  const int n_ps = (n_patch + ST - 1) / ST, n_is = (n_it + ST - 1) / ST;
  const int per_img = n_ps * n_is * ST * ST;
  const int b = bid / per_img;
  const int tloc = bid - b * per_img;
  const int sidx = tloc / (ST * ST), within = tloc - sidx * (ST * ST);
  const int it = (sidx / n_ps) * ST + within / ST;
  const int patch = (sidx % n_ps) * ST + within % ST;
  if (it >= n_it || patch >= n_patch) return;      // padding tiles of a partial supertile (whole block exits)
  const int i0 = it * BM;
  const int y0 = (patch / n_px) * PY;
  const int x0 = (patch % n_px) * PX;

  const float* f1b = f1 + static_cast<long long>(b) * C * N;
  const float* f2b = f2 + static_cast<long long>(b) * C * N;

  f32x16 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  float* As0 = smem;
  float* Bs0 = smem + BK * BM;
  float* As1 = smem + 2 * BK * BM;
  float* Bs1 = smem + 3 * BK * BM;

  TileRegs t;
  load_tile<ALIGNED>(t, f1b, f2b, 0, N, h, w, i0, y0, x0, tid);
  store_tile(t, As0, Bs0, tid);
  __syncthreads();

  const int nk = C / BK;
  const int kh = lane >> 5;       // which k of the pair this lane feeds
  const int l31 = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tile<ALIGNED>(t, f1b, f2b, (kt + 1) * BK, N, h, w, i0, y0, x0, tid);
    const float* As = (kt & 1) ? As1 : As0;
    const float* Bs = (kt & 1) ? Bs1 : Bs0;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[(kk + kh) * BM + wave * 32 + l31];
      const float* brow = Bs + (kk + kh) * BN + l31;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, brow[s * 32], acc[s], 0, 0, 0);
      }
    }
    if (more) store_tile(t, (kt & 1) ? As0 : As1, (kt & 1) ? Bs0 : Bs1, tid);
    __syncthreads();
  }

  // ---------------------------------- epilogue ----------------------------------
  float* S = smem + wave * 2304;    // [32 i][32 j]   j = yy*16 + xx within sub-tile s (patch rows 2s, 2s+1)
  float* L1 = S + 1024;             // [32 i][4 Y1][8 X1]
  float* L2 = S + 2048;             // [32 i][2 Y2][4 X2]
  const int iw = i0 + wave * 32;    // first i row of this wave
  float* p0 = pyr + info.off[0] + static_cast<long long>(b) * N * N;
  const int h1 = info.hl[1], w1 = info.wl[1];
  float* p1 = info.levels > 1 ? pyr + info.off[1] + static_cast<long long>(b) * N * h1 * w1 : nullptr;

#pragma unroll
  for (int s = 0; s < 4; ++s) {
    // accumulators -> LDS (C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      S[row * 32 + l31] = acc[s][r] * scale;
    }
    __syncthreads();
    // level 0: 32 rows x 2 segments of 16 floats
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int f = lane + 64 * p;
      const int row = f >> 3;
      const int c = (f & 7) << 2;
      const int yy = c >> 4, xx = c & 15;
      const int i = iw + row, y = y0 + 2 * s + yy, x = x0 + xx;
      if (i < N && y < h) {
        const float4 v = *reinterpret_cast<const float4*>(S + row * 32 + c);
        float* dst = p0 + static_cast<long long>(i) * N + static_cast<long long>(y) * w + x;
        if (ALIGNED && x + 3 < w) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          if (x + 0 < w) dst[0] = v.x;
          if (x + 1 < w) dst[1] = v.y;
          if (x + 2 < w) dst[2] = v.z;
          if (x + 3 < w) dst[3] = v.w;
        }
      }
    }
    // level 1: 32 rows x 8 cells (2x2 means), kept in LDS for level 2
    if (info.levels > 1) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int f = lane + 64 * p;
        const int row = f >> 3;
        const int X = f & 7;
        const float2 top = *reinterpret_cast<const float2*>(S + row * 32 + 2 * X);
        const float2 bot = *reinterpret_cast<const float2*>(S + row * 32 + 16 + 2 * X);
        const float v = (((top.x + top.y) + bot.x) + bot.y) * 0.25f;
        L1[row * 32 + s * 8 + X] = v;
        const int i = iw + row, Y1 = (y0 >> 1) + s, X1 = (x0 >> 1) + X;
        if (i < N && Y1 < h1 && X1 < w1) p1[(static_cast<long long>(i) * h1 + Y1) * w1 + X1] = v;
      }
    }
    __syncthreads();
  }
  if (info.levels > 2) {
    const int h2 = info.hl[2], w2 = info.wl[2];
    float* p2 = pyr + info.off[2] + static_cast<long long>(b) * N * h2 * w2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int f = lane + 64 * p;
      const int row = f >> 3;
      const int Y = (f & 7) >> 2, X = f & 3;
      const float* q = L1 + row * 32 + (2 * Y) * 8 + 2 * X;
      const float v = (((q[0] + q[1]) + q[8]) + q[9]) * 0.25f;
      L2[row * 8 + Y * 4 + X] = v;
      const int i = iw + row, Y2 = (y0 >> 2) + Y, X2 = (x0 >> 2) + X;
      if (i < N && Y2 < h2 && X2 < w2) p2[(static_cast<long long>(i) * h2 + Y2) * w2 + X2] = v;
    }
    __syncthreads();
    if (info.levels > 3) {
      const int h3 = info.hl[3], w3 = info.wl[3];
      float* p3 = pyr + info.off[3] + static_cast<long long>(b) * N * h3 * w3;
      const int row = lane >> 1, X = lane & 1;
      const float* q = L2 + row * 8 + 2 * X;
      const float v = (((q[0] + q[1]) + q[4]) + q[5]) * 0.25f;
      const int i = iw + row, Y3 = (y0 >> 3), X3 = (x0 >> 3) + X;
      if (i < N && Y3 < h3 && X3 < w3) p3[(static_cast<long long>(i) * h3 + Y3) * w3 + X3] = v;
    }
  }
}

}  // namespace

extern "C" {

int rnnpose_corr_pyramid_layout(int B, int h, int w, int levels, int64_t* h_offsets, int* h_hl, int* h_wl) {
  const char* fn = "rnnpose_corr_pyramid_layout";
  RP_REQUIRE(B > 0 && h > 0 && w > 0, fn, "B,h,w must be positive");
  RP_REQUIRE(levels >= 1 && levels <= RNNPOSE_MAX_LEVELS, fn, "levels must be 1..4");
  int64_t off = 0;
  const int64_t rows = static_cast<int64_t>(B) * h * w;
  for (int l = 0; l < levels; ++l) {
    const int hl = h >> l, wl = w >> l;      // chained floor(./2) == floor(./2^l)
    RP_REQUIRE(hl >= 1 && wl >= 1, fn, "feature map too small for the requested number of levels");
    if (h_offsets) h_offsets[l] = off;
    if (h_hl) h_hl[l] = hl;
    if (h_wl) h_wl[l] = wl;
    off += rows * hl * wl;
  }
  if (h_offsets) h_offsets[levels] = off;
  return 0;
}

int rnnpose_corr_pyramid_f32(const float* fmap1, const float* fmap2, int B, int C, int h, int w, int levels,
                             float* pyramid, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_corr_pyramid_f32";
  RP_REQUIRE(fmap1 && fmap2 && pyramid, fn, "null pointer");
  RP_REQUIRE(C > 0 && C % BK == 0, fn, "C must be a positive multiple of 16");
  int64_t offs[RNNPOSE_MAX_LEVELS + 1];
  PyrInfo info{};
  if (int e = rnnpose_corr_pyramid_layout(B, h, w, levels, offs, info.hl, info.wl)) return e;
  for (int l = 0; l < levels; ++l) info.off[l] = offs[l];
  info.levels = levels;
  const int N = h * w;
  const int n_it = rp::cdiv(N, BM), n_py = rp::cdiv(h, PY), n_px = rp::cdiv(w, PX);
  const long long ntiles = static_cast<long long>(B) * rp::cdiv(n_it, ST) * rp::cdiv(n_py * n_px, ST) * ST * ST;
  RP_REQUIRE(ntiles < (1LL << 31), fn, "grid too large");
  const float scale = 1.0f / sqrtf(static_cast<float>(C));
  const bool aligned = (N % 4 == 0) && (w % 4 == 0) && (reinterpret_cast<uintptr_t>(fmap1) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(fmap2) % 16 == 0) && (reinterpret_cast<uintptr_t>(pyramid) % 16 == 0);
  dim3 grid(static_cast<unsigned>(ntiles)), block(NT);
  if (aligned) {
    hipLaunchKernelGGL(corr_pyramid_kernel<true>, grid, block, 0, rp::as_stream(stream), fmap1, fmap2, pyramid, B, C, h,
                       w, n_it, n_py, n_px, scale, info);
  } else {
    hipLaunchKernelGGL(corr_pyramid_kernel<false>, grid, block, 0, rp::as_stream(stream), fmap1, fmap2, pyramid, B, C,
                       h, w, n_it, n_py, n_px, scale, info);
  }
  return rp::check_launch(fn);
}

}  // extern "C"
