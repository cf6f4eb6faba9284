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
#include "f16x3.cuh"

namespace {

constexpr int BM = 128;   // i rows per workgroup
constexpr int PY = 8;     // patch rows
constexpr int PX = 16;    // patch cols
constexpr int BN = PY * PX;
constexpr int BK = 16;
constexpr int NT = 256;
constexpr int ST = 8;     // supertile side (tiles)

// Diagnostics builds only (tools/corr_ablate.sh): RPC_ABL bits 1 = the K loop requests no operands after the first slab,
// 2 = the epilogue stores nothing, 4 = no K loop at all, 8 = no epilogue (one value per lane leaves), 16 = the K loop does not
// refill LDS.  0 in the product.
#ifndef RPC_ABL
#define RPC_ABL 0
#endif
#ifndef RPC_DB
#define RPC_DB 0          // fp16x3 kernel: 0 = single operand buffer, two barriers per slab, four workgroups per CU (r03-r06); 1 = double-buffered
                          // operand LDS, ONE barrier per slab, 80 KiB = two workgroups per CU -- built in r06 (asked for by the r03-r05 reviews) and
                          // SLOWER: 465-480 vs 445 us at B = 8 60 x 80, 7.30 vs 6.93 ms at 120 x 160, same box, bit-identical results
                          // (profiles/r06_corr_double_buffer.txt): the kernel needs its co-resident workgroups more than it needs the barrier
#endif
#ifndef RPC_NT
#define RPC_NT 1          // 1: level-0 rows leave with non-temporal stores (they are never re-read by this kernel: -13 %);
                          // 2: the pooled levels too (their 8-32 byte pieces then miss L2's merging: +30 %)
#endif
typedef float f4v __attribute__((ext_vector_type(4)));

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

// accumulators (4 sub-tiles of 32 i x 32 j per wave, C layout of the 32x32 MFMAs) -> level 0 + pooled levels.
// Shared by the fp32 and the fp16x3 kernels; `smem` is the (re-used) staging LDS, all 256 threads must call it.
template <bool ALIGNED>
__device__ __forceinline__ void pyramid_epilogue(const f32x16& acc0, const f32x16& acc1, const f32x16& acc2,
                                                 const f32x16& acc3, float* smem, float* __restrict__ pyr,
                                                 const PyrInfo& info, int b, int N, int h, int w, int i0, int y0, int x0,
                                                 float scale, int wave, int lane, int patch, int n_patch) {
  const int kh = lane >> 5, l31 = lane & 31;
  const f32x16 acc[4] = {acc0, acc1, acc2, acc3};
  const int Nm = ((RPC_ABL & 2) && scale != 1234.5f) ? 0 : N;      // diagnostics: every store below is masked by i < Nm
  const int Nm0 = ((RPC_ABL & 32) && scale != 1234.5f) ? 0 : Nm;   // 32: no level-0 stores, 64: no pooled stores
  const int NmP = ((RPC_ABL & 64) && scale != 1234.5f) ? 0 : Nm;
  float* S = smem + wave * 2304;    // [32 i][32 j]   j = yy*16 + xx within sub-tile s (patch rows 2s, 2s+1)
  float* L1 = S + 1024;             // [32 i][4 Y1][8 X1]
  float* L2 = S + 2048;             // [32 i][2 Y2][4 X2]
  const int iw = i0 + wave * 32;    // first i row of this wave
  // level 0 is stored J-PATCH-MAJOR (r04): [image][8 x 16 patch of j][i][8][16] -- the 128 i rows x one patch of a workgroup are
  // ONE contiguous 64-KB run (a wave: 16 KB), rows of other i tiles of the same patch follow it.  r03's row-major level 0 left the
  // chip as 64-byte pieces 19 KB apart (3.6-4.3 TB/s in the store-pattern probe against 5.9-6.5 for linear runs).  Patches at the
  // image border are stored whole: their columns outside the image were multiplied with zero rows.
  float* p0 = pyr + info.off[0] + (static_cast<long long>(b) * n_patch + patch) * N * BN;
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
    __builtin_amdgcn_wave_barrier();   // S / L1 / L2 are wave-private: LDS ops of one wave execute in order
    // level 0: 32 rows x 2 segments of 16 floats
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int f = lane + 64 * p;
      const int row = f >> 3;
      const int c = (f & 7) << 2;
      const int yy = c >> 4, xx = c & 15;
      const int i = iw + row;
      if (i < Nm0) {
        const float4 v = *reinterpret_cast<const float4*>(S + row * 32 + c);
        float* dst = p0 + static_cast<long long>(i) * BN + (2 * s + yy) * PX + xx;      // 16-byte aligned (layout offsets are multiples of 4)
        if (RPC_NT) {
          const f4v vv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(dst));
        } else {
          *reinterpret_cast<float4*>(dst) = v;
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
        if (i < NmP && Y1 < h1 && X1 < w1) {
          if (RPC_NT >= 2) __builtin_nontemporal_store(v, p1 + (static_cast<long long>(i) * h1 + Y1) * w1 + X1);
          else p1[(static_cast<long long>(i) * h1 + Y1) * w1 + X1] = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // S / L1 / L2 are wave-private: LDS ops of one wave execute in order
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
      if (i < NmP && Y2 < h2 && X2 < w2) {
        if (RPC_NT >= 2) __builtin_nontemporal_store(v, p2 + (static_cast<long long>(i) * h2 + Y2) * w2 + X2);
        else p2[(static_cast<long long>(i) * h2 + Y2) * w2 + X2] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();   // S / L1 / L2 are wave-private: LDS ops of one wave execute in order
    if (info.levels > 3) {
      const int h3 = info.hl[3], w3 = info.wl[3];
      float* p3 = pyr + info.off[3] + static_cast<long long>(b) * N * h3 * w3;
      const int row = lane >> 1, X = lane & 1;
      const float* q = L2 + row * 8 + 2 * X;
      const float v = (((q[0] + q[1]) + q[4]) + q[5]) * 0.25f;
      const int i = iw + row, Y3 = (y0 >> 3), X3 = (x0 >> 3) + X;
      if (i < NmP && Y3 < h3 && X3 < w3) p3[(static_cast<long long>(i) * h3 + Y3) * w3 + X3] = v;
    }
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
  pyramid_epilogue<ALIGNED>(acc[0], acc[1], acc[2], acc[3], smem, pyr, info, b, N, h, w, i0, y0, x0, scale, wave, lane, patch, n_patch);
}

// ------------------------------------------------------------------------------------------------------------------
// fp16x3 variant: the same tiling and epilogue, operands split into fp16 hi + lo halves and multiplied with three
// v_mfma_f32_32x32x16_f16 per k-slab (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulation) -- fp32-class accuracy (the
// dropped lo*lo term is 2^-22 relative) at 3/16 of the matrix-pipe time of the fp32 MFMA, which turns the build from
// MFMA-bound (62 % of 157 TF) into an HBM-write-bound kernel.  Operands arrive pixel-major (NHWC: (B, h*w, C)), the
// layout the encoder engine produces, so that a lane's 8 consecutive-k fragment is one 16-byte LDS read.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr int HBK = 32;     // channels per k-slab
constexpr int HRS = 40;     // LDS row stride in halfs (80 bytes: conflict-free ds_read_b128 over 32 consecutive rows)

__device__ __forceinline__ void split4_h3(const float4 v, float s, h4& hi, h4& lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float c = fminf(fmaxf(x[i], -65504.f), 65504.f);
    const _Float16 hh = static_cast<_Float16>(c);
    hi[i] = hh;
    lo[i] = static_cast<_Float16>(c - static_cast<float>(hh));
  }
}

// pre-pass: fp32 features -> SPLIT tensors (rnnpose_hip.h: per 8-channel group 16 bytes of fp16 hi then 16 bytes of fp16 lo,
// pixel-major, the same 4*C bytes per pixel as the fp32 map), scaled by a_scale, saturated.  Both maps in one launch
// (blockIdx.z = map * B + b).  layout 0: source is (B, C, N) (NCHW, transposed through a 32 x 33 LDS tile); 1: (B, N, C).
__global__ __launch_bounds__(256) void split_features_kernel(const float* __restrict__ src1, const float* __restrict__ src2,
                                                             _Float16* __restrict__ dst1, _Float16* __restrict__ dst2, int B,
                                                             int C, int N, int layout, float a_scale, unsigned long long* sat) {
  const int which = blockIdx.z / B, b = blockIdx.z - which * B;
  const float* src = which ? src2 : src1;
  _Float16* dst = which ? dst2 : dst1;
  if (layout == 1) {
    const long long total8 = static_cast<long long>(N) * C / 8;
    const long long i = (static_cast<long long>(blockIdx.y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    if (i >= total8) return;
    const float4* s4 = reinterpret_cast<const float4*>(src + static_cast<long long>(b) * N * C) + 2 * i;
    const float4 v0 = s4[0], v1 = s4[1];
    h4 h0, l0, h1, l1;
    split4_h3(v0, a_scale, h0, l0);
    split4_h3(v1, a_scale, h1, l1);
    if (sat && (rp::quad_saturates(v0, a_scale) || rp::quad_saturates(v1, a_scale))) atomicAdd(sat, 1ull);   // range guard
    h4* d = reinterpret_cast<h4*>(dst + (static_cast<long long>(b) * N * C) * 2) + 4 * i;
    d[0] = h0; d[1] = h1; d[2] = l0; d[3] = l1;
    return;
  }
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r, n = n0 + tx;
    tile[ty + 8 * r][tx] = (c < C && n < N) ? src[(static_cast<long long>(b) * C + c) * N + n] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + ty + 8 * r, c = c0 + tx;
    if (n < N && c < C) {
      const float raw = tile[tx][ty + 8 * r] * a_scale;
      if (sat && !(fabsf(raw) <= 65504.f)) atomicAdd(sat, 1ull);           // range guard (f16x3.cuh)
      const float x = fminf(fmaxf(raw, -65504.f), 65504.f);
      const _Float16 h = static_cast<_Float16>(x);
      _Float16* row = dst + (static_cast<long long>(b) * N + n) * C * 2 + (c >> 3) * 16 + (c & 7);
      row[0] = h;
      row[8] = static_cast<_Float16>(x - static_cast<float>(h));
    }
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Operands are SPLIT tensors: a pixel's k-slab of 32 channels is 128 contiguous bytes (4 groups x [hi 16 B | lo 16 B]).
//
// What bounds it (profiles/r03_corr_ablation.txt, r03_corr_store_patterns.txt, r03_corr_variants_not_kept.txt; B = 8, 60 x 80,
// C = 256: 475-490 us per launch): an empty pass (no K loop, no stores) 70 us, the stores alone +160 us, the K loop alone +270 us
// (fragment reads + MFMA + two barriers per slab 150 -- 75 % of the matrix pipe --, refilling LDS +30, requesting operands +60,
// the epilogue's transposes +25), and the SUM is what is measured.  The K loop reaches that rate only with all four resident
// workgroups of a CU in it (they hide each other's LDS latencies and barriers); a workgroup sitting in its epilogue until the
// memory system has taken its stores is one fewer, so the two phases hardly overlap.  The store phase itself: 978 MB as 64-byte
// row pieces (128 rows x 8 image rows per tile) go at 3.6-4.3 TB/s whatever the instruction shape, also as 128- or 256-byte
// pieces; one workgroup writing a whole 8-row stripe of an i row (2560 contiguous bytes) 5.2-6.0, a linear fill 5.9-6.5.
// Tried, measured slower or equal, not kept: a second register set (requests two slabs ahead: equal, costs a workgroup per CU);
// non-temporal stores for the pooled levels too (+30 %: their 8-32 byte pieces then miss L2's merging); workgroups walking a
// contiguous range of the tile list so that one workgroup writes whole stripes (+45 %: a tile's first operands cannot be
// consumed before the previous tile's stores are acknowledged -- gfx9 counts both on vmcnt).  Kept: non-temporal level-0 stores
// (-13 %), operands pre-split by their producer (no pre-pass).  Next: a K loop that is fast on its own (double-buffered LDS,
// one barrier per slab), then stripe-wide tiles.
#ifndef RPC_MINW
#define RPC_MINW 2        // waves per SIMD the fp16x3 kernel is compiled for (measurement: 3, 4)
#endif
template <bool ALIGNED>
__global__ __launch_bounds__(NT, RPC_MINW) void corr_pyramid_h3_kernel(const _Float16* __restrict__ f1, const _Float16* __restrict__ f2,
                                                                float* __restrict__ pyr, int B, int C, int h, int w, int n_it,
                                                                int n_py, int n_px, float scale, PyrInfo info, int sti, int stp) {
  // [A | B][hi | lo][128 rows x HRS] halfs = 40 KiB per operand buffer; the epilogue's 36 KiB of float staging aliases it.
  // RPC_DB = 0 (default): ONE buffer, the next slab waits in registers, two barriers per slab (MFMA | barrier | store | barrier), four
  // workgroups per CU.  RPC_DB = 1 (r06; measured slower, see the macro): TWO buffers, the next slab is stored while
  // the current one is being multiplied, ONE barrier per slab, two workgroups per CU (80 KiB).
  constexpr int BUFH = 2 * 2 * 128 * HRS;
  __shared__ __attribute__((aligned(16))) _Float16 sT[(RPC_DB ? 2 : 1) * BUFH];
  float* smem = reinterpret_cast<float*>(sT);
  const int N = h * w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  const int ntiles = gridDim.x;
  int bid = blockIdx.x;
  {
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  // Within an image, tiles are ordered in SUPERTILES of sti i tiles x stp patches (host: the smallest even split of the image with
  // at most 20 of each, 19 x 20 at 60 x 80): the tiles an XCD has in flight share sti + stp operand panels of 64 KB (2.5 MB of
  // its 4 MB L2).  8 x 8 supertiles (r01-r02) re-fetched every panel 4 times: 326 MB per launch for 79 MB of operands.
  const int n_patch = n_py * n_px;
  const int n_ps = (n_patch + stp - 1) / stp, n_is = (n_it + sti - 1) / sti;
  const int sts = sti * stp;
  const int per_img = n_ps * n_is * sts;
  const int b = bid / per_img;
  const int tloc = bid - b * per_img;
  const int sidx = tloc / sts, within = tloc - sidx * sts;
  const int it = (sidx / n_ps) * sti + within / stp;
  const int patch = (sidx % n_ps) * stp + within % stp;
  if (it >= n_it || patch >= n_patch) return;
  const int i0 = it * BM;
  const int y0 = (patch / n_px) * PY;
  const int x0 = (patch % n_px) * PX;

  // per-thread staging: rows (tid >> 2) + 64 r (r = 0, 1), channel group q = tid & 3 of the slab (32 bytes: hi then lo).
  // Offsets in halfs (< 2^31, host check); rows outside the problem read element 0 and are masked to zero.
  const int q = tid & 3;
  const unsigned rs = 2u * C;                       // halfs per pixel
  unsigned offa0, offa1, offb0, offb1, ma0, ma1, mb0, mb1;
  {
    const int r0 = tid >> 2, r1 = r0 + 64;
    const bool va0 = i0 + r0 < N, va1 = i0 + r1 < N;
    offa0 = va0 ? (static_cast<unsigned>(b) * N + i0 + r0) * rs + q * 16 : 0u;
    offa1 = va1 ? (static_cast<unsigned>(b) * N + i0 + r1) * rs + q * 16 : 0u;
    const int ya = y0 + (r0 >> 4), xa = x0 + (r0 & 15), yb = y0 + (r1 >> 4), xb = x0 + (r1 & 15);
    const bool vb0 = ya < h && xa < w, vb1 = yb < h && xb < w;
    offb0 = vb0 ? (static_cast<unsigned>(b) * N + ya * w + xa) * rs + q * 16 : 0u;
    offb1 = vb1 ? (static_cast<unsigned>(b) * N + yb * w + xb) * rs + q * 16 : 0u;
    ma0 = va0 ? 0xffffffffu : 0u; ma1 = va1 ? 0xffffffffu : 0u;
    mb0 = vb0 ? 0xffffffffu : 0u; mb1 = vb1 ? 0xffffffffu : 0u;
  }

  f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }

  u32x4 rah0, rah1, ral0, ral1, rbh0, rbh1, rbl0, rbl1;
#define RPH_LOAD(K0_)                                                                          \
  do {                                                                                         \
    rah0 = *reinterpret_cast<const u32x4*>(f1 + offa0 + (K0_));                                \
    ral0 = *reinterpret_cast<const u32x4*>(f1 + offa0 + (K0_) + 8);                            \
    rah1 = *reinterpret_cast<const u32x4*>(f1 + offa1 + (K0_));                                \
    ral1 = *reinterpret_cast<const u32x4*>(f1 + offa1 + (K0_) + 8);                            \
    rbh0 = *reinterpret_cast<const u32x4*>(f2 + offb0 + (K0_));                                \
    rbl0 = *reinterpret_cast<const u32x4*>(f2 + offb0 + (K0_) + 8);                            \
    rbh1 = *reinterpret_cast<const u32x4*>(f2 + offb1 + (K0_));                                \
    rbl1 = *reinterpret_cast<const u32x4*>(f2 + offb1 + (K0_) + 8);                            \
  } while (0)
  // plane p (0 = A hi, 1 = A lo, 2 = B hi, 3 = B lo) x 128 rows x HRS halfs
#define RPH_ST(V_, M_, P_, R_, BO_) *reinterpret_cast<u32x4*>(sT + (BO_) + (P_) * (128 * HRS) + ((tid >> 2) + 64 * (R_)) * HRS + q * 8) = (V_) & (M_)
#define RPH_STORE(BO_)                                                                         \
  do {                                                                                         \
    RPH_ST(rah0, ma0, 0, 0, BO_); RPH_ST(rah1, ma1, 0, 1, BO_); RPH_ST(ral0, ma0, 1, 0, BO_); RPH_ST(ral1, ma1, 1, 1, BO_); \
    RPH_ST(rbh0, mb0, 2, 0, BO_); RPH_ST(rbh1, mb1, 2, 1, BO_); RPH_ST(rbl0, mb0, 3, 0, BO_); RPH_ST(rbl1, mb1, 3, 1, BO_); \
  } while (0)
#define RPH_MFMA(BO_)                                                                          \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                           \
    const h8 ah = *reinterpret_cast<const h8*>(sAh + (BO_) + kk * 16);                         \
    const h8 al = *reinterpret_cast<const h8*>(sAh + (BO_) + 128 * HRS + kk * 16);             \
    h8 bh[4], bl[4];                                                                           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                            \
      bh[s] = *reinterpret_cast<const h8*>(sBh + (BO_) + s * 32 * HRS + kk * 16);              \
      bl[s] = *reinterpret_cast<const h8*>(sBh + (BO_) + 128 * HRS + s * 32 * HRS + kk * 16);  \
    }                                                                                          \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], acc0, 0, 0, 0);                   \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], acc1, 0, 0, 0);                   \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[2], acc2, 0, 0, 0);                   \
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[3], acc3, 0, 0, 0);                   \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], acc0, 0, 0, 0);                   \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], acc1, 0, 0, 0);                   \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[2], acc2, 0, 0, 0);                   \
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[3], acc3, 0, 0, 0);                   \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], acc0, 0, 0, 0);                   \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], acc1, 0, 0, 0);                   \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[2], acc2, 0, 0, 0);                   \
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[3], acc3, 0, 0, 0);                   \
  }

  RPH_LOAD(0);
  RPH_STORE(0);
  __syncthreads();
  const int nk = (RPC_ABL & 4) ? 0 : C / HBK;
  const _Float16* sAh = sT + (wave * 32 + l31) * HRS + lh * 8;
  const _Float16* sBh = sT + 2 * 128 * HRS + l31 * HRS + lh * 8;
  if (RPC_DB) {
    for (int kt = 1; kt < nk; ++kt) {                // slab kt travels, then lands in the OTHER buffer, while slab kt - 1 is multiplied
      const int cur = ((kt - 1) & 1) * BUFH, nxt = (kt & 1) * BUFH;
      if (!(RPC_ABL & 1)) RPH_LOAD(kt * (2 * HBK));
      RPH_MFMA(cur)
      if (!(RPC_ABL & 16)) RPH_STORE(nxt);
      __syncthreads();                               // slab kt is visible; nobody reads slab kt - 1 any more (its buffer is refilled next turn)
    }
    if (nk > 0) {
      RPH_MFMA(((nk - 1) & 1) * BUFH)
      __syncthreads();                               // every wave is done with the operands: the epilogue re-uses their LDS
    }
  } else {
    for (int kt = 1; kt < nk; ++kt) {                // slab kt travels while slab kt - 1 is multiplied
      if (!(RPC_ABL & 1)) RPH_LOAD(kt * (2 * HBK));
      RPH_MFMA(0)
      __syncthreads();
      if (!(RPC_ABL & 16)) RPH_STORE(0);
      __syncthreads();
    }
    if (nk > 0) {
      RPH_MFMA(0)
      __syncthreads();                               // every wave is done with the operands: the epilogue re-uses their LDS
    }
  }
  if (RPC_ABL & 8) {
    if (scale == 1234.5f) pyr[tid] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
    return;
  }
#undef RPH_MFMA
#undef RPH_LOAD
#undef RPH_STORE
#undef RPH_ST
  pyramid_epilogue<ALIGNED>(acc0, acc1, acc2, acc3, smem, pyr, info, b, N, h, w, i0, y0, x0, scale, wave, lane, patch, n_patch);
}


// ------------------------------------------------------------------------------------------------------------------
// r06: the same tile, products and epilogue with the operands requested by LDS-DMA (global_load_lds_dwordx4: global memory -> LDS, no
// staging registers, no ds_write).  Why: per 32-channel slab a workgroup moved 32 KB through ds_write_b128 -- ~13 LDS cycles per
// wave-instruction (MI355X_MICROARCH.md, LDS: the address / data transfer of a wide store, not the array, sets its cost: ~79 B/clk per
// CU) -- which with four workgroups per CU is ~1660 LDS cycles per slab round next to ~1300 for the fragment reads and 3072 MFMA cycles
// per SIMD: the LDS was a co-limiter of the K loop (75 % of the pipe with all four workgroups in it), and the 32 staging registers
// were what kept a second slab from being in flight.  Here:
//   * slabs of 16 channels, DOUBLE-buffered: [A hi | A lo | B hi | B lo] x 128 rows x 32 bytes = 16 KB per buffer, 32 KB per workgroup
//     (the epilogue's 36 KB of staging alias them as before: four workgroups per CU); slab k + 1 is requested before slab k's MFMAs,
//     ONE barrier per slab (16 per tile, as many as the single-buffered form's 2 x 8);
//   * a row is 32 bytes (two 8-channel groups), stored with a 1-bit XOR swizzle (chunk ^ bit 3 of the row) so that the 16 lanes of a
//     ds_read_b128 group (rows {0-3, 12-15, 20-27} of a 32-row fragment) hit 16 different 16-byte units.  The DMA image is
//     lane-linear (lane j -> LDS bytes [16 j, 16 j + 16) of its 1-KB piece), so the swizzle is applied to the SOURCE address: lane j
//     fetches chunk (j & 1) ^ ((j >> 4) & 1) of row j >> 1;
//   * wave w requests rows [32 w, 32 w + 32) of all four planes: 4 requests per wave and slab, addresses formed once per tile and
//     advanced by 64 bytes per slab; rows outside the problem read a zero page (stride 0);
//   * the same k order and the same three products per k block as the register form: results are BIT-IDENTICAL to it.
constexpr int DBK = 16;                       // channels per slab
constexpr int DPL = 128 * 32;                 // bytes per plane
constexpr int DBUF = 4 * DPL;                 // bytes per slab buffer
__device__ __attribute__((aligned(64))) const unsigned char g_cp_zero_page[64] = {0};

// (as in conv_strip_kernel.cuh: inline asm, because hipcc models the builtin as a flat access and then waits lgkmcnt(0) / vmcnt(0)
//  everywhere; M0 = the request's LDS base is saved and restored inside the statement)
__device__ __forceinline__ void cp_glds16(const void* g, unsigned dst) {
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
__device__ __forceinline__ void cp_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <bool ALIGNED>
__global__ __launch_bounds__(NT, 4) void corr_pyramid_h3dma_kernel(const _Float16* __restrict__ f1, const _Float16* __restrict__ f2,
                                                                   float* __restrict__ pyr, int B, int C, int h, int w, int n_it,
                                                                   int n_py, int n_px, float scale, PyrInfo info, int sti, int stp) {
  __shared__ __attribute__((aligned(1024))) unsigned char sD[4 * 2304 * 4];      // 36 KB: two 16-KB slab buffers; the epilogue's staging aliases them
  float* smem = reinterpret_cast<float*>(sD);
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)sD));   // LDS byte address
  const int N = h * w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  const int ntiles = gridDim.x;
  int bid = blockIdx.x;
  {
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
  const int n_patch = n_py * n_px;
  const int n_ps = (n_patch + stp - 1) / stp, n_is = (n_it + sti - 1) / sti;
  const int sts = sti * stp;
  const int per_img = n_ps * n_is * sts;
  const int b = bid / per_img;
  const int tloc = bid - b * per_img;
  const int sidx = tloc / sts, within = tloc - sidx * sts;
  const int it = (sidx / n_ps) * sti + within / stp;
  const int patch = (sidx % n_ps) * stp + within % stp;
  if (it >= n_it || patch >= n_patch) return;
  const int i0 = it * BM;
  const int y0 = (patch / n_px) * PY;
  const int x0 = (patch % n_px) * PX;

  // request addresses of this lane: tile row 32 wave + (lane >> 1), chunk (lane & 1) ^ bit 3 of the row
  const unsigned char* pa;
  const unsigned char* pb;
  unsigned inca, incb;
  {
    const int rr = lane >> 1, ch = (lane & 1) ^ ((rr >> 3) & 1);
    const int ra = wave * 32 + rr;
    const unsigned rsb = 4u * C;                      // bytes per pixel of a split tensor
    const bool va = i0 + ra < N;
    const int yb = y0 + (ra >> 4), xb = x0 + (ra & 15);
    const bool vb = yb < h && xb < w;
    pa = va ? reinterpret_cast<const unsigned char*>(f1) + (static_cast<size_t>(b) * N + i0 + ra) * rsb + ch * 32 : g_cp_zero_page;
    pb = vb ? reinterpret_cast<const unsigned char*>(f2) + (static_cast<size_t>(b) * N + yb * w + xb) * rsb + ch * 32 : g_cp_zero_page;
    inca = va ? 64u : 0u;
    incb = vb ? 64u : 0u;
  }
  const unsigned dw = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);      // (wave-uniform: the request takes its LDS base from M0)
#define RPD_REQ(BO_)                                       \
  do {                                                     \
    cp_glds16(pa, __builtin_amdgcn_readfirstlane(dw + (BO_)));                   \
    cp_glds16(pa + 16, __builtin_amdgcn_readfirstlane(dw + (BO_) + DPL));        \
    cp_glds16(pb, __builtin_amdgcn_readfirstlane(dw + (BO_) + 2 * DPL));         \
    cp_glds16(pb + 16, __builtin_amdgcn_readfirstlane(dw + (BO_) + 3 * DPL));    \
    pa += inca;                                            \
    pb += incb;                                            \
  } while (0)

  f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }

  const int sw = (lh ^ ((l31 >> 3) & 1)) * 16;
  const unsigned char* sA = sD + (wave * 32 + l31) * 32 + sw;
  const unsigned char* sB = sD + 2 * DPL + l31 * 32 + sw;
#define RPD_MFMA(BO_)                                                                          \
  {                                                                                            \
    const h8 ah = *reinterpret_cast<const h8*>(sA + (BO_));                                    \
    const h8 al = *reinterpret_cast<const h8*>(sA + (BO_) + DPL);                              \
    h8 bh[4], bl[4];                                                                           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                            \
      bh[s] = *reinterpret_cast<const h8*>(sB + (BO_) + s * 1024);                             \
      bl[s] = *reinterpret_cast<const h8*>(sB + (BO_) + DPL + s * 1024);                       \
    }                                                                                          \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], acc0, 0, 0, 0);                   \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], acc1, 0, 0, 0);                   \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[2], acc2, 0, 0, 0);                   \
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[3], acc3, 0, 0, 0);                   \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], acc0, 0, 0, 0);                   \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], acc1, 0, 0, 0);                   \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[2], acc2, 0, 0, 0);                   \
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[3], acc3, 0, 0, 0);                   \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], acc0, 0, 0, 0);                   \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], acc1, 0, 0, 0);                   \
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[2], acc2, 0, 0, 0);                   \
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[3], acc3, 0, 0, 0);                   \
  }

  const int nk = C / DBK;
  RPD_REQ(0);
  cp_wait_vm0();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = (kt & 1) * DBUF;
    // slab kt + 1 -> the other buffer: every wave finished reading slab kt - 1 from it before the barrier that ended the last turn
    if (kt + 1 < nk) RPD_REQ(DBUF - cur);
    RPD_MFMA(cur)
    cp_wait_vm0();                                   // my pieces of slab kt + 1 have landed ...
    __syncthreads();                                 // ... and so have everyone's; nobody reads slab kt any more
  }
#undef RPD_MFMA
#undef RPD_REQ
  pyramid_epilogue<ALIGNED>(acc0, acc1, acc2, acc3, smem, pyr, info, b, N, h, w, i0, y0, x0, scale, wave, lane, patch, n_patch);
}


// ------------------------------------------------------------------------------------------------------------------
// r06, variant 2: WAVE-SPECIALISED and PERSISTENT -- the form that lets the store phase of a tile run under the K loop of the next one.
// What bounds variants 0 and 1 is the SUM of a K-loop phase and an HBM-write-bound store phase per workgroup (978 MB per launch): a workgroup
// in its epilogue holds a slot and computes nothing.  A persistent workgroup could fire its stores and go on to the next tile -- but its next
// operand requests then share vmcnt with those stores (gfx9 counts loads and stores on ONE counter, and they return out of order relative to
// each other, so only vmcnt(0) is safe), and the K loop waits for store acknowledgements of a saturated write queue: r03's persistent form,
// +45 %.  Here the two kinds of request live in DIFFERENT WAVES:
//   * waves 4-7 (LOADERS) do nothing but request operand slabs by LDS-DMA -- loader w the rows [32 w, 32 w + 32) of the four planes, 4 requests
//     per 16-channel slab -- wait for them (their vmcnt counts only loads) and meet the compute waves at one barrier per slab, one slab ahead
//     (double-buffered 2 x 16 KB as in variant 1);
//   * waves 0-3 (COMPUTE) read fragments, multiply, and run the epilogue of a finished tile -- staging LDS of their own (36 KB, NOT aliasing the
//     operand buffers: the loaders are already filling them for the next tile), stores fired and never waited for -- then start the next tile;
//   * the workgroup walks its share of the tile list (grid = 2 workgroups per CU; the tiles an XCD has in flight stay neighbours in the
//     supertile order, so the operand panels stay in its L2): no workgroup dispatch per tile (the 'empty pass' of the one-tile forms was 70 us).
// LDS 68 KB: two workgroups per CU = two compute + two loader waves per SIMD, <= 128 registers.  Same k order, same products, same epilogue:
// BIT-IDENTICAL results.  Barrier protocol: barrier s = "slab s has landed and everyone is done with slab s - 1"; a loader requests slab s into
// buffer s & 1 after barrier s - 1 (the compute waves arrived there after multiplying slab s - 2, the buffer's previous content), waits, arrives
// at barrier s; a compute wave passes barrier s and multiplies slab s.  Slabs are counted across tiles; the epilogue contains no barrier.
template <bool ALIGNED>
__global__ __launch_bounds__(2 * NT, 4) void corr_pyramid_h3ws_kernel(const _Float16* __restrict__ f1, const _Float16* __restrict__ f2,
                                                                      float* __restrict__ pyr, int B, int C, int h, int w, int n_it,
                                                                      int n_py, int n_px, float scale, PyrInfo info, int sti, int stp, int ntiles) {
  __shared__ __attribute__((aligned(1024))) unsigned char sD[2 * DBUF + 4 * 2304 * 4];      // operand slabs | epilogue staging
  float* smem = reinterpret_cast<float*>(sD + 2 * DBUF);
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)sD));   // LDS byte address
  const int N = h * w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const int cw = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;

  // this workgroup's share of the tile list: XCD x (= blockIdx % 8) owns the contiguous chunk of ids the one-tile forms give it; its
  // gridDim / 8 workgroups walk the chunk interleaved, so that at any time they are on neighbouring tiles
  const int nslots = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per = ntiles >> 3, rem = ntiles & 7;
  const int t_begin = xcd * per + (xcd < rem ? xcd : rem), t_count = per + (xcd < rem ? 1 : 0);
  const int n_patch = n_py * n_px;
  const int n_ps = (n_patch + stp - 1) / stp, n_is = (n_it + sti - 1) / sti;
  const int sts = sti * stp;
  const int per_img = n_ps * n_is * sts;
  const int nk = C / DBK;
  const unsigned dw = __builtin_amdgcn_readfirstlane(lds0 + cw * 1024);
  const int sw = (lh ^ ((l31 >> 3) & 1)) * 16;
  const unsigned char* sA = sD + (cw * 32 + l31) * 32 + sw;
  const unsigned char* sB = sD + 2 * DPL + l31 * 32 + sw;
  unsigned par = 0;                                   // parity of the running slab count = operand buffer of the next slab

  for (int ti = slot; ti < t_count; ti += nslots) {
    const int bid = t_begin + ti;
    const int b = bid / per_img;
    const int tloc = bid - b * per_img;
    const int sidx = tloc / sts, within = tloc - sidx * sts;
    const int it = (sidx / n_ps) * sti + within / stp;
    const int patch = (sidx % n_ps) * stp + within % stp;
    if (it >= n_it || patch >= n_patch) continue;     // padding tile of a partial supertile (uniform over the workgroup)
    const int i0 = it * BM;
    const int y0 = (patch / n_px) * PY;
    const int x0 = (patch % n_px) * PX;
    if (loader) {
      const unsigned char* pa;
      const unsigned char* pb;
      unsigned inca, incb;
      {
        const int rr = lane >> 1, ch = (lane & 1) ^ ((rr >> 3) & 1);
        const int ra = cw * 32 + rr;
        const unsigned rsb = 4u * C;
        const bool va = i0 + ra < N;
        const int yb = y0 + (ra >> 4), xb = x0 + (ra & 15);
        const bool vb = yb < h && xb < w;
        pa = va ? reinterpret_cast<const unsigned char*>(f1) + (static_cast<size_t>(b) * N + i0 + ra) * rsb + ch * 32 : g_cp_zero_page;
        pb = vb ? reinterpret_cast<const unsigned char*>(f2) + (static_cast<size_t>(b) * N + yb * w + xb) * rsb + ch * 32 : g_cp_zero_page;
        inca = va ? 64u : 0u;
        incb = vb ? 64u : 0u;
      }
      for (int kt = 0; kt < nk; ++kt) {
        const unsigned bo = par * DBUF;
        cp_glds16(pa, __builtin_amdgcn_readfirstlane(dw + bo));
        cp_glds16(pa + 16, __builtin_amdgcn_readfirstlane(dw + bo + DPL));
        cp_glds16(pb, __builtin_amdgcn_readfirstlane(dw + bo + 2 * DPL));
        cp_glds16(pb + 16, __builtin_amdgcn_readfirstlane(dw + bo + 3 * DPL));
        pa += inca;
        pb += incb;
        par ^= 1u;
        cp_wait_vm0();
        __syncthreads();
      }
    } else {
      f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
      for (int kt = 0; kt < nk; ++kt) {
        const unsigned bo = par * DBUF;
        par ^= 1u;
        __syncthreads();
        const h8 ah = *reinterpret_cast<const h8*>(sA + bo);
        const h8 al = *reinterpret_cast<const h8*>(sA + bo + DPL);
        h8 bh[4], bl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bh[s] = *reinterpret_cast<const h8*>(sB + bo + s * 1024);
          bl[s] = *reinterpret_cast<const h8*>(sB + bo + DPL + s * 1024);
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[3], acc3, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[3], acc3, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[3], acc3, 0, 0, 0);
      }
      pyramid_epilogue<ALIGNED>(acc0, acc1, acc2, acc3, smem, pyr, info, b, N, h, w, i0, y0, x0, scale, cw, lane, patch, n_patch);
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
    // level 0: j-patch-major, whole 8 x 16 patches (see pyramid_epilogue); levels 1..: (B*h*w, hl, wl) row-major
    off += l == 0 ? rows * rp::cdiv(h, PY) * rp::cdiv(w, PX) * BN : rows * hl * wl;
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

size_t rnnpose_corr_pyramid_f16x3_workspace_bytes(int B, int C, int h, int w) {
  if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
  return static_cast<size_t>(B) * h * w * C * sizeof(_Float16) * 4;      // two split tensors (fmap1, fmap2)
}

namespace {
int g_corr_supertile = 20;      // (rnnpose_corr_supertile: measurement)
int g_corr_variant = 0;         // (rnnpose_corr_variant)
int launch_h3(const char* fn, const _Float16* f1, const _Float16* f2, int B, int C, int h, int w, int levels, float a_scale,
              float* pyramid, hipStream_t st) {
  RP_REQUIRE(C > 0 && C % HBK == 0, fn, "C must be a positive multiple of 32");
  RP_REQUIRE(a_scale > 0.f, fn, "a_scale must be positive");
  int64_t offs[RNNPOSE_MAX_LEVELS + 1];
  PyrInfo info{};
  if (int e = rnnpose_corr_pyramid_layout(B, h, w, levels, offs, info.hl, info.wl)) return e;
  for (int l = 0; l < levels; ++l) info.off[l] = offs[l];
  info.levels = levels;
  const int N = h * w;
  RP_REQUIRE(B < 65536 && 2LL * B * N * C < (1LL << 31), fn, "feature maps too large for 32-bit element offsets");
  const int n_it = rp::cdiv(N, BM), n_py = rp::cdiv(h, PY), n_px = rp::cdiv(w, PX);
  // supertile = the smallest even split of the image's (i tiles) x (patches) with at most g_corr_supertile of each
  const int n_patch = n_py * n_px;
  const int sti = rp::cdiv(n_it, rp::cdiv(n_it, g_corr_supertile)), stp = rp::cdiv(n_patch, rp::cdiv(n_patch, g_corr_supertile));
  const long long ntiles = static_cast<long long>(B) * rp::cdiv(n_it, sti) * rp::cdiv(n_patch, stp) * sti * stp;
  RP_REQUIRE(ntiles < (1LL << 31), fn, "grid too large");
  const float scale = 1.0f / (sqrtf(static_cast<float>(C)) * a_scale * a_scale);
  const bool aligned = (N % 4 == 0) && (w % 4 == 0) && (reinterpret_cast<uintptr_t>(pyramid) % 16 == 0);
  dim3 grid(static_cast<unsigned>(ntiles)), block(NT);
  if (g_corr_variant == 2) {          // wave-specialised persistent form (r06): loaders + compute waves, stores never waited for
    static int ncu_cached = 0;          // (CUs of the current device, asked once: one process drives one GPU)
    if (ncu_cached == 0) {
      int dev = 0, v = 0;
      ncu_cached = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    const int ncu = ncu_cached;
    const unsigned nwg = static_cast<unsigned>(((2 * ncu + 7) / 8) * 8);
    if (aligned) {
      hipLaunchKernelGGL(corr_pyramid_h3ws_kernel<true>, dim3(nwg), dim3(2 * NT), 0, st, f1, f2, pyramid, B, C, h, w, n_it, n_py, n_px, scale, info, sti, stp,
                         static_cast<int>(ntiles));
    } else {
      hipLaunchKernelGGL(corr_pyramid_h3ws_kernel<false>, dim3(nwg), dim3(2 * NT), 0, st, f1, f2, pyramid, B, C, h, w, n_it, n_py, n_px, scale, info, sti, stp,
                         static_cast<int>(ntiles));
    }
  } else if (g_corr_variant == 1) {          // operands by LDS-DMA (r06), bit-identical to the register form
    if (aligned) {
      hipLaunchKernelGGL(corr_pyramid_h3dma_kernel<true>, grid, block, 0, st, f1, f2, pyramid, B, C, h, w, n_it, n_py, n_px, scale, info, sti, stp);
    } else {
      hipLaunchKernelGGL(corr_pyramid_h3dma_kernel<false>, grid, block, 0, st, f1, f2, pyramid, B, C, h, w, n_it, n_py, n_px, scale, info, sti, stp);
    }
  } else if (aligned) {
    hipLaunchKernelGGL(corr_pyramid_h3_kernel<true>, grid, block, 0, st, f1, f2, pyramid, B, C, h, w, n_it, n_py, n_px, scale, info, sti, stp);
  } else {
    hipLaunchKernelGGL(corr_pyramid_h3_kernel<false>, grid, block, 0, st, f1, f2, pyramid, B, C, h, w, n_it, n_py, n_px, scale, info, sti, stp);
  }
  return rp::check_launch(fn);
}
}  // namespace

int rnnpose_corr_variant(int variant) {       // 0: operands through registers + ds_write (r03-r05); 1: operands by LDS-DMA (r06); 2: loaders + persistent compute waves (r06)
  if (variant < 0 || variant > 2) {
    rp::set_error("rnnpose_corr_variant: 0 (register staging), 1 (LDS-DMA) or 2 (wave-specialised, persistent)");
    return 1;
  }
  g_corr_variant = variant;
  return 0;
}

int rnnpose_corr_supertile(int max_side) {       // measurement: side limit of the fp16x3 kernel's tile-order supertiles (default 20; 8 = r02)
  if (max_side < 1 || max_side > 64) {
    rp::set_error("rnnpose_corr_supertile: max_side 1..64");
    return 1;
  }
  g_corr_supertile = max_side;
  return 0;
}

int rnnpose_corr_pyramid_f16x3(const float* fmap1, const float* fmap2, int layout, int B, int C, int h, int w, int levels,
                               float a_scale, void* workspace, size_t workspace_bytes, float* pyramid,
                               rnnpose_stream_t stream) {
  const char* fn = "rnnpose_corr_pyramid_f16x3";
  RP_REQUIRE(fmap1 && fmap2 && pyramid && workspace, fn, "null pointer");
  RP_REQUIRE(layout == 0 || layout == 1, fn, "layout must be 0 (B,C,h,w) or 1 (B,h,w,C)");
  RP_REQUIRE(C > 0 && C % HBK == 0, fn, "C must be a positive multiple of 32");
  RP_REQUIRE(a_scale > 0.f, fn, "a_scale must be positive");
  RP_REQUIRE(B > 0 && 2 * B < 65536, fn, "B out of range");
  RP_REQUIRE(workspace_bytes >= rnnpose_corr_pyramid_f16x3_workspace_bytes(B, C, h, w) &&
                 reinterpret_cast<uintptr_t>(workspace) % 16 == 0, fn, "workspace too small or not 16-byte aligned");
  RP_REQUIRE(layout == 0 || (reinterpret_cast<uintptr_t>(fmap1) % 16 == 0 && reinterpret_cast<uintptr_t>(fmap2) % 16 == 0), fn,
             "pixel-major feature maps must be 16-byte aligned");
  const int N = h * w;
  hipStream_t st = rp::as_stream(stream);
  const size_t plane = static_cast<size_t>(B) * N * C * 2;             // halfs per split tensor
  _Float16* ws = static_cast<_Float16*>(workspace);
  dim3 sg = layout == 1 ? dim3(1024, static_cast<unsigned>(rp::cdiv(static_cast<long long>(N) * C / 8, 256 * 1024)), 2 * B)
                        : dim3(rp::cdiv(N, 32), rp::cdiv(C, 32), 2 * B);
  hipLaunchKernelGGL(split_features_kernel, sg, dim3(256), 0, st, fmap1, fmap2, ws, ws + plane, B, C, N, layout, a_scale,
                     rp::sat_counter());
  return launch_h3(fn, ws, ws + plane, B, C, h, w, levels, a_scale, pyramid, st);
}

int rnnpose_corr_pyramid_split(const void* fmap1_split, const void* fmap2_split, int B, int C, int h, int w, int levels,
                               float a_scale, float* pyramid, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_corr_pyramid_split";
  RP_REQUIRE(fmap1_split && fmap2_split && pyramid, fn, "null pointer");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(fmap1_split) % 16 == 0 && reinterpret_cast<uintptr_t>(fmap2_split) % 16 == 0, fn,
             "split feature maps must be 16-byte aligned");
  return launch_h3(fn, static_cast<const _Float16*>(fmap1_split), static_cast<const _Float16*>(fmap2_split), B, C, h, w, levels,
                   a_scale, pyramid, rp::as_stream(stream));
}

}  // extern "C"
