// Shared by the two implicit-GEMM convolution kernel families (conv_igemm.hip: 128-row tiles, 4 waves, weights straight to
// registers; conv_strip.hip: 160-row strips, operands by LDS-DMA): the launch parameter block, which rnnpose_conv2d_nhwc_f16x3
// fills once from the public descriptor, and the split-form output store.
#pragma once
#include "common.hpp"
#include "f16x3.cuh"

namespace rpconv {

constexpr int BK = 32;                  // channels per block (one 128-byte line of an NHWC pixel)
constexpr int MAX_CB = 24;

using rp::h8; using rp::h4; using rp::h2; using rp::f32x2; using rp::f32x16; using rp::u32x2; using rp::split4;

struct Seg {
  const float* ptr;
  int cstride, coff, ccount;
};

struct KParams {
  Seg seg0, seg1, seg2, seg3;       // separate members: a dynamically indexed by-value array would be copied to LDS
  int cb1, cb2, cb3;                // first channel block of segments 1..3 (ncb if absent)
  int ncb;
  int B, U, V, su, sv;              // slow / fast axis extents and pixel strides
  int G, T, du0, dv0;               // groups (slow-axis taps) x taps per group (fast axis)
  int stride, Uin, Vin, gkw, dvg0;  // strided mode (stride 2): no tap sharing, G = kh*kw groups decoded as (g / gkw, g % gkw)
  const uint4* wpk;                 // packed weights: [g][cb][t][32-col tile][hi kk0, hi kk1, lo kk0, lo kk1][lane] x 16 bytes
  int Npad;
  const float* bias;
  int Cout;
  float a_scale, out_scale;
  int epi;
  float* dst;
  int dst_cs, dst_co;
  const float* aux0;
  int aux0_cs, aux0_co;
  const float* aux1;
  int aux1_cs, aux1_co;
  float* dst2;
  int dst2_cs, dst2_co;
  int gru_c;
  double* tstats;  // optional per-tile column statistics (linear epilogue), fp64: sum, sum of squares
  const float* addm;   // optional per-pixel bias map (NHWC), added before the epilogue
  int addm_cs, addm_co;
  int tpi;             // > 0: M is tiled PER IMAGE (tpi tiles of 128 rows each, the last one ragged): no tile straddles two images
  const float* in_mr;  // NORM variant: (B, seg0.cstride, 2) mean / rstd of source 0, applied with ReLU while staging
  int n_mt, n_nt;
  unsigned long long* sat;   // fp16x3 range guard: counter of clamped / non-finite activation quads (NULL = check off)
  int sp_tx, sp_ty;          // SPATIAL kernels: 16-pixel-wide / 8-pixel-high patches per image row / column
  int dst_hl, dst2_hl;       // dst / dst2 receive the PRE-SPLIT fp16 hi|lo form (rnnpose_hip.h, "split tensors") instead of fp32
  float* dsth;               // optional second destination of the primary result, always in split form (GRU: h' as fp32 AND split)
  int dsth_cs, dsth_co;
  int ksplit;                // > 1: gridDim.x = tiles * ksplit, workgroup (tile, s) multiplies channel blocks [nit s / ksplit, nit (s + 1) / ksplit)
  float* ks_ws;              // ksplit partial accumulators: [tile][split][acc register][thread] floats
  unsigned* ks_cnt;          // per-tile arrival counters (zero between launches)
  const uint4* wpk_strip;    // the same weights in the strip kernel's order (conv_strip.hip), behind the first copy in w_packed
  int single_product;        // strip kernels, 160-row strips, one column tile per wave: a_hi * b_hi only (cfg.raft.mixed_precision)
  int off32;                 // strip kernels: every destination / epilogue operand spans < 2^32 elements (32-bit offsets in the fast epilogue)
  int stagger;               // persistent strip launches (measurement): 100-MHz ticks of start delay per workgroup slot of a CU
  int n_tiles;               // strip kernels, r06: > 0 = a PERSISTENT launch -- gridDim.x resident workgroups walk n_tiles tiles (conv_strip_kernel.cuh)
};

// One output quad (4 consecutive channels starting at channel ch of the pixel row `row`) in split form: the 8-channel group
// g = ch / 8 occupies 32 bytes = [hi x 8 | lo x 8] fp16; a quad is the 8-byte half (ch / 4) & 1 of each plane.  nv < 4 (the
// ragged tail of a 126-channel layer): only the first nv fp16 of each plane are written -- their neighbours belong to
// another producer (the flow channels of the motion features).
__device__ __forceinline__ void store_quad_hl(float* row, int ch, float y0, float y1, float y2, float y3, int nv, float a_scale,
                                              int& sat_n) {
  h4 hi, lo;
  const float4 v = make_float4(y0, nv > 1 ? y1 : 0.f, nv > 2 ? y2 : 0.f, nv > 3 ? y3 : 0.f);
  split4(v, a_scale, hi, lo);
  sat_n += rp::quad_saturates(v, a_scale) ? 1 : 0;
  float* ph = row + (ch & ~7) + ((ch >> 2) & 1) * 2;
  if (nv == 4) {
    *reinterpret_cast<h4*>(ph) = hi;
    *reinterpret_cast<h4*>(ph + 4) = lo;
  } else {
    _Float16* hh = reinterpret_cast<_Float16*>(ph);
    _Float16* ll = reinterpret_cast<_Float16*>(ph + 4);
    if (nv > 0) { hh[0] = hi.x; ll[0] = lo.x; }
    if (nv > 1) { hh[1] = hi.y; ll[1] = lo.y; }
    if (nv > 2) { hh[2] = hi.z; ll[2] = lo.z; }
  }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// ---- weight packing: (Cout, Cin, kh, kw) fp32 -> fp16 hi / lo in the consumption order of each kernel family ----
struct PackParams {
  int Cout, Cin, kh, kw, G, T, ncb, Npad, vertical;
  unsigned char cb_seg[MAX_CB];
  short cb_c0[MAX_CB];
  short seg_start[4];     // first input channel (in the concatenated Cin order) of each segment
  short seg_count[4];
  float w_scale;
};

// conv_strip.hip: the strip kernels (160-row strips, operands by LDS-DMA)
int strip_waves(int c_out);                                    // workgroup shape: 16 * (32-column tiles per wave) + waves; 0 = unsupported width
void strip_allow_two_wave(int on);                             // measurement: two-wave workgroups (c_out <= 64) in the automatic choice (default on)
void strip_force_ni(int ni);                                   // measurement: 0 automatic, 1 / 2 column tiles per wave
void strip_allow_small(int on);                                // measurement: 32-row strips in the automatic choice (default on)
void strip_allow_s2(int on);                                   // measurement: stride-2 3x3 layers on strips over parity planes (default on)
void strip_allow_persist(int on);                              // measurement: persistent launches of the fp32-source strip forms (default OFF: slower)
void strip_allow_small32(int on);                              // measurement: 32-row strips for launches of <= 256 waves of 160-row strips (default on)
long long strip_s2_halfs(int c_in, int kh, int kw, int ncb, int Npad, int n_seg);   // size of the stride-2 copy of a layer's packed weights (0: none)
// strip height (160 / 32 rows) a launch of `batch` images takes, 0 = not a strip launch; request: 0 automatic, else the height to force
int strip_rows(int H, int W, int kh, int kw, int stride, int c_out, int batch, int request);
int strip_tiles_per_image(int H, int W, int kh, int kw, int rows);      // output tiles per image when tiled per image (tile_stats records)
// launches the strip kernel for the already filled parameter block (U, V, su, sv, T, dv0, segments, epilogue ...); sets the tiling
// members.  hlin: split-tensor sources (LDS-DMA); else fp32 sources through registers (+ fused normalisation when p.in_mr).
int strip_launch(KParams& p, int H, int W, int kh, int kw, bool hlin, bool per_image, int rows, hipStream_t st);
int strip_launch_r32(const KParams& p, int nw, int ni, bool spatial, bool hlin, bool norm, unsigned nwg, hipStream_t st);      // conv_strip_r32.hip
int strip_launch_r96(const KParams& p, int nw, int ni, bool spatial, bool hlin, bool norm, unsigned nwg, hipStream_t st);      // conv_strip_r96.hip (r06)
int strip_launch_p1(const KParams& p, int nw, int ni, bool spatial, bool hlin, bool norm, unsigned nwg, hipStream_t st);       // conv_strip_p1.hip: 160-row strips, single product
void strip_pack(const float* w, _Float16* pk, const PackParams& q, hipStream_t st);

}  // namespace rpconv
