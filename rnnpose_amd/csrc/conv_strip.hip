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
//     256 CUs, two workgroups per CU (64 / 76 KB of LDS, <= 256 registers).
//   * Every operand arrives by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction, no registers, no vector ALU):
//       - weights: WAVE-PRIVATE.  A wave only ever multiplies its own 32 columns, so its weight fragments go through its own
//         ring of 2-KB records (one record = 16 channels x 32 columns x (hi, lo) of one tap; 5 / 6 slots), requested four / five
//         steps ahead from a scalar base address and waited for with a counted s_waitcnt vmcnt -- no workgroup barrier is
//         involved in the weight stream at all;
//       - activations: from SPLIT TENSORS (rnnpose_hip.h: fp16 hi|lo per 8-channel group, written by the producer's epilogue),
//         one HALF block (16 channels = one MFMA K) of the strip with its halo at a time, two slots; the next half block is
//         requested a whole half block (5 or 9 taps) ahead.  ONE barrier per half block, placed in front of its last tap.
//     fp32 sources (and the encoder's fused instance norm + ReLU, which needs the vector ALU anyway) take the same kernel with the
//     activation half block staged through registers instead (MODE 1 / 2).
//   * LDS image of a half block / a weight record: a hi plane and a lo plane of 32-byte rows (2 groups x 8 fp16 of one pixel /
//     column), so that the lo fragment sits at a CONSTANT offset from the hi one; the 16-byte group position is XOR-swizzled --
//     with bit 3 of the row (linear strips, weight records) or the parity of the halo-tile line (3x3 patches) -- on the SOURCE
//     address of the DMA (the LDS image of a DMA is lane-linear), in the packed weight order, and in the fragment addresses:
//     a ds_read_b128 lane group (16 lanes) touches 16 different bank slots.  With the line-parity swizzle a 3x3 tap is the
//     centre address plus a compile-time offset, so that EVERY fragment read of the main loop is `one address register +
//     immediate`: 25 (1x5 / 5x1: tap masks folded in) or 10 (3x3) address registers per lane, no vector ALU work per step.
//   * A step = one tap of one half block: 15 MFMAs per wave (5 row tiles x 3 products) interleaved one to one with the 12
//     ds_read_b128 of the NEXT step's fragments (two register sets), one counted wait, one DMA statement.  Taps and pairs of half
//     blocks are unrolled: ring slots, fragment sets and every wait count are compile-time constants (round 4, first version: a
//     rolled step with ~200 scalar / vector bookkeeping instructions in front of each 15-MFMA group ran 1.1-1.9x SLOWER than the
//     128-row kernels -- a lone wave issues one instruction per ~4 cycles, the bookkeeping was 2x the MFMA time).
//   * Epilogue: the parameter block's fused forms (bias, ReLU, GRU gates, additive map, split outputs, fp64 tile statistics),
//     through a wave-private LDS tile (the wave's own weight ring) with 16-byte row-contiguous stores.
#include "conv_strip_kernel.cuh"

namespace {

// packed order of the strip kernels: [half block hb = 2 cb + kk][tap][32-column tile] records of 2 KB = the LDS image itself (the
// DMA copies it linearly): [hi plane | lo plane] x [column n: 32 bytes] x [group position: 16 bytes] x 8 fp16, position pos of
// column n holding group g = pos ^ ((n >> 3) & 1) = channels 16 kk + 8 g + j of the block
__global__ void pack_strip_kernel(const float* __restrict__ w, _Float16* __restrict__ pk, const PackParams q, int TT, int spatial) {
  const int nt32 = q.Npad / 32;
  const long long total = static_cast<long long>(q.ncb) * 2 * TT * q.Npad * 32;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j8 = static_cast<int>(i & 7);
  const int pos = static_cast<int>((i >> 3) & 1);
  const int n32 = static_cast<int>((i >> 4) & 31);
  const int part = static_cast<int>((i >> 9) & 1);
  const int ctile = static_cast<int>((i >> 10) % nt32);
  const long long st = (i >> 10) / nt32;
  const int tap = static_cast<int>(st % TT);
  const int hbk = static_cast<int>(st / TT);
  const int blk = hbk >> 1, kk = hbk & 1;
  const int g = pos ^ ((n32 >> 3) & 1);
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

// r05: the STRIDE-2 form of a 3x3 layer (one source of whole 32-channel blocks).  out(Y, X) = sum w(dy, dx) in(2Y + dy, 2X + dx) is a
// stride-1 layer on the half-resolution grid over the FOUR PARITY PLANES of the input -- plane (py, px) pixel (y, x) = in(2y + py,
// 2x + px), a strided view of the NHWC source, no copy -- with 2 x 2 taps (ty, tx), offsets (ty - 1, tx - 1) in {-1, 0}:
//     dy = -1 -> plane row 1 at y - 1 (ty 0),  dy = 0 -> plane row 0 at y (ty 1),  dy = +1 -> plane row 1 at y (ty 1);  likewise dx.
// Nine of the sixteen (plane, tap) pairs carry a weight, seven are zero.  Packed order = the strip order with K running over
// [plane p = 2 py + px][channel]: [half block of 4 Cin channels][tap t = 2 ty + tx][32-column tile] records of 2 KB.
__global__ void pack_strip_s2_kernel(const float* __restrict__ w, _Float16* __restrict__ pk, const PackParams q) {
  const int nt32 = q.Npad / 32;
  const long long total = static_cast<long long>((q.kh == 1 ? 1 : 4) * q.ncb) * 2 * 4 * q.Npad * 32;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j8 = static_cast<int>(i & 7);
  const int pos = static_cast<int>((i >> 3) & 1);
  const int n32 = static_cast<int>((i >> 4) & 31);
  const int part = static_cast<int>((i >> 9) & 1);
  const int ctile = static_cast<int>((i >> 10) % nt32);
  const long long st = (i >> 10) / nt32;
  const int tap = static_cast<int>(st % 4);
  const int hbk = static_cast<int>(st / 4);
  const int blk4 = hbk >> 1, kk = hbk & 1;
  const int plane = blk4 / q.ncb, blk = blk4 - plane * q.ncb;
  const int py = plane >> 1, px = plane & 1, ty = tap >> 1, tx = tap & 1;
  const int ky = py == 0 ? (ty == 1 ? 1 : -1) : (ty == 0 ? 0 : 2);
  const int kx = px == 0 ? (tx == 1 ? 1 : -1) : (tx == 0 ? 0 : 2);
  const int g = pos ^ ((n32 >> 3) & 1);
  const int ci = blk * 32 + kk * 16 + g * 8 + j8;
  const int n = ctile * 32 + n32;
  // The seven empty (plane, tap) slots (three of four for 1x1) carry ZERO weights and are multiplied, not skipped (skipping measured slower:
  // profiles/r05_ab_s2_skip.txt).  Their activations lie OUTSIDE the output's receptive field (e.g. in(2y-2, 2x-2)): a non-finite or
  // fp16-saturated activation there gives 0 * inf = NaN in an output that the 128-row kernel and the reference leave finite -- the two
  // forms agree for finite, in-range inputs only.  The encoder launches these layers on instance-normalised maps
  // (src_bounded: |x| <= sqrt(H W), two orders of magnitude inside the range); any other caller keeps the range guard on, which counts the event.
  float v = 0.f;
  if (q.kh == 1) {        // 1x1 stride 2 (the encoder's down-sampling branches): plane (0, 0) only, the centre tap only
    if (n < q.Cout && ci < q.Cin && tap == 3) v = w[static_cast<long long>(n) * q.Cin + ci] * q.w_scale;
  } else if (n < q.Cout && ci < q.Cin && ky >= 0 && kx >= 0) v = w[((static_cast<long long>(n) * q.Cin + ci) * 3 + ky) * 3 + kx] * q.w_scale;
  const _Float16 h = static_cast<_Float16>(v);
  pk[i] = part == 0 ? h : static_cast<_Float16>(v - static_cast<float>(h));
}

bool strip_kernel_shape(int kh, int kw) { return (kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1); }

}  // namespace

namespace rpconv {

// Column tiles per wave: 1 unless forced.  The two-tile form (one wave per SIMD, 160 x 64 per wave, 14 fragment reads per 30
// MFMAs instead of 12 per 15) is built, tested and SLOWER (r04, profiles/r04_strip_ni2.txt: GRU z|r at B = 8 84.6 vs 76.4 us, even
// with every memory operation compiled out 70 vs 56 us): a lone wave does not keep its SIMD's matrix pipe busy.
static int g_strip_two_wave = 1;     // two-wave workgroups (c_out <= 64: the encoder's 64-channel layers at 240 x 320) in the automatic choice
static int g_strip_ni = 1;           // 1 / 2: column tiles per wave (strip_force_ni; 0: the least-waste shape over both)

void strip_allow_two_wave(int on) { g_strip_two_wave = on; }
void strip_force_ni(int ni) { g_strip_ni = ni; }      // (rnnpose_conv_strip: mode 1 -> 1, mode 2 -> 1, mode 3 -> 2)

// workgroup shape for c_out output channels: NI column tiles of 32 per wave x NW waves; returns NI * 16 + NW (0 = unsupported).
// c_out <= 64: two waves of one tile each (three workgroups per CU).
// NI = 2 (one wave per SIMD, 160 x 64 per wave) wherever whole 64-column wave tiles fit; the column-tile width that wastes
// the fewest columns, the wider one on a tie.
int strip_waves(int c_out) {
  if (c_out <= 32) return 0;                          // (one-wave workgroups are not built)
  if (c_out <= 64) return g_strip_ni == 2 ? 0 : 1 * 16 + 2;      // two waves of 32 columns: the encoder's 64-channel layers (strip_auto: large maps only)
  int best = 0, best_waste = 1 << 30;
  const int cand[5][2] = {{2, 4}, {2, 3}, {2, 2}, {1, 4}, {1, 3}};       // 256, 192, 128, 128, 96 columns
  for (const auto& c : cand) {
    if (g_strip_ni != 0 && c[0] != g_strip_ni) continue;
    const int wdt = 32 * c[0] * c[1];
    const int waste = rp::cdiv(c_out, wdt) * wdt - c_out;
    if (waste < best_waste) { best_waste = waste; best = c[0] * 16 + c[1]; }
  }
  return best;
}

int strip_tiles_per_image(int H, int W, int kh, int kw, int rows) {
  if (kh == 3 && kw == 3) return rp::cdiv(W, SPW) * rp::cdiv(H, rows / 16);          // patches of (rows / 16) x 16 pixels
  return rp::cdiv(static_cast<long long>(H) * W, rows);
}

static int g_strip_small = 1;        // 32-row strips in the automatic choice (strip_allow_small)
void strip_allow_small(int on) { g_strip_small = on; }
#ifndef RS_MID96_WAVES
#define RS_MID96_WAVES 512
#endif
#ifndef RS_SMALL32_WAVES
#define RS_SMALL32_WAVES 256         // (measurement: -DRS_SMALL32_WAVES=512 also moves the GRU's q and the motion encoder's last layer at B = 4)
#endif
static int g_strip_small32 = 1;      // r06: 32-row strips for launches of <= 256 waves of 160-row strips (rnnpose_conv_strip(6): off)
static int g_strip_persist = 0;      // r06: persistent launches of the fp32-source forms (rnnpose_conv_strip(7): ON).  Built, bit-identical, and SLOWER: the
                                     // encoder's 64 -> 64 layer at B = 8 takes 190-255 us instead of 155-190 (profiles/r06_persistent_strips.txt) -- OFF by default
static int g_strip_s2 = 1;           // stride-2 3x3 layers as strips over parity planes (strip_allow_s2; 0: the 128-row kernel's tap-per-staging mode)
void strip_allow_s2(int on) { g_strip_s2 = on; }
void strip_allow_small32(int on) { g_strip_small32 = on; }
void strip_allow_persist(int on) { g_strip_persist = on; }

// Strip height for a launch of `batch` images: 160 or 32 rows -- or 0: not a strip launch.  request: 0 = automatic, else that height
// (tests, measurement).  160-row strips when they give the launch >= 240 workgroups (about one per CU) or, failing that, >= 24 per
// image with more than 8192 pixels in the launch (the headline's 128-column layers at B = 4: 120 workgroups, equal to the other
// forms); 32-row strips for launches of at most 8192 pixels that get >= 24 of them (single-image crops: 30 x 30 maps); else none.
// Measured per layer (r04, profiles/r04_strip_heights.txt): single-image crops take 12-19 us per layer on 32-row strips, 18-28 on
// the 128-row kernel, 30-48 on 160-row strips; at 14 400 pixels (16 crops) the 128-row kernel is 0.5-1 % ahead of 32-row strips on
// the step, at 19 200 (B = 4 of the headline) they are equal.  (64-row strips were built too: never the best of the three.)
int strip_rows(int H, int W, int kh, int kw, int stride, int c_out, int batch, int request) {
  const bool s2_1x1 = stride == 2 && kh == 1 && kw == 1;
  if ((stride != 1 && stride != 2) || !(strip_kernel_shape(kh, kw) || s2_1x1) || strip_waves(c_out) == 0) return 0;
  const int cfg = strip_waves(c_out), ni = cfg >> 4, nwt = (cfg & 15) * ni;         // 32-column wave tiles per workgroup
  if (stride == 2) {      // r05: the 2x2-tap form over the parity planes: 3x3 (or 1x1: one plane), even input size, 160-row strips on the OUTPUT grid, one tile per wave
    if (!g_strip_s2 || !((kh == 3 && kw == 3) || s2_1x1) || (H & 1) || (W & 1) || ni != 1) return 0;
    H >>= 1; W >>= 1;
    kh = kw = 3;          // (tiled like a 3x3 layer: 10 x 16 patches)
  }
  if (request != 0) return (request == 160 || ((request == 32 || request == 96) && ni == 1)) ? request : 0;
  if ((cfg & 15) == 2 && ni == 1 && g_strip_two_wave == 0) return 0;      // (two-wave workgroups, c_out <= 64: see strip_allow_two_wave)
  auto fits = [&](int rows) {
    if (ni == 2 && rows != 160) return false;         // (two tiles per wave: 160-row strips only)
    if (kh == 3) {                                    // ragged patches: at most 15 % of the rows wasted
      const int ph = rows / 16;
      const long long covered = static_cast<long long>(rp::cdiv(W, SPW)) * SPW * rp::cdiv(H, ph) * ph;
      if (covered * 100 > static_cast<long long>(H) * W * 115) return false;
    }
    return true;
  };
  const int ncol = rp::cdiv(c_out, 32 * nwt);
  const long long nb = batch > 0 ? batch : 1;
  if (!g_strip_small)                                 // (measurement: r04's first rule -- 160-row strips for maps that give >= 24 per image)
    return fits(160) && strip_tiles_per_image(H, W, kh, kw, 160) * ncol >= 24 ? 160 : 0;
  const long long t160 = strip_tiles_per_image(H, W, kh, kw, 160) * static_cast<long long>(ncol);
  const long long npix = nb * H * W;
  if (fits(160) && (t160 * nb >= 240 || (t160 >= 24 && npix > 8192))) {
    // r06: a launch whose 160-row strips are at most 256 WAVES (a quarter of the chip's SIMDs: convf2 128 -> 64 of a half-batch chain =
    // 120 two-wave workgroups) takes 32-row strips instead -- five times the waves; 15 vs 26 us (profiles/r05_conv_layers_alone.txt hl6 / hl5)
    if (g_strip_small32 && stride == 1 && t160 * nb * (cfg & 15) <= RS_SMALL32_WAVES && fits(32)) return 32;
    // ... and one of at most 512 waves (half the SIMDs: the 128-column layers of a half-batch chain -- GRU q, the motion encoder's last
    // layer: 120 four-wave workgroups) 96-row strips (three 32-row tiles per wave, 6 x 16 patches): 200 workgroups; q 34 -> 25 us, conv
    // 256 -> 126 53 -> 40 us at B = 4 (profiles/r06_strip_heights.txt); at B = 8 (960 waves) the 160-row form stays ahead
    if (g_strip_small32 && stride == 1 && t160 * nb * (cfg & 15) <= RS_MID96_WAVES && fits(96)) return 96;
    return 160;
  }
  if (fits(32) && npix <= 8192 && strip_tiles_per_image(H, W, kh, kw, 32) * ncol * nb >= 24) return 32;
  return 0;
}

void strip_pack(const float* w, _Float16* pk, const PackParams& q, hipStream_t st) {
  if (q.kh == 1 && q.kw == 1) {       // 1x1: no stride-1 strip form; the stride-2 form (one plane, one tap in four) right behind the first copy
    const long long t1 = strip_s2_halfs(q.Cin, 1, 1, q.ncb, q.Npad, q.seg_count[1] == 0 && q.seg_count[2] == 0 && q.seg_count[3] == 0 ? 1 : 2);
    if (t1 > 0) hipLaunchKernelGGL(pack_strip_s2_kernel, dim3(rp::cdiv(t1, 256)), dim3(256), 0, st, w, pk, q);
    return;
  }
  if (!strip_kernel_shape(q.kh, q.kw)) return;        // (the second copy stays unwritten: never read for other shapes)
  const int TT = q.kh * q.kw;
  const long long total = static_cast<long long>(q.ncb) * 2 * TT * q.Npad * 32;
  hipLaunchKernelGGL(pack_strip_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, st, w, pk, q, TT, (q.kh == 3 && q.kw == 3) ? 1 : 0);
  const long long t2 = strip_s2_halfs(q.Cin, q.kh, q.kw, q.ncb, q.Npad, q.seg_count[1] == 0 && q.seg_count[2] == 0 && q.seg_count[3] == 0 ? 1 : 2);
  if (t2 > 0) hipLaunchKernelGGL(pack_strip_s2_kernel, dim3(rp::cdiv(t2, 256)), dim3(256), 0, st, w, pk + total, q);     // (third copy: the stride-2 form)
}

// fp16 elements of the stride-2 copy of a layer's packed weights (0: the layer has none): 3x3, ONE source of whole 32-channel blocks
long long strip_s2_halfs(int c_in, int kh, int kw, int ncb, int Npad, int n_seg) {
  if (n_seg != 1 || c_in % 32 != 0) return 0;
  if (kh == 1 && kw == 1) return static_cast<long long>(ncb) * 2 * 4 * Npad * 32;          // one plane
  if (!(kh == 3 && kw == 3) || 4 * ncb > MAX_CB) return 0;
  return static_cast<long long>(4 * ncb) * 2 * 4 * Npad * 32;
}

int strip_launch(KParams& p, int H, int W, int kh, int kw, bool hlin, bool per_image, int rows, hipStream_t st) {
  const char* fn = "rnnpose_conv2d_nhwc_f16x3";
  RP_REQUIRE((strip_kernel_shape(kh, kw) && p.stride == 1) || (p.stride == 2 && ((kh == 3 && kw == 3) || (kh == 1 && kw == 1))), fn,
             "strip kernel: 3x3, 1x5 or 5x1, stride 1; 3x3 or 1x1, stride 2");
  if (p.stride == 2) kh = kw = 3;        // (the stride-2 forms run the 3x3 geometry: 10 x 16 patches, 2 x 2 taps)
  const int cfg = strip_waves(p.Cout);
  RP_REQUIRE(cfg != 0, fn, "strip kernel: c_out must exceed 32");
  const int ni = cfg >> 4, nw = cfg & 15;
  RP_REQUIRE(rows == 160 || ((rows == 32 || rows == 96) && ni == 1), fn, "strip kernel: strips of 160, 96 or 32 rows (two column tiles per wave: 160 only)");
  const bool spatial = kh == 3;
  const bool norm = p.in_mr != nullptr;
  RP_REQUIRE(!norm || (spatial && !hlin), fn, "strip kernel: the fused normalisation is the 3x3 fp32-source form");
  {
    const Seg* sg[4] = {&p.seg0, &p.seg1, &p.seg2, &p.seg3};
    const int nseg = p.cb3 < p.ncb ? 4 : (p.cb2 < p.ncb ? 3 : (p.cb1 < p.ncb ? 2 : 1));
    for (int s = 0; s < nseg; ++s) {     // whole 32-channel blocks: the source pointers advance by half blocks without a range test
      RP_REQUIRE(sg[s]->ccount % 32 == 0, fn, "strip kernel: source channel counts in multiples of 32");
      if (hlin) RP_REQUIRE(sg[s]->coff % 8 == 0 && sg[s]->cstride % 8 == 0, fn, "strip kernel, split-tensor sources: channel offsets / strides of 8");
    }
  }
  {      // 32-bit element offsets in the fast epilogue: pixel index x channel stride of every destination / operand below 2^32
    const long long npix = static_cast<long long>(p.B) * H * W;
    long long cs = p.dst_cs;
    if (p.dst2 && p.dst2_cs > cs) cs = p.dst2_cs;
    if (p.dsth && p.dsth_cs > cs) cs = p.dsth_cs;
    if (p.addm && p.addm_cs > cs) cs = p.addm_cs;
    if (p.aux0 && p.aux0_cs > cs) cs = p.aux0_cs;
    if (p.aux1 && p.aux1_cs > cs) cs = p.aux1_cs;
    p.off32 = (npix + 1) * (cs + 1) < (1LL << 32) ? 1 : 0;
  }
  p.ksplit = 1;
  p.n_nt = rp::cdiv(p.Cout, 32 * ni * nw);
  RP_REQUIRE(p.n_nt * ni * nw * 32 <= p.Npad, fn, "strip kernel: column tiles exceed the packed width");
  if (spatial) {
    p.T = p.stride == 2 ? 4 : 9; p.dv0 = 0;
    if (p.stride == 1) { p.Uin = p.U; p.Vin = p.V; p.su = p.V; p.sv = 1; }      // (source pixel of a staged row: RS_ROW_PIXEL; stride 2: set by the caller)
    p.sp_tx = rp::cdiv(W, SPW); p.sp_ty = rp::cdiv(H, rows / 16);
    p.tpi = p.sp_tx * p.sp_ty;
    p.n_mt = p.B * p.tpi;
  } else if (per_image) {
    p.tpi = rp::cdiv(static_cast<long long>(H) * W, rows);
    p.n_mt = p.B * p.tpi;
  } else {
    p.tpi = 0;
    p.n_mt = rp::cdiv(static_cast<long long>(p.B) * H * W, rows);
  }
  unsigned nwg = static_cast<unsigned>(p.n_mt) * p.n_nt;
  // r06 (measured slower, OFF unless rnnpose_conv_strip(7)): a launch of the fp32-source forms (the encoder's
  // layers) with at least 1.5 rounds of workgroups runs PERSISTENT -- as many workgroups as the chip holds at once (two-wave workgroups:
  // four per CU, else two), each walking its share of the tile list and pulling tile t + 1's first activations towards the CU in front
  // of tile t's epilogue (conv_strip_kernel.cuh).
  p.n_tiles = 0;
  if (g_strip_persist && !hlin && ni == 1 && spatial && p.stride == 1 && rows == 160 && !p.single_product) {      // (the forms that have a PERSIST instantiation)
    const unsigned slots = static_cast<unsigned>(rp::cu_count()) * (nw == 2 ? 4u : 2u) & ~7u;
    if (slots >= 8 && nwg >= slots + slots / 2) { p.n_tiles = static_cast<int>(nwg); nwg = slots; }
    static const int stagger_env = getenv("RNNPOSE_STRIP_STAGGER") ? atoi(getenv("RNNPOSE_STRIP_STAGGER")) : 0;      // (measurement)
    p.stagger = stagger_env;
  }
  int bad;
  if (rows == 160 && p.single_product && ni == 1 && p.stride == 1) bad = strip_launch_p1(p, nw, ni, spatial, hlin, norm, nwg, st);      // (the stride-2 forms keep three products)
  else if (rows == 160) bad = strip_launch_height<5>(p, nw, ni, spatial, hlin, norm, nwg, st);
  else if (rows == 96) bad = strip_launch_r96(p, nw, ni, spatial, hlin, norm, nwg, st);
  else bad = strip_launch_r32(p, nw, ni, spatial, hlin, norm, nwg, st);
  RP_REQUIRE(!bad, fn, "strip kernel: no kernel for this workgroup shape and strip height");
  return rp::check_launch(fn);
}

}  // namespace rpconv

#if RS_ABL & 512
extern "C" int rnnpose_debug_strip_clk(unsigned long long* host, int n_wg) {
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_strip_clk), static_cast<size_t>(n_wg) * 8 * sizeof(unsigned long long)));
}
#endif
