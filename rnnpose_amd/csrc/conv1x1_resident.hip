// 1x1 convolution with the activation tile RESIDENT in LDS: BasicMotionEncoder.convc1 (324 -> 256 + ReLU on the looked-up
// correlation features, thirdparty/raft/update.py:80,87) and other 1x1 layers with Cin <= 352, Cout = 256.
//
// In the implicit-GEMM kernel (csrc/conv_igemm.hip) a 1x1 layer is the worst case: every 32-channel block costs a full
// activation staging (global -> registers -> fp16 hi/lo split -> LDS -> barrier) for only 12 MFMAs per wave, and each of the
// Cout / 64 column tiles repeats that split: 12-16 % of the matrix peak (r02).  Here a workgroup owns 32 consecutive pixels
// and ALL output columns: the 32 x Cin tile is split once into LDS (<= 45 KB, 16-byte chunks XOR-swizzled), then the
// channel blocks stream through v_mfma_f32_16x16x32_f16 with the weights as B fragments straight from their packed array
// ([channel block][16-column tile][hi, lo][lane] x 16 B), two stages ahead.  4 waves x 4 column tiles x 2 pixel tiles.
// Numerics: the fp16x3 split of csrc/f16x3.cuh; the K sum runs in 32-channel MFMAs (the implicit-GEMM kernel uses two
// 16-channel ones per block), so results agree with it to fp32 round-off, not bit for bit.
#include "common.hpp"
#include "f16x3.cuh"

namespace {

using rp::h4; using rp::h8; using rp::split4;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MQ = 32;          // pixels per workgroup (= threads / 8)
constexpr int CT = 4;           // 16-column tiles per wave: Cout = 4 waves x 4 x 16 = 256
constexpr int MAX_NCB = 11;     // Cin <= 352

struct RParams {
  const float* x;               // (M, cs): channels [co, co + Cin)
  int cs, co, Cin;
  const uint4* wpk;             // record (cb * 16 + column tile) * 128 + part * 64 + lane  (part: hi, lo)
  const float* bias;            // (256)
  float a_scale, out_scale;
  float* dst;                   // (M, dcs): channels [dco, dco + 256)
  int dcs, dco, relu;
  long long M;
  unsigned long long* sat;
  int dst_hl;                   // write the output as a SPLIT tensor (fp16 hi|lo per 8-channel group, rnnpose_hip.h) for the next convolution
};

template <int NCB>
__global__ __launch_bounds__(8 * MQ) __attribute__((amdgpu_num_vgpr(160))) void conv1x1_resident_kernel(const RParams p) {
  constexpr int NCBL = NCB * 4096 >= MQ * 260 * 4 ? NCB : (MQ * 260 * 4 + 4095) / 4096;    // the epilogue staging tile (33 KB) aliases it
  __shared__ __attribute__((aligned(16))) _Float16 sA[NCBL][2][MQ * 32];    // [channel block][hi, lo][row * 32 + swizzled chunk * 8 + e]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long m0 = static_cast<long long>(blockIdx.x) * MQ;

  // ---- stage: thread -> row tid >> 3, channel quad tid & 7 of every 32-channel block; all loads first, unconditional (rows
  //      past the end re-read the last pixel, quads past Cin re-read quad 0 and become zero)
  const int c4 = tid & 7;
  float4 av[NCB];
  {
    const long long m = m0 + (tid >> 3);
    const long long mc = m < p.M ? m : p.M - 1;
    const float* src = p.x + mc * p.cs + p.co;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int c = cb * 32 + c4 * 4;
      av[cb] = *reinterpret_cast<const float4*>(src + (c < p.Cin ? c : 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  int sat_n = 0;
  {
    const int r = tid >> 3;
    const int chunk = (c4 >> 1) ^ (((r >> 2) & 1) << 1);            // rows r and r + 4 would share LDS banks: swap chunk pairs
    const int off = r * 32 + chunk * 8 + (c4 & 1) * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const bool in = cb * 32 + c4 * 4 < p.Cin;                      // (Cin % 4 == 0: a quad is inside or outside as a whole)
      const float4 v = in ? av[cb] : make_float4(0.f, 0.f, 0.f, 0.f);
      h4 hi, lo;
      split4(v, p.a_scale, hi, lo);
      if (p.sat && in) sat_n += rp::quad_saturates(v, p.a_scale) ? 1 : 0;
      *reinterpret_cast<h4*>(&sA[cb][0][off]) = hi;
      *reinterpret_cast<h4*>(&sA[cb][1][off]) = lo;
    }
  }
  if (p.sat && sat_n) atomicAdd(p.sat, static_cast<unsigned long long>(sat_n));
  __syncthreads();

  // ---- main loop: stage = one channel block: 4 column tiles x 2 pixel tiles x 3 split products = 24 MFMAs ----
  const int l15 = lane & 15, lq = lane >> 4;
  const int phys = lq ^ (((l15 >> 2) & 1) << 1);
  f32x4 acc[2][CT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 bq[2][CT][2];                                                 // ring of 2 stages x column tile x (hi, lo)
#define R_LOADB(SLOT_, CB_)                                                                       \
  {                                                                                               \
    const uint4* rec_ = p.wpk + (static_cast<long long>(CB_) * 16 + wave * CT) * 128 + lane;      \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) { bq[SLOT_][j][0] = rec_[j * 128]; bq[SLOT_][j][1] = rec_[j * 128 + 64]; } \
  }
  R_LOADB(0, 0)
  if (NCB > 1) R_LOADB(1, 1)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    h8 ah[2], al[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ah[t] = *reinterpret_cast<const h8*>(&sA[cb][0][(16 * t + l15) * 32 + phys * 8]);
      al[t] = *reinterpret_cast<const h8*>(&sA[cb][1][(16 * t + l15) * 32 + phys * 8]);
    }
    const int sl = cb & 1;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const h8 bh = __builtin_bit_cast(h8, bq[sl][j][0]), bl = __builtin_bit_cast(h8, bq[sl][j][1]);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t][j] = rp::mfma16_c32(al[t], bh, acc[t][j]);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t][j] = rp::mfma16_c32(ah[t], bl, acc[t][j]);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t][j] = rp::mfma16_c32(ah[t], bh, acc[t][j]);
    }
    // (fenced: left alone the compiler sinks the weight loads to just before their MFMAs and waits vmcnt(0) on each)
    __builtin_amdgcn_sched_barrier(0);
    if (cb + 2 < NCB) R_LOADB(sl, cb + 2)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef R_LOADB

  // ---- epilogue: accumulators (C layout of the 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + e) -> LDS tile (aliasing the
  //      activation tile) -> 16-byte stores, a whole 1-KB output row per 64 lanes.  Straight from the registers it was 32
  //      4-byte stores per lane in 64-byte runs: the store path, not the arithmetic, set the kernel's time.
  constexpr int RSF = 260;                                            // staging row stride in floats (256 + 4: conflict-free)
  float* S = reinterpret_cast<float*>(&sA[0][0][0]);
  __syncthreads();                                                    // every wave is done reading the activation tile
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
      h4 hi, lo;
      split4(y, p.a_scale, hi, lo);
      if (p.sat && rp::quad_saturates(y, p.a_scale)) atomicAdd(p.sat, 1ull);
      const int ch = p.dco + cq;
      float* ph = p.dst + m * p.dcs + (ch & ~7) + ((ch >> 2) & 1) * 2;
      *reinterpret_cast<h4*>(ph) = hi;
      *reinterpret_cast<h4*>(ph + 4) = lo;
    } else {
      *reinterpret_cast<float4*>(p.dst + m * p.dcs + p.dco + cq) = y;
    }
  }
}

// fp32 (256, Cin) weights -> [channel block][column tile (16)][hi, lo][lane][8] fp16: lane l carries column 16*ct + (l & 15) and
// the 8 channels cb*32 + 8*(l >> 4) + j (zero past Cin) -- one B operand of v_mfma_f32_16x16x32_f16.
__global__ void resident_pack_kernel(const float* __restrict__ w, int Cin, int ncb, float scale, _Float16* __restrict__ pk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncb * 16 * 1024) return;
  const int j = i & 7, ln = (i >> 3) & 63, part = (i >> 9) & 1, ct = (i >> 10) & 15, cb = i >> 14;
  const int col = 16 * ct + (ln & 15), ch = cb * 32 + 8 * (ln >> 4) + j;
  const float v = ch < Cin ? w[col * Cin + ch] * scale : 0.f;
  const _Float16 h = static_cast<_Float16>(v);
  pk[i] = part == 0 ? h : static_cast<_Float16>(v - static_cast<float>(h));
}

}  // namespace

extern "C" size_t rnnpose_conv1x1_resident_packed_bytes(int c_in) {
  if (c_in <= 0 || c_in > 32 * MAX_NCB) return 0;
  return static_cast<size_t>(rp::cdiv(c_in, 32)) * 16 * 1024 * sizeof(_Float16);
}

extern "C" int rnnpose_conv1x1_resident_pack_f16x3(const float* weight, int c_out, int c_in, float w_scale, void* packed,
                                                   rnnpose_stream_t stream) {
  const char* fn = "rnnpose_conv1x1_resident_pack_f16x3";
  RP_REQUIRE(weight && packed && reinterpret_cast<uintptr_t>(packed) % 16 == 0, fn, "null or misaligned pointer");
  RP_REQUIRE(c_out == 256 && c_in > 0 && c_in <= 32 * MAX_NCB && c_in % 4 == 0, fn, "needs c_out = 256 and c_in <= 352, c_in % 4 == 0");
  RP_REQUIRE(w_scale > 0.f, fn, "w_scale must be positive");
  const int ncb = rp::cdiv(c_in, 32), total = ncb * 16 * 1024;
  hipLaunchKernelGGL(resident_pack_kernel, dim3(rp::cdiv(total, 256)), dim3(256), 0, rp::as_stream(stream), weight, c_in, ncb, w_scale,
                     static_cast<_Float16*>(packed));
  return rp::check_launch(fn);
}

extern "C" int rnnpose_conv1x1_resident_f16x3(const float* x, int x_c_stride, int x_c_offset, int c_in, const void* w_packed,
                                              const float* bias, float a_scale, float w_scale, int relu, long long n_pixels,
                                              float* dst, int dst_c_stride, int dst_c_offset, int dst_split, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_conv1x1_resident_f16x3";
  if (dst_split) RP_REQUIRE(dst_c_stride % 8 == 0 && reinterpret_cast<uintptr_t>(dst) % 32 == 0, fn, "split-form dst: channel stride multiple of 8, 32-byte aligned");
  RP_REQUIRE(x && w_packed && bias && dst, fn, "null pointer");
  RP_REQUIRE(c_in > 0 && c_in <= 32 * MAX_NCB && c_in % 4 == 0, fn, "c_in must be a multiple of 4, at most 352");
  RP_REQUIRE(n_pixels > 0 && n_pixels < (1LL << 31), fn, "bad pixel count");
  RP_REQUIRE(x_c_offset >= 0 && x_c_offset % 4 == 0 && x_c_stride % 4 == 0 && x_c_offset + c_in <= x_c_stride, fn,
             "input channels must lie inside the row, 16-byte aligned");
  RP_REQUIRE(dst_c_offset >= 0 && dst_c_offset + 256 <= dst_c_stride && dst_c_offset % 4 == 0 && dst_c_stride % 4 == 0 &&
                 reinterpret_cast<uintptr_t>(dst) % 16 == 0, fn, "the 256 output channels must lie inside the row, 16-byte aligned");
  RP_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w_packed) % 16 == 0, fn, "x / weights must be 16-byte aligned");
  RP_REQUIRE(a_scale > 0.f && w_scale > 0.f, fn, "scales must be positive");
  RParams p{};
  p.x = x; p.cs = x_c_stride; p.co = x_c_offset; p.Cin = c_in;
  p.wpk = static_cast<const uint4*>(w_packed);
  p.bias = bias;
  p.a_scale = a_scale; p.out_scale = 1.0f / (a_scale * w_scale);
  p.dst = dst; p.dcs = dst_c_stride; p.dco = dst_c_offset; p.relu = relu;
  p.M = n_pixels;
  p.sat = rp::sat_counter();
  p.dst_hl = dst_split;
  const dim3 grid(static_cast<unsigned>(rp::cdiv(n_pixels, MQ))), block(8 * MQ);
  hipStream_t st = rp::as_stream(stream);
  switch (rp::cdiv(c_in, 32)) {
#define R_CASE(N_) case N_: hipLaunchKernelGGL(conv1x1_resident_kernel<N_>, grid, block, 0, st, p); break;
    R_CASE(1) R_CASE(2) R_CASE(3) R_CASE(4) R_CASE(5) R_CASE(6) R_CASE(7) R_CASE(8) R_CASE(9) R_CASE(10) R_CASE(11)
#undef R_CASE
    default: return rp::fail_arg(fn, "unsupported c_in");
  }
  return rp::check_launch(fn);
}
