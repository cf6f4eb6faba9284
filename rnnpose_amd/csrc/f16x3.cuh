// fp16x3 split: fp32 emulation on the fp16 matrix cores (shared by the implicit-GEMM convolution and the encoder stem).
//   x * S = hi + lo   (S a power of two), a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with fp32 accumulation.
// hi = fp16(x*S) rounded to nearest, lo = fp16(x*S - hi) (the difference is exact in fp32; 11 more bits) -> 22 significant bits.
// Range: |x*S| >= 65520 does not fit fp16.  hi is clamped to +-65504 (never inf/NaN from finite inputs) and the element
// is COUNTED as saturated: split4 returns the number of saturated elements of its quad so that callers can raise a
// sticky flag (rnnpose_conv_saturation_count): fp32 in the reference has no such cliff, so it must never pass silently.
#pragma once
#include <hip/hip_runtime.h>

namespace rp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const float4 v, float s, h4& hi, h4& lo) {
  // Scalar fp32 operations on purpose, and the files that use this are built with -fno-slp-vectorize (rnnpose_amd/build.py):
  // a v_pk_mul_f32 / v_pk_add_f32 issued next to MFMAs costs ~11 cycles more than its two scalar halves
  // (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), and plain -O3 re-packs adjacent scalar ops.  r02 same-box
  // A/B: +0.7 % on the step.  The fp16 converts and the clamp stay packed (v_cvt_pk*, v_pk_min/max_f16).
  // hi = x*s ROUNDED to nearest fp16 (round 3; r02 truncated the mantissa with one v_and): the remainder then is at most
  // half an fp16 ulp, so lo carries one more bit and the neglected lo*lo product halves -- per-product error ~2^-22 instead of
  // ~2^-21, same instruction count (two packed converts back instead of four ands).  The r03 ablation
  // showed the split's vector instructions are not what bounds the convolution kernels.
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  const h2 cap = {static_cast<_Float16>(65504.f), static_cast<_Float16>(65504.f)};
  h2 h01 = __builtin_convertvector(f32x2{x[0], x[1]}, h2);                                 // round to nearest even; overflow -> inf
  h2 h23 = __builtin_convertvector(f32x2{x[2], x[3]}, h2);
  h01 = __builtin_elementwise_max(__builtin_elementwise_min(h01, cap), -cap);              // |x*s| > 65504: clamped (and counted by the callers)
  h23 = __builtin_elementwise_max(__builtin_elementwise_min(h23, cap), -cap);
  const f32x2 t01 = __builtin_convertvector(h01, f32x2), t23 = __builtin_convertvector(h23, f32x2);
  const float l[4] = {x[0] - t01.x, x[1] - t01.y, x[2] - t23.x, x[3] - t23.y};             // exact in fp32 (in range)
  h2 q01 = __builtin_convertvector(f32x2{l[0], l[1]}, h2), q23 = __builtin_convertvector(f32x2{l[2], l[3]}, h2);
  q01 = __builtin_elementwise_max(__builtin_elementwise_min(q01, cap), -cap);              // (far out of range: the remainder saturates too --
  q23 = __builtin_elementwise_max(__builtin_elementwise_min(q23, cap), -cap);              //  never inf / NaN from finite inputs)
  hi = h4{h01.x, h01.y, h23.x, h23.y};
  lo = h4{q01.x, q01.y, q23.x, q23.y};
}

// One 16 x 16 tile step over a lane's 8 channels (A: row lane & 15, B: column lane & 15, channels 8 (lane >> 4) .. + 7 of a 32-channel block).
// RP_MFMA16_K32 = 0 (default since r06): TWO v_mfma_f32_16x16x16_f16 over the lane's channels 0-3 and 4-7 (any channel -> (lane group,
// element) assignment is a valid K order as long as A and B share it, so the operand layouts are those of the K = 32 instruction).
// RP_MFMA16_K32 = 1: ONE v_mfma_f32_16x16x32_f16 (r02-r05).  Why not the K = 32 shape although it has twice the rate: on MI355X a
// wave issuing v_mfma_f32_16x16x32_{f16,bf16} makes PACKED fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) of OTHER waves
// on the same SIMD return wrong values for groups of 16 lanes (DESIGN 4 / 5 rule 11, tools/probes/pk_f32_vs_mfma.hip).  This library has
// no packed fp32 of its own (tests/test_isa_guard.py), but a co-tenant's kernels -- an integrator's renderer built with plain -O3 -- do:
// next to mask_upsample on the K = 32 shape 1599 of 1600 launches of such a neighbour differed from its solo output, next to the whole
// refinement loop 367 of 1600 (tools/pk_neighbour_probe.py, profiles/r06_pk_neighbour.txt); next to the 16x16x16 and 32x32x16 shapes:
// none.  The three kernels built on this step are latency-bound, not matrix-bound: the price is in the same record.
#ifndef RP_MFMA16_K32
#define RP_MFMA16_K32 0
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16_c32(const h8 a, const h8 b, f32x4 c) {
#if RP_MFMA16_K32
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  const h4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
  const h4 b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, c, 0, 0, 0);
#endif
}

// true if any element of the quad leaves the fp16 range once scaled (|x*s| > 65504) or is not finite
__device__ __forceinline__ bool quad_saturates(const float4 v, float s) {
  const float lim = 65504.f / s;      // (fmaxf would drop a NaN operand: four ordered compares, NaN fails each)
  return !(fabsf(v.x) <= lim && fabsf(v.y) <= lim && fabsf(v.z) <= lim && fabsf(v.w) <= lim);
}

}  // namespace rp
