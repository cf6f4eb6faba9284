// fp16x3 split: fp32 emulation on the fp16 matrix cores (shared by the implicit-GEMM convolution and the encoder stem).
//   x * S = hi + lo   (S a power of two), a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with fp32 accumulation.
// hi = x*S with the mantissa TRUNCATED to fp16's 11 significant bits (one v_and; exactly representable, so the packed
// convert is exact), lo = fp16(x*S - hi) (exact in fp32; 11 more bits) -> 21-22 significant bits.
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
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  float t[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[i]) & 0xffffe000u);   // 11 significant bits: exact in fp16
    l[i] = x[i] - t[i];                                                                    // exact in fp32
  }
  const h2 cap = {static_cast<_Float16>(65504.f), static_cast<_Float16>(65504.f)};
  h2 h01 = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(t[0], t[1]));                 // (exactly representable: any rounding mode)
  h2 h23 = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(t[2], t[3]));
  h01 = __builtin_elementwise_max(__builtin_elementwise_min(h01, cap), -cap);
  h23 = __builtin_elementwise_max(__builtin_elementwise_min(h23, cap), -cap);
  const h2 q01 = __builtin_convertvector(f32x2{l[0], l[1]}, h2), q23 = __builtin_convertvector(f32x2{l[2], l[3]}, h2);
  hi = h4{h01.x, h01.y, h23.x, h23.y};
  lo = h4{q01.x, q01.y, q23.x, q23.y};
}

// true if any element of the quad leaves the fp16 range once scaled (|x*s| > 65504) or is not finite
__device__ __forceinline__ bool quad_saturates(const float4 v, float s) {
  const float lim = 65504.f / s;      // (fmaxf would drop a NaN operand: four ordered compares, NaN fails each)
  return !(fabsf(v.x) <= lim && fabsf(v.y) <= lim && fabsf(v.z) <= lim && fabsf(v.w) <= lim);
}

}  // namespace rp
