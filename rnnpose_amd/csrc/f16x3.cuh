// fp16x3 split: fp32 emulation on the fp16 matrix cores (shared by the implicit-GEMM convolution and the encoder stem).
//   x * S = hi + lo   (S a power of two), a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with fp32 accumulation.
// hi = x*S with the mantissa TRUNCATED to fp16's 11 significant bits (one v_and; exactly representable, so the packed
// convert is exact), lo = fp16(x*S - hi) (exact in fp32; 11 more bits) -> 21-22 significant bits.  Packed fp32 math
// (v_pk_mul_f32 / v_pk_add_f32) and v_cvt_pk_f16_f32 halve the instruction count.
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
  const f32x2 s2 = {s, s};
  const f32x2 x01 = f32x2{v.x, v.y} * s2, x23 = f32x2{v.z, v.w} * s2;
  const u32x2 m = {0xffffe000u, 0xffffe000u};
  const f32x2 t01 = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, x01) & m);
  const f32x2 t23 = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, x23) & m);
  const f32x2 l01 = x01 - t01, l23 = x23 - t23;
  const h2 cap = {static_cast<_Float16>(65504.f), static_cast<_Float16>(65504.f)};
  h2 h01 = __builtin_convertvector(t01, h2), h23 = __builtin_convertvector(t23, h2);
  h01 = __builtin_elementwise_max(__builtin_elementwise_min(h01, cap), -cap);
  h23 = __builtin_elementwise_max(__builtin_elementwise_min(h23, cap), -cap);
  const h2 q01 = __builtin_convertvector(l01, h2), q23 = __builtin_convertvector(l23, h2);
  hi = h4{h01.x, h01.y, h23.x, h23.y};
  lo = h4{q01.x, q01.y, q23.x, q23.y};
}

// true if any element of the quad leaves the fp16 range once scaled (|x*s| > 65504) or is not finite
__device__ __forceinline__ bool quad_saturates(const float4 v, float s) {
  const float lim = 65504.f / s;      // (fmaxf would drop a NaN operand: four ordered compares, NaN fails each)
  return !(fabsf(v.x) <= lim && fabsf(v.y) <= lim && fabsf(v.z) <= lim && fabsf(v.w) <= lim);
}

}  // namespace rp
