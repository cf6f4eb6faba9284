// Stand-alone probe (plain HIP, no library code), r05: do PACKED fp32 vector instructions (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32 -- what hipcc's SLP vectoriser makes of adjacent scalar fp32 multiplies and adds under plain -O3) compute the same
// values when a kernel of ANOTHER stream is issuing matrix instructions on the same SIMDs?
//
// Background (profiles/r05_determinism.txt): r04 found that with two streams active, fresh refiner instances differed in 64-byte
// runs of a weight map and read it as a kernel-to-kernel visibility problem.  r05 reduced it to the descriptor-weight kernel
// (csrc/pointwise.hip, corr_weight_kernel) alone, on CONSTANT inputs: its output differs from its own solo output in groups of 16
// lanes whenever mask_upsample or conv1x1_resident (the two kernels built on v_mfma_f32_16x16x32_f16) run next to it on another
// stream -- and never when the file is compiled with -fno-slp-vectorize (no packed fp32 instructions), never next to the
// 32x32x16 convolution kernels, never next to memory-bound kernels.  This probe asks the hardware directly.
//
//   stream A: pk_kernel -- every lane runs a chain of packed fp32 multiply-adds on lane-dependent constants (no memory traffic
//             besides one store); its output is compared, launch by launch, with the output of a launch that ran alone
//             sc_kernel  -- the same arithmetic written so that it compiles to scalar v_mul_f32 / v_add_f32 / v_fma_f32
//   stream B: a loop of v_mfma_f32_16x16x32_f16, or of v_mfma_f32_32x32x16_f16, or nothing
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probes/pk_f32_vs_mfma.hip -o tools/probes/pk_f32_vs_mfma && tools/probes/pk_f32_vs_mfma [launches=400]
//   (-fno-slp-vectorize: pk_kernel keeps its packed instructions -- they come from explicit two-wide vectors -- and sc_kernel stays scalar)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int NWG = 4800, NT = 256, CHAIN = 96;

// packed: explicit two-wide vectors -> v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32
__global__ __launch_bounds__(NT) void pk_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const int t = blockIdx.x * NT + threadIdx.x;
  f2 a = {in[t & 1023], in[(t + 17) & 1023]}, b = {in[(t + 5) & 1023], in[(t + 91) & 1023]};
  f2 s = {0.f, 0.f};
#pragma unroll 8
  for (int i = 0; i < CHAIN; ++i) {
    const f2 w = {0.25f + 0.001f * i, 0.75f - 0.001f * i};
    f2 m = a * w;                 // v_pk_mul_f32
    m = m + b * w.yx;             // v_pk_fma_f32 / mul + add with op_sel
    s = s + m * b;                // accumulate
    a = a * 0.999f + 0.001f;
    b = b.yx * 1.001f - 0.0005f;
  }
  out[t] = s.x + s.y;
}

// packed fp16 arithmetic and clamps (v_pk_fma_f16 / v_pk_mul_f16 / v_pk_max_f16 / v_pk_min_f16: what the fp16x3 split's range clamp
// compiles to in the library's convolution kernels) and v_pk_mov_b32
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(NT) void pkh_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const int t = blockIdx.x * NT + threadIdx.x;
  h2 a = {(_Float16)in[t & 1023], (_Float16)in[(t + 17) & 1023]}, b = {(_Float16)in[(t + 5) & 1023], (_Float16)in[(t + 91) & 1023]};
  h2 s = {(_Float16)0.f, (_Float16)0.f};
  const h2 hi = {(_Float16)0.9f, (_Float16)0.8f}, lo = {(_Float16)0.05f, (_Float16)0.1f};
#pragma unroll 8
  for (int i = 0; i < CHAIN; ++i) {
    const h2 w = {(_Float16)(0.25f + 0.001f * i), (_Float16)(0.75f - 0.001f * i)};
    h2 m = a * w + b * w.yx;
    m = __builtin_elementwise_max(__builtin_elementwise_min(m, hi), lo);      // v_pk_min_f16 / v_pk_max_f16
    s = s * (_Float16)0.5f + m * b;
    a = a * (_Float16)0.999f + (_Float16)0.001f;
    b = __builtin_elementwise_max(b.yx * (_Float16)1.001f - (_Float16)0.0005f, lo);
  }
  out[t] = (float)s.x + (float)s.y;
}

// the same arithmetic on scalars with the packing forbidden (separate asm-opaque values)
__global__ __launch_bounds__(NT) void sc_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const int t = blockIdx.x * NT + threadIdx.x;
  float ax = in[t & 1023], ay = in[(t + 17) & 1023], bx = in[(t + 5) & 1023], by = in[(t + 91) & 1023];
  float sx = 0.f, sy = 0.f;
#pragma unroll 8
  for (int i = 0; i < CHAIN; ++i) {
    const float wx = 0.25f + 0.001f * i, wy = 0.75f - 0.001f * i;
    float mx = ax * wx, my = ay * wy;
    asm volatile("" : "+v"(mx));  // (keeps the SLP vectoriser from pairing x and y)
    mx = mx + bx * wy;
    my = my + by * wx;
    asm volatile("" : "+v"(my));
    sx = sx + mx * bx;
    sy = sy + my * by;
    ax = ax * 0.999f + 0.001f;
    asm volatile("" : "+v"(ax));
    ay = ay * 0.999f + 0.001f;
    const float nbx = by * 1.001f - 0.0005f;
    asm volatile("" : "+v"(sx));
    const float nby = bx * 1.001f - 0.0005f;
    bx = nbx; by = nby;
  }
  out[t] = sx + sy;
}

__global__ __launch_bounds__(256) void mfma16_kernel(float* out, int n) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
  f4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 1.2345f) out[0] = s;
}
__global__ __launch_bounds__(256) void mfma32_kernel(float* out, int n) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
  f16v c0 = {}, c1 = {};
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (s == 1.2345f) out[0] = s;
}

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma16k16_kernel(float* out, int n) {      // the older 16x16x16 shape (4-element operands)
  h4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
  f4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 1.2345f) out[0] = s;
}
__global__ __launch_bounds__(256) void mfma16bf_kernel(float* out, int n) {
  b8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  f4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 1.2345f) out[0] = s;
}

// r06: the volume kernel's issue pattern -- FOUR independent 32x32x16 accumulators, twelve MFMAs back to back (3 products x 4 column
// tiles), operands that change every burst -- because tools/pk_neighbour_scan.py finds the packed-fp32 neighbour disturbed (rarely:
// 3-8 of 800 launches, ~13 000 values each) next to corr_pyramid_h3_kernel and stem_conv7x7_s2_kernel, which use this shape only
__global__ __launch_bounds__(256) void mfma32x4_kernel(float* out, int n) {
  h8 a0, a1, b[4];
  for (int i = 0; i < 8; ++i) {
    a0[i] = (_Float16)(threadIdx.x * 0.001f + i); a1[i] = (_Float16)(0.25f * i - threadIdx.x * 0.002f);
    for (int s = 0; s < 4; ++s) b[s][i] = (_Float16)(0.5f + i + 0.125f * s);
  }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[1], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[2], c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[3], c3, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[1], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[3], c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0], c3, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[1], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[2], c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[3], c3, 0, 0, 0);
    asm volatile("" : "+v"(a0), "+v"(a1));
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 1.2345f) out[0] = s;
}

__global__ void compare_kernel(const float* a, const float* ref, int n, unsigned* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && __float_as_uint(a[i]) != __float_as_uint(ref[i])) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs; %d launches of %d x %d threads per configuration\n", prop.gcnArchName, prop.multiProcessorCount, launches, NWG, NT);
  const int n = NWG * NT;
  float *in, *out, *ref, *bo;
  unsigned* bad;
  CK(hipMalloc(&in, 1024 * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&bo, 64)); CK(hipMalloc(&bad, 4));
  std::vector<float> hin(1024);
  for (int i = 0; i < 1024; ++i) hin[i] = 0.1f + 0.8f * ((i * 2654435761u) >> 8) / 16777216.f;
  CK(hipMemcpy(in, hin.data(), 1024 * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  struct Cfg { const char* name; int packed; int load; };
  const Cfg cfgs[] = {
      {"packed fp32 (v_pk_*_f32), alone", 1, 0},
      {"packed fp32, stream B: v_mfma_f32_16x16x32_f16", 1, 1},
      {"packed fp32, stream B: v_mfma_f32_32x32x16_f16", 1, 2},
      {"packed fp32, stream B: v_mfma_f32_16x16x16_f16", 1, 3},
      {"packed fp32, stream B: v_mfma_f32_16x16x32_bf16", 1, 4},
      {"packed fp32, stream B: 32x32x16_f16, 4 accumulators x 12", 1, 5},
      {"packed fp32, stream B: 32x32x16_f16 x4, 2 workgroups / CU", 1, 6},
      {"packed fp16 (v_pk_*_f16, min / max), alone", 2, 0},
      {"packed fp16, stream B: v_mfma_f32_16x16x32_f16", 2, 1},
      {"packed fp16, stream B: v_mfma_f32_16x16x32_bf16", 2, 4},
      {"packed fp16, stream B: v_mfma_f32_32x32x16_f16", 2, 2},
      {"scalar fp32, alone", 0, 0},
      {"scalar fp32, stream B: v_mfma_f32_16x16x32_f16", 0, 1},
      {"scalar fp32, stream B: v_mfma_f32_16x16x32_bf16", 0, 4},
      {"scalar fp32, stream B: v_mfma_f32_32x32x16_f16", 0, 2},
  };
  printf("%-56s %14s %16s\n", "configuration", "bad launches", "differing words");
  for (const Cfg& c : cfgs) {
    // reference: the kernel alone on an idle chip
    CK(hipDeviceSynchronize());
    if (c.packed == 2) hipLaunchKernelGGL(pkh_kernel, dim3(NWG), dim3(NT), 0, sa, in, ref);
    else if (c.packed) hipLaunchKernelGGL(pk_kernel, dim3(NWG), dim3(NT), 0, sa, in, ref);
    else hipLaunchKernelGGL(sc_kernel, dim3(NWG), dim3(NT), 0, sa, in, ref);
    CK(hipDeviceSynchronize());
    unsigned long long words = 0;
    int badl = 0;
    for (int k = 0; k < launches; ++k) {
      CK(hipMemsetAsync(bad, 0, 4, sa));
      if (c.load == 1) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(mfma16_kernel, dim3(768), dim3(256), 0, sb, bo, 4000);
      if (c.load == 3) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(mfma16k16_kernel, dim3(768), dim3(256), 0, sb, bo, 4000);
      if (c.load == 4) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(mfma16bf_kernel, dim3(768), dim3(256), 0, sb, bo, 4000);
      if (c.load == 2) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(mfma32_kernel, dim3(768), dim3(256), 0, sb, bo, 4000);
      if (c.load == 5) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(mfma32x4_kernel, dim3(768), dim3(256), 0, sb, bo, 700);
      if (c.load == 6) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(mfma32x4_kernel, dim3(512), dim3(256), 0, sb, bo, 700);
      if (c.packed == 2) hipLaunchKernelGGL(pkh_kernel, dim3(NWG), dim3(NT), 0, sa, in, out);
      else if (c.packed) hipLaunchKernelGGL(pk_kernel, dim3(NWG), dim3(NT), 0, sa, in, out);
      else hipLaunchKernelGGL(sc_kernel, dim3(NWG), dim3(NT), 0, sa, in, out);
      hipLaunchKernelGGL(compare_kernel, dim3((n + 255) / 256), dim3(256), 0, sa, out, ref, n, bad);
      unsigned hb = 0;
      CK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, sa));
      CK(hipStreamSynchronize(sa));
      if (hb) { ++badl; words += hb; }
    }
    CK(hipDeviceSynchronize());
    printf("%-56s %8d / %-4d %16llu\n", c.name, badl, launches, words);
    fflush(stdout);
  }
  return 0;
}
