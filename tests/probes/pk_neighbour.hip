// TEST-SIDE source (not part of the library): a neighbour kernel the way an integrator's own code would be compiled -- plain -O3, PACKED
// fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 from explicit two-wide vectors) -- for the co-tenancy test of
// tests/test_gpu_reproducibility.py: it runs on its own stream next to the refinement loop and ITS output is checked against a launch
// that ran alone (DESIGN 5 rule 11, INTEGRATION "packed fp32").  Same arithmetic as tools/probes/pk_f32_vs_mfma.hip's pk_kernel.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tests/probes/pk_neighbour.hip -o <tmp>/libpk_neighbour.so
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NT = 256;

__global__ __launch_bounds__(NT) void pk_kernel(const float* __restrict__ in, float* __restrict__ out, int chain) {
  const int t = blockIdx.x * NT + threadIdx.x;
  f2 a = {in[t & 1023], in[(t + 17) & 1023]}, b = {in[(t + 5) & 1023], in[(t + 91) & 1023]};
  f2 s = {0.f, 0.f};
#pragma unroll 8
  for (int i = 0; i < chain; ++i) {
    const f2 w = {0.25f + 0.001f * (i & 127), 0.75f - 0.001f * (i & 127)};
    f2 m = a * w;
    m = m + b * w.yx;
    s = s + m * b;
    a = a * 0.999f + 0.001f;
    b = b.yx * 1.001f - 0.0005f;
  }
  out[t] = s.x + s.y;
}

// out[i] != ref[i] counted on the device (one counter per launch slot): the host never synchronises inside the measured window
__global__ __launch_bounds__(NT) void pk_compare(const float* __restrict__ out, const float* __restrict__ ref, int n, unsigned* __restrict__ bad) {
  const int t = blockIdx.x * NT + threadIdx.x;
  if (t < n && __float_as_uint(out[t]) != __float_as_uint(ref[t])) atomicAdd(bad, 1u);
}

extern "C" {
int pk_run(const float* in, float* out, int nwg, int chain, void* stream) {
  hipLaunchKernelGGL(pk_kernel, dim3(nwg), dim3(NT), 0, static_cast<hipStream_t>(stream), in, out, chain);
  return static_cast<int>(hipGetLastError());
}
int pk_check(const float* out, const float* ref, int n, unsigned* bad, void* stream) {
  hipLaunchKernelGGL(pk_compare, dim3((n + NT - 1) / NT), dim3(NT), 0, static_cast<hipStream_t>(stream), out, ref, n, bad);
  return static_cast<int>(hipGetLastError());
}
}
