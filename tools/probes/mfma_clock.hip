// Stand-alone probe: the shader clock MI355X sustains under a matrix-pipe-saturating load ("add the box's sustained
// MFMA clock as a measured line").  Every workgroup runs a loop of independent v_mfma_f32_32x32x16_f16 (two waves per SIMD, like the
// strip convolution kernels) and stamps both clocks at its start and end: s_memtime counts SHADER cycles, s_memrealtime counts the
// constant 100-MHz reference clock -> effective clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, per workgroup; and the matrix-pipe
// rate = MFMAs issued / wall time against 2.5 PFLOP/s.  Operands: random fp16 (zero operands clock higher: MI355X_MICROARCH.md, DVFS).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_clock.hip -o tools/probes/mfma_clock && tools/probes/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const _Float16* __restrict__ src, float* out, unsigned long long* stamps, int n) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 8 + i) & 4095]; b[i] = src[(threadIdx.x * 8 + i + 2048) & 4095]; }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 1.2345f) out[0] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;       // MFMA groups of 4 per wave: ~2 x 4 x 32 cycles per group on a SIMD with two waves
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int nwg = 2 * prop.multiProcessorCount;        // 8 waves per CU = 2 per SIMD
  _Float16* src; float* out; unsigned long long* st;
  CK(hipMalloc(&src, 4096 * 2)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&st, nwg * 16));
  std::vector<_Float16> h(4096);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  CK(hipMemcpy(src, h.data(), 4096 * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# device %s (%s), %d CUs, nominal clock %d MHz; %d workgroups x 256 threads, %d x 4 MFMA 32x32x16 f16 per wave\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate / 1000, nwg, n);
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_loop, dim3(nwg), dim3(256), 0, 0, src, out, st, n);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hs(nwg * 2);
    CK(hipMemcpy(hs.data(), st, nwg * 16, hipMemcpyDeviceToHost));
    std::vector<double> mhz(nwg);
    for (int i = 0; i < nwg; ++i) mhz[i] = 100.0 * (double)hs[2 * i] / (double)hs[2 * i + 1];
    std::sort(mhz.begin(), mhz.end());
    const double flops = 2.0 * 32 * 32 * 16 * 4.0 * n * 4.0 * nwg;        // per MFMA x 4 per group x n x 4 waves x workgroups
    printf("run %d: %.2f ms; shader clock under load (s_memtime / s_memrealtime): min %.0f  median %.0f  max %.0f MHz; matrix pipe %.0f TFLOP/s = %.3f of 2500\n",
           rep, ms, mhz.front(), mhz[nwg / 2], mhz.back(), flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2500.0);
  }
  return 0;
}
