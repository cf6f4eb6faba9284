// Calibrates s_memtime (clock64) against wall time: how many ticks per microsecond, idle and right after MFMA-heavy work.
// hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void spin(long long ticks, long long* out) {
  const long long t0 = clock64(), w0 = wall_clock64();
  while (clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = clock64() - t0; out[1] = wall_clock64() - w0; }
}
__global__ void mfma_burn(int iters, float* sink, long long* out) {
  f32x16 acc = {};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = clock64() - t0; out[1] = wall_clock64() - w0; }
  if (acc[0] == 12345.f) sink[0] = acc[1];
}
int main() {
  long long* d; hipMalloc(&d, 16); float* s; hipMalloc(&s, 4);
  long long h[2]; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 20000000LL, d); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("spin (1 wave):        %lld ticks, %lld wall ticks (100 MHz), %.3f ms -> %.1f ticks/us\n", h[0], h[1], ms, h[0] / (ms * 1e3));
  }
  for (int rep = 0; rep < 3; ++rep) {
    const int iters = 200000;
    hipEventRecord(e0); hipLaunchKernelGGL(mfma_burn, dim3(256 * 8), dim3(256), 0, 0, iters, s, d); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double fl = 2.0 * iters * 32768.0 * 256 * 8 * 4;
    printf("mfma burn (full chip): %lld ticks, %.3f ms -> %.1f ticks/us, %.1f cycles per MFMA per SIMD(2 waves), %.0f TFLOP/s\n", h[0], ms,
           h[0] / (ms * 1e3), (double)h[0] / (2.0 * iters) / 2.0, fl / (ms * 1e-3) / 1e12);
  }
  return 0;
}
