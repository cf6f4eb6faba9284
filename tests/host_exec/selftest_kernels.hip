// TEST INFRASTRUCTURE: small kernels that pin down the execution model of harness.hpp itself (independent of the library's kernels).
// Written as HIP; compiled for the host by tests/test_kernels_on_host.py::test_harness_model.
#include "f16x3.cuh"

namespace {
using rp::f32x16;
using rp::h8;

// D (32 x 32) = A (32 x 16) * B (16 x 32) with ONE v_mfma_f32_32x32x16_f16 of one wave: operands gathered and results scattered in the
// register layout the library's kernels assume (lane l: A row l % 32, K slice 8 (l / 32) ..; B column l % 32, same slice; D register v:
// row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32)
__global__ __launch_bounds__(64) void mfma_tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D) {
  const int l = threadIdx.x, i = l & 31, g = l >> 5;
  h8 a, b;
  for (int k = 0; k < 8; ++k) {
    a[k] = static_cast<_Float16>(A[i * 16 + 8 * g + k]);
    b[k] = static_cast<_Float16>(B[(8 * g + k) * 32 + i]);
  }
  f32x16 c;
  for (int v = 0; v < 16; ++v) c[v] = 1.0f;                      // C = 1 everywhere
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * g + v % 4) * 32 + i] = c[v];
}

// threads with an odd index leave before the barrier; the others pass it, exchange through LDS and reduce by shuffles
__global__ __launch_bounds__(128) void barrier_exit_kernel(const int* __restrict__ in, int* __restrict__ out) {
  __shared__ int s[128];
  const int t = threadIdx.x;
  s[t] = in[blockIdx.x * 128 + t];
  if (t & 1) return;
  __syncthreads();                                               // only the even threads arrive: the odd ones have returned
  int v = s[(t + 2) % 128];                                      // written by another (even) thread
  for (int o = 32; o >= 2; o >>= 1) v += __shfl_xor(v, o);       // sum over the wave's even lanes (even ^ even: the lanes that returned are never read)
  if ((t & 63) == 0) out[blockIdx.x * 2 + (t >> 6)] = v;
}

// dynamic LDS + atomics across workgroups
__global__ __launch_bounds__(64) void dyn_lds_kernel(const float* __restrict__ x, unsigned long long* __restrict__ count, float* __restrict__ y, int n) {
  extern __shared__ float buf[];
  const int t = threadIdx.x;
  buf[t] = x[blockIdx.x * 64 + t];
  __syncthreads();
  y[blockIdx.x * 64 + t] = buf[63 - t];
  if (buf[t] > 0.f) atomicAdd(count, 1ull);
}
}  // namespace

extern "C" {
void selftest_mfma(const float* A, const float* B, float* D) { hipLaunchKernelGGL(mfma_tile_kernel, dim3(1), dim3(64), 0, nullptr, A, B, D); }
void selftest_barrier_exit(const int* in, int* out, int nblk) { hipLaunchKernelGGL(barrier_exit_kernel, dim3(nblk), dim3(128), 0, nullptr, in, out); }
void selftest_dyn_lds(const float* x, unsigned long long* count, float* y, int nblk) {
  hipLaunchKernelGGL(dyn_lds_kernel, dim3(nblk), dim3(64), 64 * sizeof(float), nullptr, x, count, y, nblk * 64);
}
}
