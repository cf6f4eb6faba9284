// a9-a11: the Levenberg-Marquardt / Gauss-Newton pose step, fully on device.
//   lm_normal_eq   Hm = sum v w J^T J, bv = sum v w J^T r  (fp64)   geometry/transformation.py:286-297
//   lm_solve       damping, 6x6 Cholesky, NaN->0, clamp, SE(3) exp, G <- exp(xi) G
//                  transformation.py:300-306, geometry/cholesky.py:32-50, geometry/se3.py:228-281,303-306
// The reference materialises J as a (B,1,H,W,2,6) fp64 tensor (29 MB/image at 480x640) and reduces it
// with two einsums; here every pixel's 2x6 Jacobian lives in registers, each thread accumulates the 21
// upper-triangle + 6 right-hand-side sums in fp64 over a strided set of pixels, a wave64 shuffle tree and
// one LDS hop reduce the workgroup, and per-workgroup partials are summed in FIXED order by the solve
// kernel (deterministic: no atomics).  HBM traffic = target + weight + depth read once (16 B/pixel).
#include "geometry.cuh"

namespace {

using rp::Intr;
using rp::Pose;

constexpr int NACC = 27;        // 21 (upper triangle of H, row-major) + 6 (b)
constexpr int PSTRIDE = 32;     // doubles per partial record
constexpr int LM_THREADS = 256;
// r04 sweep (B = 4 / B = 8 x 480 x 640, us per fused LM step): 1024 px per workgroup, 4 in flight 40.6 / 70.7; 2048, 4 (r03) 35.9 / 52.4;
// 4096, 8 33.9 / 40.5; 8192, 8 37.1 / 46.9; 16384, 16 58.7 / 59.6 -- fewer partial records shorten the last-arrival tail until the
// main loop of a workgroup becomes the chain
#ifndef RP_LM_BATCH
#define RP_LM_BATCH 8
#endif
#ifndef RP_LM_PPB
#define RP_LM_PPB 4096
#endif
constexpr int LM_BATCH = RP_LM_BATCH;      // pixels per thread whose loads are in flight together
constexpr int LM_PIX_PER_BLOCK = RP_LM_PPB;   // 75 workgroups per 480x640 image
constexpr int LM_MAX_BLOCKS = 256;

__host__ __device__ inline int lm_blocks_per_image(long long P) {
  long long n = (P + LM_PIX_PER_BLOCK - 1) / LM_PIX_PER_BLOCK;
  if (n < 1) n = 1;
  if (n > LM_MAX_BLOCKS) n = LM_MAX_BLOCKS;
  return static_cast<int>(n);
}

__device__ __forceinline__ double shfl_down_f64(double v, int d) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_down(lo, d);
  hi = __shfl_down(hi, d);
  return __hiloint2double(hi, lo);
}

// FUSED tail (r03): the workgroup that arrives LAST for its image (device-scope ticket, one counter per image at the end of
// the workspace) sums the partial records, damps, solves and updates the pose -- lm_finalize + lm_solve_update without their
// two launches and kernel boundaries (~13 us + 2 x ~1.5 us per LM step, 48 steps per refinement).  r02 tried "finalize + solve
// as one 256-thread launch" and lost (the serial fp64 solve waited behind the launch); here it runs in a block that is already
// resident, while the other images' blocks are still computing.  Hand-off = MI355X_MICROARCH.md's counter form: plain stores ->
// barrier -> lane-0 agent-scope release (+ asm vmcnt(0)) -> relaxed agent fetch_add; the last arriver: agent-scope acquire ->
// barrier -> plain loads.  The last arriver resets the counter (zero-initialised with the workspace).
struct LmTail {
  int* tickets;                 // (B) arrival counters, zero between launches
  double* Hm;                   // (B,6,6), (B,6): undamped system, as the unfused path returns it
  double* bv;
  const float* G_in;            // pose the Jacobians were built with (may alias G_out)
  float* G_out;
  float* xi;
  int* info;
  double ep, lm, max_update;
};
__device__ void lm_finalize_block_one(const double* __restrict__ partials, int nblk, int b, double* __restrict__ Hm, double* __restrict__ bv);
__device__ __forceinline__ void lm_solve_one(const double* Hm, const double* bv, const float* G, int b, double ep, double lm, double max_update,
                             float* G_new, float* xi_out, int* info);

// one pixel's contribution to the 21 + 6 sums (fp64), J and the residual from the TRANSFORMED point
__device__ __forceinline__ void lm_accumulate(double (&acc)[NACC], float wgt, float dep, float tx, float ty, int x, int y, int target_mode,
                                          float eps, const Intr& k, const Pose& g) {
  const float Z = dep + eps;
  if (target_mode != 0) {
    tx += static_cast<float>(x);
    ty += static_cast<float>(y);
  }
  const rp::Reproj r = rp::reproject(Z, static_cast<float>(x), static_cast<float>(y), k, g);
  const bool valid = (r.Z0 > rp::kMinDepthValid) && (r.Z1 > rp::kMinDepthValid);   // transformation.py:289
  const double vw = valid ? static_cast<double>(wgt) : 0.0;
  // J_pi in fp32 (projective_ops.py:118-124): zeros where clamped Z <= 0.02
  const bool tiny = r.Zc <= rp::kMinDepthProj + 0.01f;
  const float zi1 = tiny ? 0.f : 1.0f / r.Zc;
  const float zi2 = tiny ? 0.f : 1.0f / (r.Zc * r.Zc);
  const double a = static_cast<double>(k.fx * zi1);
  const double c = static_cast<double>(-k.fx * r.X1 * zi2);
  const double d = static_cast<double>(k.fy * zi1);
  const double e = static_cast<double>(-k.fy * r.Y1 * zi2);
  const double X1 = r.X1, Y1 = r.Y1, Z1 = r.Z1;
  // J = J_pi * J_T with J_T = [I | -[X']x] built from the TRANSFORMED point (transformation.py:27-46,85-90)
  double J0[6], J1[6];
  J0[0] = a;   J0[1] = 0.0; J0[2] = c; J0[3] = c * Y1;            J0[4] = a * Z1 + c * (-X1); J0[5] = a * (-Y1);
  J1[0] = 0.0; J1[1] = d;   J1[2] = e; J1[3] = d * (-Z1) + e * Y1; J1[4] = e * (-X1);          J1[5] = d * X1;
  const double r0 = static_cast<double>(tx) - static_cast<double>(r.u);
  const double r1 = static_cast<double>(ty) - static_cast<double>(r.v);
  int idx = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double wi0 = vw * J0[i], wi1 = vw * J1[i];
#pragma unroll
    for (int jj = i; jj < 6; ++jj) {
      acc[idx] += wi0 * J0[jj] + wi1 * J1[jj];
      ++idx;
    }
    acc[21 + i] += wi0 * r0 + wi1 * r1;
  }
}

// the workgroup's 27 sums -> its partial record; FUSED: the last workgroup of the image finalizes, solves and updates the pose
template <bool FUSED>
__device__ __forceinline__ void lm_reduce_tail(double (&acc)[NACC], double* __restrict__ partials, const LmTail& tail, int b, int nblk, int bx) {
  __shared__ double red[LM_THREADS / 64][NACC];
  __shared__ int is_last;
  // wave reduction as a butterfly REDUCE-SCATTER (r04): at offset o = 32, 16, 8, 4, 2 a lane keeps the half of its (32, 16, 8, 4,
  // 2) slots that bit o of its lane number selects, sends the other half to lane ^ o and adds what it receives -- 16 + 8 + 4 + 2 +
  // 1 exchanged doubles per lane, after which lanes 2k and 2k + 1 hold the two halves of slot k = lane >> 1 (bit order below) and one
  // last exchange completes it: 32 double exchanges per lane instead of the 27 x 6 = 162 of a full butterfly per accumulator
  // (the reduction was as long as the fp64 accumulation itself).  Fixed tree: deterministic.  Then across the 4 waves through LDS.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    double v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = i < NACC ? acc[i] : 0.0;
#pragma unroll
    for (int o = 32, n = 32; o >= 2; o >>= 1, n >>= 1) {         // n slots live before the step, n / 2 after
      const bool hi = (lane & o) != 0;
#pragma unroll
      for (int i = 0; i < n / 2; ++i) {
        const double send = hi ? v[i] : v[i + n / 2];
        const double keep = hi ? v[i + n / 2] : v[i];
        v[i] = keep + rp::shfl_xor_f64(send, o);
      }
    }
    const double tot = v[0] + rp::shfl_xor_f64(v[0], 1);
    // slot held by this lane pair: bit 5 of the lane chose the upper half of 32, bit 4 of 16, ... bit 1 of 2
    const int slot = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if (!(lane & 1) && slot < NACC) red[wave][slot] = tot;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double v = red[0][threadIdx.x];
#pragma unroll
    for (int wv = 1; wv < LM_THREADS / 64; ++wv) v += red[wv][threadIdx.x];
    partials[(static_cast<long long>(b) * nblk + bx) * PSTRIDE + threadIdx.x] = v;
  }
  if constexpr (FUSED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the compiler may drop the wait behind buffer_wbl2: guide, G16 pitfall)
      const int t = __hip_atomic_fetch_add(&tail.tickets[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = (t == nblk - 1) ? 1 : 0;
      if (is_last) {
        __hip_atomic_store(&tail.tickets[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        // ONE agent-scope acquire per workgroup: it invalidates this CU's vector L1 (buffer_inv sc1), which is per CU, not per
        // thread -- followed by the barrier below before any thread of the block loads the other workgroups' partial records
        // (cdna_hip_programming.md, Guideline 16: "consumer: one relaxed poll -> one agent acquire -> __syncthreads() -> plain loads")
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    if (!is_last) return;
    lm_finalize_block_one(partials, nblk, b, tail.Hm, tail.bv);   // (ends with the values in global memory, written by this block)
    __syncthreads();
    if (threadIdx.x == 0) lm_solve_one(tail.Hm, tail.bv, tail.G_in, b, tail.ep, tail.lm, tail.max_update, tail.G_out, tail.xi, tail.info);
  }
}

template <bool FUSED>
__global__ __launch_bounds__(LM_THREADS) void lm_normal_eq_kernel(const float* __restrict__ target, int target_mode,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ depth, float eps,
                                                                  const float* __restrict__ K,
                                                                  const float* __restrict__ G, int H, int W,
                                                                  double* __restrict__ partials, const LmTail tail) {
  const int b = blockIdx.y;
  const int nblk = gridDim.x;
  const long long P = static_cast<long long>(H) * W;
  const Intr k = rp::load_intr(K, b);
  const Pose g = rp::load_pose(G, b);
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;

  // LM_BATCH pixels per trip: their 4-5 loads each are issued (unconditionally, clamped to the last pixel) before the
  // first fp64 chain starts.  One pixel per trip was 8 dependent memory round trips per thread: 26 us per launch for
  // 20 MB (r02).  Pixels are accumulated in the same order as before, so the partial sums are bit-identical.
  const long long stride = static_cast<long long>(nblk) * LM_THREADS;
  for (long long t0 = static_cast<long long>(blockIdx.x) * LM_THREADS + threadIdx.x; t0 < P; t0 += LM_BATCH * stride) {
    float wgt_[LM_BATCH], dep_[LM_BATCH], tx_[LM_BATCH], ty_[LM_BATCH];
#pragma unroll
    for (int j = 0; j < LM_BATCH; ++j) {
      const long long tj = t0 + j * stride;
      const long long t = tj < P ? tj : P - 1;
      wgt_[j] = weight[b * P + t];
      dep_[j] = depth[b * P + t];
      if (target_mode == 0) {
        const float2 tt = *reinterpret_cast<const float2*>(target + (b * P + t) * 2);
        tx_[j] = tt.x;
        ty_[j] = tt.y;
      } else {
        tx_[j] = target[(static_cast<long long>(b) * 2 + 0) * P + t];
        ty_[j] = target[(static_cast<long long>(b) * 2 + 1) * P + t];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < LM_BATCH; ++j) {
      const long long t = t0 + j * stride;
      if (t >= P) break;
      const unsigned tu = static_cast<unsigned>(t);            // t < H*W < 2^31: 32-bit division (64-bit is a software routine)
      const int y = static_cast<int>(tu / static_cast<unsigned>(W)), x = static_cast<int>(tu - static_cast<unsigned>(y) * static_cast<unsigned>(W));
      const float wgt = wgt_[j];
      // zero-weight pixels (the descriptor weight is 0 on the rendered background) add w * (...) = 0 to every sum: a wave
      // made of such pixels skips the fp64 chain -- but only pixels whose target and depth are FINITE count as skippable: in the
      // reference 0 * NaN = NaN poisons the system and the NaN -> zero-update guard fires (geometry/cholesky.py:43-44); a
      // lane with a non-finite input keeps its wave in the chain, so that happens here too, whatever the wave is made of.
      const bool skippable = wgt == 0.f && __builtin_isfinite(tx_[j]) && __builtin_isfinite(ty_[j]) && __builtin_isfinite(dep_[j]);
      if (__builtin_amdgcn_ballot_w64(!skippable) == 0ull) continue;
      lm_accumulate(acc, wgt, dep_[j], tx_[j], ty_[j], x, y, target_mode, eps, k, g);
    }
  }
  lm_reduce_tail<FUSED>(acc, partials, tail, b, nblk, static_cast<int>(blockIdx.x));
}

// sums the block partials in fixed order and expands to full H (6x6) and b (6)
__device__ void lm_finalize_block_one(const double* __restrict__ partials, int nblk, int b, double* __restrict__ Hm,
                                      double* __restrict__ bv) {
  // 8 groups of 32 lanes walk the partial records with stride 8 (each load is one coalesced 256-byte record), then
  // the 8 group sums are added in fixed order: deterministic, and 8x shorter than one serial chain per value.
  __shared__ double grp[8][PSTRIDE];
  __shared__ double s[NACC];
  const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
  double v = 0.0;
  for (int i = g; i < nblk; i += 8) v += partials[(static_cast<long long>(b) * nblk + i) * PSTRIDE + k];
  grp[g][k] = v;
  __syncthreads();
  if (threadIdx.x < NACC) {
    double t = 0.0;
    for (int q = 0; q < 8; ++q) t += grp[q][threadIdx.x];
    s[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < 36) {
    const int i = threadIdx.x / 6, j = threadIdx.x % 6;
    const int r = i < j ? i : j, c = i < j ? j : i;
    const int idx = r * 6 - r * (r - 1) / 2 + (c - r);     // row-major upper triangle
    Hm[b * 36 + threadIdx.x] = s[idx];
  } else if (threadIdx.x < 42) {
    bv[b * 6 + (threadIdx.x - 36)] = s[21 + (threadIdx.x - 36)];
  }
}

__global__ __launch_bounds__(256) void lm_finalize_kernel(const double* __restrict__ partials, int nblk,
                                                          double* __restrict__ Hm, double* __restrict__ bv) {
  lm_finalize_block_one(partials, nblk, blockIdx.x, Hm, bv);
}

// ---- SE(3) exponential, fp32, same branch structure as geometry/se3.py:228-281 ----
__device__ void se3_exp_dev(const float* xi, float* Gout /*16*/) {
  const float v0 = xi[0], v1 = xi[1], v2 = xi[2];
  const float w0 = xi[3], w1 = xi[4], w2 = xi[5];
  const float th2 = w0 * w0 + w1 * w1 + w2 * w2;
  const float th = sqrtf(th2);
  const float th4 = th2 * th2;
  const float wx[9] = {0.f, -w2, w1, w2, 0.f, -w0, -w1, w0, 0.f};
  float wx2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) wx2[i * 3 + j] = wx[i * 3 + 0] * wx[0 * 3 + j] + wx[i * 3 + 1] * wx[1 * 3 + j] + wx[i * 3 + 2] * wx[2 * 3 + j];
  float A, Bc, Cc;   // R = I + A wx + Bc wx2 ; V = I + Bc' wx + Cc wx2
  float Bv;
  const float eps = 1e-12f;
  if (th < rp::kMinTheta) {
    A = 1.0f - (1.0f / 6.0f) * th2 + (1.0f / 120.0f) * th4;
    Bc = 0.5f - (1.0f / 12.0f) * th2 + (1.0f / 720.0f) * th4;
    Bv = 0.5f - (1.0f / 24.0f) * th2 + (1.0f / 720.0f) * th4;
    Cc = (1.0f / 6.0f) - (1.0f / 120.0f) * th2 + (1.0f / 5040.0f) * th4;
  } else {
    A = sinf(th) / (th + eps);
    Bc = (1.0f - cosf(th)) / (th2 + eps);
    Bv = Bc;
    Cc = (th - sinf(th)) / (th2 * th + eps);
  }
  float Rm[9], Vm[9];
  for (int i = 0; i < 9; ++i) {
    const float I = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
    Rm[i] = I + A * wx[i] + Bc * wx2[i];
    Vm[i] = I + Bv * wx[i] + Cc * wx2[i];
  }
  for (int i = 0; i < 3; ++i) {
    Gout[i * 4 + 0] = Rm[i * 3 + 0];
    Gout[i * 4 + 1] = Rm[i * 3 + 1];
    Gout[i * 4 + 2] = Rm[i * 3 + 2];
    Gout[i * 4 + 3] = Vm[i * 3 + 0] * v0 + Vm[i * 3 + 1] * v1 + Vm[i * 3 + 2] * v2;
  }
  Gout[12] = 0.f; Gout[13] = 0.f; Gout[14] = 0.f; Gout[15] = 1.f;
}

__device__ void mat4_mul(const float* A, const float* Bm, float* C) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * Bm[k * 4 + j];
      C[i * 4 + j] = s;
    }
}

// one thread per image: damping, Cholesky, substitutions, guards, exp, left increment
// damping, 6x6 Cholesky solve, NaN -> 0, clamp, SE(3) exponential and left increment of image b (one thread)
__device__ __forceinline__ void lm_solve_one(const double* Hm, const double* bv, const float* G, int b, double ep, double lm,
                                             double max_update, float* G_new /* may alias G */, float* xi_out, int* info) {
  // (the fused tail reads H, b that this very thread's block wrote a barrier ago: same CU, same L1 -- plain loads are current)
  // In place on the lower triangle (21 doubles instead of two 6x6 arrays: this function's registers are the floor of every kernel
  // that calls it, r04) -- the same operations in the same order as the two-array form.
  double L[21], xv[6];
#define LT(i_, j_) L[(i_) * ((i_) + 1) / 2 + (j_)]
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) LT(i, j) = Hm[b * 36 + i * 6 + j];
    xv[i] = bv[b * 6 + i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) LT(i, i) = LT(i, i) + ep + lm * LT(i, i);   // H += ep*I + lm*H*I  (transformation.py:300)
  int bad = 0;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = LT(j, j);
#pragma unroll
    for (int k = 0; k < j; ++k) s -= LT(j, k) * LT(j, k);
    if (!(s > 0.0)) {
      if (!bad) bad = j + 1;
      LT(j, j) = nan;
    } else {
      LT(j, j) = sqrt(s);
    }
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = LT(i, j);
#pragma unroll
      for (int k = 0; k < j; ++k) t -= LT(i, k) * LT(j, k);
      LT(i, j) = t / LT(j, j);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {              // forward substitution, in place (xv: rhs -> y)
    double t = xv[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= LT(i, k) * xv[k];
    xv[i] = t / LT(i, i);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {             // back substitution, in place (y -> x)
    double t = xv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) t -= LT(k, i) * xv[k];
    xv[i] = t / LT(i, i);
  }
#undef LT
  float xi[6];
  for (int i = 0; i < 6; ++i) {
    double v = xv[i];
    if (v != v) v = 0.0;                                   // NaN -> 0      (cholesky.py:43-44)
    v = v < -max_update ? -max_update : (v > max_update ? max_update : v);   // clamp (cholesky.py:45)
    xi[i] = static_cast<float>(v);
    xi_out[b * 6 + i] = xi[i];
  }
  if (info) info[b] = bad;
  float dG[16], Gin[16], Gout[16];
  se3_exp_dev(xi, dG);
  for (int i = 0; i < 16; ++i) Gin[i] = G[b * 16 + i];
  mat4_mul(dG, Gin, Gout);                                 // se3.py:303-306
  for (int i = 0; i < 16; ++i) G_new[b * 16 + i] = Gout[i];
}

__global__ __launch_bounds__(64) void lm_solve_update_kernel(const double* __restrict__ Hm, const double* __restrict__ bv,
                                                             const float* G, int B, double ep, double lm,
                                                             double max_update, float* G_new /* may alias G */,
                                                             float* __restrict__ xi_out, int* __restrict__ info) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) lm_solve_one(Hm, bv, G, b, ep, lm, max_update, G_new, xi_out, info);
}

__global__ void se3_exp_kernel(const float* __restrict__ xi, int B, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float x[6], Gm[16];
  for (int i = 0; i < 6; ++i) x[i] = xi[b * 6 + i];
  se3_exp_dev(x, Gm);
  for (int i = 0; i < 16; ++i) out[b * 16 + i] = Gm[i];
}

__global__ void se3_compose_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int B, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float a[16], c[16], o[16];
  for (int i = 0; i < 16; ++i) {
    a[i] = A[b * 16 + i];
    c[i] = Bm[b * 16 + i];
  }
  mat4_mul(a, c, o);
  for (int i = 0; i < 16; ++i) out[b * 16 + i] = o[i];
}

// [R t; 0 1]^-1 = [R^T  -R^T t; 0 1]   (geometry/se3.py:194-209)
__global__ void se3_inverse_kernel(const float* __restrict__ A, int B, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float a[16], o[16];
  for (int i = 0; i < 16; ++i) a[i] = A[b * 16 + i];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o[i * 4 + j] = a[j * 4 + i];
    float s = 0.f;
    for (int k = 0; k < 3; ++k) s += a[k * 4 + i] * a[k * 4 + 3];
    o[i * 4 + 3] = -s;
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
  for (int i = 0; i < 16; ++i) out[b * 16 + i] = o[i];
}

// a12, r06: the pose bookkeeping between two outer iterations as ONE launch (model/PoseRefiner.py:241-244) -- Ti <- Tij Ti, then the legacy start pose
// Tij <- Ti Ti^-1 (literal != 0: the reference's product, the identity up to fp32 rounding) or the exact identity.  The same three kernels' arithmetic
// in the same order (se3_compose, se3_inverse, se3_compose above): bit-identical to them; five ~6-us launches fewer per outer iteration.
__global__ void se3_outer_update_kernel(const float* __restrict__ Tij, const float* __restrict__ Ti, int B, int literal, float* __restrict__ Ti_out,
                                        float* __restrict__ Tij_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float a[16], c[16], t[16], inv[16], o[16];
  for (int i = 0; i < 16; ++i) {
    a[i] = Tij[b * 16 + i];
    c[i] = Ti[b * 16 + i];
  }
  mat4_mul(a, c, t);
  for (int i = 0; i < 16; ++i) Ti_out[b * 16 + i] = t[i];
  if (literal) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) inv[i * 4 + j] = t[j * 4 + i];
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += t[k * 4 + i] * t[k * 4 + 3];
      inv[i * 4 + 3] = -s;
    }
    inv[12] = 0.f; inv[13] = 0.f; inv[14] = 0.f; inv[15] = 1.f;
    mat4_mul(t, inv, o);
  } else {
    for (int i = 0; i < 16; ++i) o[i] = (i % 5 == 0) ? 1.f : 0.f;
  }
  for (int i = 0; i < 16; ++i) Tij_out[b * 16 + i] = o[i];
}

// workspace = [B arrival counters, 8 bytes each: zero between launches][partial records]
inline double* lm_partials(void* workspace, int B) { return static_cast<double*>(workspace) + B; }

// per-workgroup partial sums into `workspace`; -> number of partial records per image
int launch_normal_eq_partials(const float* target, int target_mode, const float* weight, const float* depth, float eps,
                              const float* K, const float* G, int B, int H, int W, void* workspace, hipStream_t st) {
  const long long P = static_cast<long long>(H) * W;
  const int nblk = lm_blocks_per_image(P);
  hipLaunchKernelGGL(lm_normal_eq_kernel<false>, dim3(nblk, B), dim3(LM_THREADS), 0, st, target, target_mode, weight, depth, eps,
                     K, G, H, W, lm_partials(workspace, B), LmTail{});
  return nblk;
}

// partial sums + (in the last-arriving workgroup of every image) finalize, damped solve, pose update: ONE launch per LM step
void launch_lm_step_fused(const float* target, int target_mode, const float* weight, const float* depth, float eps, const float* K,
                          const float* G_in, float* G_out, int B, int H, int W, double ep, double lm, double max_update,
                          void* workspace, double* Hm, double* bv, float* xi, int* info, hipStream_t st) {
  const long long P = static_cast<long long>(H) * W;
  const int nblk = lm_blocks_per_image(P);
  LmTail tail{};
  tail.tickets = static_cast<int*>(workspace);
  tail.Hm = Hm; tail.bv = bv; tail.G_in = G_in; tail.G_out = G_out; tail.xi = xi; tail.info = info;
  tail.ep = ep; tail.lm = lm; tail.max_update = max_update;
  hipLaunchKernelGGL(lm_normal_eq_kernel<true>, dim3(nblk, B), dim3(LM_THREADS), 0, st, target, target_mode, weight, depth, eps,
                     K, G_in, H, W, lm_partials(workspace, B), tail);
}

int launch_normal_eq(const float* target, int target_mode, const float* weight, const float* depth, float eps,
                     const float* K, const float* G, int B, int H, int W, void* workspace, double* Hm, double* bv,
                     hipStream_t st) {
  const int nblk = launch_normal_eq_partials(target, target_mode, weight, depth, eps, K, G, B, H, W, workspace, st);
  hipLaunchKernelGGL(lm_finalize_kernel, dim3(B), dim3(256), 0, st, lm_partials(workspace, B), nblk, Hm, bv);
  return 0;
}

}  // namespace

extern "C" {

size_t rnnpose_lm_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  // partial records + one arrival counter per image (padded to 8 bytes each); the counters must be ZERO before the first fused
  // step (rnnpose_lm_step_*): allocate the workspace zero-filled.  Every fused launch leaves them at zero.
  return static_cast<size_t>(B) * lm_blocks_per_image(static_cast<long long>(H) * W) * PSTRIDE * sizeof(double) +
         static_cast<size_t>(B) * sizeof(double);
}

static bool g_lm_fused = true;
int rnnpose_lm_fused_tail(int enable) {          // measurement switch: 0 = three launches per LM step (r02), 1 = one (default)
  g_lm_fused = enable != 0;
  return 0;
}

int rnnpose_lm_normal_eq_f64(const float* target, int target_mode, const float* weight, const float* depth,
                             float depth_eps, const float* K, const float* G, int B, int H, int W, void* workspace,
                             size_t workspace_bytes, double* Hm, double* bv, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_lm_normal_eq_f64";
  RP_REQUIRE(target && weight && depth && K && G && workspace && Hm && bv, fn, "null pointer");
  RP_REQUIRE(target_mode == 0 || target_mode == 1, fn, "target_mode must be 0 or 1");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && static_cast<long long>(H) * W < (1LL << 31), fn, "bad size");
  RP_REQUIRE(workspace_bytes >= rnnpose_lm_workspace_bytes(B, H, W), fn, "workspace too small");
  launch_normal_eq(target, target_mode, weight, depth, depth_eps, K, G, B, H, W, workspace, Hm, bv, rp::as_stream(stream));
  return rp::check_launch(fn);
}

int rnnpose_lm_solve_update_f32(const double* Hm, const double* bv, const float* G, int B, double ep_lambda,
                                double lm_lambda, double max_update, float* G_new, float* xi, int* info,
                                rnnpose_stream_t stream) {
  const char* fn = "rnnpose_lm_solve_update_f32";
  RP_REQUIRE(Hm && bv && G && G_new && xi, fn, "null pointer");
  RP_REQUIRE(B > 0, fn, "bad size");
  hipLaunchKernelGGL(lm_solve_update_kernel, dim3(rp::cdiv(B, 64)), dim3(64), 0, rp::as_stream(stream), Hm, bv, G, B,
                     ep_lambda, lm_lambda, max_update, G_new, xi, info);
  return rp::check_launch(fn);
}

static int lm_step_impl(const char* fn, const float* target, int target_mode, const float* weight, const float* depth,
                        float depth_eps, const float* K, const float* G_in, float* G_out, int B, int H, int W, int num_iters,
                        double ep_lambda, double lm_lambda, double max_update, void* workspace, size_t workspace_bytes,
                        double* Hm, double* bv, float* xi, int* info, rnnpose_stream_t stream) {
  RP_REQUIRE(target && weight && depth && K && G_in && G_out && workspace && Hm && bv && xi, fn, "null pointer");
  RP_REQUIRE(target_mode == 0 || target_mode == 1, fn, "target_mode must be 0 or 1");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && static_cast<long long>(H) * W < (1LL << 31) && num_iters >= 0, fn, "bad size");
  RP_REQUIRE(workspace_bytes >= rnnpose_lm_workspace_bytes(B, H, W), fn, "workspace too small");
  hipStream_t st = rp::as_stream(stream);
  for (int it = 0; it < num_iters; ++it) {
    const float* g = it == 0 ? G_in : G_out;            // later iterations continue in place on the output
    if (g_lm_fused && info) {
      launch_lm_step_fused(target, target_mode, weight, depth, depth_eps, K, g, G_out, B, H, W, ep_lambda, lm_lambda, max_update,
                           workspace, Hm, bv, xi, info, st);
      continue;
    }
    launch_normal_eq(target, target_mode, weight, depth, depth_eps, K, g, B, H, W, workspace, Hm, bv, st);
    // (g may alias G_out: each thread reads its whole pose before writing it.  A merged finalize + solve kernel was measured
    //  SLOWER, 26 us vs 7 + 6 us: the solve is a serial fp64 chain that then waits behind the 256-thread reduction's launch)
    hipLaunchKernelGGL(lm_solve_update_kernel, dim3(rp::cdiv(B, 64)), dim3(64), 0, st, Hm, bv, g, B, ep_lambda,
                       lm_lambda, max_update, G_out, xi, info);
  }
  return rp::check_launch(fn);
}

int rnnpose_lm_step_f32(const float* target, int target_mode, const float* weight, const float* depth, float depth_eps,
                        const float* K, float* G, int B, int H, int W, int num_iters, double ep_lambda, double lm_lambda,
                        double max_update, void* workspace, size_t workspace_bytes, double* Hm, double* bv, float* xi,
                        int* info, rnnpose_stream_t stream) {
  return lm_step_impl("rnnpose_lm_step_f32", target, target_mode, weight, depth, depth_eps, K, G, G, B, H, W, num_iters,
                      ep_lambda, lm_lambda, max_update, workspace, workspace_bytes, Hm, bv, xi, info, stream);
}

int rnnpose_lm_step_io_f32(const float* target, int target_mode, const float* weight, const float* depth, float depth_eps,
                           const float* K, const float* G_in, float* G_out, int B, int H, int W, int num_iters,
                           double ep_lambda, double lm_lambda, double max_update, void* workspace, size_t workspace_bytes,
                           double* Hm, double* bv, float* xi, int* info, rnnpose_stream_t stream) {
  RP_REQUIRE(num_iters >= 1, "rnnpose_lm_step_io_f32", "needs at least one iteration (G_out would stay unwritten)");
  return lm_step_impl("rnnpose_lm_step_io_f32", target, target_mode, weight, depth, depth_eps, K, G_in, G_out, B, H, W,
                      num_iters, ep_lambda, lm_lambda, max_update, workspace, workspace_bytes, Hm, bv, xi, info, stream);
}

int rnnpose_se3_exp_f32(const float* xi, int B, float* out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_se3_exp_f32";
  RP_REQUIRE(xi && out && B > 0, fn, "bad argument");
  hipLaunchKernelGGL(se3_exp_kernel, dim3(rp::cdiv(B, 64)), dim3(64), 0, rp::as_stream(stream), xi, B, out);
  return rp::check_launch(fn);
}

int rnnpose_se3_compose_f32(const float* A, const float* Bm, int B, float* out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_se3_compose_f32";
  RP_REQUIRE(A && Bm && out && B > 0, fn, "bad argument");
  hipLaunchKernelGGL(se3_compose_kernel, dim3(rp::cdiv(B, 64)), dim3(64), 0, rp::as_stream(stream), A, Bm, B, out);
  return rp::check_launch(fn);
}

int rnnpose_se3_outer_update_f32(const float* Tij, const float* Ti, int B, int literal, float* Ti_out, float* Tij_out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_se3_outer_update_f32";
  RP_REQUIRE(Tij && Ti && Ti_out && Tij_out && B > 0, fn, "null pointer / empty batch");
  hipLaunchKernelGGL(se3_outer_update_kernel, dim3(rp::cdiv(B, 64)), dim3(64), 0, rp::as_stream(stream), Tij, Ti, B, literal, Ti_out, Tij_out);
  return rp::check_launch(fn);
}

int rnnpose_se3_inverse_f32(const float* A, int B, float* out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_se3_inverse_f32";
  RP_REQUIRE(A && out && B > 0, fn, "bad argument");
  hipLaunchKernelGGL(se3_inverse_kernel, dim3(rp::cdiv(B, 64)), dim3(64), 0, rp::as_stream(stream), A, B, out);
  return rp::check_launch(fn);
}

}  // extern "C"
