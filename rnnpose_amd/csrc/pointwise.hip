// HBM-bound per-pixel stages of the refinement loop:
//   a5  context_prep / flow_to_coords            model/CFNet.py:124-144
//   a6  convex_upsample                          model/CFNet.py:95-106
//   a7  induced_flow / induced_coords_lowres     geometry/transformation.py:184-198, model/PoseRefiner.py:324-328
//   a8  corr_weight                              model/PoseRefiner.py:342-345
//   a4  SepConvGRU gate / state update           thirdparty/raft/update.py:45-60
// Every kernel reads each input byte once with lane-contiguous addresses and writes coalesced rows.
#include "induced.cuh"
#include "descriptor_weight.cuh"

namespace {

using rp::Intr;
using rp::Pose;

using rp::AcTap;
using rp::ac_tap;
using rp::flow_at;

// a5: ctx (B,C,H,W) -> net = tanh(first hdim ch), inp = relu(rest) at (h,w)
__global__ __launch_bounds__(256) void context_prep_kernel(const float* __restrict__ ctx, float* __restrict__ net,
                                                           float* __restrict__ inp, int B, int C, int H, int W, int h,
                                                           int w, int hdim) {
  const unsigned n = static_cast<unsigned>(B) * C * h * w;                 // < 2^31 (host check): 32-bit divisions
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const unsigned row = t / static_cast<unsigned>(w), plane = row / static_cast<unsigned>(h);
    const int X = static_cast<int>(t - row * w);
    const int Y = static_cast<int>(row - plane * h);
    const int b = static_cast<int>(plane / static_cast<unsigned>(C));
    const int c = static_cast<int>(plane - static_cast<unsigned>(b) * C);
    const AcTap ty = ac_tap(Y, H, h), tx = ac_tap(X, W, w);
    const float* p = ctx + (static_cast<long long>(b) * C + c) * H * W;
    const float v00 = p[static_cast<long long>(ty.i0) * W + tx.i0], v01 = p[static_cast<long long>(ty.i0) * W + tx.i1];
    const float v10 = p[static_cast<long long>(ty.i1) * W + tx.i0], v11 = p[static_cast<long long>(ty.i1) * W + tx.i1];
    const float top = v00 * (1.f - tx.f) + v01 * tx.f;
    const float bot = v10 * (1.f - tx.f) + v11 * tx.f;
    const float v = top * (1.f - ty.f) + bot * ty.f;
    const long long hw = static_cast<long long>(h) * w;
    if (c < hdim) {
      net[(static_cast<long long>(b) * hdim + c) * hw + Y * w + X] = tanhf(v);
    } else {
      inp[(static_cast<long long>(b) * (C - hdim) + (c - hdim)) * hw + Y * w + X] = fmaxf(v, 0.f);
    }
  }
}

// a5: coords1 = grid + resize(flow_init / ds)
__global__ __launch_bounds__(256) void flow_to_coords_kernel(const float* __restrict__ flow, float* __restrict__ coords1,
                                                             int B, int H, int W, int h, int w) {
  const long long n = static_cast<long long>(B) * 2 * h * w;
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int X = static_cast<int>(t % w);
  const int Y = static_cast<int>((t / w) % h);
  const int c = static_cast<int>((t / (static_cast<long long>(w) * h)) % 2);
  const int b = static_cast<int>(t / (static_cast<long long>(w) * h * 2));
  const float ds = static_cast<float>(W / w);
  const AcTap ty = ac_tap(Y, H, h), tx = ac_tap(X, W, w);
  const float* p = flow + (static_cast<long long>(b) * 2 + c) * H * W;
  const float v00 = p[static_cast<long long>(ty.i0) * W + tx.i0] / ds, v01 = p[static_cast<long long>(ty.i0) * W + tx.i1] / ds;
  const float v10 = p[static_cast<long long>(ty.i1) * W + tx.i0] / ds, v11 = p[static_cast<long long>(ty.i1) * W + tx.i1] / ds;
  const float top = v00 * (1.f - tx.f) + v01 * tx.f;
  const float bot = v10 * (1.f - tx.f) + v11 * tx.f;
  const float g = c == 0 ? static_cast<float>(X) : static_cast<float>(Y);
  coords1[t] = g + (top * (1.f - ty.f) + bot * ty.f);
}

// ------------------------------------------------------------------------------------------------
// a7: full-resolution induced flow (API parity with SE3Sequence.transform + PoseRefiner.py:327)
__global__ __launch_bounds__(256) void induced_flow_kernel(const float* __restrict__ depth, const float* __restrict__ K,
                                                           const float* __restrict__ G, float* __restrict__ flow,
                                                           float* __restrict__ vmask, int H, int W, float eps, int mode) {
  const int b = blockIdx.y;
  const long long P = static_cast<long long>(H) * W;
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= P) return;
  // (32-bit division: t < H*W < 2^31; a 64-bit t % W is a ~200-instruction software routine on this ISA)
  const unsigned tu = static_cast<unsigned>(t);
  const int y = static_cast<int>(tu / static_cast<unsigned>(W)), x = static_cast<int>(tu - static_cast<unsigned>(y) * static_cast<unsigned>(W));
  const Intr k = rp::load_intr(K, b);
  const Pose g = rp::load_pose(G, b);
  const float Z = depth[b * P + t] + eps;
  const rp::Reproj r = rp::reproject(Z, static_cast<float>(x), static_cast<float>(y), k, g);
  const float fg = Z > eps ? 1.f : 0.f;
  flow[(static_cast<long long>(b) * 2 + 0) * P + t] = mode == 0 ? (r.u - static_cast<float>(x)) * fg : r.u;
  flow[(static_cast<long long>(b) * 2 + 1) * P + t] = mode == 0 ? (r.v - static_cast<float>(y)) * fg : r.v;
  if (vmask) vmask[b * P + t] = (r.Z0 > rp::kMinDepthValid && r.Z1 > rp::kMinDepthValid) ? 1.f : 0.f;
}

// a7+a5 fused: only the 4 taps each 1/8-res pixel needs are re-projected (P/16 evaluations, no full-res pass)
__global__ __launch_bounds__(256) void induced_coords_lowres_kernel(const float* __restrict__ depth,
                                                                    const float* __restrict__ K,
                                                                    const float* __restrict__ G,
                                                                    float* __restrict__ coords1, int H, int W, int h,
                                                                    int w, float eps) {
  const int b = blockIdx.y;
  const int n = h * w;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int X = t % w, Y = t / w;
  const Intr k = rp::load_intr(K, b);
  const Pose g = rp::load_pose(G, b);
  const float2 c = rp::induced_coords_at(depth + static_cast<long long>(b) * H * W, X, Y, H, W, h, w, eps, k, g);       // (induced.cuh)
  coords1[(static_cast<long long>(b) * 2 + 0) * n + t] = c.x;
  coords1[(static_cast<long long>(b) * 2 + 1) * n + t] = c.y;
}

// ------------------------------------------------------------------------------------------------
// a6: convex upsampling, scale 8.  Workgroup = (b, Y, sub-row i, 64-wide X tile): the 72 mask channels
// (9 taps x 8 sub-columns) of that sub-row are read as coalesced 256-byte rows into LDS (transposed to
// [tap][X*9 + j], stride 9 -> conflict-free), then each thread produces output pixels of row 8Y+i in x
// order, so both flow channels leave as contiguous 2-KB row segments.
constexpr int UX = 64;
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ up, int h, int w) {
  __shared__ float ms[9 * (UX * 9 + 1)];
  __shared__ float fl[2 * 3 * (UX + 2)];
  const int X0 = blockIdx.x * UX;
  const int Y = blockIdx.y >> 3, si = blockIdx.y & 7;
  const int b = blockIdx.z;
  const int n = h * w;
  const int tid = threadIdx.x;
  const float* mb = mask + static_cast<long long>(b) * 576 * n + static_cast<long long>(Y) * w;
  // mask rows: channel = k*64 + si*8 + j
  for (int e = tid; e < 72 * UX; e += 256) {
    const int X = e & (UX - 1);
    const int cj = e >> 6;            // 0..71 = k*8 + j
    const int k = cj >> 3, j = cj & 7;
    float v = 0.f;
    if (X0 + X < w) v = mb[static_cast<long long>(k * 64 + si * 8 + j) * n + X0 + X];
    ms[k * (UX * 9 + 1) + X * 9 + j] = v;
  }
  // 3x3 flow neighbourhood rows Y-1..Y+1, cols X0-1..X0+UX, zero padded, pre-scaled by 8
  for (int e = tid; e < 2 * 3 * (UX + 2); e += 256) {
    const int xx = e % (UX + 2);
    const int ry = (e / (UX + 2)) % 3;
    const int c = e / (3 * (UX + 2));
    const int yy = Y + ry - 1, x = X0 + xx - 1;
    float v = 0.f;
    if (yy >= 0 && yy < h && x >= 0 && x < w) v = 8.f * flow[(static_cast<long long>(b) * 2 + c) * n + yy * w + x];
    fl[e] = v;
  }
  __syncthreads();
  const int Wf = 8 * w;
  const long long Pf = static_cast<long long>(8 * h) * Wf;
  const int row = 8 * Y + si;
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int px = tid + rep * 256;           // 0..511 within the tile's row segment
    const int X = px >> 3, j = px & 7;
    if (X0 + X >= w) continue;
    float m[9];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      m[k] = ms[k * (UX * 9 + 1) + X * 9 + j];
      mx = fmaxf(mx, m[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      m[k] = expf(m[k] - mx);
      den += m[k];
    }
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int ky = k / 3, kx = k % 3;
      const float wk = m[k] / den;                                  // softmax weight, as torch.softmax
      ax += wk * fl[(0 * 3 + ky) * (UX + 2) + X + kx];
      ay += wk * fl[(1 * 3 + ky) * (UX + 2) + X + kx];
    }
    const long long o = static_cast<long long>(row) * Wf + 8 * X0 + px;
    up[(static_cast<long long>(b) * 2 + 0) * Pf + o] = ax;
    up[(static_cast<long long>(b) * 2 + 1) * Pf + o] = ay;
  }
}

// ------------------------------------------------------------------------------------------------
// a8: reliability weight.  One thread per pixel; g1 rows are coalesced, the four g2 taps of neighbouring
// lanes are neighbouring addresses because the flow is smooth (coalesced in practice).
#ifndef RP_CW_ACQUIRE
#define RP_CW_ACQUIRE 0
#endif
#ifndef RP_CW_DEBUG
#define RP_CW_DEBUG 0
#endif
#ifndef RP_CW_SC1
#define RP_CW_SC1 0
#endif
#ifndef RP_CW_BATCH
#define RP_CW_BATCH 4
#endif
constexpr int CW_BATCH = RP_CW_BATCH;   // channels whose 5 (3 with tap pairs) loads each are in flight together
#ifndef RP_CW_PAIRS
#define RP_CW_PAIRS 0     // default of rnnpose_corr_weight_pairs: 1 = the two taps of a row as one 8-byte load (r06: bit-identical, 56.6 vs 58.6 us alone, but -0.8 % on the step: profiles/r06_corr_weight_pairs.txt)
#endif
template <bool PAIRS>     // PAIRS (r06, W >= 2): the two taps of a row as one 8-byte load (descriptor_weight.cuh)
__global__ __launch_bounds__(256) void corr_weight_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                          const float* __restrict__ target, int target_mode,
                                                          const float* __restrict__ depth,
                                                          const float* __restrict__ sigma, float* __restrict__ weight,
                                                          int D, int H, int W) {
  // XCD-contiguous block order: the workgroups an XCD runs at the same time cover neighbouring image rows, so the
  // second descriptor row of every bilinear tap pair (the first row of the pixels one line below) is an L2 hit
  // instead of a second trip to HBM (r01 PMC: 1.16 GB fetched per launch for 0.66 GB of unique bytes before this).
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int per = nblk >> 3, rem = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = xcd * per + (xcd < rem ? xcd : rem) + idx;
  }
#if RP_CW_ACQUIRE   // diagnostics build (tools/det_variants.sh): agent-scope acquire (buffer_inv sc1) before the first load
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
#endif
  const long long P = static_cast<long long>(H) * W;
  const int bpi = static_cast<int>((P + 255) / 256);       // blocks per image
  const int b = bid / bpi;
  const long long t = static_cast<long long>(bid - b * bpi) * 256 + threadIdx.x;
  if (t >= P) return;
  // (32-bit division: t < H*W < 2^31; a 64-bit t % W is a ~200-instruction software routine on this ISA)
  const unsigned tu = static_cast<unsigned>(t);
  const int y = static_cast<int>(tu / static_cast<unsigned>(W)), x = static_cast<int>(tu - static_cast<unsigned>(y) * static_cast<unsigned>(W));
  // background pixels (rendered depth <= 0) have weight exp(..) * 0 = 0 whatever the descriptors say.  A WAVE that is all
  // background (the band around the object in a zoomed crop) leaves after one depth load instead of 5 * D descriptor
  // loads.  Wave-uniform exit only: a per-lane `return` put the channel loop under a divergent branch and the compiler
  // replaced its counted waits by vmcnt(0) -- 104 -> 259 us per launch (r02, same-box A/B).  The only observable
  // difference to the reference expression is a NaN descriptor inside an all-background wave (NaN * 0 = NaN there, 0 here).
  const float fg = depth[b * P + t] > 0.f ? 1.f : 0.f;
  if (__builtin_amdgcn_ballot_w64(fg != 0.f) == 0ull) {
    weight[b * P + t] = 0.f;
    return;
  }
  float tx, ty;
  if (target_mode == 0) {
    const float2 tt = *reinterpret_cast<const float2*>(target + (b * P + t) * 2);
    tx = tt.x;
    ty = tt.y;
  } else {
#if RP_CW_SC1       // diagnostics build: the flow map through sc1 loads (served by L2 / memory, never by this CU's L1)
    tx = __hip_atomic_load(target + (static_cast<long long>(b) * 2 + 0) * P + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + static_cast<float>(x);
    ty = __hip_atomic_load(target + (static_cast<long long>(b) * 2 + 1) * P + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + static_cast<float>(y);
#else
    tx = target[(static_cast<long long>(b) * 2 + 0) * P + t] + static_cast<float>(x);
    ty = target[(static_cast<long long>(b) * 2 + 1) * P + t] + static_cast<float>(y);
#endif
  }
#if RP_CW_DEBUG == 3   // diagnostics: the descriptor path alone (the target is the pixel itself, the flow map is read and ignored)
  tx = static_cast<float>(x) + 0.f * tx; ty = static_cast<float>(y) + 0.f * ty;
#endif
  float s;
  if constexpr (PAIRS) {
    const rp::DescPairs pairs = rp::descriptor_pairs(tx, ty, H, W);
    s = rp::descriptor_dot_pairs<CW_BATCH>(g1 + static_cast<long long>(b) * D * P + t, g2 + static_cast<long long>(b) * D * P, P, D, pairs);
  } else {
    const rp::DescTaps taps = rp::descriptor_taps(tx, ty, H, W);
    s = rp::descriptor_dot<CW_BATCH>(g1 + static_cast<long long>(b) * D * P + t, g2 + static_cast<long long>(b) * D * P, P, D, taps);
  }
#if RP_CW_DEBUG == 1   // diagnostics: what this thread read from the flow map (x / y plane)
  weight[b * P + t] = tx + 0.f * s;
#elif RP_CW_DEBUG == 2
  weight[b * P + t] = ty + 0.f * s;
#else
  weight[b * P + t] = expf(-fabsf(1.f - s) / sigma[0]) * fg;
#endif
}

// ------------------------------------------------------------------------------------------------
// a4: SepConvGRU pointwise stages (update.py:47-52).  hw = h*w; all tensors (B,Cx,hw) channel-major.
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void gru_gate_kernel(const float* __restrict__ zr, const float* __restrict__ hcat,
                                                       float* __restrict__ z_out, float* __restrict__ rhx, int C,
                                                       int Ctot, int hw, long long n) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long chw = static_cast<long long>(C) * hw;
  const long long b = t / chw, r = t - b * chw;          // r = c*hw + p
  const float zp = zr[b * 2 * chw + r];
  const float rp_ = zr[b * 2 * chw + chw + r];
  const float hv = hcat[b * Ctot * hw + r];
  z_out[t] = sigmoidf_(zp);
  rhx[b * Ctot * hw + r] = sigmoidf_(rp_) * hv;
}

__global__ __launch_bounds__(256) void gru_update_kernel(const float* __restrict__ z, const float* __restrict__ q_pre,
                                                         const float* __restrict__ hcat, float* __restrict__ hout,
                                                         int C, int Ctot_in, int Ctot_out, int hw, long long n) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long chw = static_cast<long long>(C) * hw;
  const long long b = t / chw, r = t - b * chw;
  const float zv = z[t];
  const float hv = hcat[b * Ctot_in * hw + r];
  hout[b * Ctot_out * hw + r] = (1.f - zv) * hv + zv * tanhf(q_pre[t]);
}

}  // namespace

extern "C" {

int rnnpose_context_prep_f32(const float* ctx, int B, int C, int H, int W, int h, int w, int hdim, float* net, float* inp,
                             rnnpose_stream_t stream) {
  const char* fn = "rnnpose_context_prep_f32";
  RP_REQUIRE(ctx && net && inp, fn, "null pointer");
  RP_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0, fn, "non-positive size");
  RP_REQUIRE(hdim > 0 && hdim < C, fn, "hdim must be in (0,C)");
  RP_REQUIRE(static_cast<long long>(B) * C * h * w < (1LL << 31), fn, "output too large for 32-bit element indices");
  const long long n = static_cast<long long>(B) * C * h * w;
  const int blocks = static_cast<int>(n / 256 + 1 < 65536 ? n / 256 + 1 : 65536);
  hipLaunchKernelGGL(context_prep_kernel, dim3(blocks), dim3(256), 0, rp::as_stream(stream), ctx, net, inp, B, C, H, W, h,
                     w, hdim);
  return rp::check_launch(fn);
}

int rnnpose_flow_to_coords_f32(const float* flow_init, int B, int H, int W, int h, int w, float* coords1,
                               rnnpose_stream_t stream) {
  const char* fn = "rnnpose_flow_to_coords_f32";
  RP_REQUIRE(flow_init && coords1, fn, "null pointer");
  RP_REQUIRE(B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && W >= w, fn, "bad size");
  const long long n = static_cast<long long>(B) * 2 * h * w;
  hipLaunchKernelGGL(flow_to_coords_kernel, dim3(rp::cdiv(n, 256)), dim3(256), 0, rp::as_stream(stream), flow_init,
                     coords1, B, H, W, h, w);
  return rp::check_launch(fn);
}

int rnnpose_induced_flow_f32(const float* depth, const float* K, const float* G, int B, int H, int W, float depth_eps,
                             int mode, float* flow, float* vmask, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_induced_flow_f32";
  RP_REQUIRE(depth && K && G && flow, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && static_cast<long long>(H) * W < (1LL << 31), fn, "bad size");
  RP_REQUIRE(mode == 0 || mode == 1, fn, "mode must be 0 or 1");
  const long long P = static_cast<long long>(H) * W;
  hipLaunchKernelGGL(induced_flow_kernel, dim3(rp::cdiv(P, 256), B), dim3(256), 0, rp::as_stream(stream), depth, K, G,
                     flow, vmask, H, W, depth_eps, mode);
  return rp::check_launch(fn);
}

int rnnpose_induced_coords_lowres_f32(const float* depth, const float* K, const float* G, int B, int H, int W, int h,
                                      int w, float depth_eps, float* coords1, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_induced_coords_lowres_f32";
  RP_REQUIRE(depth && K && G && coords1, fn, "null pointer");
  RP_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && h > 0 && w > 0 && W >= w, fn, "bad size");
  hipLaunchKernelGGL(induced_coords_lowres_kernel, dim3(rp::cdiv(static_cast<long long>(h) * w, 256), B), dim3(256), 0,
                     rp::as_stream(stream), depth, K, G, coords1, H, W, h, w, depth_eps);
  return rp::check_launch(fn);
}

int rnnpose_convex_upsample_f32(const float* flow, const float* mask, int B, int h, int w, int scale, float* flow_up,
                                rnnpose_stream_t stream) {
  const char* fn = "rnnpose_convex_upsample_f32";
  RP_REQUIRE(flow && mask && flow_up, fn, "null pointer");
  RP_REQUIRE(scale == 8, fn, "scale must be 8");
  RP_REQUIRE(B > 0 && B < 65536 && h > 0 && w > 0 && h * 8 < 65536, fn, "bad size");
  hipLaunchKernelGGL(convex_upsample_kernel, dim3(rp::cdiv(w, UX), h * 8, B), dim3(256), 0, rp::as_stream(stream), flow,
                     mask, flow_up, h, w);
  return rp::check_launch(fn);
}

static int g_cw_pairs = RP_CW_PAIRS;
int rnnpose_corr_weight_pairs(int enable) {      // measurement / test switch: 1 (default) the two taps of a row as one 8-byte load, 0 four 4-byte tap loads
  g_cw_pairs = enable ? 1 : 0;
  return 0;
}

int rnnpose_corr_weight_f32(const float* g1, const float* g2, const float* target, int target_mode, const float* depth,
                            const float* sigma, int B, int D, int H, int W, float* weight, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_corr_weight_f32";
  RP_REQUIRE(g1 && g2 && target && depth && sigma && weight, fn, "null pointer");
  RP_REQUIRE(target_mode == 0 || target_mode == 1, fn, "target_mode must be 0 or 1");
  RP_REQUIRE(B > 0 && B < 65536 && D > 0 && H > 1 && W > 1 && static_cast<long long>(H) * W < (1LL << 31), fn, "bad size");
  const long long P = static_cast<long long>(H) * W;
  RP_REQUIRE(static_cast<long long>(rp::cdiv(P, 256)) * B < (1LL << 31), fn, "grid too large");
  if (g_cw_pairs && W >= 2)
    hipLaunchKernelGGL(corr_weight_kernel<true>, dim3(static_cast<unsigned>(rp::cdiv(P, 256) * B)), dim3(256), 0, rp::as_stream(stream), g1, g2,
                       target, target_mode, depth, sigma, weight, D, H, W);
  else
    hipLaunchKernelGGL(corr_weight_kernel<false>, dim3(static_cast<unsigned>(rp::cdiv(P, 256) * B)), dim3(256), 0, rp::as_stream(stream), g1, g2,
                       target, target_mode, depth, sigma, weight, D, H, W);
  return rp::check_launch(fn);
}

int rnnpose_gru_gate_f32(const float* zr, const float* hcat, int B, int C, int Ctot, int hw, float* z_out, float* rhx,
                         rnnpose_stream_t stream) {
  const char* fn = "rnnpose_gru_gate_f32";
  RP_REQUIRE(zr && hcat && z_out && rhx, fn, "null pointer");
  RP_REQUIRE(B > 0 && C > 0 && Ctot >= C && hw > 0, fn, "bad size");
  const long long n = static_cast<long long>(B) * C * hw;
  hipLaunchKernelGGL(gru_gate_kernel, dim3(rp::cdiv(n, 256)), dim3(256), 0, rp::as_stream(stream), zr, hcat, z_out, rhx,
                     C, Ctot, hw, n);
  return rp::check_launch(fn);
}

int rnnpose_gru_update_f32(const float* z, const float* q_pre, const float* hcat, int B, int C, int Ctot_in, int hw,
                           float* hout, int Ctot_out, rnnpose_stream_t stream) {
  const char* fn = "rnnpose_gru_update_f32";
  RP_REQUIRE(z && q_pre && hcat && hout, fn, "null pointer");
  RP_REQUIRE(B > 0 && C > 0 && Ctot_in >= C && Ctot_out >= C && hw > 0, fn, "bad size");
  const long long n = static_cast<long long>(B) * C * hw;
  hipLaunchKernelGGL(gru_update_kernel, dim3(rp::cdiv(n, 256)), dim3(256), 0, rp::as_stream(stream), z, q_pre, hcat,
                     hout, C, Ctot_in, Ctot_out, hw, n);
  return rp::check_launch(fn);
}

}  // extern "C"
