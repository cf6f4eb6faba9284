/*
 * rnnpose_hip.h -- C ABI of librnnpose_hip.so: the MI355X (gfx950) implementation of RNNPose's
 * recurrent-refinement hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_ (host);
 *   - tensors are dense, row-major ("contiguous") in the shape given in the comment;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only ENQUEUE work:
 *     no host synchronisation, no device allocation, no D2H copy happens inside the library;
 *   - return value 0 = success; nonzero = error, with a message in rnnpose_last_error()
 *     (1 = invalid argument, 2 = HIP launch/runtime error).  The library never calls exit()
 *     (the reference's only native precedent does: thirdparty/nn/src/nearest_neighborhood.cu:10-17).
 *   - inference only (tools/eval.py:526 runs under no_grad): no backward entry points.
 *
 * Each entry point names the reference code it replaces (paths relative to the RNNPose repo).
 */
#ifndef RNNPOSE_HIP_H
#define RNNPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rnnpose_stream_t;

/* 3 (round 5): rnnpose_conv_desc_t ends with tile_stats_records (the launch checks the statistics buffer it is given);
 * rnnpose_conv_tiles_per_image_desc sizes it from the descriptor itself.
 * 2 (round 4): rnnpose_conv_desc_t grew the split-tensor / K-split members, the packed weight array holds two orders
 * (rnnpose_conv_packed_halfs doubled), tile statistics are fp64, rnnpose_lm_step_* needs a zero-filled workspace: a consumer
 * built against version 1 must not pass the check. */
#define RNNPOSE_ABI_VERSION 3
#define RNNPOSE_MAX_LEVELS 4

int rnnpose_abi_version(void);
const char* rnnpose_last_error(void);
/* gfx arch name of device `dev` copied to h_name (e.g. "gfx950"), CU count to *h_cus */
int rnnpose_device_info(int dev, char* h_name, int name_len, int* h_cus);

/* ---- a1+a2: all-pairs correlation volume + pyramid ------------- thirdparty/raft/corr.py:13-34,59-67
 * level 0: corr[b,i,j] = sum_c fmap1[b,c,i]*fmap2[b,c,j] / sqrt(C); level l = 2x2 mean of level l-1
 * (floor cropping).  `pyramid` is ONE buffer holding all levels back to back; sizes/offsets (in floats) come from
 * rnnpose_corr_pyramid_layout.  Levels 1.. are laid out as (B*h*w, h_l, w_l).  LEVEL 0 (r04) is stored J-PATCH-MAJOR:
 * [b][patch py * ceil(w/16) + px][i][8][16], i.e. corr[b,i,(y,x)] sits at
 *     ((b * n_patch + (y >> 3) * ceil(w/16) + (x >> 4)) * h*w + i) * 128 + (y & 7) * 16 + (x & 15),   n_patch = ceil(h/8) ceil(w/16)
 * (whole 8 x 16 patches: cells outside the image hold zeros).  The volume kernel's 128 (i) x one-patch tile then leaves the
 * chip as one contiguous 64-KB run and the lookup's 10 x 10 window reads a few contiguous patch rows instead of ten rows
 * 4*w bytes apart.  Only the lookup entry points read it; rnnpose_amd.ops.pyramid_level0_dense un-blocks it for inspection.
 * fmap1,fmap2: (B,C,h,w) fp32, C % 4 == 0.                                                       */
int rnnpose_corr_pyramid_layout(int B, int h, int w, int levels, int64_t* h_offsets /*[levels+1]*/,
                                int* h_hl /*[levels]*/, int* h_wl /*[levels]*/);
int rnnpose_corr_pyramid_f32(const float* fmap1, const float* fmap2, int B, int C, int h, int w,
                             int levels, float* pyramid, rnnpose_stream_t stream);
/* Same volume + pyramid with the operands split into fp16 hi/lo halves and multiplied on the fp16 matrix cores with
 * fp32 accumulation (3 MFMAs per k-slab; the dropped lo*lo term is 2^-22 relative: fp32-class accuracy).  One pre-pass
 * launch writes both operands, scaled and split, as pixel-major SPLIT TENSORS (see rnnpose_conv_desc_t) into `workspace`.
 * layout 0: fmaps are (B,C,h,w) as above; layout 1: (B,h,w,C).  a_scale (a power of two, 8 in the host code) scales both
 * operands before the split; |x| * a_scale beyond 65504 saturates (range guard).  C % 32 == 0.
 * rnnpose_corr_pyramid_split: the operands ARE split tensors already ((B,h,w,C), 4*C bytes per pixel, scale a_scale), e.g.
 * written by the encoder's output convolution (rnnpose_conv1x1_resident_f16x3, dst_split): no pre-pass, no workspace. */
size_t rnnpose_corr_pyramid_f16x3_workspace_bytes(int B, int C, int h, int w);
int rnnpose_corr_pyramid_f16x3(const float* fmap1, const float* fmap2, int layout, int B, int C, int h, int w, int levels,
                               float a_scale, void* workspace, size_t workspace_bytes, float* pyramid,
                               rnnpose_stream_t stream);
int rnnpose_corr_pyramid_split(const void* fmap1_split, const void* fmap2_split, int B, int C, int h, int w, int levels,
                               float a_scale, float* pyramid, rnnpose_stream_t stream);
int rnnpose_corr_supertile(int max_side);   /* measurement switch: tile-order supertiles of at most max_side x max_side (default 20) */
int rnnpose_corr_variant(int variant);      /* which fp16x3 volume kernel rnnpose_corr_pyramid_f16x3 / _split launch: 0 = operands through registers
                                             * and ds_write (r03-r05), 1 = operands by LDS-DMA, double-buffered 16-channel slabs (r06); bit-identical results */

/* ---- a3': volume-free lookup ---- thirdparty/raft/corr.py:70-98 (AlternateCorrBlock; its alt_cuda_corr extension is not in the
 * reference tree and the reference never enables it, model/CFNet.py:63-64).  The same (levels*81)-channel window features as
 * rnnpose_corr_lookup_*, computed on the fly from fmap1 and the 2^l-pooled fmap2 (pooling is linear: equal to the pyramid
 * lookup up to fp32 summation order) -- no volume is built.  A measured alternative, not the graded path.
 * fmap1 / fmap2: (B,h,w,C) pixel-major fp32, C % 4 == 0, C <= 512, 16-byte aligned.  rnnpose_fmap_pyramid_f32 writes levels
 * 1..levels-1 of fmap2 (chained 2x2 means, floor cropping) back to back into `pooled` (rnnpose_fmap_pyramid_floats floats).
 * out: pixel-major, channels [out_c_offset, +levels*81) of rows of out_c_stride floats; coords (B,2,h,w) as for the lookup. */
size_t rnnpose_fmap_pyramid_floats(int B, int h, int w, int C, int levels);
int rnnpose_fmap_pyramid_f32(const float* fmap2_nhwc, int B, int h, int w, int C, int levels, float* pooled,
                             rnnpose_stream_t stream);
int rnnpose_corr_alt_lookup_f32(const float* fmap1_nhwc, const float* fmap2_nhwc, const float* pooled, const float* coords, int B,
                                int h, int w, int C, int levels, int radius, float* out, int out_c_stride, int out_c_offset,
                                rnnpose_stream_t stream);

/* ---- a3: pyramid lookup -------- thirdparty/raft/corr.py:36-57, thirdparty/raft/utils/utils.py:57-71
 * coords (B,2,h,w) (ch0 = x, ch1 = y) -> out (B, levels*(2r+1)^2, h, w); channel
 * l*(2r+1)^2 + i*(2r+1) + j samples level l at (x/2^l + i - r, y/2^l + j - r)  [x-major window],
 * bilinear, align_corners=True, zero padding.  radius must be 4 (the only value the reference uses). */
int rnnpose_corr_lookup_f32(const float* pyramid, const float* coords, int B, int h, int w, int levels,
                            int radius, float* out, rnnpose_stream_t stream);
int rnnpose_corr_lookup_variant(int variant);   /* r06 measurement / test switch: 1 (default) = the lookup kernel at half the instructions per wave (one address form per level
                                                 * class, scalar pixel bases, phase 2 on 48 lanes), 0 = the r01-r05 kernel.  Bit-identical results. */

/* ---- a5: GRU_CFUpdator glue -------------------------------------------- model/CFNet.py:124-144
 * context_prep: ctx (B,C,H,W) -> bilinear (align_corners=True) resize to (h,w); first `hdim` channels
 * -> net = tanh(.), remaining C-hdim -> inp = relu(.).
 * flow_to_coords: coords1 (B,2,h,w) = pixel grid + resize(flow_init / (W/w)), flow_init (B,2,H,W);
 * does NOT modify flow_init (the reference divides in place, CFNet.py:141).                        */
int rnnpose_context_prep_f32(const float* ctx, int B, int C, int H, int W, int h, int w, int hdim,
                             float* net, float* inp, rnnpose_stream_t stream);
int rnnpose_flow_to_coords_f32(const float* flow_init, int B, int H, int W, int h, int w, float* coords1,
                               rnnpose_stream_t stream);

/* ---- a6: convex upsampling --------------------------------------------- model/CFNet.py:95-106
 * flow (B,2,h,w), mask (B,9*s*s,h,w) -> flow_up (B,2,s*h,s*w); s must be 8.                       */
int rnnpose_convex_upsample_f32(const float* flow, const float* mask, int B, int h, int w, int scale,
                                float* flow_up, rnnpose_stream_t stream);

/* ---- a7: induced flow of the relative pose --- geometry/transformation.py:184-198,
 *      geometry/projective_ops.py:68-114, model/PoseRefiner.py:313,324-328
 * depth (B,1,H,W) RAW rendered depth (0 = background; the kernel adds depth_eps as PoseRefiner.py:313
 * does), K (B,3,3), G (B,4,4) -> flow (B,2,H,W) = (project(G*backproject) - grid) * [depth+eps > eps],
 * vmask (B,H,W) float = [Z>0.1][Z'>0.1] (may be NULL).  mode 0: masked flow as above (PoseRefiner.py:327);
 * mode 1: the raw re-projected coordinates (u,v) of SE3.transform, unmasked (transformation.py:184-198).
 * induced_coords_lowres fuses a7 with flow_to_coords and evaluates only the 4 taps each 1/8-res pixel
 * needs: coords1 (B,2,h,w).                                                                        */
int rnnpose_induced_flow_f32(const float* depth, const float* K, const float* G, int B, int H, int W,
                             float depth_eps, int mode, float* flow, float* vmask, rnnpose_stream_t stream);
int rnnpose_induced_coords_lowres_f32(const float* depth, const float* K, const float* G, int B, int H,
                                      int W, int h, int w, float depth_eps, float* coords1,
                                      rnnpose_stream_t stream);

/* ---- a8: descriptor reliability weight ---- model/PoseRefiner.py:342-345, projective_ops.py:11-23
 * g1,g2 (B,D,H,W); weight (B,H,W) = exp(-|1 - <g1, bilinear(g2, target)>| / sigma) * [depth > 0];
 * g2 is sampled at pixel ((2x/(W-1)-1+1)*W-1)/2 (the reference's align_corners mismatch), zero pad.
 * target_mode 0: `target` is (B,H,W,2) absolute pixel coordinates (x,y interleaved);
 * target_mode 1: `target` is a planar flow (B,2,H,W) and the pixel grid is added in-kernel.
 * sigma: device pointer to one float (the nn.Parameter PoseRefiner.sigma[0]).                      */
int rnnpose_corr_weight_f32(const float* g1, const float* g2, const float* target, int target_mode,
                            const float* depth, const float* sigma, int B, int D, int H, int W,
                            float* weight, rnnpose_stream_t stream);
int rnnpose_corr_weight_pairs(int enable);    /* r06 measurement / test switch: 1 (default) = the two taps of an image row arrive as ONE 8-byte load (3 load
                                               * instructions per channel and pixel instead of 5; same arithmetic, bit-identical weights for finite descriptors), 0 = four 4-byte tap loads */

/* ---- a9: Gauss-Newton normal equations (fp64) -- geometry/transformation.py:274-297,
 *      geometry/projective_ops.py:116-124, geometry/transformation.py:27-46
 * Hm (B,6,6) = sum v*w*J^T J, bv (B,6) = sum v*w*J^T (target - x'), UNDAMPED, fp64.
 * weight (B,H,W) fp32; target/target_mode as above; depth raw (+depth_eps inside).
 * workspace: at least rnnpose_lm_workspace_bytes(B,H,W) bytes of device memory ([B arrival counters][block partials]). */
size_t rnnpose_lm_workspace_bytes(int B, int H, int W);
int rnnpose_lm_normal_eq_f64(const float* target, int target_mode, const float* weight, const float* depth,
                             float depth_eps, const float* K, const float* G, int B, int H, int W,
                             void* workspace, size_t workspace_bytes, double* Hm, double* bv,
                             rnnpose_stream_t stream);

/* ---- a10+a11: damped 6x6 Cholesky solve + SE(3) exp + left increment -- transformation.py:300-306,
 *      geometry/cholesky.py:32-50, geometry/se3.py:228-281,303-306
 * Hd = Hm + ep_lambda*I + lm_lambda*diag(Hm); xi = float(clamp(nan_to_zero(Hd^-1 bv), +-max_update));
 * G_new = exp(xi) * G.  info[b] = 0, or k>0 if the leading minor of order k is not positive (the
 * reference's torch.cholesky raises there; here xi becomes NaN->0 and the flag is left for the host). */
int rnnpose_lm_solve_update_f32(const double* Hm, const double* bv, const float* G, int B, double ep_lambda,
                                double lm_lambda, double max_update, float* G_new, float* xi, int* info,
                                rnnpose_stream_t stream);

/* fused a9+a10+a11: `num_iters` GN steps (reprojction_optim), G updated in place (B,4,4).
 * Hm/bv/xi/info receive the LAST iteration's values.                                               */
int rnnpose_lm_step_f32(const float* target, int target_mode, const float* weight, const float* depth,
                        float depth_eps, const float* K, float* G, int B, int H, int W, int num_iters,
                        double ep_lambda, double lm_lambda, double max_update, void* workspace,
                        size_t workspace_bytes, double* Hm, double* bv, float* xi, int* info,
                        rnnpose_stream_t stream);
/* same with separate input and output poses (num_iters >= 1; G_out may alias G_in): saves the caller a device copy */
int rnnpose_lm_step_io_f32(const float* target, int target_mode, const float* weight, const float* depth,
                           float depth_eps, const float* K, const float* G_in, float* G_out, int B, int H, int W,
                           int num_iters, double ep_lambda, double lm_lambda, double max_update, void* workspace,
                           size_t workspace_bytes, double* Hm, double* bv, float* xi, int* info,
                           rnnpose_stream_t stream);
/* The fused steps run ONE launch per Gauss-Newton iteration: the workgroup that arrives last for an image (device-scope ticket
 * in the workspace, which must therefore be ZERO-FILLED when it is allocated) sums the partial records, solves and updates the
 * pose.  rnnpose_lm_fused_tail(0) restores the three-launch form (normal equations, finalize, solve) for measurements. */
int rnnpose_lm_fused_tail(int enable);

/* ---- a11/a12 helpers: batched SE(3) ------------------------- geometry/se3.py:194-209,228-306
 * se3_exp: xi (B,6) -> (B,4,4);  se3_compose: out = A*Bm (B,4,4);  se3_inverse: out = A^-1.        */
int rnnpose_se3_exp_f32(const float* xi, int B, float* out, rnnpose_stream_t stream);
int rnnpose_se3_compose_f32(const float* A, const float* Bm, int B, float* out, rnnpose_stream_t stream);
int rnnpose_se3_inverse_f32(const float* A, int B, float* out, rnnpose_stream_t stream);
/* r06: the pose bookkeeping between two outer iterations as one launch (model/PoseRefiner.py:241-244): Ti_out = Tij Ti; Tij_out = Ti_out Ti_out^-1 (literal != 0:
 * the reference's legacy product) or the exact identity (literal == 0).  Bit-identical to rnnpose_se3_compose_f32 / _inverse_f32 / _compose_f32 in that order.
 * Tij, Ti, Ti_out, Tij_out: (B,4,4) fp32 row-major; the outputs must not alias the inputs. */
int rnnpose_se3_outer_update_f32(const float* Tij, const float* Ti, int B, int literal, float* Ti_out, float* Tij_out, rnnpose_stream_t stream);

/* ---- a4: SepConvGRU pointwise stages ---------------------------- thirdparty/raft/update.py:45-60
 * zr (B,2C,h,w) = pre-activation outputs of the fused z|r convolution, hcat (B,Ctot,h,w) whose first
 * C channels are h: writes rh = sigmoid(r)*h into rhx[:, :C] (rhx (B,Ctot,h,w), channels >= C untouched)
 * and z = sigmoid(z_pre) into z_out (B,C,h,w).
 * gru_update: h_new = (1-z)*h + z*tanh(q_pre), written to hout (a (B,Ctot_out,h,w) buffer, first C ch). */
int rnnpose_gru_gate_f32(const float* zr, const float* hcat, int B, int C, int Ctot, int hw, float* z_out,
                         float* rhx, rnnpose_stream_t stream);
int rnnpose_gru_update_f32(const float* z, const float* q_pre, const float* hcat, int B, int C, int Ctot_in,
                           int hw, float* hout, int Ctot_out, rnnpose_stream_t stream);

/* ---- a4: dense convolutions of the update block, NHWC, fp16x3-split MFMA (fp32-class accuracy) ----------
 *      thirdparty/raft/update.py:6-14 (FlowHead), :33-60 (SepConvGRU), :79-97 (BasicMotionEncoder), :172-187
 * Zero padding kh/2 x kw/2, odd kernel sizes up to 7 (1x1, 3x3, 1x5, 5x1 are what the update block uses), stride 1 or 2.
 * Input = virtual concat of up to 4 NHWC tensors (B,H,W,c_stride), using channels [c_offset, c_offset+c_count).
 * Weights are packed once by rnnpose_conv_pack_weights_f16x3 from the PyTorch layout (c_out, c_in, kh, kw),
 * with the SAME segment channel counts; w_scale (a power of two) must be the one given to the packer.
 * Output y = conv(x) + bias goes through `epilogue`:
 *   0 linear / 1 ReLU            -> dst[pixel, dst_c_offset + n]
 *   2 GRU z|r (c_out = 2*gru_c)  -> n <  gru_c: dst  = sigmoid(y)                     (z)
 *                                   n >= gru_c: dst2 = sigmoid(y) * aux0[pixel, n-gru_c]   (r*h, aux0 = h)
 *   3 GRU state update           -> dst = (1 - aux1) * aux0 + aux1 * tanh(y)          (aux1 = z, aux0 = h)
 * All destination / aux tensors are NHWC with their own channel stride and offset (so results land directly
 * in a slice of a wider tensor).  dst must not alias a source of the same launch.                      */
typedef struct {
  const float* ptr;
  int c_stride, c_offset, c_count;
} rnnpose_conv_src_t;

typedef struct {
  rnnpose_conv_src_t src[4];
  int n_src;
  int B, H, W;                 /* INPUT spatial size */
  int kh, kw;
  int stride;                  /* 1, or 2 (output = ceil(H/2) x ceil(W/2), padding kh/2, kw/2: the encoder's strided convs) */
  const void* w_packed;        /* from rnnpose_conv_pack_weights_f16x3 (fp16 hi and lo parts, MFMA-fragment order) */
  const float* bias;
  int c_out;
  float a_scale, w_scale;
  int epilogue;
  float* dst;
  int dst_c_stride, dst_c_offset;
  const float* aux0;
  int aux0_c_stride, aux0_c_offset;
  const float* aux1;
  int aux1_c_stride, aux1_c_offset;
  float* dst2;
  int dst2_c_stride, dst2_c_offset;
  int gru_c;
  double* tile_stats;          /* optional (NULL = off), linear epilogue only: (B * tiles_per_image, c_out, 2) FP64 receives, per
                                  128-row output tile, the column sums and sums of squares of the outputs (bias included)
                                  -- the instance-norm statistics pass of the encoder without re-reading the tensor
                                  (rnnpose_instnorm_tiles_nhwc_f32).  With tile_stats (or src0_mean_rstd) the output rows
                                  are tiled PER IMAGE: tiles_per_image = rnnpose_conv_tiles_per_image(...), the last tile of an
                                  image ragged, so no tile straddles two images. */
  const float* add_map;        /* optional (NULL = off): NHWC tensor added to y before the epilogue, y += add_map[pixel,
                                  add_c_offset + n] -- a per-pixel bias.  Used to hoist the part of a convolution whose input
                                  does not change between calls (the context half `inp` of the GRU input, constant over the
                                  inner iterations: update.py:181 concatenates it anew every step) out of the loop: conv is
                                  linear, conv([h|inp|m]) = conv([h|m]) + conv(inp).  Needs c_out % 4 == 0, 16-byte alignment. */
  int add_c_stride, add_c_offset;
  const float* src0_mean_rstd; /* optional (NULL = off): (B, src[0].c_stride, 2) mean / rstd per image and channel of source 0:
                                  the convolution reads relu((x - mean) * rstd) instead of x -- the instance norm + ReLU
                                  between conv1 and conv2 of a ResidualBlock (extractor.py:48-52) applied in the load, so
                                  the normalised tensor is never written.  One source, 3x3, stride 1 only. */
  /* ---- SPLIT TENSORS (round 3): activations that only feed further convolutions live in HBM already split for the
   * fp16 matrix cores, written once by their producer's epilogue instead of being re-split by every consumer tile (a 3x3
   * layer with 256 outputs re-split each input 12 times).  A split tensor has the shape, strides and SIZE of its fp32
   * NHWC counterpart (4 bytes per element, channel counts / offsets in multiples of 8); every 8-channel group of a pixel
   * occupies 32 bytes = [hi x 8 | lo x 8] fp16 with  x * a_scale = hi + lo  (hi = fp16(x * a_scale) rounded to
   * nearest, lo = fp16(x * a_scale - hi); |x * a_scale| > 65504 is clamped and counted by the range guard).  a_scale must be
   * the same for the producer and every consumer of a tensor (the package uses 8). */
  int src_hl;                  /* != 0: ALL sources are split tensors (stride 1, 1/3/5 taps per group, no src0_mean_rstd) */
  int dst_hl;                  /* != 0: dst receives the split form (no tile_stats) */
  int dst2_hl;                 /* != 0: dst2 (r*h of the GRU z|r epilogue) receives the split form */
  float* dst_split;            /* optional (NULL = off): a second copy of the primary result in split form (GRU state update:
                                  h' is needed as fp32 by the next gate epilogue AND split by the next convolution); not
                                  with epilogue 2 */
  int dst_split_c_stride, dst_split_c_offset;
  int src_bounded;             /* != 0: the caller guarantees |source| < 65504 / a_scale, the range guard skips this launch.  The
                                  encoder sets it: every convolution input there is an instance-normalised map or a ReLU sum of a few
                                  (|(x - mean) * rstd| <= sqrt(H*W); extractor.py:48-58), two orders below the fp16x3 range */
  int tile;                    /* 0 = automatic; 1 = 128x64, 2 = 128x128 as 4 column waves, 3 = 128x128 as 2x2 waves, 4 = 128x64 with
                                  a block-deep register pipeline (3, 4: split sources only); 5 / 6 / 7 = the STRIP kernels with 160- / 32- / 96-row strips (160 / 32 / 96 output
                                  pixels x 64 / 96 / 128 columns per workgroup, weights and split-tensor activations by LDS-DMA:
                                  stride 1, 3x3 / 1x5 / 5x1, c_out > 32, source channel counts in multiples of 32; a launch with tile_stats / src0_mean_rstd tiles every image into
                                  rnnpose_conv_tiles_per_image_ex(..., tile, B) tiles).  The automatic choice takes 160-row strips for
                                  these layer shapes when they give the launch >= 240 workgroups (or >= 24 per image) -- r06: 32-row strips
                                  instead when those are at most 256 waves, 96-row strips when at most 512 (the thin layers of a
                                  half-batch chain) --, 32-row strips for launches of <= 8192 pixels (single-image crops), else the 128-row kernels
                                  (rnnpose_conv_strip(0): never strips). */
  void* ksplit_ws;             /* optional (NULL = off): workspace of rnnpose_conv_ksplit_workspace_bytes() bytes, 16-byte aligned,
                                  its first 1024 bytes ZERO before the first launch (the kernel leaves them zero).  With it, a
                                  stride-1 launch of few tiles (B = 1 crops: 8-64 workgroups on 256 CUs) splits its K loop over up
                                  to 8 workgroups per tile; the last one to arrive sums the partial accumulators in split order
                                  (deterministic) and runs the epilogue.  Launches that may run CONCURRENTLY (two streams) need
                                  separate workspaces. */
  size_t ksplit_ws_bytes;
  int single_product;          /* != 0: ONE fp16 product per multiply-add (a_hi * b_hi, fp32 accumulation) instead of the three of the
                                  fp16x3 split -- cfg.raft.mixed_precision, the arithmetic the reference runs on a GPU (model/CFNet.py:
                                  47,126,152: fp16 autocast around encoder and update block).  Honoured by the 160-row strip kernels
                                  (the layers the automatic choice gives them: the update block and the encoder's residual layers at
                                  the headline shapes); every other kernel keeps the three-product form.  ~2^-11 relative per product:
                                  NOT the arithmetic of the parity tolerances. */
  int tile_stats_records;      /* with tile_stats: how many (c_out, 2) records the buffer holds.  The launch REFUSES a buffer smaller
                                  than B * rnnpose_conv_tiles_per_image_desc(this descriptor) -- which kernel family a launch takes
                                  (and with it the tile count: 32-row strips write four times the records of 128-row tiles) is the
                                  library's choice, so the buffer is checked where the choice is made. */
} rnnpose_conv_desc_t;

size_t rnnpose_conv_ksplit_workspace_bytes(void);
int rnnpose_conv_ksplit(int enable);       /* measurement switch (default 1): 0 = never split K */
int rnnpose_conv_ksplit_limits(int max_tiles, int max_splits);   /* measurement: launches of <= max_tiles (default 24, <= 96) tiles
                                                                     split into <= max_splits (default 4, 2..8) ranges */

/* Output tiles per image of a convolution launch = records per image of its `tile_stats`: 3x3 stride-1 layers run on 8 x 16
 * image PATCHES (ceil(W/16) * ceil(H/8) tiles per image, nine taps on one staged halo tile), everything else on runs of 128
 * output pixels (ceil(H_out*W_out/128) when tiled per image).  rnnpose_conv_spatial_tiles(0) switches the patch tiling off
 * (measurement: the r02 row-major tiling for 3x3 layers too). */
int rnnpose_conv_tiles_per_image(int H, int W, int kh, int kw, int stride);      /* (the 128-row kernels' rule only: it does NOT size
                                                                                      tile_stats of a tile = 0 launch since ABI 2) */
/* Records per image of `tile_stats` for exactly the launch this descriptor describes -- the kernel choice looks at B, H, W, kh, kw,
 * stride, c_out, tile, the sources' channel counts and whether src0_mean_rstd is set (pointers are not dereferenced, dst / weights
 * may be NULL): the one sizing rule that cannot disagree with rnnpose_conv2d_nhwc_f16x3.  -1 on bad arguments. */
int rnnpose_conv_tiles_per_image_desc(const rnnpose_conv_desc_t* h_desc);
/* fp16 MFMA products per multiply-add the launch this descriptor describes executes: 3 (fp16x3 split) or 1 (single_product honoured:
 * 160-row strips with one column tile per wave).  For flop accounting (bench.py's pipe_util); -1 on bad arguments. */
int rnnpose_conv_products_desc(const rnnpose_conv_desc_t* h_desc);
/* The same for the kernel a launch of `batch` images with this c_out and `tile` request (0 automatic .. 7) will take: the strip
 * kernels tile an image into ceil(W/16) * ceil(H/10) patches of 10 x 16 pixels (3x3) or ceil(H*W/160) runs of 160 pixels (96-row strips: patches of 6 x 16 pixels, runs of 96; 32-row
 * strips: patches of 2 x 16 pixels, runs of 32).  Shape-only: it assumes source channel counts in multiples of 32 (launches whose
 * sources are not fall back to the 128-row kernels: use rnnpose_conv_tiles_per_image_desc).  STRIDE 2: -1 whenever the shape alone would
 * allow the strip form -- whether a stride-2 launch takes it depends on its sources: only rnnpose_conv_tiles_per_image_desc answers.
 * A launch with statistics requires tile_stats_records == B * (that answer), not merely >= (ABI 3, tightened in r06). */
int rnnpose_conv_tiles_per_image_ex(int H, int W, int kh, int kw, int stride, int c_out, int tile, int batch);
int rnnpose_conv_spatial_tiles(int enable);
int rnnpose_conv_strip(int mode);          /* measurement switch: 0 = the automatic choice never takes the strip kernels, 1 = default,
                                              2 = automatic without the two-wave workgroups of 64-channel layers, 3 = two 32-column
                                              tiles per wave (one workgroup per CU), 4 = 160-row strips only (>= 24 per image), 5 = automatic
                                              without the stride-2 form (r05: stride-2 3x3 layers as 2x2-tap strips over the four parity
                                              planes of the input; one fp32 source of whole 32-channel blocks, even H and W), 6 = automatic without r06's
                                              32-row strips for launches whose 160-row strips would be at most 256 waves, 7 = automatic WITH persistent launches of the
                                              fp32-source strip forms (r06: resident workgroups walking the tile list; measured slower, off by default) */
/* number of fp16 elements of the packed weight array (hi and lo parts interleaved, in the fragment order of the 128-row kernels
 * followed by the record order of the strip kernels); -1 on bad arguments */
long long rnnpose_conv_packed_halfs(int c_out, int kh, int kw, const int* h_seg_counts, int n_seg);
int rnnpose_conv_pack_weights_f16x3(const float* w_oihw, int c_out, int c_in, int kh, int kw, const int* h_seg_counts,
                                    int n_seg, float w_scale, void* w_packed, rnnpose_stream_t stream);
int rnnpose_conv2d_nhwc_f16x3(const rnnpose_conv_desc_t* h_desc, rnnpose_stream_t stream);

/* fp16x3 range guard.  Activations are split as x*a_scale = hi + lo in fp16: |x*a_scale| > 65504 does not fit, the hi
 * part is clamped and the result is WRONG (the reference computes in fp32 and has no such cliff).  With the check
 * enabled, every launch of the fp16x3 kernels (convolution, encoder stem, correlation-volume operand split) counts the
 * input quads it had to clamp (or that were not finite) in a device counter; rnnpose_f16x3_saturation_count
 * synchronises `stream`, returns the count since the last reset and optionally resets it.  Off by default (the test costs
 * ~3 vector instructions per staged float4).  Process-global switch; not meant to be toggled concurrently with launches. */
int rnnpose_f16x3_saturation_check(int enable);
int rnnpose_f16x3_saturation_count(unsigned long long* h_count, int reset, rnnpose_stream_t stream);
/* Device-side read of the same counter: copies it into *d_count (DEVICE memory) on `stream`, no host synchronisation --
 * callers fold it into their outputs (PoseRefiner returns it as "f16x3_range_events") so that a clamp cannot pass unseen. */
int rnnpose_f16x3_saturation_peek(unsigned long long* d_count, rnnpose_stream_t stream);

/* ---- f1: RAFT encoder stem -- input normalisation + 7x7 stride-2 convolution 3 -> 64 in one kernel -------------
 *      model/CFNet.py:42-43 (image = 2*(image/255) - 1), thirdparty/raft/extractor.py:131,197 (conv1)
 * img (N,3,H,W) fp32 NCHW -> out (N, ceil(H/2), ceil(W/2), 64) fp32 NHWC = conv7x7_s2_p3(normalize ? 2*(img/255)-1 : img) + bias.
 * Weights (64,3,7,7) packed once by rnnpose_stem_pack_weights_f16x3 (2 arrays of rnnpose_stem_packed_halfs() fp16).
 * tile_stats (optional): (N * tiles_per_image, 64, 2) per-tile column sums / sums of squares for
 * rnnpose_instnorm_tiles_nhwc_f32 (8 x 16 output tiles; pixels of a ragged tile outside the image never enter the sums, so the
 * statistics hold for every size -- rnnpose_stem_tiles reports h_exact = 1 always since r03; the field is kept for callers).
 * r06: the kernel is PERSISTENT -- two workgroups per CU walk the tile list with the packed weights resident in LDS and the patch split
 * into fp16 hi / lo once per tile; rnnpose_stem_workgroups(n) caps the number of workgroups (tests: many tiles per workgroup at small
 * sizes; 0 = default).  Results do not depend on it. */
long long rnnpose_stem_packed_halfs(void);
int rnnpose_stem_workgroups(int max_workgroups);
int rnnpose_stem_pack_weights_f16x3(const float* w_oihw, float w_scale, void* w_hi, void* w_lo, rnnpose_stream_t stream);
int rnnpose_stem_tiles(int H, int W, int* h_tiles_per_image, int* h_exact);
int rnnpose_stem_conv7x7_s2_f16x3(const float* img_nchw, int N, int H, int W, int normalize, const void* w_hi,
                                  const void* w_lo, const float* bias, float a_scale, float w_scale, float* out_nhwc,
                                  double* tile_stats, rnnpose_stream_t stream);

/* ---- NHWC companions of the fused update-block engine ----------------------------------------------------
 * corr_lookup_nhwc: a3 with the output laid out (B,h,w,levels*81) (thirdparty/raft/corr.py:36-57).
 * nchw_to_nhwc / nhwc_to_nchw: (B,C,HW) <-> channel window [c_offset, c_offset+C) of a (B,HW,c_stride) tensor.
 * flow_prep: flow = coords1 - grid (model/CFNet.py:150; subtract_grid = 0: coords1 already holds the flow, the
 *   BasicUpdateBlock.forward boundary of update.py:178) written as (B,h,w,4) [fx,fy,0,0] (input of the 7x7 flow
 *   convolution, update.py:84) and into channels [motion_c_offset, +2) of the motion-feature tensor (update.py:97).
 * flow_conv7x7_relu: BasicMotionEncoder.convf1 + ReLU (7x7, 2 -> c_out <= 128, update.py:84,91) as a direct fp32
 *   convolution (K = 98 is too thin for the matrix cores): flow4 from flow_prep, w_t = the (c_out,2,7,7) weight
 *   transposed to (98, c_out), output into channels [out_c_offset, +c_out) of an NHWC tensor.
 * flow_head_out: FlowHead.conv2 (3x3, c_in -> 2, update.py:10,14) on channels [x_c_offset, +c_in) of x, fused with
 *   coords1 += delta (CFNet.py:157): delta (B,h,w,2), coords1_out (B,2,h,w) (may be NULL), flow_lr (B,h,w,2) =
 *   coords1_out - grid.  w_oihw is the PyTorch (2,c_in,3,3) weight.
 * convex_upsample_nhwc: a6 (model/CFNet.py:95-106) with mask (B,h,w,576) and flow_lr (B,h,w,2) -> (B,2,8h,8w). */
int rnnpose_corr_lookup_nhwc_f32(const float* pyramid, const float* coords, int B, int h, int w, int levels, int radius,
                                 float* out, rnnpose_stream_t stream);
/* NHWC lookup of the images [b0, b1) of a pyramid built for batch B: coords (b1-b0,2,h,w) and out (b1-b0,h,w,levels*81)
 * are the sub-batch tensors, `pyramid` the whole buffer (thirdparty/raft/corr.py:36-57 per image). */
int rnnpose_corr_lookup_nhwc_part_f32(const float* pyramid, const float* coords, int B, int b0, int b1, int h, int w,
                                      int levels, int radius, float* out, rnnpose_stream_t stream);
/* r06: the same lookup FORMING its coordinates itself: coords1 = grid + down-sampled pose-induced flow (exactly
 * rnnpose_induced_coords_lowres_f32's values: geometry/transformation.py:184-198, model/PoseRefiner.py:324-328, model/CFNet.py:136-144) from
 * depth (b1-b0,1,H,W), K (b1-b0,3,3), G (b1-b0,4,4) of the part's images; coords_out (b1-b0,2,h,w) receives them for the later consumers of
 * the iteration.  One launch and one dependent kernel boundary fewer per GRU iteration. */
int rnnpose_corr_lookup_induced_nhwc_part_f32(const float* pyramid, const float* depth, const float* K, const float* G, int H, int W,
                                              float eps, int B, int b0, int b1, int h, int w, int levels, int radius, float* coords_out,
                                              float* out, rnnpose_stream_t stream);
int rnnpose_nchw_to_nhwc_f32(const float* src, int B, int C, int HW, float* dst, int dst_c_stride, int dst_c_offset,
                             rnnpose_stream_t stream);
int rnnpose_nhwc_to_nchw_f32(const float* src, int B, int C, int HW, int src_c_stride, int src_c_offset, float* dst,
                             rnnpose_stream_t stream);
int rnnpose_flow_prep_f32(const float* coords1, int subtract_grid, int B, int h, int w, float* flow4, float* motion,
                          int motion_c_stride, int motion_c_offset, rnnpose_stream_t stream);
int rnnpose_flow_conv7x7_relu_f32(const float* flow4, const float* w_t, const float* bias, int B, int h, int w, int c_out,
                                  float* out, int out_c_stride, int out_c_offset, rnnpose_stream_t stream);
/* flow_prep + flow_conv7x7_relu in one launch: coords1 (B,2,h,w) planar (grid subtracted here when subtract_grid != 0) ->
 * relu(convf1(flow)) into `out`, and the flow itself into channels [motion_c_offset, +2) of `motion` (update.py:84,91,97). */
int rnnpose_flow_features_f32(const float* coords1, int subtract_grid, const float* w_t, const float* bias, int B, int h, int w,
                              int c_out, float* out, int out_c_stride, int out_c_offset, float* motion, int motion_c_stride,
                              int motion_c_offset, int out_split, int motion_split, float a_scale, rnnpose_stream_t stream);
/* r06: rnnpose_flow_features_f32 forming coords1 itself from depth (B,1,H,W), K (B,3,3), G (B,4,4) (see
 * rnnpose_corr_lookup_induced_nhwc_part_f32), grid subtracted: flow = coords1 - grid -> relu(convf1(flow)), motion[..., co:co+2]. */
int rnnpose_flow_features_induced_f32(const float* depth, const float* K, const float* G, int H, int W, float eps, const float* w_t,
                                      const float* bias, int B, int h, int w, int c_out, float* out, int out_c_stride, int out_c_offset,
                                      float* motion, int motion_c_stride, int motion_c_offset, int out_split, int motion_split,
                                      float a_scale, rnnpose_stream_t stream);
/* out_split / motion_split != 0: `out` / `motion` are split tensors (rnnpose_conv_desc_t, "SPLIT TENSORS") with scale a_scale.
 * rnnpose_split_hl_f32: channels [src_c_offset, +c_count) of an fp32 NHWC tensor -> the split form in channels
 * [dst_c_offset, +c_count) of `dst` (c_count, dst offsets / strides multiples of 8): hidden state / context input once per
 * outer iteration (model/CFNet.py:131-133 produce them in fp32). */
int rnnpose_split_hl_f32(const float* src, int src_c_stride, int src_c_offset, long long n_pixels, int c_count, float a_scale,
                         float* dst, int dst_c_stride, int dst_c_offset, rnnpose_stream_t stream);
int rnnpose_flow_head_out_f32(const float* x, int x_c_stride, int x_c_offset, int c_in, const float* w_oihw,
                              const float* bias, const float* coords1, int B, int h, int w, float* delta,
                              float* coords1_out, float* flow_lr, rnnpose_stream_t stream);
int rnnpose_convex_upsample_nhwc_f32(const float* flow_lr, const float* mask, int B, int h, int w, float* flow_up,
                                     rnnpose_stream_t stream);
/* 1x1 convolution (+ bias, optional ReLU) with the activation tile resident in LDS: 32 pixels x ALL 256 output columns per
 * workgroup, the tile split into fp16 hi / lo once (the implicit-GEMM kernel re-splits it per 64-column tile).  Used for
 * BasicMotionEncoder.convc1 (thirdparty/raft/update.py:80,87: 324 -> 256).  pack: weight (256, c_in) fp32 -> MFMA fragments
 * (rnnpose_conv1x1_resident_packed_bytes(c_in) bytes, 16-byte aligned) of w_scale * weight; c_in % 4 == 0, c_in <= 352.
 * x: n_pixels rows of x_c_stride floats, channels [x_c_offset, +c_in); dst: rows of dst_c_stride, channels [dst_c_offset, +256).
 * Agrees with rnnpose_conv2d_nhwc_f16x3 to fp32 round-off (the K sum is grouped differently).  dst_split != 0: dst is written
 * as a split tensor (rnnpose_conv_desc_t, "SPLIT TENSORS") with the launch's a_scale. */
size_t rnnpose_conv1x1_resident_packed_bytes(int c_in);
int rnnpose_conv1x1_resident_pack_f16x3(const float* weight, int c_out, int c_in, float w_scale, void* packed,
                                        rnnpose_stream_t stream);
int rnnpose_conv1x1_resident_f16x3(const float* x, int x_c_stride, int x_c_offset, int c_in, const void* w_packed,
                                   const float* bias, float a_scale, float w_scale, int relu, long long n_pixels, float* dst,
                                   int dst_c_stride, int dst_c_offset, int dst_split, rnnpose_stream_t stream);
/* r06: the window lookup and convc1 in ONE launch -- corr = CorrBlock.__call__(coords) (thirdparty/raft/corr.py:36-57, bilinear_sampler:
 * thirdparty/raft/utils/utils.py:57-71) followed by relu(convc1(corr)) (thirdparty/raft/update.py:80,87), as GRU_CFUpdator.forward
 * chains them (model/CFNet.py:147-152).  The 4 x 81 window features of a 32-pixel tile are computed level by level straight into an LDS ring of
 * channel blocks that the 1x1 convolution's MFMAs consume (csrc/corr_convc1.hip); the (B,h,w,324) tensor is never written.  (Measured equal
 * to the two-kernel path, not faster: the engine keeps the two kernels by default.)  pyramid: the buffer of rnnpose_corr_pyramid_* built for B
 * images; [b0, b1): the images of this launch; coords (b1-b0, 2, h, w) fp32 (channel 0 = x): window centres; w_packed / bias / scales:
 * convc1 packed by rnnpose_conv1x1_resident_pack_f16x3 with c_in = 324; dst (b1-b0, h, w, dst_c_stride), channels [dst_c_offset, +256),
 * fp32 or (dst_split) a split tensor.  levels must be 4, radius 4.  Same values as rnnpose_corr_lookup_nhwc_part_f32 followed by
 * rnnpose_conv1x1_resident_f16x3 (identical operations per element; tests compare them to 1e-6). */
int rnnpose_corr_lookup_convc1_f16x3(const float* pyramid, const float* coords, int B, int b0, int b1, int h, int w, int levels,
                                     int radius, const void* w_packed, const float* bias, float a_scale, float w_scale, int relu,
                                     float* dst, int dst_c_stride, int dst_c_offset, int dst_split, rnnpose_stream_t stream);
/* mask.2 + convex up-sampling in ONE kernel: mask = post_scale * mask.2(x) (1x1, 256 -> 576, thirdparty/raft/update.py:183-187)
 * is consumed in registers by the softmax / 3x3 convex combination of model/CFNet.py:95-106; the (B,h,w,576) mask tensor is
 * never written.  pack: weight (576,256) fp32 -> fp16 hi / lo MFMA fragments (rnnpose_mask_upsample_packed_bytes() bytes,
 * 16-byte aligned) of post_scale * w_scale * weight.  x (B,h,w,x_c_stride): channels [x_c_offset, +256) = relu(mask.0(h));
 * bias (576) already multiplied by post_scale; flow_lr (B,h,w,2); flow_up (B,2,8h,8w).  The softmax over the 9 taps is
 * evaluated online (running max / sums): agrees with conv2d + rnnpose_convex_upsample_nhwc_f32 to fp32 round-off. */
size_t rnnpose_mask_upsample_packed_bytes(void);
int rnnpose_mask_upsample_pack_f16x3(const float* weight, float post_scale, float w_scale, void* packed, rnnpose_stream_t stream);
int rnnpose_mask_upsample_f16x3(const float* x, int x_c_stride, int x_c_offset, const void* w_packed, int c_out,
                                const float* bias, float a_scale, float w_scale, const float* flow_lr, int B, int h, int w,
                                float* flow_up, rnnpose_stream_t stream);

/* ---- f1 (adjacent): instance norm of the RAFT encoder on NHWC tensors ---- thirdparty/raft/extractor.py:28-31,48-58
 * x (B,HW,C) -> out = act((x - mean_bc) * rsqrt(var_bc + eps)) (biased variance, no affine, as nn.InstanceNorm2d);
 * relu != 0 applies ReLU; if residual != NULL: out = relu(residual + out) (ResidualBlock tail, extractor.py:58).
 * mean_rstd (B,C,2) receives the statistics.  Deterministic (fixed-order fp64 partial sums in `workspace`). */
size_t rnnpose_instnorm_workspace_bytes(int B, int HW, int C);
int rnnpose_instnorm_nhwc_f32(const float* x, int B, int HW, int C, float eps, int relu, const float* residual,
                              void* workspace, size_t workspace_bytes, float* mean_rstd, float* out,
                              rnnpose_stream_t stream);
/* Same normalisation with the statistics taken from the producing convolution's `tile_stats` (tiles_per_image records per
 * image, as the producer laid them out): only the finalize + apply passes run.  out == NULL: statistics only (mean_rstd
 * for a consumer that normalises in its load, rnnpose_conv_desc_t.src0_mean_rstd).
 * residual_mean_rstd (optional, (B,C,2)): `residual` is itself a RAW convolution output whose instance norm [+ ReLU if
 * residual_relu] is applied on the fly: out = relu(act_r((residual - mean_r) * rstd_r) + act((x - mean) * rstd)) -- the
 * stem output / the down-sampling branch of a ResidualBlock are then never written in normalised form (extractor.py:54-58). */
int rnnpose_instnorm_tiles_nhwc_f32(const float* x, int B, int HW, int C, float eps, int relu, const float* residual,
                                    const float* residual_mean_rstd, int residual_relu, const double* tile_stats,
                                    int tiles_per_image, float* mean_rstd, float* out,
                                    rnnpose_stream_t stream);

/* ---- f3 ("next"): brute-force nearest neighbour for ADD-S ------ thirdparty/nn/src/nearest_neighborhood.cu:48-163
 * idxs[b,q] = argmin_r |ref[b,r,:] - que[b,q,:]|^2, first minimum wins, optional r != q; dim 2 or 3.
 * rnnpose_nn_search_f32: DEVICE pointers, asynchronous.
 * findNearestPointIdxLauncher: the reference's own cffi symbol (thirdparty/nn/src/ext.h:1-10): HOST pointers,
 * synchronous, same argument list -- thirdparty/nn/nn_utils.py binds to this library unchanged.              */
int rnnpose_nn_search_f32(const float* ref_pts, const float* que_pts, int* idxs, int b, int pn1, int pn2, int dim,
                          int exclude_self, rnnpose_stream_t stream);
void findNearestPointIdxLauncher(float* ref_pts, float* que_pts, int* idxs, int b, int pn1, int pn2, int dim,
                                 int exclude_self);

/* ---- f2 ("next"): LINEMOD pose metrics on device -------------------- utils/eval_metric.py:28-37,102-192
 * model (P,3), pose_pred / pose_gt (B,3,4), K (3,3) -> out (B,5) fp64 = [mean ADD distance, mean ADD-S distance
 * (nearest predicted point of every target point; -1 unless symmetric), mean 2-D projection error in px,
 * translation error in cm, rotation error in degrees].  Thresholding (0.1/0.02/0.05 diameter, 5 px, 5 cm 5 deg)
 * is host logic (rnnpose_amd/evaluator.py).  workspace is only needed when symmetric != 0.                    */
size_t rnnpose_pose_metrics_workspace_bytes(int B, int P);
int rnnpose_pose_metrics_f64(const float* model, int P, const float* pose_pred, const float* pose_gt, const float* K, int B,
                             int symmetric, void* workspace, size_t workspace_bytes, double* out, rnnpose_stream_t stream);

/* ---- f4 ("next"): zoom-crop of every outer iteration on device ------- model/PoseRefiner.py:145-218,286-291
 * mask_bbox: bbox (B,4) int32 = [xmin, ymin, xmax, ymax] of depth (B,1,H,W) > 0 ([INT_MAX,INT_MAX,-1,-1] if empty).
 * zoom_crop_params: get_affine_transformation + gen_zoom_crop_grids without the host round trip: K (B,3,3), T (B,4,4)
 *   -> theta (B,2,3) (the F.affine_grid matrices) and K_crop (B,3,3) = inverse(window transform) @ K.
 * zoom_crop: out (B,C,crop_h,crop_w) = F.grid_sample(in (B,C,H,W), F.affine_grid(theta)) (bilinear, zero padding,
 *   align_corners=False); grid_out (B,crop_h,crop_w,2) optionally receives the sampling grid; out may be NULL.  */
/* pointcloud_depth: DiffRender.render_pointcloud (geometry/diff_render_optim.py:369-401) for a batch: image b splats
 *   the vertices verts[vert_offsets[b] .. vert_offsets[b+1]) (device int array of B+1 entries; max_verts >= the largest
 *   count) with pose T (B,4,4) and intrinsics K (B,3,3): out (B,1,H,W) = 0 where nothing lands, else the depth of the
 *   HIGHEST-index vertex on that pixel (the reference leaves the winner of a collision unspecified); pixels are round-half-even and
 *   clamped into the image like the reference's.  workspace: B*H*W ints.  This is the foreground source of the zoom crop. */
size_t rnnpose_pointcloud_depth_workspace_bytes(int B, int H, int W);
int rnnpose_pointcloud_depth_f32(const float* verts, const int* vert_offsets, int max_verts, const float* T, const float* K, int B,
                                 int H, int W, void* workspace, size_t workspace_bytes, float* out, rnnpose_stream_t stream);
int rnnpose_mask_bbox_f32(const float* depth, int B, int H, int W, int* bbox, rnnpose_stream_t stream);
int rnnpose_zoom_crop_params_f32(const int* bbox, const float* K, const float* T, int B, int H, int W, int crop_h, int crop_w,
                                 float margin_ratio, float* theta, float* K_crop, rnnpose_stream_t stream);
int rnnpose_zoom_crop_f32(const float* in, const float* theta, int B, int C, int H, int W, int crop_h, int crop_w, float* out,
                          float* grid_out, rnnpose_stream_t stream);

/* ---- f4 (second half): triangle-mesh rasteriser of the render hand-off ------------------------------------------------
 *      geometry/diff_render_optim.py:283-367 (DiffRender.forward / render_depth on PyTorch3D's MeshRasterizer,
 *      faces_per_pixel = 1, blur 0) and model/PoseRefiner.py:118-143 (render).  PARITY UNPINNED (PyTorch3D absent).
 * Geometry of a batch: verts (V,3) = the vertices of all loaded models back to back; faces (F,3) int32 = LOCAL vertex
 * indices, models back to back; image b uses vertices from vert_off[b], faces [face_off[b], face_off[b] + face_cnt[b])
 * (device int32 arrays of B entries); T (B,4,4) object->camera, K (B,3,3).  A pixel (x, y) samples the ray through
 * (x + pixel_center, y + pixel_center) (PyTorch3D: 0.5); the nearest face by interpolated camera z wins, equal depths go
 * to the lower face index; faces with a vertex at z <= near are dropped.
 * rnnpose_raster_mesh_f32: pass 1, fills the z-buffer `workspace` (rnnpose_raster_workspace_bytes).
 * rnnpose_raster_resolve_f32: pass 2 from the same workspace and geometry arguments:
 *   out_zbuf   (B,1,H,W)  interpolated depth, `empty_depth` where no face covers the pixel (the reference uses -1);
 *   out_vdepth (B,1,H,W)  camera z of the vertex with the largest barycentric weight, 0 where empty (`render_depth`);
 *   out_attr   (B, 3*with_color + C, H, W): [shaded vertex colour |] attr interpolated with the (perspective-correct if
 *              requested) barycentrics, 0 where empty.  attr = per-vertex rows of C floats, image b's rows start at element
 *              attr_off[b] (device int64 array); colors (V,3) per-vertex albedo aligned with verts (NULL = white);
 *              shade != 0: (0.5 + 0.3 |n.l|) * albedo + 0.2 with the flat face normal and a point light at (1,1,-1) in
 *              object space (PyTorch3D's default Phong terms at shininess 0).  Any output pointer may be NULL. */
size_t rnnpose_raster_workspace_bytes(int B, int H, int W);
int rnnpose_raster_mesh_f32(const float* verts, const int* faces, const int* vert_off, const int* face_off,
                            const int* face_cnt, int max_faces, const float* T, const float* K, int B, int H, int W,
                            float near, float pixel_center, int perspective_correct, void* workspace,
                            size_t workspace_bytes, rnnpose_stream_t stream);
int rnnpose_raster_resolve_f32(const float* verts, const int* faces, const int* vert_off, const int* face_off, const float* T,
                               const float* K, int B, int H, int W, float near, float pixel_center, int perspective_correct,
                               const void* workspace, const float* attr, const long long* attr_off, int C,
                               const float* colors, int with_color, int shade, float empty_depth, float* out_attr,
                               float* out_zbuf, float* out_vdepth, rnnpose_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RNNPOSE_HIP_H */
