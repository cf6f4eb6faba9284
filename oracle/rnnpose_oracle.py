"""CPU ORACLE for the RNNPose recurrent-refinement hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch-CPU / numpy, fp32 with the fp64 Levenberg-Marquardt
section of the reference) of the algorithm SURVEY.md §8(a) / Appendix A describes.  It exists to CHECK
the HIP path; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it.  Nothing under `rnnpose_amd/` imports this module, and the product path raises if the HIP library
is missing -- it never falls back to this code.

Parity pinning: the reference ships no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference ITSELF, produced in the build container by importing
the reference's Python modules (tests/golden/gen_golden.py, committed) and stored as small .npz
fixtures under tests/golden/.  tests/test_oracle_golden.py replays them on every CPU run.

Every function cites the reference file:line it restates (paths relative to /root/reference).
Bilinear sampling is written out as explicit index arithmetic (no grid_sample) so that the sampling
conventions (align_corners, zero padding, x-major window order) are stated, not inherited.  Dense
convolutions / matmuls use torch-CPU library calls: they are plain fp32 contractions.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# constants, with their sources
EPS_DEPTH = 1e-5          # model/PoseRefiner.py:21
MIN_DEPTH_VALID = 0.1     # geometry/transformation.py:16
MIN_DEPTH_PROJ = 0.01     # geometry/projective_ops.py:9
LM_LMBDA = 1e-4           # config/default.py:54
EP_LMBDA = 100.0          # config/default.py:55
MAX_UPDATE = 1.0          # geometry/cholesky.py:32
MIN_THETA = 1e-4          # geometry/se3.py:10


_DTYPE = torch.float32      # working precision of the restatement (the reference's CPU path is fp32 + fp64 LM)


def _t(x, dtype=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(_DTYPE if dtype is None else dtype)


class precision:
    """`with precision(torch.float64): refine(...)` evaluates the SAME arithmetic in fp64: the value that both fp32
    evaluations -- this oracle's and the GPU's -- approximate.  Used to put a number on the reference's own fp32 round-off
    (bench.py's parity block, tools/error_budget.py): two implementations cannot be asked to agree more closely than the
    reference agrees with its exact self."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _DTYPE
        self._old, self._old_default = _DTYPE, torch.get_default_dtype()
        _DTYPE = self.dtype
        torch.set_default_dtype(self.dtype)
        return self

    def __exit__(self, *exc):
        global _DTYPE
        _DTYPE = self._old
        torch.set_default_dtype(self._old_default)
        return False


# --------------------------------------------------------------------------------------------------
# bilinear sampling with zero padding at pixel coordinates (explicit)
# --------------------------------------------------------------------------------------------------
def bilinear_zero_pad(img: torch.Tensor, px: torch.Tensor, py: torch.Tensor) -> torch.Tensor:
    """img (M,Hs,Ws); px,py (M,...) pixel coords -> (M,...).  Zero outside, weights (1-fx)(1-fy)...
    Restates the interpolation of F.grid_sample(mode='bilinear', padding_mode='zeros') once the
    coordinates are in pixel units."""
    M, Hs, Ws = img.shape
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    fx = px - x0
    fy = py - y0
    x0 = x0.clamp(-4, Ws + 4).long()   # clamp only to keep the integer conversion safe
    y0 = y0.clamp(-4, Hs + 4).long()
    flat = img.reshape(M, Hs * Ws)
    out = torch.zeros_like(px)
    sh = px.shape
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi = x0 + dx
            yi = y0 + dy
            ok = (xi >= 0) & (xi < Ws) & (yi >= 0) & (yi < Hs)
            idx = (yi.clamp(0, Hs - 1) * Ws + xi.clamp(0, Ws - 1)).reshape(M, -1)
            v = torch.gather(flat, 1, idx).reshape(sh)
            out = out + torch.where(ok, v * (wx * wy), torch.zeros_like(v))
    bad = ~(torch.isfinite(px) & torch.isfinite(py))
    return torch.where(bad, torch.zeros_like(out), out)


def resize_bilinear_ac(x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    """(B,C,H,W) -> (B,C,oh,ow), bilinear with align_corners=True: src = dst*(in-1)/(out-1).
    Restates F.interpolate(..., mode='bilinear', align_corners=True) (model/CFNet.py:129,142)."""
    B, C, H, W = x.shape
    sy = (H - 1) / (oh - 1) if oh > 1 else 0.0
    sx = (W - 1) / (ow - 1) if ow > 1 else 0.0
    ys = torch.arange(oh, dtype=torch.float32) * np.float32(sy)
    xs = torch.arange(ow, dtype=torch.float32) * np.float32(sx)
    y0 = torch.floor(ys).long().clamp(0, H - 1)
    x0 = torch.floor(xs).long().clamp(0, W - 1)
    y1 = (y0 + 1).clamp(max=H - 1)
    x1 = (x0 + 1).clamp(max=W - 1)
    fy = (ys - y0.float()).view(1, 1, oh, 1)
    fx = (xs - x0.float()).view(1, 1, 1, ow)
    r0 = x[:, :, y0]
    r1 = x[:, :, y1]
    top = r0[:, :, :, x0] * (1 - fx) + r0[:, :, :, x1] * fx
    bot = r1[:, :, :, x0] * (1 - fx) + r1[:, :, :, x1] * fx
    return top * (1 - fy) + bot * fy


# --------------------------------------------------------------------------------------------------
# a1/a2: all-pairs correlation + pyramid                       thirdparty/raft/corr.py:13-34,59-67
# --------------------------------------------------------------------------------------------------
def corr_pyramid(fmap1, fmap2, num_levels: int = 4):
    """-> list of (B*h*w, h_l, w_l) fp32; level 0 = fmap1^T fmap2 / sqrt(C); level l = 2x2 mean of l-1
    with floor cropping (avg_pool2d(2, stride=2))."""
    f1, f2 = _t(fmap1), _t(fmap2)
    B, C, h, w = f1.shape
    corr = torch.matmul(f1.reshape(B, C, h * w).transpose(1, 2), f2.reshape(B, C, h * w))
    corr = corr / math.sqrt(C)
    lvl = corr.reshape(B * h * w, h, w)
    pyr = [lvl]
    for _ in range(num_levels - 1):
        hh, ww = lvl.shape[1] // 2, lvl.shape[2] // 2
        c = lvl[:, : hh * 2, : ww * 2].reshape(-1, hh, 2, ww, 2)
        lvl = c.sum(dim=(2, 4)) * 0.25
        pyr.append(lvl)
    return pyr


# --------------------------------------------------------------------------------------------------
# a3: pyramid lookup                 thirdparty/raft/corr.py:36-57, thirdparty/raft/utils/utils.py:57-71
# --------------------------------------------------------------------------------------------------
def corr_lookup(pyramid, coords, radius: int = 4):
    """coords (B,2,h,w) (ch0=x, ch1=y) -> (B, L*(2r+1)^2, h, w).
    Channel l*81 + i*9 + j samples x-offset (i-r), y-offset (j-r)  (x-major: `delta` is built from
    meshgrid(dy,dx) but added to (x,y), corr.py:44-50).  The reference normalises with 2x/(W-1)-1 and
    grid_sample(align_corners=True) maps back with ((g+1)/2)*(W-1); both are replayed in fp32."""
    coords = _t(coords)
    B, _, h, w = coords.shape
    n = 2 * radius + 1
    cx = coords[:, 0].reshape(B * h * w, 1, 1)
    cy = coords[:, 1].reshape(B * h * w, 1, 1)
    d = torch.arange(-radius, radius + 1, dtype=torch.float32)
    off_x = d.view(1, n, 1)   # index i -> x offset
    off_y = d.view(1, 1, n)   # index j -> y offset
    outs = []
    for l, lvl in enumerate(pyramid):
        lvl = _t(lvl)
        Hl, Wl = lvl.shape[-2:]
        x = cx / (2 ** l) + off_x + torch.zeros(1, n, n)
        y = cy / (2 ** l) + off_y + torch.zeros(1, n, n)
        xg = 2 * x / (Wl - 1) - 1
        yg = 2 * y / (Hl - 1) - 1
        px = ((xg + 1) / 2) * (Wl - 1)
        py = ((yg + 1) / 2) * (Hl - 1)
        s = bilinear_zero_pad(lvl, px, py)                 # (B*h*w, n, n) indexed [i,j]
        outs.append(s.reshape(B, h, w, n * n))
    out = torch.cat(outs, dim=-1)
    return out.permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------------------
# a4: BasicUpdateBlock                                   thirdparty/raft/update.py:79-97,33-60,6-14,164-188
# --------------------------------------------------------------------------------------------------
UPDATE_BLOCK_SHAPES = {
    "encoder.convc1.weight": (256, 324, 1, 1), "encoder.convc1.bias": (256,),
    "encoder.convc2.weight": (192, 256, 3, 3), "encoder.convc2.bias": (192,),
    "encoder.convf1.weight": (128, 2, 7, 7), "encoder.convf1.bias": (128,),
    "encoder.convf2.weight": (64, 128, 3, 3), "encoder.convf2.bias": (64,),
    "encoder.conv.weight": (126, 256, 3, 3), "encoder.conv.bias": (126,),
    "gru.convz1.weight": (128, 384, 1, 5), "gru.convz1.bias": (128,),
    "gru.convr1.weight": (128, 384, 1, 5), "gru.convr1.bias": (128,),
    "gru.convq1.weight": (128, 384, 1, 5), "gru.convq1.bias": (128,),
    "gru.convz2.weight": (128, 384, 5, 1), "gru.convz2.bias": (128,),
    "gru.convr2.weight": (128, 384, 5, 1), "gru.convr2.bias": (128,),
    "gru.convq2.weight": (128, 384, 5, 1), "gru.convq2.bias": (128,),
    "flow_head.conv1.weight": (256, 128, 3, 3), "flow_head.conv1.bias": (256,),
    "flow_head.conv2.weight": (2, 256, 3, 3), "flow_head.conv2.bias": (2,),
    "mask.0.weight": (256, 128, 3, 3), "mask.0.bias": (256,),
    "mask.2.weight": (576, 256, 1, 1), "mask.2.bias": (576,),
}


def _conv(x, W, name, pad):
    return F.conv2d(x, _t(W[name + ".weight"]), _t(W[name + ".bias"]), padding=pad)


def update_block(W: dict, net, inp, corr, flow):
    """-> (net, mask, delta_flow).  W: {state_dict key: array} with the keys of UPDATE_BLOCK_SHAPES."""
    net, inp, corr, flow = _t(net), _t(inp), _t(corr), _t(flow)
    # BasicMotionEncoder (update.py:88-97)
    cor = F.relu(_conv(corr, W, "encoder.convc1", 0))
    cor = F.relu(_conv(cor, W, "encoder.convc2", 1))
    flo = F.relu(_conv(flow, W, "encoder.convf1", 3))
    flo = F.relu(_conv(flo, W, "encoder.convf2", 1))
    out = F.relu(_conv(torch.cat([cor, flo], 1), W, "encoder.conv", 1))
    motion = torch.cat([out, flow], 1)
    x = torch.cat([inp, motion], 1)                                  # update.py:181
    # SepConvGRU (update.py:45-60): horizontal (1x5) then vertical (5x1)
    h = net
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(hx, W, "gru.convz" + sfx, pad))
        r = torch.sigmoid(_conv(hx, W, "gru.convr" + sfx, pad))
        q = torch.tanh(_conv(torch.cat([r * h, x], 1), W, "gru.convq" + sfx, pad))
        h = (1 - z) * h + z * q
    # heads (update.py:184-187)
    dflow = _conv(F.relu(_conv(h, W, "flow_head.conv1", 1)), W, "flow_head.conv2", 1)
    mask = 0.25 * _conv(F.relu(_conv(h, W, "mask.0", 1)), W, "mask.2", 0)
    return h, mask, dflow


# --------------------------------------------------------------------------------------------------
# a5/a6: GRU_CFUpdator glue + convex upsampling                          model/CFNet.py:95-106,109-173
# --------------------------------------------------------------------------------------------------
def context_prep(ctx, hdim: int = 128):
    """ctx (B,256,H,W) -> net=tanh(first 128), inp=relu(rest) at 1/8 res   (CFNet.py:124-133)."""
    ctx = _t(ctx)
    B, C, H, W = ctx.shape
    cnet = resize_bilinear_ac(ctx, H // 8, W // 8)
    return torch.tanh(cnet[:, :hdim]), torch.relu(cnet[:, hdim:])


def coords_grid_lowres(B, h, w):
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    return torch.stack([xs, ys], 0)[None].repeat(B, 1, 1, 1)          # raft/utils/utils.py:74-77


def flow_init_to_coords1(flow_init):
    """full-res flow_init (B,2,H,W) -> coords1 (B,2,h,w) = grid + resize(flow_init/8)  (CFNet.py:136-144).
    (The reference divides in place; this does not mutate its input.)"""
    flow_init = _t(flow_init)
    B, _, H, W = flow_init.shape
    h, w = H // 8, W // 8
    ds = W // w
    return coords_grid_lowres(B, h, w) + resize_bilinear_ac(flow_init / ds, h, w)


def convex_upsample(flow, mask, scale: int = 8):
    """flow (B,2,h,w), mask (B,9*s*s,h,w) -> (B,2,s*h,s*w)   (CFNet.py:95-106).
    up[c, s*Y+i, s*X+j] = sum_k softmax_k(mask[k*s*s+i*s+j, Y, X]) * s*flow[c, Y+k//3-1, X+k%3-1]."""
    flow, mask = _t(flow), _t(mask)
    B, _, h, w = flow.shape
    s = scale
    m = torch.softmax(mask.reshape(B, 9, s, s, h, w), dim=1)
    fp = F.pad(s * flow, (1, 1, 1, 1))
    up = torch.zeros(B, 2, s, s, h, w)
    for k in range(9):
        ky, kx = k // 3, k % 3
        nb = fp[:, :, ky:ky + h, kx:kx + w]                          # flow[c, Y+ky-1, X+kx-1]
        up = up + m[:, k][:, None] * nb[:, :, None, None]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(B, 2, s * h, s * w)


# --------------------------------------------------------------------------------------------------
# a7: induced flow of the current relative pose
#     geometry/transformation.py:184-198, geometry/projective_ops.py:68-114, model/PoseRefiner.py:324-328
# --------------------------------------------------------------------------------------------------
def _pix_grid(H, W):
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                            torch.arange(W, dtype=torch.float32), indexing="ij")
    return xs, ys


def _intr(K):
    K = _t(K)
    return (K[:, 0, 0].view(-1, 1, 1), K[:, 1, 1].view(-1, 1, 1),
            K[:, 0, 2].view(-1, 1, 1), K[:, 1, 2].view(-1, 1, 1))


def backproject(Z, K):
    """Z (B,H,W) (already depth+EPS) -> X,Y,Z each (B,H,W)   (projective_ops.py:68-99)."""
    B, H, W = Z.shape
    xs, ys = _pix_grid(H, W)
    fx, fy, cx, cy = _intr(K)
    X = Z * (xs - cx) / fx
    Y = Z * (ys - cy) / fy
    return X, Y, Z


def transform_points(G, X, Y, Z):
    """G (B,4,4) fp32; einsum('aijk,ai...k->ai...j') on homogeneous points (transformation.py:78-86)."""
    G = _t(G).reshape(-1, 4, 4)
    g = lambda r, c: G[:, r, c].view(-1, 1, 1)
    X1 = g(0, 0) * X + g(0, 1) * Y + g(0, 2) * Z + g(0, 3)
    Y1 = g(1, 0) * X + g(1, 1) * Y + g(1, 2) * Z + g(1, 3)
    Z1 = g(2, 0) * X + g(2, 1) * Y + g(2, 2) * Z + g(2, 3)
    return X1, Y1, Z1


def project(X1, Y1, Z1, K):
    fx, fy, cx, cy = _intr(K)
    Zc = torch.clamp(Z1, min=MIN_DEPTH_PROJ)                          # projective_ops.py:107
    return fx * (X1 / Zc) + cx, fy * (Y1 / Zc) + cy, Zc


def induced_flow(depth, K, G):
    """depth (B,1,H,W) raw rendered depth (0=bg), K (B,3,3), G (B,1,4,4) ->
    flow_init (B,2,H,W) = (reproj - grid)*(depth+EPS > EPS), vmask (B,H,W) float."""
    D = _t(depth)[:, 0]
    Z = D + EPS_DEPTH                                                 # PoseRefiner.py:313
    X, Y, Z = backproject(Z, K)
    X1, Y1, Z1 = transform_points(G, X, Y, Z)
    u, v, _ = project(X1, Y1, Z1, K)
    xs, ys = _pix_grid(*D.shape[-2:])
    fg = (Z > EPS_DEPTH).float()
    flow = torch.stack([(u - xs) * fg, (v - ys) * fg], 1)
    vmask = ((Z > MIN_DEPTH_VALID) & (Z1 > MIN_DEPTH_VALID)).float()
    return flow, vmask


# --------------------------------------------------------------------------------------------------
# a8: descriptor reliability weight   model/PoseRefiner.py:342-345, geometry/projective_ops.py:11-23
# --------------------------------------------------------------------------------------------------
def corr_weight(g1, g2, target, depth, sigma):
    """g1,g2 (B,D,H,W); target (B,H,W,2) pixel coords; depth (B,1,H,W) raw; sigma scalar ->
    w (B,H,W) = exp(-|1 - <g1, sample(g2,target)>|/sigma) * (depth>0).
    Sampling: normalize_coords_grid uses 2x/(W-1)-1 (align_corners=True formula) but grid_sample runs
    with its default align_corners=False, i.e. the tap lands at ((g+1)*W-1)/2."""
    g1, g2, target = _t(g1), _t(g2), _t(target)
    B, Dn, H, W = g2.shape
    tx, ty = target[..., 0], target[..., 1]
    gx = 2 * tx / (W - 1) - 1
    gy = 2 * ty / (H - 1) - 1
    px = ((gx + 1) * W - 1) / 2
    py = ((gy + 1) * H - 1) / 2
    pxr = px[:, None].expand(B, Dn, H, W).reshape(B * Dn, H, W)
    pyr = py[:, None].expand(B, Dn, H, W).reshape(B * Dn, H, W)
    warped = bilinear_zero_pad(g2.reshape(B * Dn, H, W), pxr, pyr).reshape(B, Dn, H, W)
    s = (g1 * warped).sum(1)
    sig = float(np.asarray(sigma).reshape(-1)[0])
    return torch.exp(-torch.abs(1 - s) / sig) * (_t(depth)[:, 0] > 0).float()


# --------------------------------------------------------------------------------------------------
# a9: normal equations (fp64)     geometry/transformation.py:265-300, projective_ops.py:116-124
# --------------------------------------------------------------------------------------------------
def lm_normal_eq(target, weight, depth, K, G):
    """target (B,H,W,2), weight (B,H,W) fp32, depth (B,1,H,W) raw, K (B,3,3), G (B,1,4,4) ->
    H (B,6,6) fp64 (undamped), b (B,6) fp64."""
    tgt = _t(target, torch.float64)
    wgt = _t(weight, torch.float64)
    Z = _t(depth)[:, 0] + EPS_DEPTH
    X0, Y0, Z0 = backproject(Z, K)
    X1, Y1, Z1 = transform_points(G, X0, Y0, Z0)
    u, v, Zc = project(X1, Y1, Z1, K)
    fx, fy, cx, cy = _intr(K)
    valid = ((Z0 > MIN_DEPTH_VALID) & (Z1 > MIN_DEPTH_VALID)).double()
    small = Zc <= MIN_DEPTH_PROJ + 0.01
    zi1 = torch.where(small, torch.zeros_like(Zc), 1.0 / Zc)
    zi2 = torch.where(small, torch.zeros_like(Zc), 1.0 / Zc ** 2)
    o = torch.zeros_like(Zc)
    # J_pi rows in fp32 (projective_ops.py:122-124), note X1/Y1 unclamped, zi from clamped Z
    Jp = torch.stack([torch.stack([fx * zi1, o, -fx * X1 * zi2], -1),
                      torch.stack([o, fy * zi1, -fy * Y1 * zi2], -1)], -2).double()   # (B,H,W,2,3)
    i = torch.ones_like(Zc)
    # J_T columns j1..j6 (transformation.py:27-46), built from the TRANSFORMED point
    JT = torch.stack([torch.stack([i, o, o], -1), torch.stack([o, i, o], -1), torch.stack([o, o, i], -1),
                      torch.stack([o, -Z1, Y1], -1), torch.stack([Z1, o, -X1], -1),
                      torch.stack([-Y1, X1, o], -1)], -1).double()                     # (B,H,W,3,6)
    J = torch.matmul(Jp, JT)                                                          # (B,H,W,2,6)
    r = tgt - torch.stack([u, v], -1).double()                                        # (B,H,W,2)
    vw = (valid * wgt)[..., None, None]
    Hm = torch.einsum("bhwrj,bhwrk->bjk", vw * J, J)
    bv = torch.einsum("bhwrj,bhwr->bj", vw * J, r)
    return Hm, bv


# --------------------------------------------------------------------------------------------------
# a10: damped solve                geometry/transformation.py:300-302, geometry/cholesky.py:32-50
# --------------------------------------------------------------------------------------------------
def cholesky_solve6(Hm: np.ndarray, b: np.ndarray):
    """Plain fp64 Cholesky L L^T = H, forward/back substitution.  NaN if not positive definite."""
    n = Hm.shape[-1]
    L = np.zeros_like(Hm)
    with np.errstate(invalid="ignore", divide="ignore"):
        for j in range(n):
            s = Hm[j, j] - np.dot(L[j, :j], L[j, :j])
            L[j, j] = np.sqrt(s) if s > 0 else np.nan
            for i in range(j + 1, n):
                L[i, j] = (Hm[i, j] - np.dot(L[i, :j], L[j, :j])) / L[j, j]
        y = np.zeros(n)
        for i in range(n):
            y[i] = (b[i] - np.dot(L[i, :i], y[:i])) / L[i, i]
        x = np.zeros(n)
        for i in reversed(range(n)):
            x[i] = (y[i] - np.dot(L[i + 1:, i], x[i + 1:])) / L[i, i]
    return x


def lm_solve(Hm, b, ep_lmbda=EP_LMBDA, lm_lmbda=LM_LMBDA, max_update=MAX_UPDATE):
    """H (B,6,6), b (B,6) fp64 -> xi (B,6) fp32 = clamp(nan_to_zero(solve(H + ep*I + lm*diag(H), b)))."""
    Hm = np.asarray(Hm, dtype=np.float64).copy()
    b = np.asarray(b, dtype=np.float64)
    out = np.zeros(b.shape, dtype=np.float32)
    for n in range(Hm.shape[0]):
        Hd = Hm[n] + ep_lmbda * np.eye(6) + lm_lmbda * Hm[n] * np.eye(6)
        x = cholesky_solve6(Hd, b[n])
        x = np.where(np.isnan(x), 0.0, x)
        out[n] = np.clip(x, -max_update, max_update).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------------------
# a11: SE(3) exponential and left increment                          geometry/se3.py:228-281,303-306
# --------------------------------------------------------------------------------------------------
def se3_exp(xi):
    """xi (B,6) fp32 (upsilon, omega) -> (B,4,4) fp32, same formulas / thresholds as the reference."""
    xi = _t(xi).reshape(-1, 6)
    v, w = xi[:, :3], xi[:, 3:]
    th2 = (w ** 2).sum(1).view(-1, 1, 1)
    th = torch.sqrt(th2)
    th4 = th2 * th2
    z = torch.zeros_like(w[:, 0])
    wx = torch.stack([torch.stack([z, -w[:, 2], w[:, 1]], -1),
                      torch.stack([w[:, 2], z, -w[:, 0]], -1),
                      torch.stack([-w[:, 1], w[:, 0], z], -1)], -2)
    wx2 = torch.matmul(wx, wx)
    I = torch.eye(3).repeat(xi.shape[0], 1, 1)
    eps = 1e-12
    R1 = I + (1.0 - (1.0 / 6.0) * th2 + (1.0 / 120.0) * th4) * wx + (0.5 - (1.0 / 12.0) * th2 + (1.0 / 720.0) * th4) * wx2
    V1 = I + (0.5 - (1.0 / 24.0) * th2 + (1.0 / 720.0) * th4) * wx + ((1.0 / 6.0) - (1.0 / 120.0) * th2 + (1.0 / 5040.0) * th4) * wx2
    R2 = I + (torch.sin(th) / (th + eps)) * wx + ((1 - torch.cos(th)) / (th2 + eps)) * wx2
    V2 = I + ((1 - torch.cos(th)) / (th2 + eps)) * wx + ((th - torch.sin(th)) / (th2 * th + eps)) * wx2
    R = torch.where(th < MIN_THETA, R1, R2)
    V = torch.where(th < MIN_THETA, V1, V2)
    t = torch.matmul(V, v[..., None])
    G = torch.eye(4).repeat(xi.shape[0], 1, 1)
    G[:, :3, :3] = R
    G[:, :3, 3:] = t
    return G


def se3_increment(G, xi):
    """G <- exp(xi) G   (se3.py:303-306)."""
    G = _t(G)
    sh = G.shape
    return torch.matmul(se3_exp(xi), G.reshape(-1, 4, 4)).reshape(sh)


def lm_step(target, weight, depth, K, G, num_iters: int = 1):
    """SE3Sequence.reprojction_optim (transformation.py:265-316): num_iters damped GN steps.
    -> (G_new (B,1,4,4), list of (H,b,xi) per iteration)."""
    G = _t(G).reshape(-1, 1, 4, 4)
    trace = []
    for _ in range(num_iters):
        Hm, b = lm_normal_eq(target, weight, depth, K, G)
        xi = lm_solve(Hm.numpy(), b.numpy())
        G = se3_increment(G, xi)
        trace.append((Hm, b, torch.from_numpy(xi)))
    return G, trace


# --------------------------------------------------------------------------------------------------
# f1 (adjacent): RAFT BasicEncoder with instance norm          thirdparty/raft/extractor.py:6-58,118-232
# --------------------------------------------------------------------------------------------------
def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def _resblock(x, W, pfx, stride):
    y = F.relu(_inorm(F.conv2d(x, _t(W[pfx + ".conv1.weight"]), _t(W[pfx + ".conv1.bias"]), stride=stride, padding=1)))
    y = F.relu(_inorm(F.conv2d(y, _t(W[pfx + ".conv2.weight"]), _t(W[pfx + ".conv2.bias"]), padding=1)))
    if stride != 1:
        x = _inorm(F.conv2d(x, _t(W[pfx + ".downsample.0.weight"]), _t(W[pfx + ".downsample.0.bias"]), stride=stride))
    return F.relu(x + y)


def encoder_shapes(input_dim=3, output_dim=256):
    s = {"conv1.weight": (64, input_dim, 7, 7), "conv1.bias": (64,)}
    inp = 64
    for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
        for bi, (cin, st) in enumerate(((inp, stride), (dim, 1))):
            p = f"layer{li}.{bi}"
            s[p + ".conv1.weight"] = (dim, cin, 3, 3); s[p + ".conv1.bias"] = (dim,)
            s[p + ".conv2.weight"] = (dim, dim, 3, 3); s[p + ".conv2.bias"] = (dim,)
            if st != 1:
                s[p + ".downsample.0.weight"] = (dim, cin, 1, 1); s[p + ".downsample.0.bias"] = (dim,)
        inp = dim
    s["conv2.weight"] = (output_dim, 128, 1, 1); s["conv2.bias"] = (output_dim,)
    return s


def image_encoder(W: dict, image1, image2):
    """ImageFeaEncoder.forward (model/CFNet.py:41-49) incl. the 2*(x/255)-1 re-normalisation quirk."""
    x = torch.cat([_t(image1), _t(image2)], 0)
    x = 2 * (x / 255.0) - 1.0
    x = F.relu(_inorm(F.conv2d(x, _t(W["conv1.weight"]), _t(W["conv1.bias"]), stride=2, padding=3)))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _resblock(x, W, f"layer{li}.0", stride)
        x = _resblock(x, W, f"layer{li}.1", 1)
    x = F.conv2d(x, _t(W["conv2.weight"]), _t(W["conv2.bias"]))
    B = x.shape[0] // 2
    return x[:B], x[B:]


# --------------------------------------------------------------------------------------------------
# FAST variants for the cpu_baseline leg of bench.py only: the same stages written with the library calls the
# reference itself uses on CPU (grid_sample / einsum), so the timed CPU baseline reflects the reference's own
# CPU code path rather than this file's explicit (slow) index arithmetic.  tests/test_oracle_golden.py proves
# them equal to the explicit restatements above.
# --------------------------------------------------------------------------------------------------
def corr_lookup_fast(pyramid, coords, radius: int = 4):
    """thirdparty/raft/corr.py:36-57 with F.grid_sample(align_corners=True) (utils/utils.py:57-71)."""
    coords = _t(coords).permute(0, 2, 3, 1)
    B, h, w, _ = coords.shape
    n = 2 * radius + 1
    d = torch.linspace(-radius, radius, n)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, n, n, 2)
    outs = []
    for l, lvl in enumerate(pyramid):
        lvl = _t(lvl)
        Hl, Wl = lvl.shape[-2:]
        c = coords.reshape(B * h * w, 1, 1, 2) / 2 ** l + delta
        gx = 2 * c[..., 0] / (Wl - 1) - 1
        gy = 2 * c[..., 1] / (Hl - 1) - 1
        s = F.grid_sample(lvl[:, None], torch.stack([gx, gy], -1), align_corners=True)
        outs.append(s.view(B, h, w, -1))
    return torch.cat(outs, -1).permute(0, 3, 1, 2).contiguous()


def corr_weight_fast(g1, g2, target, depth, sigma):
    """model/PoseRefiner.py:342-345 with F.grid_sample (default align_corners=False)."""
    g1, g2, target = _t(g1), _t(g2), _t(target)
    B, Dn, H, W = g2.shape
    grid = torch.stack([2 * target[..., 0] / (W - 1) - 1, 2 * target[..., 1] / (H - 1) - 1], -1)
    warped = F.grid_sample(g2, grid, align_corners=False)
    s = (g1 * warped).sum(1)
    sig = float(np.asarray(sigma).reshape(-1)[0])
    return torch.exp(-torch.abs(1 - s) / sig) * (_t(depth)[:, 0] > 0).float()


def convex_upsample_fast(flow, mask, scale: int = 8):
    """model/CFNet.py:95-106 verbatim in structure (softmax + unfold)."""
    flow, mask = _t(flow), _t(mask)
    N_, _, H, W = flow.shape
    m = torch.softmax(mask.view(N_, 1, 9, scale, scale, H, W), dim=2)
    up = F.unfold(scale * flow, [3, 3], padding=1).view(N_, 2, 9, 1, 1, H, W)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N_, 2, scale * H, scale * W)


_FAST = {"corr_lookup": corr_lookup_fast, "corr_weight": corr_weight_fast, "convex_upsample": convex_upsample_fast}


# --------------------------------------------------------------------------------------------------
# a12: the loop                                                     model/PoseRefiner.py:239-365
# --------------------------------------------------------------------------------------------------
def refine(inp: dict, W: dict, outer: int = 3, inner: int = 8, optim_iters: int = 1,
           capture: bool = False, stage_timer=None, fast: bool = False, literal_legacy_pose: bool = True):
    """One batched refinement on STATIC synthetic renderings (the renderer is out of scope, so depth /
    ctx / g1 / fmaps are not re-rendered between outer iterations).
    inp: dict from rnnpose_amd.synthetic.make_inputs (fmap1,fmap2,ctx,g1,g2,depth,K,G0,sigma) --
    if it holds img_render/img_target and W holds 'enc' weights the encoder runs per outer iteration.
    literal_legacy_pose (default True since r06: it is what the reference computes): start every outer iteration from
    Tij = Ti * Ti.inv() as the reference's legacy branch literally does (model/PoseRefiner.py:243-244; geometry/se3.py:194-208:
    inverse = [R^T | -R^T t], product = 4x4 matmul); False: the exact identity that product stands for.
    Returns dict(G=(B,1,4,4) final Ti, flow_up=last, weight=last, trace=[per-iteration captures])."""
    import time
    tm = stage_timer if stage_timer is not None else {}
    lookup_fn = corr_lookup_fast if fast else corr_lookup
    weight_fn = corr_weight_fast if fast else corr_weight
    upsample_fn = convex_upsample_fast if fast else convex_upsample

    def tick(name, t0):
        tm[name] = tm.get(name, 0.0) + time.perf_counter() - t0

    B = inp["depth"].shape[0]
    Hh, Ww = inp["depth"].shape[-2:]
    Ti = _t(inp["G0"]).reshape(B, 1, 4, 4)
    Tij = torch.eye(4).repeat(B, 1, 1, 1)
    K = inp["K"]
    xs, ys = _pix_grid(Hh, Ww)
    grid = torch.stack([xs, ys], -1)[None]
    trace = []
    flow_up = wgt = None
    for _o in range(outer):
        Ti = torch.matmul(Tij, Ti)                                    # PoseRefiner.py:241
        Tij = torch.eye(4).repeat(B, 1, 1, 1)                         # :242
        if literal_legacy_pose:                                       # :243-244 (legacy): Ti * Ti.inv()
            Rt = Ti[..., :3, :3].transpose(-1, -2)
            Ginv = torch.cat([torch.cat([Rt, -torch.matmul(Rt, Ti[..., :3, 3:])], -1), Tij[..., 3:, :].to(Ti.dtype)], -2)
            Tij = torch.matmul(Ti, Ginv)
        t0 = time.perf_counter()
        if "img_render" in inp and "enc" in W:
            fmap1, fmap2 = image_encoder(W["enc"], inp["img_render"], inp["img_target"])
        else:
            fmap1, fmap2 = _t(inp["fmap1"]), _t(inp["fmap2"])
        tick("encoder", t0)
        for i in range(inner):
            t0 = time.perf_counter()
            flow_init, _ = induced_flow(inp["depth"], K, Tij)
            tick("induced_flow", t0)
            if i == 0:
                t0 = time.perf_counter()
                pyr = corr_pyramid(fmap1, fmap2)
                net, cinp = context_prep(inp["ctx"])
                tick("corr_build_ctx", t0)
            t0 = time.perf_counter()
            coords0 = coords_grid_lowres(B, Hh // 8, Ww // 8)
            coords1 = flow_init_to_coords1(flow_init)
            corr = lookup_fn(pyr, coords1)
            tick("lookup", t0)
            t0 = time.perf_counter()
            net, mask, dflow = update_block(W["upd"], net, cinp, corr, coords1 - coords0)
            tick("update_block", t0)
            t0 = time.perf_counter()
            coords1 = coords1 + dflow
            flow_up = upsample_fn(coords1 - coords0, mask)
            tick("upsample", t0)
            t0 = time.perf_counter()
            target = flow_up.permute(0, 2, 3, 1) + grid
            wgt = weight_fn(inp["g1"], inp["g2"], target, inp["depth"], inp["sigma"])
            tick("weight", t0)
            t0 = time.perf_counter()
            Tij, lm_trace = lm_step(target, wgt, inp["depth"], K, Tij, optim_iters)
            tick("lm", t0)
            if capture:
                trace.append(dict(corr=corr, net=net, dflow=dflow, mask=mask, flow_up=flow_up, weight=wgt,
                                  H=lm_trace[-1][0], b=lm_trace[-1][1], xi=lm_trace[-1][2], Tij=Tij.clone()))
    Ti = torch.matmul(Tij, Ti)                                        # PoseRefiner.py:365
    return dict(G=Ti, Tij=Tij, flow_up=flow_up, weight=wgt, trace=trace)
