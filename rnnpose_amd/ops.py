"""Torch-tensor front end of the C ABI (include/rnnpose_hip.h): device memory and streams come from
PyTorch-ROCm, every computation is a hand-written HIP kernel in librnnpose_hip.so.

All ops take CUDA(=HIP) tensors, enqueue on torch's current stream, never synchronise and never fall back
to a CPU or ATen implementation: a non-GPU tensor or a missing library raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
from contextlib import contextmanager

import torch

from . import _lib

F32 = torch.float32
F64 = torch.float64

# ---- optional per-op HIP-event timing (used by bench.py for the roofline object) -------------------------
_prof = None   # None or dict name -> list[(start_evt, end_evt)]


_event_pool = []        # timing events are reused across profile() contexts: creating ~1300 of them costs ~10 ms per step


@contextmanager
def profile(names=None):
    """Record HIP events around every op call (or only the ops in `names`) on the current stream.
    Yields a dict; after the context exits and the device is synchronised, call summarize()."""
    global _prof
    prev = _prof
    rec = {"_only": set(names) if names else None, "_next": 0}
    _prof = rec
    try:
        yield rec
    finally:
        _prof = prev


def profile_begin(names=None):
    """Non-context form of profile(): start recording, return the record dict; stop with profile_end()."""
    global _prof
    rec = {"_only": set(names) if names else None, "_next": 0, "_prev": _prof}
    _prof = rec
    return rec


def profile_end(rec):
    global _prof
    _prof = rec.get("_prev")


def profiling() -> bool:
    """True while a profile() context is recording (HIP events cannot be recorded into a captured graph)."""
    return _prof is not None


def summarize(rec) -> dict:
    """-> {name: (n_calls, mean_ms, total_ms, total_work, total_bytes)} (synchronises).  total_work / total_bytes = sums of
    the ALGORITHMIC flops / HBM bytes each launch declared for ITS OWN arguments (a half-batch launch declares half)."""
    torch.cuda.synchronize()
    out = {}
    for k, evs in rec.items():
        if k.startswith("_"):
            continue
        ms = [a.elapsed_time(b) for a, b, _, _ in evs]
        out[k] = (len(ms), sum(ms) / max(len(ms), 1), sum(ms), sum(w for _, _, w, _ in evs), sum(n for _, _, _, n in evs))
    return out


def _launch(name, *args, work=0, nbytes=0, products=3):
    """products: fp16 MFMA products per algorithmic multiply-add this launch executes (3 = fp16x3 split; 1 = a single-product
    strip launch) -- summed into rec["_exec"][name] as EXECUTED flops next to the algorithmic `work`."""
    rec = _prof
    if rec is not None and (rec["_only"] is None or name in rec["_only"]):
        ex = rec.setdefault("_exec", {})
        ex[name] = ex.get(name, 0.0) + products * work
        i = rec["_next"]
        rec["_next"] = i + 2
        while len(_event_pool) < i + 2:
            _event_pool.append(torch.cuda.Event(enable_timing=True))
        a, b = _event_pool[i], _event_pool[i + 1]
        a.record()
        _lib.call(name, *args)
        b.record()
        rec.setdefault(name, []).append((a, b, work, nbytes))
    else:
        _lib.call(name, *args)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, dtype=F32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); rnnpose_amd has no CPU path")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---- a1+a2 ---------------------------------------------------------------------------------------------
def pyramid_layout(B, h, w, levels=4):
    offs = (C.c_int64 * (levels + 1))()
    hl = (C.c_int * levels)()
    wl = (C.c_int * levels)()
    _lib.call("rnnpose_corr_pyramid_layout", B, h, w, levels, offs, hl, wl)
    return list(offs), list(hl), list(wl)


def pyramid_level0_dense(buf, B, h, w):
    """Level 0 of a pyramid buffer in the reference's shape (B*h*w, 1, h, w): the buffer holds it j-patch-major
    ([b][8 x 16 patch][i][8][16], include/rnnpose_hip.h); a COPY made with torch indexing -- for the CorrBlock facade's
    `corr_pyramid` attribute and for tests, never on the refinement path (1 GB at the headline shape)."""
    N, npy, npx = h * w, -(-h // 8), -(-w // 16)
    v = buf[:B * npy * npx * N * 128].view(B, npy, npx, N, 8, 16)
    return v.permute(0, 3, 1, 4, 2, 5).reshape(B * N, 1, npy * 8, npx * 16)[:, :, :h, :w].contiguous()


def pyramid_from_levels(levels):
    """Inverse of PyramidLevels: the reference-shaped list [(B*h*w, 1, h_l, w_l)] -> the flat buffer the lookup kernels read
    (level 0 re-blocked j-patch-major with zero padding).  For the dispatcher-level operator set (torch_ops), which carries the
    pyramid as Tensor[]; the refinement path never leaves the buffer form."""
    l0 = levels[0]
    BN, _, h, w = l0.shape
    N = h * w
    B = BN // N
    npy, npx = -(-h // 8), -(-w // 16)
    p = torch.nn.functional.pad(l0[:, 0], (0, npx * 16 - w, 0, npy * 8 - h))
    blocked = p.view(B, N, npy, 8, npx, 16).permute(0, 2, 4, 1, 3, 5).reshape(-1)
    return torch.cat([blocked] + [lv.reshape(-1) for lv in levels[1:]])


class PyramidLevels:
    """The levels of a pyramid buffer as the reference's list of (B*h*w, 1, h_l, w_l) tensors (thirdparty/raft/corr.py:23-34),
    materialised on access: levels 1.. are views of the buffer, level 0 is un-blocked into a copy (pyramid_level0_dense)."""

    def __init__(self, buf, offs, hl, wl, B, h, w):
        self.buf, self.offs, self.hl, self.wl, self.B, self.h, self.w = buf, offs, hl, wl, B, h, w

    def __len__(self):
        return len(self.hl)

    def __getitem__(self, l):
        if isinstance(l, slice):
            return [self[i] for i in range(*l.indices(len(self)))]
        l = range(len(self))[l]
        if l == 0:
            return pyramid_level0_dense(self.buf, self.B, self.h, self.w)
        return self.buf[self.offs[l]:self.offs[l + 1]].view(self.B * self.h * self.w, 1, self.hl[l], self.wl[l])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def _pyramid_bytes(B, Cc, h, w, levels, hl, wl):
    """algorithmic bytes of a volume build: both operands read once, every pyramid cell written once (fp32) --
    without the padding cells of level 0's border patches"""
    return 4.0 * (2 * B * h * w * Cc + B * h * w * sum(hl[l] * wl[l] for l in range(levels)))


def corr_pyramid(fmap1, fmap2, levels: int = 4, out=None, precision: str = "f32"):
    """fmap1,fmap2 (B,C,h,w) -> (flat buffer, [views (B*h*w,1,h_l,w_l)])   thirdparty/raft/corr.py:13-34.
    `out`: an existing flat buffer of the right size to overwrite (keeps the address stable for captured graphs).
    precision "f32": fp32 MFMA kernel on the NCHW maps; "f16x3": transpose to pixel-major + fp16x3-split kernel."""
    fmap1, fmap2 = _chk(fmap1, "fmap1"), _chk(fmap2, "fmap2")
    if fmap1.shape != fmap2.shape or fmap1.dim() != 4:
        raise ValueError("fmap1/fmap2 must both be (B,C,h,w)")
    B, Cc, h, w = fmap1.shape
    if precision == "f16x3":
        return _corr_pyramid_f16x3(fmap1, fmap2, 0, B, Cc, h, w, levels, out, A_SCALE)
    if precision != "f32":
        raise ValueError("precision must be 'f32' or 'f16x3'")
    offs, hl, wl = pyramid_layout(B, h, w, levels)
    buf = out if (out is not None and out.numel() == offs[-1] and out.device == fmap1.device) else \
        torch.empty(offs[-1], device=fmap1.device, dtype=F32)
    _launch("rnnpose_corr_pyramid_f32", _ptr(fmap1), _ptr(fmap2), B, Cc, h, w, levels, _ptr(buf), _stream(),
            work=2.0 * B * (h * w) ** 2 * Cc, nbytes=_pyramid_bytes(B, Cc, h, w, levels, hl, wl))
    return buf, PyramidLevels(buf, offs, hl, wl, B, h, w)


_pyr_ws = {}


CORR_VARIANT_DEFAULT = 0      # (the library's own default: csrc/corr_pyramid.hip g_corr_variant)


def corr_variant(variant: int):
    """Measurement switch (RNNPOSE_CORR_VARIANT): which fp16x3 volume kernel is launched -- 0 operands through registers + ds_write,
    1 operands by LDS-DMA (r06); bit-identical results."""
    _apply_conv_env()
    _lib.call("rnnpose_corr_variant", int(variant))


def _corr_pyramid_f16x3(f1, f2, layout, B, Cc, h, w, levels, out, a_scale):
    _apply_conv_env()
    n = int(_lib.load().rnnpose_corr_pyramid_f16x3_workspace_bytes(B, Cc, h, w))
    key = (f1.device, n)
    ws = _pyr_ws.get(key)
    if ws is None:
        range_guard_arm(f1.device)
        ws = _pyr_ws[key] = torch.empty(n // 2, device=f1.device, dtype=torch.float16)
    offs, hl, wl = pyramid_layout(B, h, w, levels)
    buf = out if (out is not None and out.numel() == offs[-1] and out.device == f1.device) else \
        torch.empty(offs[-1], device=f1.device, dtype=F32)
    _launch("rnnpose_corr_pyramid_f16x3", _ptr(f1), _ptr(f2), layout, B, Cc, h, w, levels, float(a_scale), _ptr(ws), n,
            _ptr(buf), _stream(), work=2.0 * B * (h * w) ** 2 * Cc, nbytes=_pyramid_bytes(B, Cc, h, w, levels, hl, wl))
    return buf, PyramidLevels(buf, offs, hl, wl, B, h, w)


def corr_pyramid_nhwc(f1, f2, levels: int = 4, out=None, a_scale: float = 8.0):
    """f1,f2 (B,h,w,C) pixel-major -> same (buffer, views) as corr_pyramid, fp16x3-split MFMA (fp32-class accuracy)."""
    f1, f2 = _nhwc(f1, "f1"), _nhwc(f2, "f2")
    if f1.shape != f2.shape:
        raise ValueError("f1/f2 must have the same (B,h,w,C) shape")
    B, h, w, Cc = f1.shape
    return _corr_pyramid_f16x3(f1, f2, 1, B, Cc, h, w, levels, out, a_scale)


class SplitTensor:
    """A (B,C,h,w) feature map held as a pixel-major split tensor (rnnpose_hip.h "SPLIT TENSORS": per 8-channel group fp16
    hi x 8 | fp16 lo x 8 of a_scale * x, in a float32-typed (B,h,w,C) buffer): what the encoder's output convolution writes
    for the volume build.  `shape` is the NCHW shape of the map it stands for; dense() converts back (tests, facade)."""

    def __init__(self, data, a_scale: float = 8.0):
        self.data = _nhwc(data, "split tensor")
        self.a_scale = float(a_scale)

    @property
    def shape(self):
        B, h, w, Cc = self.data.shape
        return torch.Size((B, Cc, h, w))

    @property
    def device(self):
        return self.data.device

    def float(self):
        return self

    def __getitem__(self, sl):
        if not isinstance(sl, slice):
            raise TypeError("SplitTensor supports batch slices only")
        return SplitTensor(self.data[sl], self.a_scale)

    def dense(self):
        return unsplit_hl(self.data, self.a_scale).permute(0, 3, 1, 2).contiguous()


def corr_pyramid_split(f1: SplitTensor, f2: SplitTensor, levels: int = 4, out=None):
    """Volume + pyramid from operands that are split tensors already (no pre-pass): (buffer, views) as corr_pyramid."""
    _apply_conv_env()
    if f1.shape != f2.shape or f1.a_scale != f2.a_scale:
        raise ValueError("f1/f2 must have the same shape and scale")
    B, Cc, h, w = f1.shape
    offs, hl, wl = pyramid_layout(B, h, w, levels)
    dev = f1.device
    buf = out if (out is not None and out.numel() == offs[-1] and out.device == dev) else torch.empty(offs[-1], device=dev, dtype=F32)
    _launch("rnnpose_corr_pyramid_split", _ptr(f1.data), _ptr(f2.data), B, Cc, h, w, levels, f1.a_scale, _ptr(buf), _stream(),
            work=2.0 * B * (h * w) ** 2 * Cc, nbytes=_pyramid_bytes(B, Cc, h, w, levels, hl, wl))
    return buf, PyramidLevels(buf, offs, hl, wl, B, h, w)


# ---- a3' (volume-free lookup: a measured alternative, thirdparty/raft/corr.py:70-98) ------------------------
def fmap_pyramid(f2, levels: int = 4):
    """f2 (B,h,w,C) pixel-major -> flat buffer holding levels 1..levels-1 of the 2x2-mean pyramid of the feature map."""
    f2 = _nhwc(f2, "f2")
    B, h, w, Cc = f2.shape
    n = int(_lib.load().rnnpose_fmap_pyramid_floats(B, h, w, Cc, levels))
    buf = torch.empty(max(n, 4), device=f2.device, dtype=F32)
    _launch("rnnpose_fmap_pyramid_f32", _ptr(f2), B, h, w, Cc, levels, _ptr(buf), _stream())
    return buf


def corr_alt_lookup(f1, f2, pooled, coords, levels: int = 4, radius: int = 4, out=None):
    """Window features computed on the fly (no volume): f1, f2 (B,h,w,C) pixel-major, pooled = fmap_pyramid(f2, levels),
    coords (B,2,h,w) -> (B,h,w,levels*81) pixel-major (the layout corr_lookup_nhwc writes)."""
    f1, f2 = _nhwc(f1, "f1"), _nhwc(f2, "f2")
    coords = _chk(coords, "coords")
    B, h, w, Cc = f1.shape
    if f2.shape != f1.shape or tuple(coords.shape) != (B, 2, h, w):
        raise ValueError("f1/f2 must share (B,h,w,C) and coords must be (B,2,h,w)")
    nch = levels * (2 * radius + 1) ** 2
    if out is None:
        out = torch.empty(B, h, w, nch, device=f1.device, dtype=F32)
    _launch("rnnpose_corr_alt_lookup_f32", _ptr(f1), _ptr(f2), _ptr(pooled), _ptr(coords), B, h, w, Cc, levels, radius, _ptr(out),
            out.shape[3], 0, _stream(), work=2.0 * B * h * w * levels * 100 * Cc)
    return out


# ---- a3 ------------------------------------------------------------------------------------------------
def _lookup_bytes(B, h, w, levels, radius):
    """per pixel and level a (2r+2)^2 texel footprint read + (2r+1)^2 outputs written, + 2 coords."""
    return 4.0 * B * h * w * (levels * (2 * radius + 2) ** 2 + levels * (2 * radius + 1) ** 2 + 2)


def corr_lookup(pyramid_buf, coords, levels: int = 4, radius: int = 4):
    """coords (B,2,h,w) -> (B, levels*81, h, w)                       thirdparty/raft/corr.py:36-57"""
    coords = _chk(coords, "coords")
    pyramid_buf = _chk(pyramid_buf, "pyramid")
    B, two, h, w = coords.shape
    if two != 2:
        raise ValueError("coords must be (B,2,h,w)")
    offs, _, _ = pyramid_layout(B, h, w, levels)
    if pyramid_buf.numel() != offs[-1]:
        raise ValueError("pyramid buffer does not match coords shape")
    out = torch.empty(B, levels * (2 * radius + 1) ** 2, h, w, device=coords.device, dtype=F32)
    _launch("rnnpose_corr_lookup_f32", _ptr(pyramid_buf), _ptr(coords), B, h, w, levels, radius, _ptr(out), _stream(),
            nbytes=_lookup_bytes(B, h, w, levels, radius))
    return out


# ---- a5 ------------------------------------------------------------------------------------------------
def context_prep(ctx, h, w, hdim: int = 128):
    ctx = _chk(ctx, "context_fea")
    B, Cc, H, W = ctx.shape
    net = torch.empty(B, hdim, h, w, device=ctx.device, dtype=F32)
    inp = torch.empty(B, Cc - hdim, h, w, device=ctx.device, dtype=F32)
    _launch("rnnpose_context_prep_f32", _ptr(ctx), B, Cc, H, W, h, w, hdim, _ptr(net), _ptr(inp), _stream(),
            nbytes=4.0 * B * Cc * h * w * 5)        # 4 taps read + 1 value written per low-res element
    return net, inp


def flow_to_coords(flow_init, h, w):
    flow_init = _chk(flow_init, "flow_init")
    B, two, H, W = flow_init.shape
    out = torch.empty(B, 2, h, w, device=flow_init.device, dtype=F32)
    _launch("rnnpose_flow_to_coords_f32", _ptr(flow_init), B, H, W, h, w, _ptr(out), _stream())
    return out


# ---- a6 ------------------------------------------------------------------------------------------------
def convex_upsample(flow, mask, scale: int = 8):
    flow, mask = _chk(flow, "flow"), _chk(mask, "mask")
    B, _, h, w = flow.shape
    if mask.shape != (B, 9 * scale * scale, h, w):
        raise ValueError(f"mask must be (B,{9*scale*scale},h,w)")
    out = torch.empty(B, 2, scale * h, scale * w, device=flow.device, dtype=F32)
    _launch("rnnpose_convex_upsample_f32", _ptr(flow), _ptr(mask), B, h, w, scale, _ptr(out), _stream(),
            nbytes=4.0 * B * h * w * (9 * scale * scale + 2 + 2 * scale * scale))
    return out


# ---- a7 ------------------------------------------------------------------------------------------------
def induced_flow(depth, K, G, eps: float = 1e-5, want_vmask: bool = True, absolute: bool = False):
    """depth (B,1,H,W) raw, K (B,3,3), G (B,[1,]4,4) -> flow (B,2,H,W), vmask (B,H,W) or None.
    absolute=True returns the raw re-projected coordinates (SE3.transform) instead of the masked flow."""
    depth, K, G = _chk(depth, "depth"), _chk(K, "intrinsics"), _chk(G, "G")
    B, H, W = depth.shape[0], depth.shape[-2], depth.shape[-1]
    flow = torch.empty(B, 2, H, W, device=depth.device, dtype=F32)
    vmask = torch.empty(B, H, W, device=depth.device, dtype=F32) if want_vmask else None
    _launch("rnnpose_induced_flow_f32", _ptr(depth), _ptr(K), _ptr(G), B, H, W, eps, int(absolute), _ptr(flow), _ptr(vmask),
            _stream())
    return flow, vmask


def induced_coords_lowres(depth, K, G, h, w, eps: float = 1e-5, out=None):
    depth, K, G = _chk(depth, "depth"), _chk(K, "intrinsics"), _chk(G, "G")
    B, H, W = depth.shape[0], depth.shape[-2], depth.shape[-1]
    if out is None:
        out = torch.empty(B, 2, h, w, device=depth.device, dtype=F32)
    _launch("rnnpose_induced_coords_lowres_f32", _ptr(depth), _ptr(K), _ptr(G), B, H, W, h, w, eps, _ptr(out), _stream())
    return out


# ---- a8 ------------------------------------------------------------------------------------------------
def _target_mode(target, H, W):
    if target.dim() >= 3 and target.shape[-1] == 2 and target.shape[-3:-1] == (H, W):
        return 0      # (B,[K,]H,W,2) absolute coordinates
    if target.dim() == 4 and target.shape[1] == 2 and target.shape[-2:] == (H, W):
        return 1      # (B,2,H,W) planar flow, grid added in-kernel
    raise ValueError(f"target has shape {tuple(target.shape)}; expected (B,H,W,2) or (B,2,H,W)")


def corr_weight(g1, g2, target, depth, sigma, out=None):
    g1, g2, target, depth = _chk(g1, "g1"), _chk(g2, "g2"), _chk(target, "target"), _chk(depth, "depth")
    sigma = _chk(sigma.reshape(-1)[:1], "sigma")
    B, D, H, W = g1.shape
    mode = _target_mode(target, H, W)
    if out is None:
        out = torch.empty(B, H, W, device=g1.device, dtype=F32)
    _launch("rnnpose_corr_weight_f32", _ptr(g1), _ptr(g2), _ptr(target), mode, _ptr(depth), _ptr(sigma), B, D, H, W,
            _ptr(out), _stream(), nbytes=4.0 * B * H * W * (2 * D + 2 + 1 + 1))   # g1, g2 (unique texels), target, depth, w
    return out


# ---- a9-a11 --------------------------------------------------------------------------------------------
_ws_cache = {}


def _workspace(B, H, W, device, slot=0):
    """slot: calls that may run concurrently on different streams (the two batch halves) need distinct workspaces."""
    n = int(_lib.load().rnnpose_lm_workspace_bytes(B, H, W))
    # keyed on the SHAPE, not on the byte count: the first B doubles are the arrival counters of the fused LM step (zero between
    # launches), the partial records follow -- two shapes of equal size would find each other's partial sums where their counters
    # should be zero and the fused tail would never fire
    key = (device, int(B), int(H), int(W), slot)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.zeros(n // 8, device=device, dtype=F64)      # (zero: the arrival counters live in the first B doubles)
        _ws_cache[key] = ws
    return ws, n


def lm_normal_eq(target, weight, depth, K, G, eps: float = 1e-5):
    """-> Hm (B,6,6) fp64 (undamped), bv (B,6) fp64             geometry/transformation.py:286-297"""
    target, weight, depth = _chk(target, "target"), _chk(weight, "weight"), _chk(depth, "depth")
    K, G = _chk(K, "intrinsics"), _chk(G, "G")
    B, H, W = depth.shape[0], depth.shape[-2], depth.shape[-1]
    mode = _target_mode(target, H, W)
    ws, n = _workspace(B, H, W, depth.device)
    Hm = torch.empty(B, 6, 6, device=depth.device, dtype=F64)
    bv = torch.empty(B, 6, device=depth.device, dtype=F64)
    _launch("rnnpose_lm_normal_eq_f64", _ptr(target), mode, _ptr(weight), _ptr(depth), eps, _ptr(K), _ptr(G), B, H, W,
            _ptr(ws), n, _ptr(Hm), _ptr(bv), _stream(), nbytes=16.0 * B * H * W, work=200.0 * B * H * W)
    return Hm, bv


def lm_solve_update(Hm, bv, G, ep_lambda=100.0, lm_lambda=1e-4, max_update=1.0):
    """-> G_new (B,4,4), xi (B,6) fp32, info (B,) int32"""
    Hm, bv, G = _chk(Hm, "H", F64), _chk(bv, "b", F64), _chk(G, "G")
    B = Hm.shape[0]
    G_new = torch.empty(B, 4, 4, device=G.device, dtype=F32)
    xi = torch.empty(B, 6, device=G.device, dtype=F32)
    info = torch.empty(B, device=G.device, dtype=torch.int32)
    _launch("rnnpose_lm_solve_update_f32", _ptr(Hm), _ptr(bv), _ptr(G), B, float(ep_lambda), float(lm_lambda),
            float(max_update), _ptr(G_new), _ptr(xi), _ptr(info), _stream())
    return G_new, xi, info


def lm_step(target, weight, depth, K, G, num_iters=1, ep_lambda=100.0, lm_lambda=1e-4, max_update=1.0, eps=1e-5, out=None,
            slot=0):
    """num_iters fused GN steps; returns (G_new (B,4,4), Hm, bv, xi, info) of the last iteration.
    out: optional preallocated (G_new, Hm, bv, xi, info) views to write into; slot: workspace slot (see _workspace)."""
    _apply_lm_env()
    target, weight, depth = _chk(target, "target"), _chk(weight, "weight"), _chk(depth, "depth")
    K = _chk(K, "intrinsics")
    B, H, W = depth.shape[0], depth.shape[-2], depth.shape[-1]
    mode = _target_mode(target, H, W)
    ws, n = _workspace(B, H, W, depth.device, slot)
    if out is not None and int(num_iters) >= 1:
        Gd, Hm, bv, xi, info = out          # (every step writes all five outputs for every image)
        Gin = _chk(G, "G").reshape(-1, 4, 4)
        if not Gin.is_contiguous():
            Gin = Gin.contiguous()
        _launch("rnnpose_lm_step_io_f32", _ptr(target), mode, _ptr(weight), _ptr(depth), eps, _ptr(K), _ptr(Gin), _ptr(Gd), B, H, W,
                int(num_iters), float(ep_lambda), float(lm_lambda), float(max_update), _ptr(ws), n, _ptr(Hm), _ptr(bv),
                _ptr(xi), _ptr(info), _stream(), nbytes=16.0 * B * H * W * int(num_iters), work=200.0 * B * H * W * int(num_iters))
        return Gd, Hm, bv, xi, info
    if out is not None:
        Gd, Hm, bv, xi, info = out
        Gd.copy_(_chk(G, "G").reshape(-1, 4, 4))
        G = Gd
        xi.zero_()
        info.zero_()
    else:
        G = _chk(G, "G").reshape(-1, 4, 4).clone()
        Hm = torch.empty(B, 6, 6, device=depth.device, dtype=F64)
        bv = torch.empty(B, 6, device=depth.device, dtype=F64)
        xi = torch.zeros(B, 6, device=depth.device, dtype=F32)
        info = torch.zeros(B, device=depth.device, dtype=torch.int32)
    _launch("rnnpose_lm_step_f32", _ptr(target), mode, _ptr(weight), _ptr(depth), eps, _ptr(K), _ptr(G), B, H, W,
            int(num_iters), float(ep_lambda), float(lm_lambda), float(max_update), _ptr(ws), n, _ptr(Hm), _ptr(bv),
            _ptr(xi), _ptr(info), _stream(), nbytes=16.0 * B * H * W * int(num_iters), work=200.0 * B * H * W * int(num_iters))
    return G, Hm, bv, xi, info


_lm_fused_env_applied = False


def _apply_lm_env():
    """RNNPOSE_LM_FUSED=0/1 selects the LM step form at first use (same-box A/B); default: the library's."""
    global _lm_fused_env_applied
    if not _lm_fused_env_applied:
        _lm_fused_env_applied = True
        v = _os.environ.get("RNNPOSE_LM_FUSED")
        if v is not None:
            _lib.call("rnnpose_lm_fused_tail", int(v != "0"))


def lm_fused_tail(enable: bool = True):
    """One launch per LM step (default) or the r02 three-launch form (measurement switch; drop captured graphs when toggling)."""
    _lib.call("rnnpose_lm_fused_tail", int(bool(enable)))


def se3_exp(xi):
    xi = _chk(xi, "xi")
    sh = xi.shape[:-1]
    x = xi.reshape(-1, 6)
    out = torch.empty(x.shape[0], 4, 4, device=xi.device, dtype=F32)
    _launch("rnnpose_se3_exp_f32", _ptr(x), x.shape[0], _ptr(out), _stream())
    return out.reshape(*sh, 4, 4)


def se3_compose(A, Bm):
    A, Bm = _chk(A, "A"), _chk(Bm, "B")
    if A.shape != Bm.shape:
        raise ValueError("se3_compose needs equal shapes")
    out = torch.empty_like(A)
    _launch("rnnpose_se3_compose_f32", _ptr(A), _ptr(Bm), A.numel() // 16, _ptr(out), _stream())
    return out


def se3_outer_update(Tij, Ti, literal: bool = True):
    """-> (Ti_new, Tij_new) = (Tij Ti, literal ? Ti_new Ti_new^-1 : I): the pose bookkeeping between two outer iterations
    (model/PoseRefiner.py:241-244) as one launch; bit-identical to se3_compose / se3_inverse / se3_compose."""
    Tij, Ti = _chk(Tij, "Tij"), _chk(Ti, "Ti")
    if Tij.shape != Ti.shape:
        raise ValueError("se3_outer_update needs equal shapes")
    Ti_new, Tij_new = torch.empty_like(Ti), torch.empty_like(Tij)
    _launch("rnnpose_se3_outer_update_f32", _ptr(Tij), _ptr(Ti), Ti.numel() // 16, int(bool(literal)), _ptr(Ti_new), _ptr(Tij_new), _stream())
    return Ti_new, Tij_new


def se3_inverse(A):
    A = _chk(A, "A")
    out = torch.empty_like(A)
    _launch("rnnpose_se3_inverse_f32", _ptr(A), A.numel() // 16, _ptr(out), _stream())
    return out


# ---- a4 pointwise ---------------------------------------------------------------------------------------
def gru_gate(zr, hcat, z_out, rhx, C: int = 128):
    """zr (B,2C,h,w) pre-activations; hcat (B,Ctot,h,w) with h in [:C]; writes z_out (B,C,h,w), rhx[:, :C]."""
    B, Ctot = hcat.shape[0], hcat.shape[1]
    hw = hcat.shape[2] * hcat.shape[3]
    _launch("rnnpose_gru_gate_f32", _ptr(zr), _ptr(hcat), B, C, Ctot, hw, _ptr(z_out), _ptr(rhx), _stream())


def gru_update(z, q_pre, hcat, hout, C: int = 128):
    """hout[:, :C] = (1-z)*hcat[:, :C] + z*tanh(q_pre)  (hout may be hcat itself)."""
    B = hcat.shape[0]
    hw = hcat.shape[2] * hcat.shape[3]
    _launch("rnnpose_gru_update_f32", _ptr(z), _ptr(q_pre), _ptr(hcat), B, C, hcat.shape[1], hw, _ptr(hout), hout.shape[1],
            _stream())


# ---- a4 dense convolutions: NHWC implicit GEMM on the fp16 matrix cores with fp16x3 split (fp32-class accuracy) ----
EPI_LINEAR, EPI_RELU, EPI_GRU_ZR, EPI_GRU_Q = 0, 1, 2, 3
# Activation scale of the fp16x3 split (x * A_SCALE = hi + lo in fp16).  Range: |x| <= 65504 / A_SCALE = 8188; full 21-22
# bit precision while lo stays a normal fp16 (|x| >= 2^-3 / A_SCALE = 0.016), below that the ABSOLUTE error stays under
# 2^-25 / A_SCALE = 4e-9 -- fp32-class for O(1) activations either way.  (r01 used 64: range 1023, too tight for
# real correlation magnitudes; saturation is counted by the range guard, ops.saturation_check.)
A_SCALE = 8.0


class PackedConv:
    """Weights of one convolution, split into fp16 hi/lo and laid out for csrc/conv_igemm.hip.
    weight (Cout,Cin,kh,kw), bias (Cout,), seg_counts = channel counts of the virtual-concat sources (sum = Cin).
    post_scale is folded into weight and bias (the 0.25 of the mask head, update.py:187)."""

    def __init__(self, weight, bias, seg_counts=None, post_scale: float = 1.0, a_scale: float = A_SCALE):
        import math
        range_guard_arm(weight.device)
        w = _chk(weight.detach(), "weight") * post_scale
        b = _chk(bias.detach(), "bias") * post_scale
        self.c_out, self.c_in, self.kh, self.kw = w.shape
        self.c_in_real = self.c_in          # engines that zero-pad input channels overwrite this for flop accounting
        self.seg_counts = [int(c) for c in (seg_counts or [self.c_in])]
        if sum(self.seg_counts) != self.c_in:
            raise ValueError("seg_counts must sum to the input channels")
        wmax = float(w.abs().max())
        self.w_scale = float(2.0 ** math.floor(math.log2(1024.0 / wmax))) if wmax > 0 else 1.0
        self.a_scale = float(a_scale)
        segs = (C.c_int * len(self.seg_counts))(*self.seg_counts)
        n = int(_lib.load().rnnpose_conv_packed_halfs(self.c_out, self.kh, self.kw, segs, len(self.seg_counts)))
        if n <= 0:
            raise ValueError("unsupported convolution shape")
        self.w_packed = torch.empty(n, device=w.device, dtype=torch.float16)
        self.bias = b.contiguous()
        _lib.call("rnnpose_conv_pack_weights_f16x3", _ptr(w), self.c_out, self.c_in, self.kh, self.kw, segs,
                  len(self.seg_counts), self.w_scale, _ptr(self.w_packed), _stream())


def _nhwc(t, name):
    if not (t.is_cuda and t.dtype == F32 and t.is_contiguous() and t.dim() == 4):
        raise ValueError(f"{name} must be a contiguous fp32 CUDA tensor shaped (B,H,W,C)")
    return t


def conv2d_nhwc(pc: PackedConv, srcs, dst, epilogue: int = EPI_LINEAR, aux0=None, aux1=None, dst2=None, gru_c: int = 0,
                stride: int = 1, tile_stats=None, add_map=None, in_norm=None, src_hl: bool = False, dst_hl: bool = False,
                dst2_hl: bool = False, dst_split=None, tile: int = 0, src_bounded: bool = False, ksplit_ws=None,
                single_product=None):
    """srcs: list of (tensor (B,H,W,C), c_offset) matched with pc.seg_counts; dst/aux0/aux1/dst2: (tensor, c_offset).
    ksplit_ws: a conv_ksplit_workspace() buffer -- lets a launch of few tiles (B = 1 crops) split its K loop over several
    workgroups per tile; launches that may run concurrently (two streams) need separate buffers.
    Writes in place into dst (and dst2); returns nothing.
    src_hl / dst_hl / dst2_hl: the sources / dst / dst2 are SPLIT tensors (fp16 hi|lo per 8-channel group in a buffer of the
    fp32 tensor's shape: include/rnnpose_hip.h; split_hl / unsplit_hl convert); dst_split = (tensor, c_offset): an additional
    split-form copy of the primary result; tile: 0 auto, 1..4 tile shapes of the 128-row kernels, 5 / 6 / 7 the strip kernels with 160- / 32- / 96-row strips
    (160-row strips, operands by LDS-DMA: include/rnnpose_hip.h)."""
    _apply_conv_env()
    d = _lib.ConvDesc()
    if len(srcs) != len(pc.seg_counts):
        raise ValueError("number of sources differs from the packed segment list")
    B, H, W, _ = srcs[0][0].shape
    for i, ((t, off), cnt) in enumerate(zip(srcs, pc.seg_counts)):
        _nhwc(t, f"src{i}")
        if t.shape[:3] != (B, H, W):
            raise ValueError("sources must share (B,H,W)")
        d.src[i] = _lib.ConvSrc(t.data_ptr(), t.shape[3], off, cnt)
    d.n_src = len(srcs)
    d.B, d.H, d.W, d.kh, d.kw, d.stride = B, H, W, pc.kh, pc.kw, stride
    d.w_packed, d.bias = pc.w_packed.data_ptr(), pc.bias.data_ptr()
    d.c_out, d.a_scale, d.w_scale, d.epilogue = pc.c_out, pc.a_scale, pc.w_scale, epilogue

    def put(prefix, spec):
        if spec is None:
            return
        t, off = spec
        _nhwc(t, prefix)
        setattr(d, prefix, t.data_ptr())
        setattr(d, prefix + "_c_stride", t.shape[3])
        setattr(d, prefix + "_c_offset", off)

    put("dst", dst)
    put("aux0", aux0)
    put("aux1", aux1)
    put("dst2", dst2)
    if add_map is not None:          # (tensor (B,H,W,Cs), c_offset): per-pixel bias added before the epilogue
        t, off = add_map
        _nhwc(t, "add_map")
        d.add_map, d.add_c_stride, d.add_c_offset = t.data_ptr(), t.shape[3], off
    d.gru_c = gru_c
    if in_norm is not None:          # (B, C_src, 2) mean / rstd of source 0: read relu((x - mean) * rstd) instead of x
        if not (in_norm.is_cuda and in_norm.dtype == F32 and in_norm.is_contiguous() and tuple(in_norm.shape) == (B, srcs[0][0].shape[3], 2)):
            raise ValueError("in_norm must be a contiguous fp32 CUDA tensor of (B, C_source, 2)")
        d.src0_mean_rstd = in_norm.data_ptr()
    d.src_hl, d.dst_hl, d.dst2_hl, d.tile = int(bool(src_hl)), int(bool(dst_hl)), int(bool(dst2_hl)), int(tile)
    if tile_stats is not None:
        # (B * tiles per image, c_out, 2) fp64 per-tile column sums / sums of squares for instnorm_tiles_nhwc.  The tile count is the one
        # of the kernel THIS launch takes: asked from the library with the descriptor itself (the launch checks it again)
        tpi = int(_lib.load().rnnpose_conv_tiles_per_image_desc(C.byref(d)))
        if not (tile_stats.is_cuda and tile_stats.dtype == F64 and tile_stats.is_contiguous()
                and tile_stats.numel() == B * tpi * pc.c_out * 2):      # exact: the consumer takes the tiling from this shape
            raise ValueError("tile_stats must be a contiguous FP64 CUDA tensor of (B * conv_tiles_per_image(H, W, kh, kw, stride, c_out, tile, B, "
                             f"src_counts, fused_norm) = {B * tpi}, c_out, 2): the tile count depends on the kernel the launch takes")
        d.tile_stats, d.tile_stats_records = tile_stats.data_ptr(), B * tpi
    d.src_bounded = int(bool(src_bounded))      # caller's guarantee that the sources cannot leave the fp16x3 range: no range check
    if dst_split is not None:
        t, off = dst_split
        _nhwc(t, "dst_split")
        d.dst_split, d.dst_split_c_stride, d.dst_split_c_offset = t.data_ptr(), t.shape[3], off
    if ksplit_ws is not None:
        d.ksplit_ws, d.ksplit_ws_bytes = ksplit_ws.data_ptr(), ksplit_ws.numel() * ksplit_ws.element_size()
    d.single_product = int(_single_product if single_product is None else bool(single_product))
    Ho, Wo = -(-H // stride), -(-W // stride)
    # algorithmic bytes of the launch: every input element and weight read once, every output written once, the
    # epilogue's operands (additive map; h of the gate, h and z of the state update) read once -- 4 bytes each
    nb = B * H * W * pc.c_in_real + pc.c_out * pc.c_in_real * pc.kh * pc.kw + B * Ho * Wo * pc.c_out
    nb += B * Ho * Wo * pc.c_out * (add_map is not None) + B * Ho * Wo * (gru_c if epilogue == EPI_GRU_ZR else 0)
    nb += 2 * B * Ho * Wo * pc.c_out * (epilogue == EPI_GRU_Q) + B * Ho * Wo * pc.c_out * (dst_split is not None)
    products = 3
    if d.single_product and _prof is not None:       # (flop accounting only: which launches the library runs single-product)
        products = int(_lib.load().rnnpose_conv_products_desc(C.byref(d)))
    _launch("rnnpose_conv2d_nhwc_f16x3", C.byref(d), _stream(),
            work=2.0 * B * Ho * Wo * pc.c_out * pc.c_in_real * pc.kh * pc.kw, nbytes=4.0 * nb, products=products)


_single_product = False


def single_product(enable=None) -> bool:
    """Process-wide default of conv2d_nhwc's `single_product` for callers that do not pass it (measurement scripts).  The package itself
    never sets it: since r05 the mode is carried by the engines a PoseRefiner configures (engine.UpdateEngine / EncoderEngine
    .single_product) and passed explicitly with every launch.  -> the current value."""
    global _single_product
    if enable is not None:
        _single_product = bool(enable)
    return _single_product


def conv_tiles_per_image(H, W, kh, kw, stride=1, c_out=None, tile: int = 0, batch: int = 1, src_counts=None, fused_norm: bool = False) -> int:
    """Records per image of a convolution's tile_stats.  128-row kernels: 3x3 stride 1 on 8 x 16 patches, else runs of 128
    output pixels; with c_out given: for the kernel a launch of `batch` images of that width and `tile` request takes -- the strip
    kernels (csrc/conv_strip*.hip, tile=5 / 6 / 7 or the automatic choice) tile an image into 10 / 2 / 6 x 16 patches or runs of 160 / 32 / 96 pixels.
    src_counts (channel counts of the sources) and fused_norm (the launch passes in_norm) complete the description: sources that are
    not whole 32-channel blocks send an automatic launch to the 128-row kernels (rnnpose_conv_tiles_per_image_desc)."""
    _apply_conv_env()
    if c_out is None:
        return int(_lib.load().rnnpose_conv_tiles_per_image(int(H), int(W), int(kh), int(kw), int(stride)))
    if src_counts is not None:
        d = _lib.ConvDesc()
        for i, c in enumerate(src_counts):
            d.src[i] = _lib.ConvSrc(None, int(c), 0, int(c))
        d.n_src = len(src_counts)
        d.B, d.H, d.W, d.kh, d.kw, d.stride, d.c_out, d.tile = int(batch), int(H), int(W), int(kh), int(kw), int(stride), int(c_out), int(tile)
        d.src0_mean_rstd = 16 if fused_norm else None        # (only its null-ness is looked at)
        return int(_lib.load().rnnpose_conv_tiles_per_image_desc(C.byref(d)))
    return int(_lib.load().rnnpose_conv_tiles_per_image_ex(int(H), int(W), int(kh), int(kw), int(stride), int(c_out), int(tile), int(batch)))


_conv_env_applied = False


def _apply_conv_env():
    """RNNPOSE_SPATIAL_TILES=0 restores the row-major tiling of 3x3 layers (same-box A/B)."""
    global _conv_env_applied
    if not _conv_env_applied:
        _conv_env_applied = True
        v = _os.environ.get("RNNPOSE_SPATIAL_TILES")
        if v is not None:
            _lib.call("rnnpose_conv_spatial_tiles", int(v != "0"))
        v = _os.environ.get("RNNPOSE_KSPLIT")
        if v is not None:
            _lib.call("rnnpose_conv_ksplit", int(v != "0"))
        v = _os.environ.get("RNNPOSE_LOOKUP_VARIANT")      # 0: the r01-r05 window-lookup kernel (same-box A/B; bit-identical results)
        if v is not None:
            _lib.call("rnnpose_corr_lookup_variant", int(v != "0"))
        v = _os.environ.get("RNNPOSE_STRIP")               # 0: the automatic tile choice never takes the strip kernels (same-box A/B);
        if v is not None:                                  # 2 / 3: strips with one / two column tiles per wave only
            _lib.call("rnnpose_conv_strip", int(v))
        v = _os.environ.get("RNNPOSE_CORR_VARIANT")           # 0 / 1: volume kernel with register-staged / LDS-DMA operands (same-box A/B)
        if v is not None:
            _lib.call("rnnpose_corr_variant", int(v))
        v = _os.environ.get("RNNPOSE_KSPLIT_LIMITS")          # "max_tiles,max_splits" (measurement)
        if v:
            a, b = (int(t) for t in v.split(","))
            _lib.call("rnnpose_conv_ksplit_limits", a, b)


def conv_spatial_tiles(enable: bool = True):
    _lib.call("rnnpose_conv_spatial_tiles", int(bool(enable)))


def conv_strip(mode=True):
    """Measurement switch (RNNPOSE_STRIP): False / 0 = the automatic tile choice never takes the strip kernels; 2 = not the two-wave
    workgroups of 64-channel layers; 3 = two 32-column tiles per wave; 4 = 160-row strips only (no 32-row strips); 5 = automatic
    without the stride-2 form (stride-2 3x3 layers then run the 128-row kernel's tap-per-staging mode, as until r04); 6 = automatic
    without r06's 32- / 96-row strips for launches of few waves; 7 = automatic WITH persistent launches of the fp32-source strip forms
    (r06, measured slower: off by default)."""
    _apply_conv_env()
    _lib.call("rnnpose_conv_strip", int(mode))


def conv_ksplit(enable: bool = True):
    """Measurement switch: False = launches never split K even when a workspace is passed (RNNPOSE_KSPLIT=0)."""
    _lib.call("rnnpose_conv_ksplit", int(bool(enable)))


def conv_ksplit_workspace(device):
    """Zeroed workspace for conv2d_nhwc(ksplit_ws=...): one per chain of launches that can run concurrently with another."""
    n = int(_lib.load().rnnpose_conv_ksplit_workspace_bytes())
    return torch.zeros((n + 3) // 4, device=device, dtype=torch.int32)


_nchw_pc = {}


def conv2d_nchw(weight, bias, x, relu: bool = False):
    """NCHW facade of the implicit-GEMM kernel: F.conv2d(x, weight, bias, stride=1, padding=k//2) [+ ReLU] for the odd
    "same" convolutions of the update block (thirdparty/raft/update.py).  Layout transposes on both sides, channel counts
    padded to multiples of 4; packed weights are cached per parameter identity.  The fused NHWC engine does not use this."""
    x = _chk(x, "x")
    B, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = weight.shape
    if Cin_w != Cin:
        raise ValueError(f"weight expects {Cin_w} input channels, got {Cin}")
    cin_p, cout_p = -(-Cin // 4) * 4, -(-Cout // 4) * 4
    # the entry keeps the parameter tensors alive: a freed module's storage could otherwise be handed to a new parameter of the
    # same shape at the same address and version, and the facade would compute with the previous module's packed weights
    key = (weight.data_ptr(), weight._version, bias.data_ptr(), bias._version, tuple(weight.shape))
    ent = _nchw_pc.get(key)
    pc = ent[0] if (ent is not None and ent[1] is weight and ent[2] is bias) else None
    if pc is None:
        w = _chk(weight.detach(), "weight")
        if cin_p != Cin:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin_p - Cin))
        pc = PackedConv(w, bias, [cin_p])
        pc.c_in_real = Cin
        if len(_nchw_pc) >= 64:
            _nchw_pc.pop(next(iter(_nchw_pc)))
        _nchw_pc[key] = (pc, weight, bias)
    xn = (torch.zeros if cin_p != Cin else torch.empty)(B, H, W, cin_p, device=x.device, dtype=F32)
    nchw_to_nhwc(x, xn)
    out = torch.empty(B, H, W, cout_p, device=x.device, dtype=F32)
    conv2d_nhwc(pc, [(xn, 0)], (out, 0), EPI_RELU if relu else EPI_LINEAR)
    return nhwc_to_nchw(out, 0, Cout)


# ---- NHWC companions ------------------------------------------------------------------------------------------
def corr_lookup_nhwc(pyramid_buf, coords, out=None, levels: int = 4, radius: int = 4):
    """coords (B,2,h,w) -> (B,h,w,levels*81)"""
    coords = _chk(coords, "coords")
    B, _, h, w = coords.shape
    if out is None:
        out = torch.empty(B, h, w, levels * (2 * radius + 1) ** 2, device=coords.device, dtype=F32)
    _launch("rnnpose_corr_lookup_nhwc_f32", _ptr(pyramid_buf), _ptr(coords), B, h, w, levels, radius, _ptr(out), _stream(),
            nbytes=_lookup_bytes(B, h, w, levels, radius))
    return out


def corr_lookup_nhwc_part(pyramid_buf, coords, out, B, b0, b1, levels: int = 4, radius: int = 4):
    """NHWC lookup of images [b0, b1) of a batch-B pyramid: coords (b1-b0,2,h,w) and out (b1-b0,h,w,levels*81) are the
    SUB-BATCH tensors."""
    coords = _chk(coords, "coords")
    n, _, h, w = coords.shape
    if n != b1 - b0:
        raise ValueError("coords must hold the b1 - b0 images of the part")
    _launch("rnnpose_corr_lookup_nhwc_part_f32", _ptr(pyramid_buf), _ptr(coords), B, b0, b1, h, w, levels, radius, _ptr(out),
            _stream(), nbytes=_lookup_bytes(b1 - b0, h, w, levels, radius))
    return out


class InducedCoords:
    """Where an iteration's coords1 comes from when its first consumers form it themselves (r06): depth (n,1,H,W), K (n,3,3), G (n,4,4) of
    the chain's images, the low-resolution size, and `out` (n,2,h,w) = the tensor the lookup launch writes the coordinates to (the flow
    head reads them there).  Same values as induced_coords_lowres(depth, K, G, h, w, eps), bit for bit (csrc/induced.cuh)."""

    def __init__(self, depth, K, G, h, w, eps, out):
        self.depth, self.K, self.G = _chk(depth, "depth"), _chk(K, "K"), _chk(G, "G")
        self.h, self.w, self.eps, self.out = int(h), int(w), float(eps), out
        self.H, self.W = int(depth.shape[-2]), int(depth.shape[-1])
        n = depth.shape[0]
        if tuple(K.shape[-2:]) != (3, 3) or tuple(G.shape[-2:]) != (4, 4) or K.reshape(-1, 3, 3).shape[0] != n or G.reshape(-1, 4, 4).shape[0] != n:
            raise ValueError("K (n,3,3) and G (n,4,4) must match depth (n,1,H,W)")
        if tuple(out.shape) != (n, 2, self.h, self.w) or not out.is_contiguous():
            raise ValueError("out must be a contiguous (n,2,h,w) tensor")


def corr_lookup_induced_nhwc_part(pyramid_buf, ic: InducedCoords, out, B, b0, b1, levels: int = 4, radius: int = 4):
    """corr_lookup_nhwc_part with the coordinates formed inside the launch (and written to ic.out)."""
    if ic.depth.shape[0] != b1 - b0:
        raise ValueError("the coordinate source must hold the b1 - b0 images of the part")
    _launch("rnnpose_corr_lookup_induced_nhwc_part_f32", _ptr(pyramid_buf), _ptr(ic.depth), _ptr(ic.K), _ptr(ic.G), ic.H, ic.W, ic.eps, B, b0, b1,
            ic.h, ic.w, levels, radius, _ptr(ic.out), _ptr(out), _stream(), nbytes=_lookup_bytes(b1 - b0, ic.h, ic.w, levels, radius))
    return out


def flow_features_induced(ic: InducedCoords, w_t, bias, out, motion, motion_c_offset, out_c_offset=0, out_split: bool = False,
                          motion_split: bool = False, a_scale: float = 8.0):
    """flow_features with coords1 formed inside the launch (grid subtracted)."""
    B, c_out = ic.depth.shape[0], w_t.shape[1]
    _launch("rnnpose_flow_features_induced_f32", _ptr(ic.depth), _ptr(ic.K), _ptr(ic.G), ic.H, ic.W, ic.eps, _ptr(w_t), _ptr(bias), B, ic.h, ic.w,
            c_out, _ptr(out), out.shape[-1], out_c_offset, _ptr(motion), motion.shape[3], motion_c_offset, int(bool(out_split)),
            int(bool(motion_split)), float(a_scale), _stream(), work=2.0 * 98 * c_out * B * ic.h * ic.w)
    return out


def nchw_to_nhwc(src, dst=None, c_offset: int = 0):
    """src (B,C,H,W) -> channels [c_offset, c_offset+C) of dst (B,H,W,Cs) (allocated (B,H,W,C) if None)."""
    src = _chk(src, "src")
    B, Cc, H, W = src.shape
    if dst is None:
        dst = torch.empty(B, H, W, Cc, device=src.device, dtype=F32)
    _launch("rnnpose_nchw_to_nhwc_f32", _ptr(src), B, Cc, H * W, _ptr(dst), dst.shape[3], c_offset, _stream())
    return dst


def nhwc_to_nchw(src, c_offset: int = 0, c_count: int | None = None, out=None):
    """channels [c_offset, c_offset+c_count) of src (B,H,W,Cs) -> (B,c_count,H,W)"""
    B, H, W, Cs = src.shape
    c_count = Cs - c_offset if c_count is None else c_count
    dst = out if out is not None else torch.empty(B, c_count, H, W, device=src.device, dtype=F32)
    _launch("rnnpose_nhwc_to_nchw_f32", _ptr(src), B, c_count, H * W, Cs, c_offset, _ptr(dst), _stream())
    return dst


def flow_prep(coords1, flow4, motion, motion_c_offset, subtract_grid: bool = True):
    """coords1 (B,2,h,w): absolute coordinates (subtract_grid) or the flow itself -> flow4 (B,h,w,4), motion[..., co:co+2]."""
    B, _, h, w = coords1.shape
    _launch("rnnpose_flow_prep_f32", _ptr(coords1), int(subtract_grid), B, h, w, _ptr(flow4), _ptr(motion), motion.shape[3],
            motion_c_offset, _stream())


def flow_conv7x7_relu(flow4, w_t, bias, out, out_c_offset=0):
    """relu(convf1(flow)) (update.py:84,91): flow4 (B,h,w,4), w_t (98,c_out) -> channels of NHWC `out`."""
    B, h, w, _ = flow4.shape
    c_out = w_t.shape[1]
    _launch("rnnpose_flow_conv7x7_relu_f32", _ptr(flow4), _ptr(w_t), _ptr(bias), B, h, w, c_out, _ptr(out), out.shape[-1],
            out_c_offset, _stream(), work=2.0 * 98 * c_out * B * h * w)
    return out


def flow_features(coords1, w_t, bias, out, motion, motion_c_offset, out_c_offset=0, subtract_grid: bool = True,
                  out_split: bool = False, motion_split: bool = False, a_scale: float = 8.0):
    """flow_prep + flow_conv7x7_relu in one launch: coords1 (B,2,h,w) -> relu(convf1(flow)) into `out` (NHWC), flow into
    motion[..., co:co+2] (update.py:84,91,97).  out_split / motion_split: the destinations are split tensors."""
    B, _, h, w = coords1.shape
    c_out = w_t.shape[1]
    _launch("rnnpose_flow_features_f32", _ptr(coords1), int(subtract_grid), _ptr(w_t), _ptr(bias), B, h, w, c_out, _ptr(out),
            out.shape[-1], out_c_offset, _ptr(motion), motion.shape[3], motion_c_offset, int(bool(out_split)), int(bool(motion_split)),
            float(a_scale), _stream(), work=2.0 * 98 * c_out * B * h * w)
    return out


# ---- split tensors (fp16 hi|lo per 8-channel group; include/rnnpose_hip.h "SPLIT TENSORS") ------------------------------
def split_hl(src, dst=None, src_c_offset: int = 0, c_count: int | None = None, dst_c_offset: int = 0, a_scale: float = 8.0):
    """channels [src_c_offset, +c_count) of the fp32 NHWC tensor `src` -> split form in channels [dst_c_offset, +c_count) of
    `dst` (allocated with c_count channels if None).  The split tensor is an opaque float32-typed buffer of the same shape."""
    _nhwc(src, "src")
    B, H, W, Cs = src.shape
    c_count = Cs - src_c_offset if c_count is None else c_count
    if dst is None:
        dst = torch.empty(B, H, W, c_count, device=src.device, dtype=F32)
    _lib.call("rnnpose_split_hl_f32", _ptr(src), Cs, int(src_c_offset), B * H * W, int(c_count), float(a_scale), _ptr(dst),
              dst.shape[3], int(dst_c_offset), _stream())
    return dst


def unsplit_hl(t, a_scale: float = 8.0):
    """Split tensor (..., C) -> the fp32 values (hi + lo) / a_scale it stands for (torch arithmetic: tests and the NCHW facade)."""
    sh = t.shape
    v = t.contiguous().view(torch.float16).view(*sh[:-1], sh[-1] // 8, 2, 8).float()
    return ((v[..., 0, :] + v[..., 1, :]) / a_scale).reshape(sh)


def flow_head_out(x, x_c_offset, c_in, weight, bias, coords1, delta, coords1_out, flow_lr):
    B, h, w, cs = x.shape
    _launch("rnnpose_flow_head_out_f32", _ptr(x), cs, x_c_offset, c_in, _ptr(weight), _ptr(bias), _ptr(coords1), B, h, w,
            _ptr(delta), _ptr(coords1_out), _ptr(flow_lr), _stream())


def convex_upsample_nhwc(flow_lr, mask, out=None):
    B, h, w, _ = mask.shape
    if out is None:
        out = torch.empty(B, 2, 8 * h, 8 * w, device=mask.device, dtype=F32)
    _launch("rnnpose_convex_upsample_nhwc_f32", _ptr(flow_lr), _ptr(mask), B, h, w, _ptr(out), _stream(),
            nbytes=4.0 * B * h * w * (576 + 2 + 128))
    return out


class PackedConv1x1:
    """Weights of a 1x1 convolution with 256 outputs and <= 352 inputs for the LDS-resident kernel (csrc/conv1x1_resident.hip)."""

    def __init__(self, weight, bias, a_scale: float = A_SCALE):
        import math
        range_guard_arm(weight.device)
        w = _chk(weight.detach(), "weight")
        if w.dim() != 4 or w.shape[0] != 256 or w.shape[2:] != (1, 1) or w.shape[1] % 4 or w.shape[1] > 352:
            raise ValueError("needs a (256, c_in <= 352, 1, 1) weight with c_in % 4 == 0")
        self.c_in = int(w.shape[1])
        wmax = float(w.abs().max())
        self.w_scale = float(2.0 ** math.floor(math.log2(1024.0 / wmax))) if wmax > 0 else 1.0
        self.a_scale = float(a_scale)
        self.bias = _chk(bias.detach(), "bias").contiguous()
        n = int(_lib.load().rnnpose_conv1x1_resident_packed_bytes(self.c_in))
        self.w_packed = torch.empty(n // 2, device=w.device, dtype=torch.float16)
        _lib.call("rnnpose_conv1x1_resident_pack_f16x3", _ptr(w), 256, self.c_in, self.w_scale, _ptr(self.w_packed), _stream())


def conv1x1_resident(pc: PackedConv1x1, src, dst, relu: bool = True, dst_split: bool = False):
    """src, dst: (tensor (B,h,w,C), c_offset).  dst[..., off:off+256] = act(conv1x1(src[..., off:off+c_in])).
    dst_split: dst is written as a split tensor (for a consuming convolution with src_hl)."""
    x, xo = src
    y, yo = dst
    _nhwc(x, "src"); _nhwc(y, "dst")
    n = x.shape[0] * x.shape[1] * x.shape[2]
    if y.shape[:3] != x.shape[:3]:
        raise ValueError("src and dst must share (B,h,w)")
    _launch("rnnpose_conv1x1_resident_f16x3", _ptr(x), x.shape[3], int(xo), pc.c_in, _ptr(pc.w_packed), _ptr(pc.bias), pc.a_scale,
            pc.w_scale, int(bool(relu)), n, _ptr(y), y.shape[3], int(yo), int(bool(dst_split)), _stream(), work=2.0 * n * 256 * pc.c_in)


def corr_lookup_convc1(pc: PackedConv1x1, pyramid_buf, coords, dst, B_total, b0, b1, levels=4, radius=4, relu: bool = True,
                       dst_split: bool = False):
    """r06: window lookup + convc1 in one launch (model/CFNet.py:147-152): dst[..., off:off+256] = act(convc1(corr_fn(coords))) for the
    images [b0, b1) of a pyramid built for B_total images; coords (b1-b0,2,h,w); dst = (tensor (b1-b0,h,w,C), c_offset).  The
    (B,h,w,324) window-feature tensor is never materialised (csrc/corr_convc1.hip).  Not faster than the two kernels (engine.py): an option."""
    y, yo = dst
    _nhwc(y, "dst")
    coords = _chk(coords, "coords")
    nb, two, h, w = coords.shape
    if pc.c_in != 324 or two != 2 or nb != b1 - b0 or tuple(y.shape[:3]) != (nb, h, w):
        raise ValueError("corr_lookup_convc1: weights packed for c_in = 324, coords (b1-b0,2,h,w), dst (b1-b0,h,w,C)")
    n = nb * h * w
    _launch("rnnpose_corr_lookup_convc1_f16x3", _ptr(pyramid_buf), _ptr(coords), int(B_total), int(b0), int(b1), h, w, int(levels), int(radius),
            _ptr(pc.w_packed), _ptr(pc.bias), pc.a_scale, pc.w_scale, int(bool(relu)), _ptr(y), y.shape[3], int(yo), int(bool(dst_split)),
            _stream(), work=2.0 * n * 256 * 324, nbytes=4.0 * n * (levels * 100 + 2 + 256))


class PackedMaskHead:
    """mask.2 weights (576,256,1,1) + bias, post-scaled (the 0.25 of update.py:187) and split into fp16 hi/lo MFMA fragments
    for the fused mask + up-sampling kernel (csrc/mask_upsample.hip)."""

    def __init__(self, weight, bias, post_scale: float = 0.25, a_scale: float = A_SCALE):
        import math
        range_guard_arm(weight.device)
        w = _chk(weight.detach(), "weight")
        if tuple(w.shape) != (576, 256, 1, 1):
            raise ValueError("mask.2 weight must be (576,256,1,1)")
        wmax = float(w.abs().max()) * post_scale
        self.w_scale = float(2.0 ** math.floor(math.log2(1024.0 / wmax))) if wmax > 0 else 1.0
        self.a_scale = float(a_scale)
        self.bias = (_chk(bias.detach(), "bias") * post_scale).contiguous()
        n = int(_lib.load().rnnpose_mask_upsample_packed_bytes())
        self.w_packed = torch.empty(n // 2, device=w.device, dtype=torch.float16)
        _lib.call("rnnpose_mask_upsample_pack_f16x3", _ptr(w), float(post_scale), self.w_scale, _ptr(self.w_packed), _stream())


def mask_upsample(pm: PackedMaskHead, x, x_c_offset, flow_lr, out=None):
    """mask.2 + convex 8x up-sampling in one kernel (update.py:183-187, CFNet.py:95-106).  x (B,h,w,C) with relu(mask.0(h))
    in channels [x_c_offset, +256); flow_lr (B,h,w,2) -> (B,2,8h,8w).  Agrees with the 1x1 convolution + convex_upsample_nhwc
    to fp32 round-off (online softmax); the mask tensor is never materialised."""
    _nhwc(x, "x")
    B, h, w, cs = x.shape
    if out is None:
        out = torch.empty(B, 2, 8 * h, 8 * w, device=x.device, dtype=F32)
    _launch("rnnpose_mask_upsample_f16x3", _ptr(x), cs, int(x_c_offset), _ptr(pm.w_packed), 576, _ptr(pm.bias), pm.a_scale,
            pm.w_scale, _ptr(_chk(flow_lr, "flow_lr")), B, h, w, _ptr(out), _stream(), work=2.0 * B * h * w * 576 * 256,
            nbytes=4.0 * B * h * w * (256 + 2 + 128))
    return out


# ---- fp16x3 range guard ------------------------------------------------------------------------------------------------
# ON BY DEFAULT (round 3): every kernel that splits fp32 values into fp16 hi|lo counts the quads it had to clamp in a sticky
# device counter; PoseRefiner.forward returns the counter as a device tensor ("f16x3_range_events"), so a silent clamp --
# something the fp32 reference cannot do -- is visible in the public output without a host synchronisation.
# RNNPOSE_RANGE_GUARD=0 switches it off (measurement only).
import os as _os
_guard_env = _os.environ.get("RNNPOSE_RANGE_GUARD", "1") != "0"
_guard_on = False
_guard_devices = set()


def range_guard_arm(device=None):
    """Allocate the current device's counter and switch counting on (idempotent; never called under stream capture: the
    Packed* constructors and PoseRefiner.forward call it before any launch)."""
    global _guard_on
    if not _guard_env:
        return
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev in _guard_devices and _guard_on:
        return
    with torch.cuda.device(dev):
        _lib.call("rnnpose_f16x3_saturation_check", 1)
    _guard_devices.add(dev)
    _guard_on = True


def range_guard_state() -> bool:
    """Part of every captured graph's key: the counter pointer is baked into the captured launches."""
    return _guard_on


def saturation_check(enable: bool = True):
    """Switch the fp16x3 range guard on/off (process-global): with it on, every fp16x3 kernel counts the activation quads
    whose scaled magnitude left the fp16 range (|x * a_scale| > 65504) -- values the fp32 reference would handle."""
    global _guard_on
    _lib.call("rnnpose_f16x3_saturation_check", int(bool(enable)))
    _guard_on = bool(enable)
    if enable:
        _guard_devices.add(torch.cuda.current_device())


def saturation_events(out=None):
    """-> device tensor (1,) int64 holding the sticky counter NOW in stream order; no host synchronisation."""
    if out is None:
        out = torch.zeros(1, device="cuda", dtype=torch.int64)
    _lib.call("rnnpose_f16x3_saturation_peek", _ptr(out), _stream())
    return out


def saturation_count(reset: bool = True) -> int:
    """Clamped / non-finite quads seen since the last reset (synchronises the current stream)."""
    n = C.c_ulonglong(0)
    _lib.call("rnnpose_f16x3_saturation_count", C.byref(n), int(bool(reset)), _stream())
    return int(n.value)


# ---- f1: encoder stem ---------------------------------------------------------------------------------------------------
class PackedStem:
    """conv1 of the RAFT encoder (64,3,7,7) split into fp16 hi/lo MFMA fragments for csrc/stem.hip."""

    def __init__(self, weight, bias, a_scale: float = 8.0):
        import math
        range_guard_arm(weight.device)
        w = _chk(weight.detach(), "weight")
        if tuple(w.shape) != (64, 3, 7, 7):
            raise ValueError("the stem kernel is the RAFT encoder's conv1: weight must be (64,3,7,7)")
        wmax = float(w.abs().max())
        self.w_scale = float(2.0 ** math.floor(math.log2(1024.0 / wmax))) if wmax > 0 else 1.0
        self.a_scale = float(a_scale)
        n = int(_lib.load().rnnpose_stem_packed_halfs())
        self.w_hi = torch.empty(n, device=w.device, dtype=torch.float16)
        self.w_lo = torch.empty(n, device=w.device, dtype=torch.float16)
        self.bias = _chk(bias.detach(), "bias")
        _lib.call("rnnpose_stem_pack_weights_f16x3", _ptr(w), self.w_scale, _ptr(self.w_hi), _ptr(self.w_lo), _stream())


def stem_conv(ps: PackedStem, img, normalize: bool = True):
    """img (N,3,H,W) NCHW -> (out (N,ceil(H/2),ceil(W/2),64) NHWC, tile_stats or None)
    = conv7x7 stride 2 pad 3 of (2*(img/255)-1 if normalize else img) + bias   (model/CFNet.py:42-43, extractor.py:131,197)."""
    img = _chk(img, "image")
    N, Cc, H, W = img.shape
    if Cc != 3:
        raise ValueError("image must have 3 channels")
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    tiles, exact = C.c_int(0), C.c_int(0)
    _lib.call("rnnpose_stem_tiles", H, W, C.byref(tiles), C.byref(exact))
    out = torch.empty(N, Ho, Wo, 64, device=img.device, dtype=F32)
    ts = torch.empty(N * tiles.value, 64, 2, device=img.device, dtype=F64) if exact.value else None      # fp64 sums (cancellation in E[x^2] - mean^2)
    _launch("rnnpose_stem_conv7x7_s2_f16x3", _ptr(img), N, H, W, int(bool(normalize)), _ptr(ps.w_hi), _ptr(ps.w_lo),
            _ptr(ps.bias), ps.a_scale, ps.w_scale, _ptr(out), _ptr(ts), _stream(), work=2.0 * N * Ho * Wo * 64 * 147)
    return out, ts


# ---- f1: instance norm on NHWC --------------------------------------------------------------------------------
_in_ws = {}


def instnorm_nhwc(x, relu=True, residual=None, out=None, eps: float = 1e-5):
    """x (B,H,W,C) -> act((x-mean)*rstd) [then relu(residual + .)], nn.InstanceNorm2d semantics (no affine)."""
    _nhwc(x, "x")
    B, H, W, Cc = x.shape
    n = int(_lib.load().rnnpose_instnorm_workspace_bytes(B, H * W, Cc))
    key = (x.device, n, torch.cuda.current_stream().cuda_stream)     # (calls on different streams may run concurrently)
    ws = _in_ws.get(key)
    if ws is None:
        ws = _in_ws[key] = torch.empty(n // 8, device=x.device, dtype=F64)
    if out is None:
        out = torch.empty_like(x)
    stats = torch.empty(B, Cc, 2, device=x.device, dtype=F32)
    _launch("rnnpose_instnorm_nhwc_f32", _ptr(x), B, H * W, Cc, eps, int(bool(relu)), _ptr(residual), _ptr(ws), n,
            _ptr(stats), _ptr(out), _stream(), nbytes=4.0 * x.numel() * (3 + (residual is not None)))
    return out


def instnorm_tiles_nhwc(x, tile_stats, relu=True, residual=None, out=None, eps: float = 1e-5, stats_only: bool = False,
                        residual_norm=None, residual_relu: bool = False):
    """instnorm_nhwc with the statistics pass replaced by the producing convolution's tile_stats (B * ceil(HW/128) records).
    stats_only: -> mean_rstd (B,C,2) only (for a consumer that normalises in its load: conv2d_nhwc(in_norm=...)).
    residual_norm (B,C,2): the residual is a RAW convolution output, normalised [+ ReLU] on the fly with these statistics."""
    _nhwc(x, "x")
    B, H, W, Cc = x.shape
    tpi = tile_stats.shape[0] // B              # records per image, as the producing convolution tiled it
    stats = torch.empty(B, Cc, 2, device=x.device, dtype=F32)
    if stats_only:
        _launch("rnnpose_instnorm_tiles_nhwc_f32", _ptr(None), B, H * W, Cc, eps, int(bool(relu)), _ptr(None), _ptr(None), 0,
                _ptr(tile_stats), tpi, _ptr(stats), _ptr(None), _stream())
        return stats
    if out is None:
        out = torch.empty_like(x)
    _launch("rnnpose_instnorm_tiles_nhwc_f32", _ptr(x), B, H * W, Cc, eps, int(bool(relu)), _ptr(residual), _ptr(residual_norm),
            int(bool(residual_relu)), _ptr(tile_stats), tpi, _ptr(stats), _ptr(out), _stream(),
            nbytes=4.0 * x.numel() * (2 + (residual is not None)))
    return out


# ---- f2/f3: evaluator metrics ------------------------------------------------------------------------------------
def nn_search(ref_pts, que_pts, exclude_self: bool = False):
    """ref (B,N1,D), que (B,N2,D) fp32 on the GPU, D in {2,3} -> idx (B,N2) int32: first nearest reference point."""
    ref_pts, que_pts = _chk(ref_pts, "ref_pts"), _chk(que_pts, "que_pts")
    B, n1, dim = ref_pts.shape
    n2 = que_pts.shape[1]
    idx = torch.empty(B, n2, device=ref_pts.device, dtype=torch.int32)
    _launch("rnnpose_nn_search_f32", _ptr(ref_pts), _ptr(que_pts), _ptr(idx), B, n1, n2, dim, int(exclude_self), _stream())
    return idx


def pose_metrics(model, pose_pred, pose_gt, K, symmetric: bool = False):
    """model (P,3), pose_pred/pose_gt (B,3,4), K (3,3) -> (B,5) fp64 [ADD, ADD-S (-1 if not symmetric), proj2d px,
    translation cm, rotation deg]   (utils/eval_metric.py:102-192)."""
    model, pose_pred, pose_gt, K = _chk(model, "model"), _chk(pose_pred, "pose_pred"), _chk(pose_gt, "pose_gt"), _chk(K, "K")
    P, B = model.shape[0], pose_pred.shape[0]
    n = int(_lib.load().rnnpose_pose_metrics_workspace_bytes(B, P)) if symmetric else 0
    ws = torch.empty(max(n, 4) // 4, device=model.device, dtype=F32)
    out = torch.empty(B, 5, device=model.device, dtype=F64)
    _launch("rnnpose_pose_metrics_f64", _ptr(model), P, _ptr(pose_pred), _ptr(pose_gt), _ptr(K), B, int(symmetric),
            _ptr(ws), n, _ptr(out), _stream())
    return out


# ---- f4: zoom-crop on device ---------------------------------------------------------------------------------------
def pointcloud_depth(verts, vert_offsets, T, K, size):
    """verts (P,3) fp32 (all models concatenated), vert_offsets (B+1,) int32 device tensor, T (B,4,4), K (B,3,3), size (H,W)
    -> (B,1,H,W) depth splat of every image's vertices (geometry/diff_render_optim.py:369-401)."""
    verts, T, K = _chk(verts, "verts"), _chk(T, "T"), _chk(K, "K")
    if not (vert_offsets.is_cuda and vert_offsets.dtype == torch.int32 and vert_offsets.is_contiguous()):
        raise RuntimeError("vert_offsets must be a contiguous int32 GPU tensor of B+1 entries")
    B = T.shape[0]
    H, W = int(size[0]), int(size[1])
    n = int(_lib.load().rnnpose_pointcloud_depth_workspace_bytes(B, H, W))
    ws = torch.empty(n // 4, device=T.device, dtype=torch.int32)
    out = torch.empty(B, 1, H, W, device=T.device, dtype=F32)
    _launch("rnnpose_pointcloud_depth_f32", _ptr(verts), _ptr(vert_offsets), int(verts.shape[0]), _ptr(T), _ptr(K), B, H, W, _ptr(ws), n,
            _ptr(out), _stream())
    return out


def mask_bbox(depth):
    """depth (B,1,H,W) -> (B,4) int32 [xmin, ymin, xmax, ymax] of depth > 0 (model/PoseRefiner.py:154-158,259)."""
    depth = _chk(depth, "depth")
    B, _, H, W = depth.shape
    bbox = torch.empty(B, 4, device=depth.device, dtype=torch.int32)
    _launch("rnnpose_mask_bbox_f32", _ptr(depth), B, H, W, _ptr(bbox), _stream())
    return bbox


def zoom_crop_params(bbox, K, T, image_size, crop_size, margin_ratio=0.4):
    """-> theta (B,2,3) for F.affine_grid and K_crop (B,3,3)   (model/PoseRefiner.py:145-218, no host round trip)."""
    K, T = _chk(K, "K"), _chk(T, "T")
    if not (bbox.is_cuda and bbox.dtype == torch.int32 and bbox.is_contiguous()):
        raise RuntimeError("bbox must be a contiguous int32 GPU tensor (from mask_bbox)")
    B = K.shape[0]
    theta = torch.empty(B, 2, 3, device=K.device, dtype=F32)
    K_crop = torch.empty(B, 3, 3, device=K.device, dtype=F32)
    _launch("rnnpose_zoom_crop_params_f32", _ptr(bbox), _ptr(K), _ptr(T), B, int(image_size[0]), int(image_size[1]),
            int(crop_size[0]), int(crop_size[1]), float(margin_ratio), _ptr(theta), _ptr(K_crop), _stream())
    return theta, K_crop


def zoom_crop(x, theta, crop_size, want_grid=False):
    """F.grid_sample(x, F.affine_grid(theta, (B,C,*crop_size))) fused (bilinear, zeros, align_corners=False).
    x (B,C,H,W) or None (grid only) -> out (B,C,hc,wc) [, grid (B,hc,wc,2)]."""
    theta = _chk(theta, "theta")
    B = theta.shape[0]
    hc, wc = int(crop_size[0]), int(crop_size[1])
    out = None
    C_, H, W = 0, 1, 1
    if x is not None:
        x = _chk(x, "x")
        _, C_, H, W = x.shape
        out = torch.empty(B, C_, hc, wc, device=theta.device, dtype=F32)
    grid = torch.empty(B, hc, wc, 2, device=theta.device, dtype=F32) if want_grid else None
    _launch("rnnpose_zoom_crop_f32", _ptr(x), _ptr(theta), B, C_, H, W, hc, wc, _ptr(out), _ptr(grid), _stream())
    return (out, grid) if want_grid else out
