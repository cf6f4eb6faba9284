#!/usr/bin/env python3
"""The inner iteration's kernels that are NOT implicit-GEMM convolutions, each alone on the chip at the headline shape
(480x640, 1/8 maps 60x80), half batch (B=4: as the two-chain schedule launches them) and full batch (B=8): wall-clock
microseconds over 50 launches.  r04: these are ~45 % of an inner iteration (profiles/r04_inner_iteration.txt).
    python tools/tail_kernels.py [names,comma,separated]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
dev = "cuda"
H, W = 480, 640
h, w = H // 8, W // 8
g = torch.Generator(device=dev)
g.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


f1, f2 = r(8, 256, h, w), r(8, 256, h, w)
buf, _ = ops.corr_pyramid(f1, f2, precision="f16x3")
mhead = ops.PackedMaskHead(r(576, 256, 1, 1) * 0.09, r(576) * 0.1)
c1r = ops.PackedConv1x1(r(256, 324, 1, 1) * 0.08, r(256) * 0.1)
w7, b7 = (r(128, 2, 7, 7) * 0.1).reshape(128, 98).t().contiguous(), r(128) * 0.1
w2, b2 = r(2, 256, 3, 3) * 0.05, r(2) * 0.1
sigma = torch.ones(1, device=dev)
rows = []
for B in (4, 8):
    g1, g2 = r(B, 32, H, W), r(B, 32, H, W)
    g1 /= g1.norm(dim=1, keepdim=True)
    g2 /= g2.norm(dim=1, keepdim=True)
    depth = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.3 + 0.9
    depth[:, :, : H // 4] = 0
    K = torch.tensor([[572.4, 0, W / 2], [0, 573.6, H / 2], [0, 0, 1]], device=dev).repeat(B, 1, 1)
    G = ops.se3_exp(r(B, 6) * 0.02)
    heads = r(B, h, w, 512).clamp_(min=0)
    flow_lr = r(B, h, w, 2)
    corr = torch.empty(B, h, w, 324, device=dev)
    cor1 = torch.empty(B, h, w, 256, device=dev)
    flo1 = torch.empty(B, h, w, 128, device=dev)
    motion = torch.empty(B, h, w, 128, device=dev)
    delta, c1out = torch.empty(B, h, w, 2, device=dev), torch.empty(B, 2, h, w, device=dev)
    c1 = ops.induced_coords_lowres(depth, K, G, h, w)
    up = ops.mask_upsample(mhead, heads, 256, flow_lr)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    smooth = torch.stack([3.0 * torch.sin(xs / 70.0) + 1.5 * torch.cos(ys / 45.0), 2.0 * torch.cos(xs / 90.0) - 2.5 * torch.sin(ys / 60.0)])[None].repeat(B, 1, 1, 1).float().contiguous()
    wm = ops.corr_weight(g1, g2, smooth, depth, sigma)         # (a SMOOTH correspondence field, as the loop produces: the gather's locality depends on it)
    cases = {
        "induced_coords_lowres": lambda: ops.induced_coords_lowres(depth, K, G, h, w, out=c1),
        "corr_lookup": lambda: ops.corr_lookup_nhwc_part(buf, c1, corr, 8, 0, B),
        "conv1x1_resident": lambda: ops.conv1x1_resident(c1r, (corr, 0), (cor1, 0)),
        "lookup+convc1 fused": lambda: ops.corr_lookup_convc1(c1r, buf, c1, (cor1, 0), 8, 0, B),
        "flow_features_7x7": lambda: ops.flow_features(c1, w7, b7, flo1, motion, 126),
        "flow_head_out": lambda: ops.flow_head_out(heads, 0, 256, w2, b2, c1, delta, c1out, flow_lr),
        "mask_upsample": lambda: ops.mask_upsample(mhead, heads, 256, flow_lr, out=up),
        "corr_weight": lambda: ops.corr_weight(g1, g2, smooth, depth, sigma, out=wm),
        "lm_step": lambda: ops.lm_step(smooth, wm, depth, K, G),
    }
    for name, fn in cases.items():
        if only and name not in only:
            continue
        rows.append((B, name, timed(fn)))
for name in dict.fromkeys(n for _, n, _ in rows):
    print(f"{name:24s} " + "   ".join(f"B={B}: {t:7.1f} us" for B, n, t in rows if n == name), flush=True)
