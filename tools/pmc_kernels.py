#!/usr/bin/env python3
"""Launches every hand-written kernel a few times at the BASELINE config-2 shapes (B=8, 480x640) so that
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) can attribute HBM traffic per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

dev = "cuda"
B, H, W = 8, 480, 640
h, w = H // 8, W // 8
g = torch.Generator(device=dev)
g.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
f1, f2 = r(B, 256, h, w), r(B, 256, h, w)
ctx = r(B, 256, H, W) * 0.1
g1, g2 = r(B, 32, H, W), r(B, 32, H, W)
depth = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.3 + 0.9
K = torch.tensor([[572.4, 0, W / 2], [0, 573.6, H / 2], [0, 0, 1]], device=dev).repeat(B, 1, 1)
G = ops.se3_exp(r(B, 6) * 0.02)
mask = r(B, 576, h, w)
sigma = torch.ones(1, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    buf, _ = ops.corr_pyramid(f1, f2)
    c1 = ops.induced_coords_lowres(depth, K, G, h, w)
    corr = ops.corr_lookup(buf, c1)
    net, inp = ops.context_prep(ctx, h, w)
    up = ops.convex_upsample(c1, mask)
    wm = ops.corr_weight(g1, g2, up, depth, sigma)
    ops.lm_step(up, wm, depth, K, G)
torch.cuda.synchronize()
print("ok")
