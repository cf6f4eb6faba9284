#!/usr/bin/env python3
"""Launches every hand-written kernel of the refinement loop a few times, as bench.py's schedule does (BASELINE config-2
shapes: 480x640, full batch 8 for the per-outer kernels, half batch 4 for the per-iteration ones), so that rocprofv3 --pmc
passes (tools/pmc_sq.sh) can attribute HBM bytes and SQ activity per launch.  The convolutions: tools/conv_layers.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

dev = "cuda"
B, H, W = 8, 480, 640
h, w = H // 8, W // 8
Bh = B // 2
g = torch.Generator(device=dev)
g.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
f1, f2 = r(B, 256, h, w), r(B, 256, h, w)
ctx = r(B, 256, H, W) * 0.1
g1, g2 = r(Bh, 32, H, W), r(Bh, 32, H, W)
depth = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.3 + 0.9
K = torch.tensor([[572.4, 0, W / 2], [0, 573.6, H / 2], [0, 0, 1]], device=dev).repeat(B, 1, 1)
G = ops.se3_exp(r(B, 6) * 0.02)
mask = r(Bh, h, w, 576)
flow_lr = r(Bh, h, w, 2)
corr = torch.empty(Bh, h, w, 324, device=dev)
img = torch.rand(B, 3, H, W, device=dev, generator=g)
stem = ops.PackedStem(r(64, 3, 7, 7) * 0.1, torch.zeros(64, device=dev))
x64 = r(B, H // 2, W // 2, 64)
sigma = torch.ones(1, device=dev)
heads = r(Bh, h, w, 512).clamp_(min=0)
mhead = ops.PackedMaskHead(r(576, 256, 1, 1) * 0.09, r(576) * 0.1)
c1r = ops.PackedConv1x1(r(256, 324, 1, 1) * 0.08, r(256) * 0.1)
cor1 = torch.empty(Bh, h, w, 256, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    buf, _ = ops.corr_pyramid(f1, f2, precision="f16x3")
    c1 = ops.induced_coords_lowres(depth[:Bh], K[:Bh], G[:Bh], h, w)
    ops.corr_lookup_nhwc_part(buf, c1, corr, B, 0, Bh)
    ops.context_prep(ctx, h, w)
    up = ops.convex_upsample_nhwc(flow_lr, mask)
    ops.mask_upsample(mhead, heads, 256, flow_lr)
    ops.conv1x1_resident(c1r, (corr, 0), (cor1, 0))
    wm = ops.corr_weight(g1, g2, up, depth[:Bh], sigma)
    ops.lm_step(up, wm, depth[:Bh], K[:Bh], G[:Bh])
    y, ts = ops.stem_conv(stem, img)
    ops.instnorm_tiles_nhwc(y, ts, relu=True)
    ops.instnorm_nhwc(x64, relu=True)
torch.cuda.synchronize()
print("ok")
