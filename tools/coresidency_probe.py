#!/usr/bin/env python3
"""Do a strip convolution (matrix-bound, two waves per SIMD) and corr_weight (memory-bound, 80 registers, no LDS) of ANOTHER stream share the chip?
Stream A launches N convolutions back to back, stream B N descriptor-weight kernels; alone and together, for the 160-row strip form (234-256 registers:
two waves fill the register file) and the 96-row form (201-206: two waves leave 96 registers per lane).  together ~ max(A, B): co-resident;
together ~ A + B: time-shared.   python tools/coresidency_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

dev = "cuda"
B, H, W, h, w = 4, 480, 640, 60, 80
g = torch.Generator(device=dev); g.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
g1, g2 = r(B, 32, H, W), r(B, 32, H, W)
ys, xs_ = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
flow = torch.stack([3.0 + 0.01 * ys, -2.0 + 0.005 * xs_], 0)[None].expand(B, 2, H, W).contiguous()      # a SMOOTH field, as the loop produces (the gather's locality depends on it)
depth = (torch.rand(B, 1, H, W, device=dev, generator=g) > 0.25).float() * 1.1
sigma = torch.tensor([0.7], device=dev)
wmap = torch.empty(B, H, W, device=dev)
cw = lambda: ops.corr_weight(g1, g2, flow, depth, sigma, out=wmap)
layers = {"gru zr 1x5 256->256": ([128, 128], 256, 1, 5), "heads 3x3 128->512": ([128], 512, 3, 3), "convc2 3x3 256->192": ([256], 192, 3, 3)}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 40


def timed(fa, fb):
    for f in (fa, fb):
        if f:
            for _ in range(3):
                f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


for _ in range(3):
    tb = timed(None, cw)           # (the first passes also bring the clocks up)
print(f"corr_weight alone (B=4): {tb:.1f} us per launch")
for name, (segs, co, kh, kw) in layers.items():
    ci = sum(segs)
    wt = torch.randn(co, ci, kh, kw, device=dev) * (2.0 / (ci * kh * kw)) ** 0.5
    pc = ops.PackedConv(wt, torch.randn(co, device=dev), segs)
    xs = [(ops.split_hl(r(B, h, w, c)), 0) for c in segs]
    out = torch.empty(B, h, w, (co + 7) // 8 * 8, device=dev)
    for tile, label in ((5, "160-row strips"), (7, "96-row strips")):
        conv = lambda: ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU, src_hl=True, dst_hl=True, tile=tile)
        timed(conv, cw)
        ta = timed(conv, None)
        tab = timed(conv, cw)
        tb = timed(None, cw)
        print(f"{name:24s} {label:15s}: conv alone {ta:6.1f} us, together with corr_weight {tab:6.1f} us per pair  (sum {ta + tb:6.1f}, max {max(ta, tb):6.1f})")
