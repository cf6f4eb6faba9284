#!/usr/bin/env python3
"""Times the hand-written NHWC fp16x3 conv against torch/MIOpen fp32 at the update-block shapes (B=8, 60x80)."""
import sys, os, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops

B, h, w = 8, 60, 80
shapes = [("convc1 1x1 324->256", [324], 256, 1, 1), ("convc2 3x3 256->192", [256], 192, 3, 3),
          ("convf2 3x3 128->64", [128], 64, 3, 3), ("conv 3x3 256->126", [192, 64], 126, 3, 3),
          ("gru zr 1x5 384->256", [128, 128, 128], 256, 1, 5), ("gru q 1x5 384->128", [128, 128, 128], 128, 1, 5),
          ("gru zr 5x1 384->256", [128, 128, 128], 256, 5, 1), ("gru q 5x1 384->128", [128, 128, 128], 128, 5, 1),
          ("heads 3x3 128->512", [128], 512, 3, 3), ("mask2 1x1 256->576", [256], 576, 1, 1)]
for name, segs, co, kh, kw in shapes:
    ci = sum(segs)
    x = torch.randn(B, ci, h, w, device="cuda")
    wt = torch.randn(co, ci, kh, kw, device="cuda") * (2.0 / (ci * kh * kw)) ** 0.5
    bias = torch.randn(co, device="cuda")
    pc = ops.PackedConv(wt, bias, segs)
    xs, off = [], 0
    for c in segs:
        xs.append((x[:, off:off + c].permute(0, 2, 3, 1).contiguous(), 0)); off += c
    out = torch.empty(B, h, w, (co + 3) // 4 * 4, device="cuda")
    def run_mine(): ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU)
    def run_ref(): return F.relu_(F.conv2d(x, wt, bias, padding=(kh // 2, kw // 2)))
    res = []
    for f in (run_mine, run_ref):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20 * 1e3)
    fl = 2.0 * B * h * w * co * ci * kh * kw
    y = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), padding=(kh // 2, kw // 2)))
    e1 = float((out[..., :co].permute(0, 3, 1, 2).double() - y).abs().max()); e2 = float((run_ref().double() - y).abs().max())
    print(f"{name:24s} mine {res[0]:7.3f} ms ({fl/res[0]/1e9:6.1f} TF-eq)  miopen+relu {res[1]:7.3f} ms ({fl/res[1]/1e9:6.1f} TF)  err {e1:.2e} vs {e2:.2e}", flush=True)
