#!/usr/bin/env python3
"""Per-layer conv timing of the encoder at the BASELINE shape (16 images of 480x640 -> stem output 240x320x64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = 16
layers = [("l1 3x3 64->64 @240x320 (x4)", 240, 320, 64, 64, 3, 1, 4), ("l2 3x3 64->96 s2 @240x320", 240, 320, 64, 96, 3, 2, 1),
          ("l2 1x1 64->96 s2", 240, 320, 64, 96, 1, 2, 1), ("l2 3x3 96->96 @120x160 (x3)", 120, 160, 96, 96, 3, 1, 3),
          ("l3 3x3 96->128 s2 @120x160", 120, 160, 96, 128, 3, 2, 1), ("l3 1x1 96->128 s2", 120, 160, 96, 128, 1, 2, 1),
          ("l3 3x3 128->128 @60x80 (x3)", 60, 80, 128, 128, 3, 1, 3), ("out 1x1 128->256 @60x80", 60, 80, 128, 256, 1, 1, 1)]
tot = 0.0
for name, H, W, ci, co, k, st, cnt in layers:
    x = torch.randn(B, H, W, ci, device="cuda")
    w = torch.randn(co, ci, k, k, device="cuda") * 0.05
    pc = ops.PackedConv(w, torch.zeros(co, device="cuda"), [ci])
    Ho, Wo = -(-H // st), -(-W // st)
    out = torch.empty(B, Ho, Wo, co, device="cuda")
    ts = torch.empty(B * Ho * Wo // 128, co, 2, device="cuda") if (Ho * Wo) % 128 == 0 else None
    t = timeit(lambda: ops.conv2d_nhwc(pc, [(x, 0)], (out, 0), 0, stride=st, tile_stats=ts))
    fl = 2.0 * B * Ho * Wo * co * ci * k * k
    tot += t * cnt
    print(f"{name:36s} {t:.3f} ms  {fl / t / 1e9:6.1f} TF-eq   in {x.numel()*4/1e6:.0f} MB out {out.numel()*4/1e6:.0f} MB", flush=True)
print(f"sum over one encoder pass: {tot:.2f} ms")
