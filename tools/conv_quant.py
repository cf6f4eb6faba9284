#!/usr/bin/env python3
"""Tile-quantisation probe: times the GRU 1x5 convs at row counts that give whole / ragged rounds of workgroups."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for co, kh, kw in ((256, 1, 5), (128, 1, 5), (256, 1, 1), (192, 3, 3)):
    for (B, h, w) in ((4, 64, 64), (6, 64, 64), (7, 64, 64), (8, 64, 64), (8, 60, 80), (9, 64, 64), (10, 64, 64), (12, 64, 64), (16, 64, 64)):
        segs = [128, 128, 128] if kw == 5 else ([324] if kh == 1 else [256])
        ci = sum(segs)
        wt = torch.randn(co, ci, kh, kw, device="cuda") * 0.02
        pc = ops.PackedConv(wt, torch.zeros(co, device="cuda"), segs)
        xs = [(torch.randn(B, h, w, c, device="cuda"), 0) for c in segs]
        out = torch.empty(B, h, w, co, device="cuda")
        t = timeit(lambda: ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU))
        M = B * h * w
        nb = -(-M // 128) * -(-co // (128 if (co % 128 == 0 and -(-M // 128) * (co // 128) >= 512) else 64))
        fl = 2.0 * M * co * ci * kh * kw
        print(f"co {co} k {kh}x{kw} M {M:6d} blocks {nb:5d}  {t:.4f} ms  {fl/t/1e9:6.1f} TF-eq  ({t/M*1e6:.3f} ns/row)", flush=True)
