#!/usr/bin/env python3
"""Per-stage timing of the encoder engine vs the nn.Module (MIOpen) path, and strided conv timing."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops
from rnnpose_amd.cfnet import ImageFeaEncoder

def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

a, b = torch.rand(8, 3, 480, 640, device="cuda"), torch.rand(8, 3, 480, 640, device="cuda")
for be in ("hip", "miopen"):
    enc = ImageFeaEncoder(conv_backend=be).cuda().eval()
    with torch.no_grad():
        print(f"encoder {be}: {t(lambda: enc(a, b)):.2f} ms")
for (B, H, W, ci, co, k) in ((16, 240, 320, 64, 96, 3), (16, 240, 320, 64, 96, 1), (16, 120, 160, 96, 128, 3)):
    x = torch.randn(B, ci, H, W, device="cuda"); w = torch.randn(co, ci, k, k, device="cuda") * 0.05; bias = torch.zeros(co, device="cuda")
    xn = x.permute(0, 2, 3, 1).contiguous(); pc = ops.PackedConv(w, bias, [ci]); out = torch.empty(B, H // 2, W // 2, co, device="cuda")
    xc = x.contiguous(memory_format=torch.channels_last)
    print(f"s2 {k}x{k} {ci}->{co} @{H}x{W}: mine {t(lambda: ops.conv2d_nhwc(pc, [(xn, 0)], (out, 0), 0, stride=2)):.3f} ms  "
          f"miopen cl {t(lambda: F.conv2d(xc, w, bias, stride=2, padding=k // 2)):.3f} ms  miopen nchw {t(lambda: F.conv2d(x, w, bias, stride=2, padding=k // 2)):.3f} ms")
for (B, H, W, c) in ((16, 240, 320, 64), (16, 120, 160, 96), (16, 60, 80, 128)):
    x = torch.randn(B, H, W, c, device="cuda")
    print(f"instnorm {c}ch @{H}x{W}: {t(lambda: ops.instnorm_nhwc(x, True)):.3f} ms ({x.numel()*4*3/1e9:.2f} GB moved)")
