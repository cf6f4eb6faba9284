#!/usr/bin/env python3
"""Times torch/MIOpen fp32 convolutions at the update-block and encoder shapes (decides whether the convs
need hand-written MFMA kernels; SURVEY.md section 7 hard part 1)."""
import json
import sys
import time

import torch
import torch.nn.functional as F

dev = "cuda"
B, h, w = 8, 60, 80
shapes = [  # name, Cin, Cout, kh, kw, H, W, stride
    ("convc1 1x1 324->256", 324, 256, 1, 1, h, w, 1),
    ("convc2 3x3 256->192", 256, 192, 3, 3, h, w, 1),
    ("convf1 7x7 2->128", 2, 128, 7, 7, h, w, 1),
    ("convf2 3x3 128->64", 128, 64, 3, 3, h, w, 1),
    ("conv 3x3 256->126", 256, 126, 3, 3, h, w, 1),
    ("gru zr 1x5 384->256", 384, 256, 1, 5, h, w, 1),
    ("gru q 1x5 384->128", 384, 128, 1, 5, h, w, 1),
    ("gru zr 5x1 384->256", 384, 256, 5, 1, h, w, 1),
    ("gru q 5x1 384->128", 384, 128, 5, 1, h, w, 1),
    ("head 3x3 128->256", 128, 256, 3, 3, h, w, 1),
    ("mask2 1x1 256->576", 256, 576, 1, 1, h, w, 1),
    ("flow2 3x3 256->2", 256, 2, 3, 3, h, w, 1),
    ("enc conv1 7x7s2 3->64 (16 img)", 3, 64, 7, 7, 480, 640, 2),
    ("enc l1 3x3 64->64 (16 img)", 64, 64, 3, 3, 240, 320, 1),
    ("enc l2 3x3 96->96 (16 img)", 96, 96, 3, 3, 120, 160, 1),
    ("enc l3 3x3 128->128 (16 img)", 128, 128, 3, 3, 60, 80, 1),
]
out = []
for name, ci, co, kh, kw, H, W, st in shapes:
    nb = 16 if "16 img" in name else B
    for fmt in ("nchw", "nhwc"):
        x = torch.randn(nb, ci, H, W, device=dev)
        wt = torch.randn(co, ci, kh, kw, device=dev) * 0.05
        bias = torch.randn(co, device=dev)
        if fmt == "nhwc":
            x = x.contiguous(memory_format=torch.channels_last)
            wt = wt.contiguous(memory_format=torch.channels_last)
        pad = (kh // 2, kw // 2)
        for _ in range(3):
            y = F.conv2d(x, wt, bias, stride=st, padding=pad)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            y = F.conv2d(x, wt, bias, stride=st, padding=pad)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        fl = 2.0 * nb * co * ci * kh * kw * y.shape[-1] * y.shape[-2]
        out.append(dict(name=name, fmt=fmt, ms=round(ms, 4), tflops=round(fl / ms / 1e9, 2)))
        print(f"{name:34s} {fmt}  {ms:8.3f} ms  {fl/ms/1e9:7.2f} TFLOP/s", flush=True)
# plain fp32 GEMM reference point (hipBLASLt/rocBLAS): N=38400 x K x M
for (m, k, n_) in ((38400, 1920, 256), (38400, 324, 256), (38400, 2304, 192)):
    a = torch.randn(m, k, device=dev)
    b = torch.randn(k, n_, device=dev)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c = a @ b
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"gemm {m}x{k}x{n_}: {ms:.3f} ms {2*m*k*n_/ms/1e9:.2f} TFLOP/s", flush=True)
    out.append(dict(name=f"gemm {m}x{k}x{n_}", fmt="-", ms=round(ms, 4), tflops=round(2 * m * k * n_ / ms / 1e9, 2)))
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
