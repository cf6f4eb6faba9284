#!/usr/bin/env python3
"""Fixed cost of one convolution launch: 1x1 convolutions with K = 32 .. 512 input channels at M = 19200 rows (B=4, 60x80),
N = 256 -- the MFMA work scales with K, whatever does not is prologue / epilogue / dispatch.  Run under rocprofv3
--kernel-trace and read tools/kernel_times.py (host timing cannot resolve a 20-us kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

B, h, w = 4, 60, 80
for cin in (32, 64, 128, 256, 512):
    for co in (64, 256):
        wt = torch.randn(co, cin, 1, 1, device="cuda") * 0.05
        pc = ops.PackedConv(wt, torch.zeros(co, device="cuda"), [cin])
        x = torch.randn(B, h, w, cin, device="cuda")
        out = torch.empty(B, h, w, co, device="cuda")
        for _ in range(5):
            ops.conv2d_nhwc(pc, [(x, 0)], (out, 0), ops.EPI_RELU)
        torch.cuda.synchronize()
        print("cin", cin, "cout", co)
