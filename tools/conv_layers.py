#!/usr/bin/env python3
"""The update-block convolutions as the engine launches them (hoisted GRU input, half batch B=4 and full batch B=8 at
60x80), each alone: time, executed fp16 TFLOP/s, fraction of the 2.5 PF peak.  Also the target of the SQ counter passes
(tools/pmc_sq.sh): `python tools/conv_layers.py 3` launches every layer 3 times and prints nothing else."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
h, w = 60, 80
shapes = [("convc1 1x1 324->256", [324], 256, 1, 1), ("convc2 3x3 256->192", [256], 192, 3, 3),
          ("convf2 3x3 128->64", [128], 64, 3, 3), ("conv 3x3 256->126", [192, 64], 126, 3, 3),
          ("gru zr 1x5 256->256", [128, 128], 256, 1, 5), ("gru q 1x5 256->128", [128, 128], 128, 1, 5),
          ("gru zr 5x1 256->256", [128, 128], 256, 5, 1), ("gru q 5x1 256->128", [128, 128], 128, 5, 1),
          ("heads 3x3 128->512", [128], 512, 3, 3), ("mask2 1x1 256->576", [256], 576, 1, 1),
          ("enc l1 3x3 64->64 @240x320", [64], 64, 3, 3)]
for B in (4, 8):
    for name, segs, co, kh, kw in shapes:
        hh, ww = (240, 320) if "@240" in name else (h, w)
        ci = sum(segs)
        wt = torch.randn(co, ci, kh, kw, device="cuda") * (2.0 / (ci * kh * kw)) ** 0.5
        pc = ops.PackedConv(wt, torch.randn(co, device="cuda"), segs)
        xs = [(torch.randn(B, hh, ww, c, device="cuda"), 0) for c in segs]
        out = torch.empty(B, hh, ww, (co + 3) // 4 * 4, device="cuda")
        run = lambda: ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU)
        if reps:
            for _ in range(reps):
                run()
            continue
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        fl = 3 * 2.0 * B * hh * ww * co * ci * kh * kw
        print(f"B={B} {name:28s} {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF executed  {fl/ms/1e9/2500*100:5.1f} % of peak", flush=True)
torch.cuda.synchronize()
