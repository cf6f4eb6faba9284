#!/usr/bin/env python3
"""The update-block convolutions as the engine launches them (hoisted GRU input, half batch B=4 and full batch B=8 at
60x80), each alone: time, executed fp16 TFLOP/s, fraction of the 2.5 PF peak -- for fp32 sources (on-the-fly split) and for
split-tensor sources in every tile shape (1 = 128x64, 2 = 128x128 as 4 column waves, 3 = 128x128 as 2x2 waves).
Also the target of the SQ counter passes (tools/pmc_sq.sh): `python tools/conv_layers.py 3 [mode]` launches every layer 3
times in one mode (f32 | hl0 | hl1 | hl2 | hl3 | f32ks) and prints nothing else."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
only = sys.argv[2] if len(sys.argv) > 2 else None
flt = [f for f in os.environ.get("CONV_LAYERS_FILTER", "").split(",") if f]      # substrings of layer names to keep
batches = [int(b) for b in os.environ.get("CONV_LAYERS_B", "4,8,1").split(",")]
h, w = 60, 80
shapes = [("convc2 3x3 256->192", [256], 192, 3, 3),
          ("convf2 3x3 128->64", [128], 64, 3, 3), ("conv 3x3 256->126", [192, 64], 126, 3, 3),
          ("gru zr 1x5 256->256", [128, 128], 256, 1, 5), ("gru q 1x5 256->128", [128, 128], 128, 1, 5),
          ("gru zr 5x1 256->256", [128, 128], 256, 5, 1), ("gru q 5x1 256->128", [128, 128], 128, 5, 1),
          ("heads 3x3 128->512", [128], 512, 3, 3), ("inp 1x5 128->384", [128], 384, 1, 5),
          ("enc l1 3x3 64->64 @240x320", [64], 64, 3, 3), ("enc l2 3x3 96->96 @120x160", [96], 96, 3, 3),
          ("enc l3 3x3 128->128 @60x80", [128], 128, 3, 3)]
modes = [("f32", False, 0), ("f32t5", False, 5), ("hl5", True, 5), ("hl6", True, 6), ("f32t6", False, 6), ("f32t1", False, 1), ("f32t2", False, 2), ("hl0", True, 0), ("hl1", True, 1), ("hl2", True, 2), ("hl3", True, 3), ("hl4", True, 4), ("hl7", True, 7), ("f32t7", False, 7),
         ("f32ks", False, 0)]          # f32ks: fp32 sources with a K-split workspace (small launches split their K loop)
ksws = ops.conv_ksplit_workspace("cuda")
for B in batches:
    for name, segs, co, kh, kw in shapes:
        if flt and not any(f in name for f in flt):
            continue
        hh, ww = [int(v) for v in name.split("@")[1].split("x")] if "@" in name else ((30, 30) if B == 1 else (h, w))
        ci = sum(segs)
        wt = torch.randn(co, ci, kh, kw, device="cuda") * (2.0 / (ci * kh * kw)) ** 0.5
        pc = ops.PackedConv(wt, torch.randn(co, device="cuda"), segs)
        xf = [(torch.randn(B, hh, ww, c, device="cuda"), 0) for c in segs]
        xs = [(ops.split_hl(t), 0) for t, _ in xf]
        out = torch.empty(B, hh, ww, (co + 7) // 8 * 8, device="cuda")
        line = f"B={B} {name:28s}"
        for mname, hl, tile in modes:
            if only and mname not in only.split(","):
                continue
            if tile >= 5 and co <= 32:
                continue                      # (strips: c_out > 64; the 30 x 30 crops do not fill the chip with them)
            run = lambda: ops.conv2d_nhwc(pc, xs if hl else xf, (out, 0), ops.EPI_RELU, src_hl=hl, dst_hl=hl, tile=tile,
                                          ksplit_ws=ksws if mname.endswith("ks") else None)
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
            line += f" | {mname} {ms*1e3:6.1f} us {fl/ms/1e9/2500*100:4.1f}%"
        if not reps:
            print(line, flush=True)
torch.cuda.synchronize()
