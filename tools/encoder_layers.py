#!/usr/bin/env python3
"""Every launch of ONE image set (B=8, 480x640 by default) through EncoderEngine, each alone on the chip, in launch order:
microseconds (mean of `reps` passes, HIP events around each generator step of EncoderEngine._forward_gen).
    python tools/encoder_layers.py [B] [H] [W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402
from rnnpose_amd.engine import EncoderEngine  # noqa: E402
from rnnpose_amd.extractor import BasicEncoder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
torch.manual_seed(0)
fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=0.0).cuda()
eng = EncoderEngine(fnet)
Wt = eng._weights()
img = torch.rand(B, 3, H, W, device="cuda") * 255
out = torch.empty(B, H // 8, W // 8, 256, device="cuda")
names = ["stem", "stem.finalize"]
for li, nb in ((1, 2), (2, 2), (3, 2)):
    for bi in range(nb):
        down = li > 1 and bi == 0
        names += [f"l{li}.{bi}.c1", f"l{li}.{bi}.c1.finalize"] + ([f"l{li}.{bi}.down", f"l{li}.{bi}.down.finalize"] if down else []) + [f"l{li}.{bi}.c2", f"l{li}.{bi}.apply"]
names += ["out 1x1 (split)"]
reps = 10
tot = {}
for rep in range(reps + 2):
    g = eng._forward_gen(Wt, img, out, True, True, ks=None)
    evs = [torch.cuda.Event(enable_timing=True)]
    evs[0].record()
    n = 0
    try:
        while True:
            next(g)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
    except StopIteration:
        pass
    torch.cuda.synchronize()
    if rep >= 2:
        for i in range(len(evs) - 1):
            tot[i] = tot.get(i, 0.0) + evs[i].elapsed_time(evs[i + 1]) * 1e3
s = 0.0
for i in sorted(tot):
    t = tot[i] / reps
    s += t
    print(f"{names[i] if i < len(names) else '?':24s} {t:8.1f} us", flush=True)
print(f"{'sum':24s} {s:8.1f} us   (B={B}, {H}x{W}, one image set, launches back to back on one stream)")
