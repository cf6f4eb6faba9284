#!/usr/bin/env python3
"""The packed-fp32 neighbour (tests/probes/pk_neighbour.hip) next to the volume kernel / the stem alone (RNNPOSE_LIB selects an ablation build).
Usage (GPU box): [RNNPOSE_LIB=...] [RNNPOSE_CORR_VARIANT=1] python tools/pk_neighbour_corr.py [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "probes"))
from pk_neighbour import Neighbour, next_to  # noqa: E402
from rnnpose_amd import ops  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 800
dev = torch.device("cuda:0")
nb = Neighbour()
g = torch.Generator(device=dev).manual_seed(5)
B, h, w, C = 8, 60, 80, 256
f1, f2 = torch.randn(B, h, w, C, device=dev, generator=g) * 2, torch.randn(B, h, w, C, device=dev, generator=g) * 2
s1, s2 = ops.SplitTensor(ops.split_hl(f1), 8.0), ops.SplitTensor(ops.split_hl(f2), 8.0)
buf, _ = ops.corr_pyramid_split(s1, s2, 4)
tag = os.path.basename(os.environ.get("RNNPOSE_LIB", "in-tree")) + " variant " + os.environ.get("RNNPOSE_CORR_VARIANT", "0")
print(f"{tag:40s} next to corr_pyramid_split: %d of %d differ (%d values)" % next_to(nb, lambda: ops.corr_pyramid_split(s1, s2, 4, out=buf), launches, per=40))
img = torch.rand(8, 3, 480, 640, device=dev, generator=g) * 255
from rnnpose_amd.cfnet import ImageFeaEncoder  # noqa: E402
enc = ImageFeaEncoder().to(dev).eval()
eng = enc.engine() if hasattr(enc, "engine") else None
W = eng._weights() if eng is not None else None
if W is not None:
    stem = lambda: ops.stem_conv(W["stem"], img, True)
    stem()
    print(f"{tag:40s} next to stem_conv:          %d of %d differ (%d values)" % next_to(nb, stem, launches, per=10))
