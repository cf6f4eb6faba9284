#!/usr/bin/env python3
"""Volume-free window features (AlternateCorrBlock, csrc/corr_alt.hip) against the materialised path at the bench shape:
per GRU iteration, on-the-fly lookup vs (volume build / iterations per outer iteration + pyramid lookup).
    python tools/corr_alt_bench.py [B h w]   -> one table row per batch size (default: 4 and 8 at 60x80, 1 at 30x30)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnnpose_amd import ops
from rnnpose_amd.corr import coords_grid

ITERS_PER_OUTER = 8
C = 256


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def row(B, h, w):
    g = torch.Generator(device="cpu").manual_seed(3)
    f1 = torch.randn(B, h, w, C, generator=g).cuda()
    f2 = torch.randn(B, h, w, C, generator=g).cuda()
    flow = torch.randn(B, 2, h, w, generator=g).cuda() * 3.0
    coords = coords_grid(B, h, w, device="cuda") + flow
    s1, s2 = ops.SplitTensor(ops.split_hl(f1), 8.0), ops.SplitTensor(ops.split_hl(f2), 8.0)
    buf, _ = ops.corr_pyramid_split(s1, s2, 4)
    out_m = torch.empty(B, h, w, 324, device="cuda")
    t_build = timed(lambda: ops.corr_pyramid_split(s1, s2, 4, out=buf))
    t_look = timed(lambda: ops.corr_lookup_nhwc_part(buf, coords, out_m, B, 0, B, 4, 4))
    pooled = ops.fmap_pyramid(f2, 4)
    out_a = torch.empty(B, h, w, 324, device="cuda")
    t_pool = timed(lambda: ops.fmap_pyramid(f2, 4))
    t_alt = timed(lambda: ops.corr_alt_lookup(f1, f2, pooled, coords, 4, 4, out=out_a))
    torch.cuda.synchronize()
    err = float((out_a - out_m).abs().max())
    mat = t_build / ITERS_PER_OUTER + t_look
    alt = t_pool / ITERS_PER_OUTER + t_alt
    print(f"B={B} {h}x{w}: materialised: build {t_build:7.1f} us / {ITERS_PER_OUTER} + lookup {t_look:6.1f} us = {mat:7.1f} us per iteration | "
          f"on the fly: pooling {t_pool:5.1f} us / {ITERS_PER_OUTER} + lookup {t_alt:7.1f} us = {alt:7.1f} us per iteration "
          f"({alt / mat:4.1f}x) | max |difference| {err:.1e}  (volume: {4.0 * buf.numel() / 1e6:.0f} MB never written)")


if len(sys.argv) >= 4:
    row(*(int(v) for v in sys.argv[1:4]))
else:
    for B, h, w in ((4, 60, 80), (8, 60, 80), (1, 30, 30)):
        row(B, h, w)
