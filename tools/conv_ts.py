#!/usr/bin/env python3
"""Per-phase cycle counts inside the conv kernel (diagnostics build: RNNPOSE_HIPCC_EXTRA=-DRP_CONV_TS).
Prints, for wave 0 of a mid-problem tile, the s_memtime deltas of the first pipeline stages:
  loadA = issue of next activation rows, mma = LDS fragment reads + MFMAs, loadB = weight fragment requests,
  storeA = split + ds_write of the next tile (last tap only), bar = barrier wait."""
import os, sys, ctypes as C
assert "-DRP_CONV_TS" in os.environ.get("RNNPOSE_HIPCC_EXTRA", ""), "run with RNNPOSE_HIPCC_EXTRA=-DRP_CONV_TS"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import build, ops, _lib
build.build()
lib = _lib.load()
fn = lib.rnnpose_conv_dbg_timestamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
for name, (B, h, w), segs, co, kh, kw in (("zr1x5", (8, 60, 80), [128, 128, 128], 256, 1, 5),
                                          ("zr1x5 1 round", (8, 64, 64), [128, 128, 128], 256, 1, 5),
                                          ("zr1x5 tiny", (1, 32, 32), [128, 128, 128], 256, 1, 5),
                                          ("q1x5", (8, 60, 80), [128, 128, 128], 128, 1, 5),
                                          ("convc1", (8, 60, 80), [324], 256, 1, 1)):
    ci = sum(segs)
    wt = torch.randn(co, ci, kh, kw, device="cuda") * 0.02
    pc = ops.PackedConv(wt, torch.zeros(co, device="cuda"), segs)
    xs = [(torch.randn(B, h, w, c, device="cuda"), 0) for c in segs]
    out = torch.empty(B, h, w, co, device="cuda")
    for _ in range(3):
        ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU)
    torch.cuda.synchronize()
    ts = np.zeros((64, 8), dtype=np.int64)
    assert fn(ts.ctypes.data) == 0
    T = kw * kh if kh == 1 or kw == 1 else kw
    n = min(64, (ci + 31) // 32 * kh * kw) - 1
    d = np.diff(ts[:, :6], axis=1)            # loadA, mma, loadB, (storeA), (bar)
    print(f"== {name}: {n + 1} stages; stage length (cycles) mean {np.diff(ts[:n + 1, 0]).mean():.0f}")
    for i in range(min(n, 12)):
        last = (i % T == T - 1)
        la, mma, lb = d[i, 0], d[i, 1], d[i, 2]
        if last:
            print(f"  stage {i:2d}: loadA {la:5d} mma {mma:5d} loadB {lb:5d} storeA {ts[i,4]-ts[i,3]:5d} bar {ts[i,5]-ts[i,4]:5d}")
        else:
            print(f"  stage {i:2d}: loadA {la:5d} mma {mma:5d} loadB {lb:5d} rest {ts[i,5]-ts[i,3]:5d}")
    tot = ts[n, 0] - ts[0, 0]
    print(f"  totals over {n} stages: {tot} cycles; mma {d[:n,1].sum()} ({d[:n,1].sum()/tot:.0%}) loadA {d[:n,0].sum()/tot:.0%} loadB {d[:n,2].sum()/tot:.0%}")
