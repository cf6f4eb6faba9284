#!/usr/bin/env python3
"""Ablation timing of the conv kernel (results are WRONG for dbg != 0; only the time is meaningful).
bits: 1 no weight fragment loads, 4 no MFMA block (incl. LDS reads), 8 no activation LDS store, 16 no barriers,
32 force 64-wide tiles, 64 force 128-wide tiles, 256 no epilogue, 512 no activation global loads.
NOTE: below ~0.03 ms the Python/ctypes launch path (not the GPU) sets the floor of this loop.
r01 findings (zr 1x5, 0.17 ms): MFMA phase 0.083 (55 % of the fp16 MFMA rate inside the phase), weight loads 0.025,
activation split+store 0.017, activation loads 0.011, epilogue 0.019 -- the parts add up (no overlap across the two
resident workgroups); de-phasing them (s_sleep) or static s_setprio changed nothing."""
import os, sys, time, subprocess
if len(sys.argv) == 1:
    for dbg in (0, 1, 4, 8, 16, 256, 512, 4 + 256, 1 + 8 + 256, 31, 31 + 256 + 512, 32, 64):
        out = subprocess.run([sys.executable, __file__, str(dbg)], capture_output=True, text=True,
                             env=dict(os.environ, RNNPOSE_CONV_DBG=str(dbg))).stdout.strip()
        print(f"dbg={dbg:3d}  {out}", flush=True)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops
B, h, w = 8, 60, 80
res = []
for name, segs, co, kh, kw in (("zr1x5", [128, 128, 128], 256, 1, 5), ("q1x5", [128, 128, 128], 128, 1, 5),
                               ("heads3x3", [128], 512, 3, 3), ("convc1", [324], 256, 1, 1)):
    ci = sum(segs)
    x = torch.randn(B, h, w, ci, device="cuda")
    wt = torch.randn(co, ci, kh, kw, device="cuda") * 0.02
    pc = ops.PackedConv(wt, torch.zeros(co, device="cuda"), segs)
    xs, off = [], 0
    for c in segs:
        xs.append((x[..., off:off + c].contiguous(), 0)); off += c
    out = torch.empty(B, h, w, co, device="cuda")
    for _ in range(3): ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU)
    torch.cuda.synchronize()
    res.append(f"{name} {(time.perf_counter()-t0)/20*1e3:.3f}ms")
print("  ".join(res))
