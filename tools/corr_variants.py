#!/usr/bin/env python3
"""The correlation-volume build alone at the bench shape (B=8, 60x80, C=256, 4 levels), under the library RNNPOSE_LIB names:
per launch of the C-ABI call (2 split pre-passes + the volume kernel) µs, GB/s of algorithmic bytes, max |diff| to the fp64
product on a sample of rows.   python tools/corr_variants.py [B h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnnpose_amd import ops
B, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 60, 80)
if os.environ.get("CORR_SUPERTILE"):
    from rnnpose_amd import _lib
    _lib.call("rnnpose_corr_supertile", int(os.environ["CORR_SUPERTILE"]))
C = 256
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)
f1 = torch.randn(B, h, w, C, generator=g).to(dev) * 2.0
f2 = torch.randn(B, h, w, C, generator=g).to(dev) * 2.0
buf, views = ops.corr_pyramid_nhwc(f1, f2, 4)
torch.cuda.synchronize()
N = h * w
rows = torch.arange(0, N, 37, device=dev)
ref = torch.einsum("brc,bnc->brn", f1.view(B, N, C)[:, rows].double(), f2.view(B, N, C).double()) / 16.0
got = views[0].view(B, N, N)[:, rows].double()
err0 = (got - ref).abs().max().item()
r1 = torch.nn.functional.avg_pool2d(ref.view(-1, 1, h, w), 2).view(B, len(rows), -1)
err1 = (views[1].view(B, N, -1)[:, rows].double() - r1).abs().max().item()
r3 = torch.nn.functional.avg_pool2d(torch.nn.functional.avg_pool2d(torch.nn.functional.avg_pool2d(ref.view(-1, 1, h, w), 2), 2), 2)
err3 = (views[3].view(B, N, -1)[:, rows].double() - r3.view(B, len(rows), -1)).abs().max().item()
n = 30
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(3):
    e0.record()
    for _ in range(n):
        ops.corr_pyramid_nhwc(f1, f2, 4, out=buf)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / n * 1e3)
t = min(ts)
s1, s2 = ops.SplitTensor(ops.split_hl(f1), 8.0), ops.SplitTensor(ops.split_hl(f2), 8.0)
b2, _ = ops.corr_pyramid_split(s1, s2, 4)
same = bool(torch.equal(b2, buf))
tss = []
for rep in range(3):
    e0.record()
    for _ in range(n):
        ops.corr_pyramid_split(s1, s2, 4, out=b2)
    e1.record(); torch.cuda.synchronize()
    tss.append(e0.elapsed_time(e1) / n * 1e3)
t2 = min(tss)
nbytes = 4.0 * buf.numel() + 2 * 4.0 * B * N * C
print(f"{os.path.basename(os.environ.get('RNNPOSE_LIB', 'in-tree')) + (' st=' + os.environ['CORR_SUPERTILE'] if os.environ.get('CORR_SUPERTILE') else ''):28s} {t:8.1f} us  {nbytes / t / 1e3:7.1f} GB/s  err L0 {err0:.2e} L1 {err1:.2e} L3 {err3:.2e} | split operands {t2:7.1f} us {nbytes / t2 / 1e3:7.1f} GB/s same={same}")
if os.environ.get("CORR_FILL"):
    for rep in range(2):
        e0.record()
        for _ in range(n):
            buf.fill_(1.0)
        e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / n * 1e3
    print(f"fill_ of the pyramid buffer     {tf:8.1f} us  {4.0 * buf.numel() / tf / 1e3:7.1f} GB/s (write only)")
