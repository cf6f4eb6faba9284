#!/usr/bin/env python3
"""Does running two half-batch convolution chains on two streams beat one full-batch chain?  (Single-round kernels lose
~30 % to ramp-up / lock-step prologues and epilogues; two out-of-phase kernels in flight could fill those gaps.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops

h, w = 60, 80
def mk(B, segs, co, kh, kw):
    ci = sum(segs)
    wt = torch.randn(co, ci, kh, kw, device="cuda") * 0.02
    pc = ops.PackedConv(wt, torch.zeros(co, device="cuda"), segs)
    xs = [(torch.randn(B, h, w, c, device="cuda"), 0) for c in segs]
    out = torch.empty(B, h, w, co, device="cuda")
    return lambda: ops.conv2d_nhwc(pc, xs, (out, 0), ops.EPI_RELU)

def bench(fns_per_stream, n=20):
    streams = [torch.cuda.Stream() for _ in fns_per_stream]
    def run():
        ev = torch.cuda.Event(); ev.record()
        for s, fns in zip(streams, fns_per_stream):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                for _ in range(n):
                    for f in fns: f()
        for s in streams: torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    run(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3 / n

for name, segs, co, kh, kw in (("zr 1x5", [128, 128, 128], 256, 1, 5), ("q 1x5", [128, 128, 128], 128, 1, 5),
                               ("convc2 3x3", [256], 192, 3, 3), ("convc1 1x1", [324], 256, 1, 1), ("heads 3x3", [128], 512, 3, 3)):
    full = bench([[mk(8, segs, co, kh, kw)]])
    two = bench([[mk(4, segs, co, kh, kw)], [mk(4, segs, co, kh, kw)]])
    four = bench([[mk(2, segs, co, kh, kw)] for _ in range(4)])
    print(f"{name:12s} one stream B=8: {full:.4f} ms   two streams B=4+4: {two:.4f} ms   four streams B=2x4: {four:.4f} ms", flush=True)
# a dependent chain of different convs (as in the update step), full batch vs two half-batch chains
chain = lambda B: [mk(B, [128, 128, 128], 256, 1, 5), mk(B, [128, 128, 128], 128, 1, 5), mk(B, [128, 128, 128], 256, 5, 1),
                   mk(B, [128, 128, 128], 128, 5, 1), mk(B, [128], 512, 3, 3), mk(B, [256], 576, 1, 1)]
print(f"GRU+heads chain: one stream B=8 {bench([chain(8)], n=5):.4f} ms   two streams B=4+4 {bench([chain(4), chain(4)], n=5):.4f} ms")
