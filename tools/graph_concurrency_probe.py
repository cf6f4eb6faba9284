#!/usr/bin/env python3
"""Do two hipGraphs replayed on two streams overlap on this runtime?  20 half-batch GRU convolutions per chain (300 tiles
each: one launch leaves most of the 768 workgroup slots free), timed as: one chain alone, two chains eagerly on two
streams, two linear graphs on two streams, one two-branch graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402

dev = "cuda"
B, h, w, N = 4, 60, 80, 20
wt = torch.randn(128, 256, 1, 5, device=dev) * 0.03
pc = ops.PackedConv(wt, torch.zeros(128, device=dev), [128, 128])
bufs = [[torch.randn(B, h, w, 128, device=dev) for _ in range(3)] for _ in range(2)]


def chain(k):
    a, b, o = bufs[k]
    for _ in range(N):
        ops.conv2d_nhwc(pc, [(a, 0), (b, 0)], (o, 0), ops.EPI_RELU)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def two_eager():
    m = torch.cuda.current_stream()
    s1.wait_stream(m); s2.wait_stream(m)
    with torch.cuda.stream(s1):
        chain(0)
    with torch.cuda.stream(s2):
        chain(1)
    m.wait_stream(s1); m.wait_stream(s2)


graphs = []
for k in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(k)
    graphs.append(g)


def two_graphs():
    m = torch.cuda.current_stream()
    s1.wait_stream(m); s2.wait_stream(m)
    with torch.cuda.stream(s1):
        graphs[0].replay()
    with torch.cuda.stream(s2):
        graphs[1].replay()
    m.wait_stream(s1); m.wait_stream(s2)


gb = torch.cuda.CUDAGraph()
with torch.cuda.graph(gb):
    two_eager()

print(f"one chain alone (eager)        {timed(lambda: chain(0)):.3f} ms")
print(f"one chain alone (graph)        {timed(lambda: graphs[0].replay()):.3f} ms")
print(f"two chains, eager, two streams {timed(two_eager):.3f} ms")
print(f"two linear graphs, two streams {timed(two_graphs):.3f} ms")
print(f"one two-branch graph           {timed(lambda: gb.replay()):.3f} ms")
