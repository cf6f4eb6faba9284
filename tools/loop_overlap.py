#!/usr/bin/env python3
"""Do the two batch halves of the inner loop really run concurrently?  Wall-clock timing (synchronize before / after, no
events inside, no profiler: both perturb the queues) of one outer iteration's inner loop (8 iterations, B=8, 480x640):
  half graph alone | both half graphs on two streams | eager two streams | eager one stream (unsplit)
Usage (GPU box): python tools/loop_overlap.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, default_config  # noqa: E402
from rnnpose_amd.transformation import SE3Sequence  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B, H, W = 8, 480, 640
rend, K, G0 = bench.synth_views(B, H, W, dev, 0, True)
ref = PoseRefiner(default_config(RENDER_ITER_COUNT=3, ITER_COUNT=8, OPTIM_ITER_COUNT=1), renderer=rend).to(dev).eval()
step = lambda: ref(rend.views["image_crop"], SE3Sequence(matrix=G0.clone()), K)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"plain steps: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per step")
v = rend.views
args = (v["syn_depth"], K, v["geofea1"], v["geofea2_crop"], torch.eye(4, device=dev).repeat(B, 1, 1, 1), 60, 80, 100.0, 1e-4, 8)


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


# in-situ: events on the MAIN stream only, at the outer-iteration boundaries of real steps
import rnnpose_amd.pose_refiner as PR
marks = []
orig_outer, orig_inner = ref._outer, ref._inner_loop
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def outer_w(views):
    marks.append(("o0", ev())); r = orig_outer(views); marks.append(("o1", ev())); return r
def inner_w(*a):
    r = orig_inner(*a); marks.append(("i1", ev())); return r
ref._outer, ref._inner_loop = outer_w, inner_w
for _ in range(3):
    step()
torch.cuda.synchronize()
ref._outer, ref._inner_loop = orig_outer, orig_inner
ms = [marks[i][1].elapsed_time(marks[i + 1][1]) for i in range(len(marks) - 1)]
print("in-situ segments (ms):", " ".join(f"{marks[i][0]}>{marks[i+1][0]}:{m:.2f}" for i, m in enumerate(ms[-9:])))
gr = ref._graph            # (record of the most recently used shape)
print("graphs:", [(str(st)) for _, st in gr["graphs"]])
g0, s0 = gr["graphs"][0]
g1, s1 = gr["graphs"][1]


def only(g, st):
    with torch.cuda.stream(st):
        g.replay()


print(f"half 0 graph alone            {timed(lambda: only(g0, s0)):.3f} ms")
print(f"half 1 graph alone            {timed(lambda: only(g1, s1)):.3f} ms")
print(f"both (refiner._inner_loop)    {timed(lambda: ref._inner_loop(*args)):.3f} ms")
ref.use_graph = False
print(f"eager, two streams            {timed(lambda: ref._inner_loop(*args)):.3f} ms")
ref.cf_net.engine().split_batch = False
print(f"eager, one chain (unsplit)    {timed(lambda: ref._inner_loop(*args)):.3f} ms")


# r06: does a START OFFSET between the two chains change the overlap?  (identical chains forked together run in phase: convolutions
# next to convolutions, tails next to tails; offset by part of an iteration one chain's tail runs under the other's convolutions)
if os.environ.get("LOOP_OVERLAP_STAGGER", "1") != "0":
    ref.use_graph = True
    ref.cf_net.engine().split_batch = True
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(1_000_000); e1.record(); torch.cuda.synchronize()
    us_per_mcycle = e0.elapsed_time(e1) * 1e3
    print(f"torch.cuda._sleep(1e6) = {us_per_mcycle:.1f} us")

    def both(delay_us):
        with torch.cuda.stream(s0):
            g0.replay()
        with torch.cuda.stream(s1):
            if delay_us:
                torch.cuda._sleep(int(delay_us / us_per_mcycle * 1e6))
            g1.replay()

    for d in (0, 50, 100, 150, 200, 250, 300, 400, 600):
        print(f"both graphs, chain 1 delayed by {d:4d} us   {timed(lambda: both(d), reps=8):.3f} ms")
