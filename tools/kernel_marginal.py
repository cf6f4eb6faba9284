#!/usr/bin/env python3
"""What does each launch contribute to the STEP (not: how long does it run alone)?  TIMING ABLATION: the named C-ABI launches are
skipped (their outputs keep what the previous step left there, so values stay sane and results are WRONG) and the whole step is timed as
bench.py times it -- graphs re-captured per configuration.  The difference to the complete step is the most a fusion / removal of that
launch can gain on the default schedule (two loop chains + two encoder streams: a launch that runs under another chain's convolutions
contributes less than its duration).
Usage (GPU box): python tools/kernel_marginal.py [B H W outer inner]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rnnpose_amd import _lib, ops  # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, default_config  # noqa: E402
from rnnpose_amd.transformation import SE3Sequence  # noqa: E402

a = [int(v) for v in sys.argv[1:6]] + [8, 480, 640, 3, 8][len(sys.argv) - 1:]
B, H, W, OUTER, INNER = a
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
rend, K, G0 = bench.synth_views(B, H, W, dev, 0, True)
ref = PoseRefiner(default_config(RENDER_ITER_COUNT=OUTER, ITER_COUNT=INNER, OPTIM_ITER_COUNT=1), renderer=rend).to(dev).eval()
step = lambda: ref(rend.views["image_crop"], SE3Sequence(matrix=G0.clone()), K)

orig_call = _lib.call
skip = {}          # name -> predicate over the argument tuple (or True)
counts = {}


def call(name, *args):
    counts[name] = counts.get(name, 0) + 1
    s = skip.get(name)
    if s is not None and (s is True or s(args)):
        return 0
    return orig_call(name, *args)


_lib.call = call


def timed(reps):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


reps = 20 if B * H * W > 500_000 else 60
base = timed(reps)
names = sorted(counts, key=lambda n: -counts[n])
print(f"B={B} {H}x{W} {OUTER}x{INNER}: complete step {base:.3f} ms; C-ABI launches seen while capturing / warming up: "
      + ", ".join(f"{n.replace('rnnpose_', '')} x{counts[n]}" for n in names))
stats_only = lambda args: not args[0].value          # rnnpose_instnorm_tiles_nhwc_f32 with x == NULL: the finalize-only launch
cases = [("instnorm finalize-only launches", {"rnnpose_instnorm_tiles_nhwc_f32": stats_only})]
for n in names:
    if any(k in n for k in ("conv2d", "conv_strip", "pack", "workspace", "tiles_per", "products", "split_hl")):
        continue
    cases.append((n.replace("rnnpose_", ""), {n: True}))
for label, sk in cases:
    skip.clear()
    skip.update(sk)
    ref._drop_graphs()
    try:
        t = timed(reps)
        print(f"  without {label:44s} {t:8.3f} ms   ({base - t:+.3f} ms = {100 * (base - t) / base:+.2f} %)")
    except Exception as e:  # noqa: BLE001
        print(f"  without {label:44s} failed: {e!r}")
skip.clear()
ref._drop_graphs()
print(f"complete step again {timed(reps):.3f} ms")
