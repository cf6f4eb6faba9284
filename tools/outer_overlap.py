#!/usr/bin/env python3
"""Do the two encoder image sets of the per-outer-iteration unit (encoder x 2 + volume build + context prep) run concurrently?
Wall-clock timing (synchronize before / after) of: the captured outer graph | the eager unit on two streams | each image set alone |
both sets eager on two streams | one merged batch of 2B images.  (r06: the two-BRANCH outer graph replayed almost serially.)
Usage (GPU box): python tools/outer_overlap.py"""
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


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


v = rend.views
print(f"outer unit, as the refiner runs it (graphs) {timed(lambda: ref._outer(v)):.3f} ms")
print(f"outer unit, eager                          {timed(lambda: ref._outer_body(v)):.3f} ms")
enc = ref.image_fea_enc
print(f"encoder, both sets (eager, engine's choice) {timed(lambda: enc.forward_split(v['syn_img'], v['image_crop'])):.3f} ms")
eng = enc.engine()
print(f"encoder, ONE set alone (eager)              {timed(lambda: eng([v['syn_img']], normalize=True, split_out=True)):.3f} ms")
eng.merge_sets = True
print(f"encoder, both sets as one batch of {2 * B}       {timed(lambda: enc.forward_split(v['syn_img'], v['image_crop'])):.3f} ms")
eng.merge_sets = None
f1, f2 = enc.forward_split(v["syn_img"], v["image_crop"])
print(f"volume build + context prep + hoist (eager) {timed(lambda: ref.cf_net.prepare(f1, f2, v['cfea'])):.3f} ms")
