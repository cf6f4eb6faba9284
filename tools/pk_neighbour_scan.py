#!/usr/bin/env python3
"""Which launches of the refinement loop disturb a packed-fp32 co-tenant (tests/probes/pk_neighbour.hip)?  The loop runs with ONE C-ABI
launch type skipped at a time (as tools/kernel_marginal.py: results wrong, schedule otherwise unchanged) next to the neighbour; a launch
type whose removal takes the differing launches to zero is a trigger.  Second pass: ONLY that launch type (plus what it needs) ...
Usage (GPU box): python tools/pk_neighbour_scan.py [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "probes"))
import bench  # noqa: E402
from pk_neighbour import Neighbour, next_to  # noqa: E402
from rnnpose_amd import _lib  # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, default_config  # noqa: E402
from rnnpose_amd.transformation import SE3Sequence  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 800
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
nb = Neighbour()
B, H, W = 8, 480, 640
rend, K, G0 = bench.synth_views(B, H, W, dev, 0, True)
ref = PoseRefiner(default_config(RENDER_ITER_COUNT=3, ITER_COUNT=8, OPTIM_ITER_COUNT=1), renderer=rend).to(dev).eval()
step = lambda: ref(rend.views["image_crop"], SE3Sequence(matrix=G0.clone()), K)
orig_call = _lib.call
skip, only, counts = set(), None, {}


def call(name, *args):
    counts[name] = counts.get(name, 0) + 1
    if name in skip or (only is not None and name not in only and name in launchers):
        return 0
    return orig_call(name, *args)


launchers = set()
_lib.call = call
for _ in range(2):
    step()
torch.cuda.synchronize()
names = [n for n in sorted(counts, key=lambda n: -counts[n]) if not any(k in n for k in ("pack", "workspace", "tiles_per", "products", "layout", "saturation", "stem_tiles"))]
launchers = set(names)
print("complete loop:                          %d of %d neighbour launches differ (%d values)" % next_to(nb, step, launches, per=400))
for n in names:
    skip.clear(); skip.add(n)
    ref._drop_graphs()
    step(); torch.cuda.synchronize()
    print(f"without {n.replace('rnnpose_', ''):32s}" + "%d of %d differ (%d values)" % next_to(nb, step, launches, per=400))
skip.clear()
for n in names:
    only = {n}
    ref._drop_graphs()
    try:
        step(); torch.cuda.synchronize()
        print(f"ONLY    {n.replace('rnnpose_', ''):32s}" + "%d of %d differ (%d values)" % next_to(nb, step, launches, per=400))
    except Exception as e:  # noqa: BLE001
        print(f"ONLY    {n:32s} failed: {e!r}"[:200])
