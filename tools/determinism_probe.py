#!/usr/bin/env python3
"""Run-to-run reproducibility of the refinement loop at the headline shape (B=8, 480x640, feature maps given, 1 outer x 2 inner
iterations, eager launches): N fresh PoseRefiner instances on identical inputs, every per-iteration output (flow, weight map, G, H,
b, xi) compared bit for bit with the first instance; for every instance the recorded weight map is also recomputed from the FINAL
flow of the same iteration (ops.corr_weight) and compared.

r04 finding (profiles/r04_determinism.txt): with the TWO-chain schedule (the default until then) (two half-batch loops on two streams) every instance
differs from the first in 15-32 pixels of a weight map -- one or two 64-byte runs of one image row -- and in what follows from them
(pose 2e-9, flow <= 3.5e-5 px after two iterations); the recorded weight of those pixels is NOT the weight of the final flow: the
weight kernel saw other (close) flow values in that sector.  With ONE chain 12 of 12 instances are bit-identical; an explicit agent-scope release at the
end of mask_upsample's workgroups cut the two-chain rate from 14 of 14 to 0-1 of 12 (+12 us per launch): with two streams active, the
end-of-kernel write-back of one XCD's L2 is not always in time for the next kernel of the same stream on another XCD.  Since then the
loop runs as one chain and the encoder as one batch by default.
r05: shapes and launch modes by environment -- DET_B / DET_H / DET_W (default 8 x 480 x 640; DET_B=1 DET_H=240 DET_W=240 is the reference's
own working size, where the flow-feature / flow-head side chain may run on a helper stream: RNNPOSE_SIDE_STREAM), DET_ITERS inner
iterations (2), DET_GRAPH=1 hipGraph replay instead of eager launches.
    python tools/determinism_probe.py [trials]        (RNNPOSE_SPLIT_BATCH=1: the two-chain schedule; DET_ENCODER=1: encoder in the loop)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops, synthetic as syn  # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config  # noqa: E402
from rnnpose_amd.transformation import SE3Sequence  # noqa: E402
from oracle import rnnpose_oracle as orc  # noqa: E402  (weights generator only)

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, H, W = int(os.environ.get("DET_B", "8")), int(os.environ.get("DET_H", "480")), int(os.environ.get("DET_W", "640"))
ITERS, GRAPH = int(os.environ.get("DET_ITERS", "2")), os.environ.get("DET_GRAPH", "0") != "0"
d = syn.make_inputs(B, H, W, seed=21)
D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
z3 = torch.zeros(B, 3, H, W, device="cuda")
kw = dict(syn_img=z3, image_crop=z3, cfea=D(d["ctx"]), geofea1=D(d["g1"]), geofea2_crop=D(d["g2"]), syn_depth=D(d["depth"]),
          intrinsics_crop=D(d["K"]), fmap1=D(d["fmap1"]), fmap2=D(d["fmap2"]))
wts = {k: torch.from_numpy(v) for k, v in syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0).items()}
G0, K = D(d["G0"]), D(d["K"])
keep = []           # (keeps the allocator from handing every instance the same blocks)


if os.environ.get("DET_ENCODER", "0") != "0":        # encoder in the loop: images instead of given feature maps
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    kw["syn_img"] = torch.rand(B, 3, H, W, device="cuda", generator=gen) * 255
    kw["image_crop"] = torch.rand(B, 3, H, W, device="cuda", generator=gen) * 255
    del kw["fmap1"], kw["fmap2"]


def run(tag):
    torch.manual_seed(0)                                # (identical encoder initialisation in every instance)
    cfg = default_config(RENDER_ITER_COUNT=1, ITER_COUNT=ITERS, OPTIM_ITER_COUNT=1)
    ref = PoseRefiner(cfg, renderer=SyntheticRenderer(**kw), use_graph=GRAPH).cuda().eval()
    ref.cf_net.update_block.load_state_dict(wts)
    rec = {}
    orig = PoseRefiner._loop_buffers

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        r["big"].zero_(), r["small"].zero_(), r["coords"].zero_()
        rec["bufs"] = r
        return r
    PoseRefiner._loop_buffers = spy
    if os.environ.get("DET_STREAM", "0") != "0":          # the whole forward on a pool stream instead of the default (null) stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = ref(None, SE3Sequence(matrix=G0), K)
        torch.cuda.current_stream().wait_stream(side)
    else:
        out = ref(None, SE3Sequence(matrix=G0), K)
    PoseRefiner._loop_buffers = orig
    torch.cuda.synchronize()
    res = {"flow_last": out["flow_last"].clone()}
    for i, vw in enumerate(rec["bufs"]["views"]):
        for nm, t in zip(("flow_up", "wmap", "G", "Hm", "bv", "xi", "info"), vw):
            res[f"it{i}.{nm}"] = t.clone()
        chk = ops.corr_weight(kw["geofea1"], kw["geofea2_crop"], vw[0], kw["syn_depth"], ref.sigma[0])
        bad = (chk != vw[1]).nonzero()
        if len(bad):
            print(f"  instance {tag} it{i}: recorded weight != corr_weight(final flow) in {len(bad)} pixels, first {bad[0].tolist()}", flush=True)
    keep.append(torch.full((1 << 26,), float(len(keep)), device="cuda"))
    return res


first = run("first")
n = 0
for t in range(trials):
    cur = run(t)
    bad = [k for k in first if int((first[k] != cur[k]).sum())]
    n += bool(bad)
    if bad:
        print(f"instance {t}: differs from the first in {bad[:6]}; flow after two iterations by {float((first['flow_last'] - cur['flow_last']).abs().max()):.3g} px, "
              f"pose by {float((first[f'it{ITERS - 1}.G'] - cur[f'it{ITERS - 1}.G']).abs().max()):.3g}", flush=True)
print(f"{n} of {trials} instances differ from the first (B={B} {H}x{W}, {ITERS} iterations, {'graph replay' if GRAPH else 'eager'}, "
      f"{'two chains' if os.environ.get('RNNPOSE_SPLIT_BATCH', '0') != '0' else 'one chain'}, side stream {os.environ.get('RNNPOSE_SIDE_STREAM', '1')}, "
      f"lib {os.path.basename(os.environ.get('RNNPOSE_LIB', 'in-tree'))}, AMD_OPT_FLUSH={os.environ.get('AMD_OPT_FLUSH', 'default')})", flush=True)
