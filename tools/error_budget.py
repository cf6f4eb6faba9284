#!/usr/bin/env python3
"""fp64 truth vs the fp32 CPU oracle vs the GPU (volume in fp16x3 or exact fp32 MFMA): first-iteration correlation features
and flow, as the feature magnitude grows.  Answers: is the GPU's distance to the CPU oracle the CPU's own fp32 round-off, the
fp16x3 volume, or the fp16x3 convolutions?   -> gpurun_out/error_budget.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rnnpose_oracle as orc                      # noqa: E402  (measurement tool)
from rnnpose_amd import synthetic as syn                      # noqa: E402
from rnnpose_amd.cfnet import AttrDict                        # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config   # noqa: E402
from rnnpose_amd.transformation import SE3Sequence            # noqa: E402

B, H, W = 2, 240, 320
torch.set_num_threads(min(os.cpu_count() or 1, 64))
d0 = syn.make_inputs(B, H, W, seed=9)
updW = syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))


def oracle(d, f64):
    old = orc._t
    if f64:
        torch.set_default_dtype(torch.float64)
        orc._t = lambda x, dtype=torch.float64: old(x, torch.float64)
    try:
        r = orc.refine(d, {"upd": updW}, outer=1, inner=1, capture=True)
    finally:
        orc._t = old
        torch.set_default_dtype(torch.float32)
    tr = r["trace"][0]
    return tr["flow_up"].double(), tr["corr"].double(), tr["net"].double()


def gpu(d, precision):
    D = lambda x: T(x).cuda()
    z3 = torch.zeros(B, 3, H, W, device="cuda")
    rend = SyntheticRenderer(syn_img=z3, image_crop=z3, cfea=D(d["ctx"]), geofea1=D(d["g1"]), geofea2_crop=D(d["g2"]), syn_depth=D(d["depth"]),
                             intrinsics_crop=D(d["K"]), fmap1=D(d["fmap1"]), fmap2=D(d["fmap2"]))
    cfg = default_config(RENDER_ITER_COUNT=1, ITER_COUNT=1, OPTIM_ITER_COUNT=1,
                         raft=AttrDict(pretrained_model=None, mixed_precision=False, fea_net="default", corr_precision=precision))
    ref = PoseRefiner(cfg, renderer=rend, use_graph=False).cuda().eval()
    ref.cf_net.update_block.load_state_dict({k: T(v) for k, v in updW.items()})
    out = ref(None, SE3Sequence(matrix=D(d["G0"])), D(d["K"]))
    eng = ref.cf_net.engine()
    corr = eng._b["corr"].permute(0, 3, 1, 2).double().cpu()        # the looked-up features of the (only) iteration, NHWC -> NCHW
    return out["flow"][0].double().cpu(), corr, eng.hidden_nchw().double().cpu()


res = {}
for scale in (1.0, 3.0, 8.0):
    d = dict(d0, fmap1=d0["fmap1"] * scale, fmap2=d0["fmap2"] * scale)
    f64, c64, n64 = oracle(d, True)
    f32, c32, n32 = oracle(d, False)
    row = {"corr_max": float(c64.abs().max()), "flow_max": float(f64.abs().max()),
           "cpu_fp32_vs_fp64": {"corr": float((c32 - c64).abs().max()), "hidden": float((n32 - n64).abs().max()), "flow": float((f32 - f64).abs().max())}}
    for prec in ("f16x3", "f32"):
        fg, cg, ng = gpu(d, prec)
        row[f"gpu_volume_{prec}_vs_fp64"] = {"corr": float((cg - c64).abs().max()), "hidden": float((ng - n64).abs().max()), "flow": float((fg - f64).abs().max())}
        row[f"gpu_volume_{prec}_vs_cpu_fp32"] = {"corr": float((cg - c32).abs().max()), "flow": float((fg - f32).abs().max())}
    res[f"feature_scale_{scale:g}"] = row
    print(scale, json.dumps(row))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "error_budget.json"), "w"), indent=1)
