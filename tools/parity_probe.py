#!/usr/bin/env python3
"""Where does the first-iteration flow difference of the TIMED configuration (B=8, 480x640, encoder in the loop) come from?
GPU refiner vs the CPU oracle in its library-call form (what bench.py's parity block compares) and in its explicit form, with
the encoder in the loop and with the CPU encoder's feature maps handed to both sides.  -> gpurun_out/parity_probe.json"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rnnpose_oracle as orc                      # noqa: E402  (measurement tool: same standing as bench.py's cpu_baseline leg)
from rnnpose_amd import synthetic as syn                      # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config   # noqa: E402
from rnnpose_amd.transformation import SE3Sequence            # noqa: E402

B, H, W = int(os.environ.get("PROBE_B", 8)), 480, 640
torch.set_num_threads(min(os.cpu_count() or 1, 64))
dt = syn.make_inputs_t(B, H, W, seed=7, device="cuda", with_images=True)
d = {k: v.cpu().numpy() for k, v in dt.items()}
encW = syn.make_module_weights(orc.encoder_shapes(), seed=3)
updW = syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
res = {}


def gpu(with_encoder, fm=None):
    rend = SyntheticRenderer(syn_img=dt["img_render"], image_crop=dt["img_target"], cfea=dt["ctx"], geofea1=dt["g1"], geofea2_crop=dt["g2"],
                             syn_depth=dt["depth"], intrinsics_crop=dt["K"], fmap1=None if with_encoder else fm[0], fmap2=None if with_encoder else fm[1])
    ref = PoseRefiner(default_config(RENDER_ITER_COUNT=1, ITER_COUNT=1, OPTIM_ITER_COUNT=1), renderer=rend).cuda().eval()
    ref.cf_net.update_block.load_state_dict({k: T(v) for k, v in updW.items()})
    ref.image_fea_enc.fnet.load_state_dict({k: T(v) for k, v in encW.items()})
    out = ref(dt["img_target"], SE3Sequence(matrix=dt["G0"]), dt["K"])
    return out["flow"][0].float().cpu(), out["Ti_pred"].G.cpu(), ref


t0 = time.time()
f1c, f2c = orc.image_encoder(encW, d["img_render"], d["img_target"])
res["cpu_encoder_s"] = time.time() - t0
fg, Gg, ref = gpu(True)
dense = lambda t: t.dense() if hasattr(t, 'dense') else t          # (the encoder hands split tensors to the volume build)
f1g, f2g = dense(ref.cf_net.fmap1).cpu(), dense(ref.cf_net.fmap2).cpu()
res["fmap_max_abs"] = float(f1c.abs().max())
res["fmap_gpu_vs_cpu"] = [float((f1g - f1c).abs().max()), float((f2g - f2c).abs().max())]
dd = {k: v for k, v in d.items() if k not in ("fmap1", "fmap2")}
for form, fast in (("library", True), ("explicit", False)):
    t0 = time.time()
    w_enc = orc.refine(dd, {"upd": updW, "enc": encW}, outer=1, inner=1, fast=fast, capture=True)
    res[f"oracle_{form}_s"] = time.time() - t0
    df = (fg - w_enc["trace"][0]["flow_up"]).abs()
    i = np.unravel_index(int(df.argmax()), df.shape)
    res[f"encoder_in_loop/gpu_vs_{form}"] = dict(flow=float(df.max()), at=[int(x) for x in i], gpu=float(fg[i]), cpu=float(w_enc["trace"][0]["flow_up"][i]),
                                                 pose=float((Gg - w_enc["G"]).abs().max()), q999=float(df.flatten().kthvalue(int(df.numel() * 0.999))[0]))
    if form == "library":
        lib = w_enc["trace"][0]["flow_up"]
    else:
        res["library_vs_explicit"] = float((lib - w_enc["trace"][0]["flow_up"]).abs().max())
# identical feature maps (the CPU encoder's) on both sides: the recurrent path alone
fg2, Gg2, _ = gpu(False, (f1c.cuda(), f2c.cuda()))
d2 = dict(dd, fmap1=f1c.numpy(), fmap2=f2c.numpy())
for k in ("img_render", "img_target"):
    d2.pop(k)
for form, fast in (("library", True), ("explicit", False)):
    w2 = orc.refine(d2, {"upd": updW}, outer=1, inner=1, fast=fast, capture=True)
    df = (fg2 - w2["trace"][0]["flow_up"]).abs()
    res[f"given_fmaps/gpu_vs_{form}"] = dict(flow=float(df.max()), pose=float((Gg2 - w2["G"]).abs().max()))
res["flow_max_abs"] = float(fg.abs().max())
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "parity_probe.json"), "w"), indent=1)
