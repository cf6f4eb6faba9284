#!/usr/bin/env python3
"""bench.py's own configuration (torch default-init weights, torch.rand inputs, B=8 480x640), first inner iteration: where does
the GPU's distance to the fp64 evaluation come from?  Stage by stage against the CPU oracle evaluated in fp64."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from oracle import rnnpose_oracle as orc                      # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config   # noqa: E402
from rnnpose_amd.transformation import SE3Sequence            # noqa: E402

B = int(os.environ.get("PROBE_B", 8))
H, W = 480, 640
torch.set_num_threads(min(os.cpu_count() or 1, 64))
dev = torch.device("cuda")
torch.manual_seed(0)
rend, K, G0 = bench.synth_views(B, H, W, dev, seed=0, with_encoder=True)
cfg = default_config(RENDER_ITER_COUNT=1, ITER_COUNT=1, OPTIM_ITER_COUNT=1)
ref = PoseRefiner(cfg, renderer=rend, use_graph=False).to(dev).eval()
out = ref(rend.views["image_crop"], SE3Sequence(matrix=G0.clone()), K)
v = rend.views
inp = {"ctx": v["cfea"], "g1": v["geofea1"], "g2": v["geofea2_crop"], "depth": v["syn_depth"], "K": K, "G0": G0, "sigma": ref.sigma[0].detach(),
       "img_render": v["syn_img"], "img_target": v["image_crop"]}
inp = {k: t.detach().cpu().numpy() for k, t in inp.items()}
Wt = {"upd": {k: p.detach().cpu().numpy() for k, p in ref.cf_net.update_block.state_dict().items()},
      "enc": {k: p.detach().cpu().numpy() for k, p in ref.image_fea_enc.fnet.state_dict().items()}}
res = {}
eng = ref.cf_net.engine()
_dense = lambda t: t.dense() if hasattr(t, 'dense') else t      # (the encoder hands split tensors to the volume build)
g = dict(fmap1=_dense(ref.cf_net.fmap1).double().cpu(), fmap2=_dense(ref.cf_net.fmap2).double().cpu(), corr=eng._b["corr"].permute(0, 3, 1, 2).double().cpu(),
         net=eng.hidden_nchw().double().cpu(), dflow=eng._b["delta"].permute(0, 3, 1, 2).double().cpu(), flow_up=out["flow"][0].double().cpu())
with orc.precision(torch.float64):
    f1, f2 = orc.image_encoder(Wt["enc"], inp["img_render"], inp["img_target"])
    t64 = orc.refine(inp, Wt, outer=1, inner=1, fast=True, capture=True)["trace"][0]
t32 = orc.refine(inp, Wt, outer=1, inner=1, fast=True, capture=True)["trace"][0]
f1c, f2c = orc.image_encoder(Wt["enc"], inp["img_render"], inp["img_target"])
mx = lambda a: float(a.abs().max())
res["fmap"] = dict(max=mx(f1), gpu_vs_fp64=max(mx(g["fmap1"] - f1), mx(g["fmap2"] - f2)), cpu32_vs_fp64=max(mx(f1c.double() - f1), mx(f2c.double() - f2)))
for k in ("corr", "net", "dflow", "flow_up"):
    res[k] = dict(max=mx(t64[k]), gpu_vs_fp64=mx(g[k] - t64[k]), cpu32_vs_fp64=mx(t32[k].double() - t64[k]))
d = (g["flow_up"] - t64["flow_up"]).abs()
i = np.unravel_index(int(d.argmax()), d.shape)
res["worst_pixel"] = dict(at=[int(x) for x in i], gpu=float(g["flow_up"][i]), fp64=float(t64["flow_up"][i]), cpu32=float(t32["flow_up"][i]),
                          q999=float(d.flatten().kthvalue(int(d.numel() * 0.999))[0]), n_over_5e5=int((d > 5e-5).sum()))
# the recurrent path alone: the GPU's OWN feature maps handed to the fp64 oracle
inp2 = {k: a for k, a in inp.items() if not k.startswith("img_")}
inp2["fmap1"], inp2["fmap2"] = g["fmap1"].float().numpy(), g["fmap2"].float().numpy()
with orc.precision(torch.float64):
    u64 = orc.refine(inp2, {"upd": Wt["upd"]}, outer=1, inner=1, fast=True, capture=True)["trace"][0]
res["recurrent_path_given_gpu_fmaps"] = {k: mx(g[k] - u64[k]) for k in ("corr", "net", "dflow", "flow_up")}
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bench_parity_probe.json"), "w"), indent=1)
