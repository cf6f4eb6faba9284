#!/usr/bin/env python3
"""Free-running drift over the REAL horizon (3 outer x 8 inner iterations): GPU path vs CPU oracle on the same inputs,
per iteration max |pose difference| and max |flow difference| (B=2, 128x160 and B=1, 240x240).  GPU box only; prints a table
and writes gpurun_out/drift_probe.json (copied to profiles/ by tools/summarize_profiles.py).
Background: per-stage errors are ~1e-6, but the pose feeds back through the projection with
d(flow)/d(pose) ~ f + x^2/f px per unit, so the flow of later iterations moves by a few 1e-4 px while the pose itself stays
within its 1e-5 tolerance."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rnnpose_oracle as orc  # noqa: E402
from rnnpose_amd import synthetic as syn  # noqa: E402
from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config  # noqa: E402
from rnnpose_amd.transformation import SE3Sequence  # noqa: E402

dev = "cuda"
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
out = {}
for (B, H, W, seed) in ((2, 128, 160, 22), (1, 240, 240, 23)):
    d = syn.make_inputs(B, H, W, seed=seed)
    Wt = syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0)
    want = orc.refine(d, {"upd": Wt}, outer=3, inner=8, optim_iters=1, capture=True)
    z3 = torch.zeros(B, 3, H, W, device=dev)
    rend = SyntheticRenderer(syn_img=z3, image_crop=z3, cfea=t(d["ctx"]), geofea1=t(d["g1"]), geofea2_crop=t(d["g2"]),
                             syn_depth=t(d["depth"]), intrinsics_crop=t(d["K"]), fmap1=t(d["fmap1"]), fmap2=t(d["fmap2"]))
    ref = PoseRefiner(default_config(RENDER_ITER_COUNT=3, ITER_COUNT=8, OPTIM_ITER_COUNT=1), renderer=rend).to(dev).eval()
    ref.cf_net.update_block.load_state_dict({k: torch.from_numpy(v) for k, v in Wt.items()}, strict=True)
    res = ref(None, SE3Sequence(matrix=t(d["G0"])), t(d["K"]))
    rows = []
    for i, (tr, Tij, fl) in enumerate(zip(want["trace"], ref.residual_pose_history, ref.flow_history)):
        dG = float((Tij.G.cpu() - tr["Tij"]).abs().max())
        dF = float((fl[0].cpu() - tr["flow_up"]).abs().max())
        rows.append({"iteration": i, "outer": i // 8, "inner": i % 8, "max_abs_dpose": dG, "max_abs_dflow_px": dF,
                     "max_abs_flow_px": float(tr["flow_up"].abs().max())})
        print(f"{B}x{H}x{W} it {i:2d}  |dG| {dG:.2e}  |dflow| {dF:.2e} px  (|flow| up to {rows[-1]['max_abs_flow_px']:.1f} px)", flush=True)
    dfin = float((res["Ti_pred"].G.cpu() - want["G"]).abs().max())
    print(f"{B}x{H}x{W} final pose |dG| {dfin:.2e}")
    out[f"B{B}_{H}x{W}"] = {"per_iteration": rows, "final_pose_max_abs_diff": dfin,
                            "worst_dpose": max(r["max_abs_dpose"] for r in rows), "worst_dflow_px": max(r["max_abs_dflow_px"] for r in rows)}
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "drift_probe.json"), "w"), indent=1)
