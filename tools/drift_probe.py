#!/usr/bin/env python3
"""Where does GPU-vs-oracle error come from?  Teacher-forced per-stage errors + free-running drift on the
loop_128 golden case (B=2, 128x128, 1 outer x 3 inner).  GPU box only; prints a table, writes JSON."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rnnpose_oracle as orc  # noqa: E402
from rnnpose_amd import ops, synthetic as syn  # noqa: E402
from rnnpose_amd.cfnet import GRU_CFUpdator  # noqa: E402

dev = "cuda"
D = lambda x: (torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x).to(dev, torch.float32)
md = lambda a, b: float((a.detach().cpu().double() - (b.detach().cpu() if torch.is_tensor(b) else torch.from_numpy(b)).double()).abs().max())
res = {}


def rec(k, v):
    res[k] = v
    print(f"{k:60s} {v:.3e}", flush=True)


B, H, W = 2, 128, 128
d = syn.make_inputs(B, H, W, seed=21)
Wt = syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0)
net = GRU_CFUpdator(dict(pretrained_model=None)).to(dev).eval()
net.update_block.load_state_dict({k: torch.from_numpy(v) for k, v in Wt.items()})
trace = orc.refine(d, {"upd": Wt}, outer=1, inner=3, optim_iters=1, capture=True)["trace"]

# ---- per-conv-layer error: same inputs, GPU conv vs CPU conv ----
pyr = orc.corr_pyramid(d["fmap1"], d["fmap2"])
hid, cinp = orc.context_prep(d["ctx"])
c0 = orc.coords_grid_lowres(B, H // 8, W // 8)
G = torch.eye(4).repeat(B, 1, 1, 1)
flow_init, _ = orc.induced_flow(d["depth"], d["K"], G)
c1 = orc.flow_init_to_coords1(flow_init)
corr = orc.corr_lookup(pyr, c1)
x_in = {"encoder.convc1": (corr, 0), "encoder.convf1": (c1 - c0 + 0.3, 3)}
with torch.no_grad():
    t = lambda a: torch.from_numpy(Wt[a])
    def conv_err(name, x, pad):
        ref = F.conv2d(x, t(name + ".weight"), t(name + ".bias"), padding=pad)
        got = F.conv2d(D(x), D(Wt[name + ".weight"]), D(Wt[name + ".bias"]), padding=pad)
        rec(f"conv {name} |gpu-cpu| (max|ref|={float(ref.abs().max()):.2f})", md(got, ref))
        return ref
    cor = F.relu(conv_err("encoder.convc1", corr, 0))
    cor2 = F.relu(conv_err("encoder.convc2", cor, 1))
    flo = F.relu(conv_err("encoder.convf1", c1 - c0 + 0.37, 3))
    flo2 = F.relu(conv_err("encoder.convf2", flo, 1))
    out = F.relu(conv_err("encoder.conv", torch.cat([cor2, flo2], 1), 1))
    hx = torch.cat([hid, cinp, out, c1 - c0], 1)
    conv_err("gru.convz1", hx, (0, 2))
    conv_err("gru.convq2", hx, (2, 0))
    fh = F.relu(conv_err("flow_head.conv1", hid, 1))
    conv_err("flow_head.conv2", fh, 1)
    mk = F.relu(conv_err("mask.0", hid, 1))
    conv_err("mask.2", mk, 0)
    # matmul-based 1x1 for comparison
    got = torch.einsum("oc,bchw->bohw", D(Wt["encoder.convc1.weight"])[:, :, 0, 0], D(corr)) + D(Wt["encoder.convc1.bias"])[None, :, None, None]
    ref = F.conv2d(corr, t("encoder.convc1.weight"), t("encoder.convc1.bias"))
    rec("einsum 1x1 convc1 |gpu-cpu|", md(got, ref))

# ---- my kernels, teacher forced with the oracle state of iteration 0 ----
buf, views = ops.corr_pyramid(D(d["fmap1"]), D(d["fmap2"]))
for l in range(4):
    rec(f"corr_pyramid level {l}", md(views[l][:, 0], pyr[l]))
nk, ik = ops.context_prep(D(d["ctx"]), H // 8, W // 8)
rec("context_prep net", md(nk, hid)); rec("context_prep inp", md(ik, cinp))
rec("induced_coords_lowres (G=I)", md(ops.induced_coords_lowres(D(d["depth"]), D(d["K"]), D(G), H // 8, W // 8, 1e-5), c1))
# lookup on the ORACLE pyramid copied into the buffer (isolates the lookup kernel)
offs, hl, wl = ops.pyramid_layout(B, H // 8, W // 8)
buf2 = torch.cat([D(p).reshape(-1) for p in pyr])
for it, tr in enumerate(trace):
    pass
rec("corr_lookup @grid coords (oracle pyramid)", md(ops.corr_lookup(buf2, D(c1)), corr))
csub = c1 + torch.from_numpy(syn.uniform("p", tuple(c1.shape), 3, -2.5, 2.5))
rec("corr_lookup @subpixel coords (oracle pyramid)", md(ops.corr_lookup(buf2, D(csub)), orc.corr_lookup(pyr, csub)))

# ---- update block, teacher forced per iteration, and free-running drift ----
with torch.no_grad():
    st_h, st_c1 = hid, c1
    net.net, net.inp = D(hid), D(cinp)
    net.corr_fn = type("C", (), {"__call__": lambda s, c: ops.corr_lookup(buf, c)})()
    fr_c1 = D(c1)
    for it, tr in enumerate(trace):
        # teacher forced: oracle inputs of this iteration
        corr_o = orc.corr_lookup(pyr, st_c1)
        n2, m2, df2 = net.update_block(D(st_h), D(cinp), D(corr_o), D(st_c1 - c0))
        rec(f"it{it} update_block TF net", md(n2, tr["net"]))
        rec(f"it{it} update_block TF mask", md(m2, tr["mask"]))
        rec(f"it{it} update_block TF dflow", md(df2, tr["dflow"]))
        up = ops.convex_upsample(D(st_c1 + tr["dflow"] - c0), D(tr["mask"]))
        rec(f"it{it} convex_upsample TF", md(up, tr["flow_up"]))
        wk = ops.corr_weight(D(d["g1"]), D(d["g2"]), D(tr["flow_up"]), D(d["depth"]), D(d["sigma"]))
        rec(f"it{it} corr_weight TF", md(wk, tr["weight"]))
        # free running
        fr_c1, fr_up = net.step(D(c0), fr_c1)
        rec(f"it{it} FREE flow_up drift", md(fr_up, tr["flow_up"]))
        rec(f"it{it} FREE net drift", md(net.net, tr["net"]))
        # next oracle state: coords from the oracle's pose
        st_h = tr["net"]
        fi, _ = orc.induced_flow(d["depth"], d["K"], tr["Tij"])
        st_c1 = orc.flow_init_to_coords1(fi)
        fr_c1 = D(st_c1)   # keep the pose teacher-forced; only the hidden state runs free
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/null", "w"), indent=1)
