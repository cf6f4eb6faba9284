#!/usr/bin/env python3
"""The producer / consumer pair of r04's reproducibility finding, alone, thousands of times: on stream A
    mask_upsample (writes the flow map) -> corr_weight (reads it)
while stream B runs the same pair on its own buffers (the two-chain schedule reduced to the two kernels in question), at the
half-batch headline shape (B = 4, 480 x 640).  Every iteration gets its own low-resolution flow (so every flow map differs from
what its buffer held before) and rotates through NB output buffers; after each block of NB iterations the stream is synchronised
and every recorded weight map is compared, bit for bit, with corr_weight of the flow map that is now in memory.  A weight map
that differs was computed from a flow map that was not (yet) what mask_upsample wrote.
    python tools/visibility_pair_probe.py [blocks=150]      PAIR_ONE_STREAM=1: no stream B (control);  RNNPOSE_LIB=...: another build
    PAIR_B=pair|tiny|producer|consumer: what stream B runs per iteration (the same pair (default); eight tiny element-wise kernels;
    only mask_upsample; only corr_weight; fill: a 20-MB torch fill; copy: a 20-MB torch copy)
    PAIR_CONSUMER=copy: the consumer is a plain device copy of the flow map (compared with the flow map after the block);
    PAIR_DELAY=<cycles>: a spin kernel (torch.cuda._sleep) between producer and consumer on stream A
Prints: launches, launches with a stale weight map, the stale pixel runs (image, row, first column, length)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops  # noqa: E402
from rnnpose_amd.streams import reserve  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 150
NB = 6
dev = "cuda"
B, H, W = 4, 480, 640
h, w = H // 8, W // 8
g = torch.Generator(device=dev)
g.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
one = os.environ.get("PAIR_ONE_STREAM", "0") != "0"
ss = reserve(torch.device(dev))
sA, sB = torch.cuda.current_stream(), ss.chain[0]
mhead = ops.PackedMaskHead(r(576, 256, 1, 1) * 0.09, r(576) * 0.1)
sigma = torch.ones(1, device=dev)


def make():
    g1, g2 = r(B, 32, H, W), r(B, 32, H, W)
    g1 /= g1.norm(dim=1, keepdim=True)
    g2 /= g2.norm(dim=1, keepdim=True)
    depth = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.3 + 0.9
    depth[:, :, : H // 4] = 0
    heads = r(B, h, w, 512).clamp_(min=0)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    base = torch.stack([0.4 * torch.sin(xs / 9.0) + 0.2 * torch.cos(ys / 6.0), 0.3 * torch.cos(xs / 11.0) - 0.3 * torch.sin(ys / 7.0)], -1)[None].repeat(B, 1, 1, 1).float()
    flows = [(base + 0.01 * k).contiguous() for k in range(NB)]            # low-resolution flow of rotation slot k
    return dict(g1=g1, g2=g2, depth=depth, heads=heads, flows=flows, up=[torch.zeros(B, 2, H, W, device=dev) for _ in range(NB)],
                wm=[torch.zeros(B, H, W, device=dev) for _ in range(NB)])


a, b = make(), make()
chk = torch.empty(B, H, W, device=dev)


cmode = os.environ.get("PAIR_CONSUMER", "weight")
delay = int(os.environ.get("PAIR_DELAY", "0"))
for d in (a, b):
    d["snap"] = [torch.zeros(B, 2, H, W, device=dev) for _ in range(NB)]
big1, big2 = torch.zeros(5 << 20, device=dev), torch.ones(5 << 20, device=dev)


def pair(d, k, shift):
    fl = d["flows"][k]
    fl += shift                                     # (a tiny kernel on the same stream: every pass writes new values)
    ops.mask_upsample(mhead, d["heads"], 256, fl, out=d["up"][k])
    if delay:
        torch.cuda._sleep(delay)
    if cmode == "copy":
        d["snap"][k].copy_(d["up"][k])
    else:
        ops.corr_weight(d["g1"], d["g2"], d["up"][k], d["depth"], sigma, out=d["wm"][k])


bmode = os.environ.get("PAIR_B", "pair")
tiny = torch.zeros(64, device=dev)


def stream_b(k):
    if bmode == "pair":
        pair(b, k, 1e-3)
    elif bmode == "tiny":
        for _ in range(8):
            tiny.add_(1.0)
    elif bmode == "producer":
        ops.mask_upsample(mhead, b["heads"], 256, b["flows"][k], out=b["up"][k])
    elif bmode == "c1x1":
        ops.conv1x1_resident(X["c1r"], (X["corr"], 0), (X["cor1"], 0))
    elif bmode == "conv":
        ops.conv2d_nhwc(X["pc"], [(X["cx"], 0)], (X["cy"], 0), ops.EPI_RELU)
    elif bmode == "flowfeat":
        ops.flow_features(X["c1"], X["w7"], X["b7"], X["flo1"], X["motion"], 126)
    elif bmode == "lm":
        ops.lm_step(b["up"][k], b["wm"][k], b["depth"], X["K"], X["G"])
    elif bmode == "fill":
        big1.fill_(float(k))
    elif bmode == "copy":
        big1.copy_(big2)
    elif bmode == "consumer":
        ops.corr_weight(b["g1"], b["g2"], b["up"][k], b["depth"], sigma, out=b["wm"][k])


for d in (a, b):
    d["wm_prev"] = [torch.zeros(B, H, W, device=dev) for _ in range(NB)]
X = {}
if bmode == "c1x1":
    X.update(c1r=ops.PackedConv1x1(r(256, 324, 1, 1) * 0.08, r(256) * 0.1), corr=r(B, h, w, 324), cor1=torch.empty(B, h, w, 256, device=dev))
elif bmode == "conv":
    X.update(pc=ops.PackedConv(r(256, 256, 1, 5) * 0.03, r(256) * 0.1, [256]), cx=r(B, h, w, 256), cy=torch.empty(B, h, w, 256, device=dev))
elif bmode == "flowfeat":
    X.update(c1=r(B, 2, h, w), w7=(r(128, 2, 7, 7) * 0.1).reshape(128, 98).t().contiguous(), b7=r(128) * 0.1,
             flo1=torch.empty(B, h, w, 128, device=dev), motion=torch.empty(B, h, w, 128, device=dev))
elif bmode == "lm":
    X.update(K=torch.tensor([[572.4, 0, W / 2], [0, 573.6, H / 2], [0, 0, 1]], device=dev).repeat(B, 1, 1), G=ops.se3_exp(r(B, 6) * 0.02))
launches = stale = diag = 0
runs = []
for blk in range(blocks):
    for k in range(NB):
        pair(a, k, 1e-3)
        if not one:
            with torch.cuda.stream(sB):
                stream_b(k)
    torch.cuda.synchronize()
    for d, tag in ((a, "A"),) if (one or bmode != "pair") else ((a, "A"), (b, "B")):
        for k in range(NB):
            if cmode == "copy":
                bad = (d["snap"][k] != d["up"][k]).any(1).nonzero()
            else:
                ops.corr_weight(d["g1"], d["g2"], d["up"][k], d["depth"], sigma, out=chk)
                bad = (chk != d["wm"][k]).nonzero()
            launches += 1
            if len(bad) and cmode != "copy" and diag < 6:
                # what ARE the differing weights?  (i) the weights this buffer held before this launch (wm_prev: the launch did not
                # write them), (ii) the weights of the flow map the buffer held before (the launch read old flow), (iii) neither
                diag += 1
                i, y, x = (bad[:, 0], bad[:, 1], bad[:, 2])
                got, want, prev = d["wm"][k][i, y, x], chk[i, y, x], d["wm_prev"][k][i, y, x]
                up_old = ops.mask_upsample(mhead, d["heads"], 256, d["flows"][k] - 1e-3)
                w_old = ops.corr_weight(d["g1"], d["g2"], up_old, d["depth"], sigma)[i, y, x]
                print(f"    diag {tag}{blk}.{k}: {len(bad)} px; equal to the buffer's previous content: {int((got == prev).sum())}; equal to the weight of the "
                      f"previous flow map: {int((got == w_old).sum())}; got {got[:3].tolist()} want {want[:3].tolist()} prev {prev[:3].tolist()} w(old flow) {w_old[:3].tolist()}")
            d["wm_prev"][k].copy_(d["wm"][k])
            if len(bad):
                stale += 1
                if len(runs) < 12:
                    i, y = int(bad[0, 0]), int(bad[0, 1])
                    cols = bad[(bad[:, 0] == i) & (bad[:, 1] == y)][:, 2]
                    runs.append(f"{tag}{blk}.{k}: {len(bad)} px, image {i} row {y} cols {int(cols.min())}..{int(cols.max())}")
    torch.cuda.synchronize()
print(f"{stale} of {launches} producer->consumer pairs read a flow map that was not the producer's output "
      f"({'one stream' if one else 'two streams, B runs: ' + bmode}, consumer {cmode}, delay {delay}, lib {os.path.basename(os.environ.get('RNNPOSE_LIB', 'in-tree'))})")
for s in runs:
    print("   ", s)
