#!/usr/bin/env python3
"""Co-tenancy probe: a plain -O3 kernel made of PACKED fp32 arithmetic (tests/probes/pk_neighbour.hip -- an
integrator's own code, NOT built with this library's -fno-slp-vectorize) runs on its own stream while the refinement loop runs on the
library's streams; every launch of the neighbour is compared, on the device, with its solo output.  Prints how many launches differ
  * next to nothing (control), * next to the full loop, * next to mask_upsample / conv1x1_resident alone (the two
  v_mfma_f32_16x16x32_f16 kernels), * next to a strip convolution alone (v_mfma_f32_32x32x16_f16).
Usage (GPU box): python tools/pk_neighbour_probe.py [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


sys.path.insert(0, os.path.join(ROOT, "tests", "probes"))
from pk_neighbour import Neighbour, next_to  # noqa: E402


def main():
    import bench
    from rnnpose_amd import ops
    from rnnpose_amd.pose_refiner import PoseRefiner, default_config
    from rnnpose_amd.transformation import SE3Sequence
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    nb = Neighbour()
    print("neighbour:", nb.so, f"{nb.nwg} workgroups x 256 threads, chain {nb.chain}")
    print("control (nothing else on the chip):      %d of %d launches differ (%d values)" % next_to(nb, lambda: None, launches))
    B, H, W = 8, 480, 640
    rend, K, G0 = bench.synth_views(B, H, W, dev, 0, True)
    ref = PoseRefiner(default_config(RENDER_ITER_COUNT=3, ITER_COUNT=8, OPTIM_ITER_COUNT=1), renderer=rend).to(dev).eval()
    step = lambda: ref(rend.views["image_crop"], SE3Sequence(matrix=G0.clone()), K)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    print("next to the refinement loop (B=8, 3x8):  %d of %d launches differ (%d values)" % next_to(nb, step, launches, per=400))
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    h, w = H // 8, W // 8
    Bh = 4
    mhead = ops.PackedMaskHead(r(576, 256, 1, 1) * 0.09, r(576) * 0.1)
    heads, flow_lr, up = r(Bh, h, w, 512).clamp_(min=0), r(Bh, h, w, 2), torch.empty(Bh, 2, H, W, device=dev)
    c1r = ops.PackedConv1x1(r(256, 324, 1, 1) * 0.08, r(256) * 0.1)
    corr, cor1 = r(Bh, h, w, 324), torch.empty(Bh, h, w, 256, device=dev)
    pc = ops.PackedConv(r(192, 256, 3, 3) * 0.03, r(192) * 0.1, [256])
    x, y = r(Bh, h, w, 256), torch.empty(Bh, h, w, 192, device=dev)
    loads = {
        "mask_upsample (16x16x32 f16)": lambda: [ops.mask_upsample(mhead, heads, 256, flow_lr, out=up) for _ in range(4)],
        "conv1x1_resident (16x16x32 f16)": lambda: [ops.conv1x1_resident(c1r, (corr, 0), (cor1, 0)) for _ in range(4)],
        "strip convolution (32x32x16 f16)": lambda: [ops.conv2d_nhwc(pc, [(x, 0)], (y, 0), ops.EPI_RELU) for _ in range(4)],
    }
    for name, load in loads.items():
        print(f"next to {name:34s} %d of %d launches differ (%d values)" % next_to(nb, load, launches, per=4))


if __name__ == "__main__":
    main()
