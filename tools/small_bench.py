"""Micro-benchmarks of the small per-iteration kernels at the S2 shape (B=8, 60x80): lookup, 7x7 flow conv, flow head."""
import torch
from rnnpose_amd import build, ops

build.build()
B, h, w = 8, 60, 80
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
f1 = torch.randn(B, 256, h, w, device=dev, generator=g) * 0.5
f2 = torch.randn(B, 256, h, w, device=dev, generator=g) * 0.5
buf, _ = ops.corr_pyramid(f1, f2, 4)
from rnnpose_amd.corr import coords_grid
coords = coords_grid(B, h, w, dev) + torch.randn(B, 2, h, w, device=dev, generator=g) * 4.0
out = torch.empty(B, h, w, 324, device=dev)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("pyramid f32   %.4f ms" % timeit(lambda: ops.corr_pyramid(f1, f2, 4, out=buf), n=10))
f1n, f2n = ops.nchw_to_nhwc(f1), ops.nchw_to_nhwc(f2)
print("pyramid f16x3 %.4f ms" % timeit(lambda: ops.corr_pyramid_nhwc(f1n, f2n, 4, out=buf), n=10))
print("lookup nhwc  %.4f ms" % timeit(lambda: ops.corr_lookup_nhwc(buf, coords, out, 4, 4)))
print("lookup nchw  %.4f ms" % timeit(lambda: ops.corr_lookup(buf, coords, 4, 4)))
flow4 = torch.randn(B, h, w, 4, device=dev, generator=g)
wt = torch.randn(98, 128, device=dev, generator=g) * 0.1
bias = torch.randn(128, device=dev, generator=g)
flo1 = torch.empty(B, h, w, 128, device=dev)
print("flow conv7x7 %.4f ms" % timeit(lambda: ops.flow_conv7x7_relu(flow4, wt, bias, flo1)))

# corr_weight at full resolution with a smooth flow field (B=8, 480x640, 32-ch descriptors)
H, W = 480, 640
g1 = torch.randn(B, 32, H, W, device=dev, generator=g)
g2 = torch.randn(B, 32, H, W, device=dev, generator=g)
yy, xx = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
flow = torch.stack([3.3 + 0.01 * yy, -2.7 + 0.005 * xx], 0)[None].repeat(B, 1, 1, 1).contiguous()
depth = torch.ones(B, 1, H, W, device=dev)
sigma = torch.ones(1, device=dev)
print("corr_weight  %.4f ms" % timeit(lambda: ops.corr_weight(g1, g2, flow, depth, sigma)))

# FlowHead.conv2 + coords update (3x3, 256 -> 2) on the heads buffer (B,h,w,512)
heads = torch.randn(B, h, w, 512, device=dev, generator=g)
w2 = torch.randn(2, 256, 3, 3, device=dev, generator=g) * 0.05
b2 = torch.randn(2, device=dev, generator=g)
delta = torch.empty(B, h, w, 2, device=dev); c1o = torch.empty(B, 2, h, w, device=dev); flr = torch.empty(B, h, w, 2, device=dev)
print("flow_head_out %.4f ms" % timeit(lambda: ops.flow_head_out(heads, 0, 256, w2, b2, coords, delta, c1o, flr)))
