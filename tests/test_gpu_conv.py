"""GPU tests of the hand-written NHWC implicit-GEMM convolution (csrc/conv_igemm.hip): fp16x3-split MFMA must be
fp32-class accurate.  Reference = torch conv2d in fp64; the fp32 MIOpen result gives the error scale to beat."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from rnnpose_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from rnnpose_amd import build, ops as _ops
    build.build()
    return _ops


def D(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def check(y, y64, y32, what):
    err = float((y.double() - y64).abs().max())
    e32 = float((y32.double() - y64).abs().max())
    scale = float(y64.abs().max())
    print(f"{what}: split err {err:.3e}, fp32(MIOpen) err {e32:.3e}, max|y| {scale:.2f}")
    assert err <= max(10 * e32, 3e-6 * scale), f"{what}: fp16x3 error {err:.3e} vs fp32 error {e32:.3e} (max|y|={scale:.2f})"


@pytest.mark.parametrize("B,H,W,segs,cout,kh,kw", [
    (2, 9, 13, [324], 256, 1, 1),          # 1x1, K tail (324 = 10*32 + 4), M tail (234 rows)
    (2, 12, 20, [192, 64], 126, 3, 3),     # 3x3, two sources, Cout not a multiple of 128
    (1, 16, 16, [128, 128, 128], 256, 1, 5),
    (3, 7, 11, [128, 128, 128], 128, 5, 1),
    (1, 20, 24, [128], 512, 3, 3),
    (1, 8, 8, [256], 576, 1, 1),
])
def test_conv_linear_relu(ops, B, H, W, segs, cout, kh, kw):
    cin = sum(segs)
    x = syn.normal("x", (B, cin, H, W), 1, std=1.5)
    w = syn.normal("w", (cout, cin, kh, kw), 1, std=float(np.sqrt(2.0 / (cin * kh * kw))))
    b = syn.uniform("b", (cout,), 1, -0.5, 0.5)
    xd, wd, bd = D(x), D(w), D(b)
    y64 = F.conv2d(xd.double(), wd.double(), bd.double(), padding=(kh // 2, kw // 2))
    y32 = F.conv2d(xd, wd, bd, padding=(kh // 2, kw // 2))
    pc = ops.PackedConv(wd, bd, segs)
    xs, off = [], 0
    for c in segs:                       # sources live inside wider tensors at a channel offset
        t = torch.zeros(B, H, W, c + 8, device="cuda")
        t[..., 4:4 + c] = nhwc(xd[:, off:off + c])
        xs.append((t, 4))
        off += c
    out = torch.full((B, H, W, (cout + 15) // 4 * 4), 7.0, device="cuda")     # channel stride must be a multiple of 4
    ops.conv2d_nhwc(pc, xs, (out, 8), ops.EPI_LINEAR)
    check(nchw(out[..., 8:8 + cout]), y64, y32, f"{kh}x{kw} linear")
    assert float((out[..., :8] - 7).abs().max()) == 0 and float((out[..., 8 + cout:] - 7).abs().max()) == 0
    ops.conv2d_nhwc(pc, xs, (out, 8), ops.EPI_RELU)
    check(nchw(out[..., 8:8 + cout]), y64.clamp(min=0), y32.clamp(min=0), f"{kh}x{kw} relu")


@pytest.mark.parametrize("B,H,W,cin,cout,k", [(2, 16, 24, 64, 96, 3), (1, 15, 21, 96, 128, 3), (2, 16, 24, 64, 96, 1),
                                              (1, 9, 7, 96, 128, 1)])
def test_conv_stride2(ops, B, H, W, cin, cout, k):
    """The encoder's strided convolutions (extractor.py:9,44: 3x3 stride 2 pad 1; 1x1 stride 2 pad 0)."""
    x = syn.normal("x", (B, cin, H, W), 3, std=1.5)
    w = syn.normal("w", (cout, cin, k, k), 3, std=float(np.sqrt(2.0 / (cin * k * k))))
    b = syn.uniform("b", (cout,), 3, -0.5, 0.5)
    xd, wd, bd = D(x), D(w), D(b)
    y64 = F.conv2d(xd.double(), wd.double(), bd.double(), stride=2, padding=k // 2)
    y32 = F.conv2d(xd, wd, bd, stride=2, padding=k // 2)
    pc = ops.PackedConv(wd, bd, [cin])
    out = torch.empty(B, (H + 1) // 2, (W + 1) // 2, cout, device="cuda")
    assert tuple(out.shape[1:3]) == tuple(y64.shape[2:])
    ops.conv2d_nhwc(pc, [(nhwc(xd), 0)], (out, 0), ops.EPI_LINEAR, stride=2)
    check(nchw(out), y64, y32, f"{k}x{k} stride 2")


@pytest.mark.parametrize("kh,kw", [(1, 5), (5, 1)])
def test_conv_gru_epilogues(ops, kh, kw):
    B, H, W, C = 2, 10, 14, 128
    h = np.tanh(syn.normal("h", (B, C, H, W), 2))
    x = syn.normal("x", (B, 2 * C, H, W), 2)
    wz, wr, wq = (syn.normal(n, (C, 3 * C, kh, kw), 2, std=0.03) for n in ("wz", "wr", "wq"))
    bz, br, bq = (syn.uniform(n, (C,), 2, -0.2, 0.2) for n in ("bz", "br", "bq"))
    hd, xd = D(h), D(x)
    pad = (kh // 2, kw // 2)
    hx = torch.cat([hd, xd], 1).double()
    z64 = torch.sigmoid(F.conv2d(hx, D(wz).double(), D(bz).double(), padding=pad))
    r64 = torch.sigmoid(F.conv2d(hx, D(wr).double(), D(br).double(), padding=pad))
    q64 = torch.tanh(F.conv2d(torch.cat([r64 * hd.double(), xd.double()], 1), D(wq).double(), D(bq).double(), padding=pad))
    h64 = (1 - z64) * hd.double() + z64 * q64
    pzr = ops.PackedConv(torch.cat([D(wz), D(wr)], 0), torch.cat([D(bz), D(br)], 0), [C, C, C])
    pq = ops.PackedConv(D(wq), D(bq), [C, C, C])
    hN, xN = nhwc(hd), nhwc(xd)
    z = torch.empty(B, H, W, C, device="cuda")
    rh = torch.empty(B, H, W, C, device="cuda")
    hnew = torch.empty(B, H, W, C, device="cuda")
    ops.conv2d_nhwc(pzr, [(hN, 0), (xN, 0), (xN, C)], (z, 0), ops.EPI_GRU_ZR, aux0=(hN, 0), dst2=(rh, 0), gru_c=C)
    assert float((nchw(z).double() - z64).abs().max()) < 2e-6
    assert float((nchw(rh).double() - r64 * hd.double()).abs().max()) < 2e-6
    ops.conv2d_nhwc(pq, [(rh, 0), (xN, 0), (xN, C)], (hnew, 0), ops.EPI_GRU_Q, aux0=(hN, 0), aux1=(z, 0))
    assert float((nchw(hnew).double() - h64).abs().max()) < 3e-6


@pytest.mark.parametrize("mag", [2.0e3, 7.9e3, 3.0e-3])
def test_conv_range_accuracy(ops, mag):
    """fp16x3 range: activations of magnitude 2e3 and 7.9e3 (r01's a_scale = 64 clipped everything above 1023)
    and of magnitude 3e-3 (the lo part becomes an fp16 subnormal) must be ACCURATE -- relative error vs fp64 within 3x of a
    plain fp32 convolution's -- and the range guard must stay silent."""
    B, H, W, cin, cout = 2, 12, 20, 96, 128
    x = syn.normal("x", (B, cin, H, W), 7, std=1.0 / 3.6) * mag   # the hash generator's normal is bounded by 3.47 sigma: |x| < mag
    x[0, 3, 2, 5] = mag * 1.03                                   # the largest element, near the top of the range for 7.9e3
    w = syn.normal("w", (cout, cin, 3, 3), 7, std=float(np.sqrt(2.0 / (cin * 9))))
    b = syn.uniform("b", (cout,), 7, -0.5, 0.5) * mag
    xd, wd, bd = D(x), D(w), D(b)
    y64 = F.conv2d(xd.double(), wd.double(), bd.double(), padding=1)
    y32 = F.conv2d(xd, wd, bd, padding=1)
    ops.saturation_check(True)
    ops.saturation_count(reset=True)
    try:
        pc = ops.PackedConv(wd, bd, [cin])
        out = torch.empty(B, H, W, cout, device="cuda")
        ops.conv2d_nhwc(pc, [(nhwc(xd), 0)], (out, 0), ops.EPI_LINEAR)
        assert ops.saturation_count() == 0
    finally:
        ops.saturation_check(True)          # (the guard is on by default since round 3)
    check(nchw(out), y64, y32, f"|x| ~ {mag:g}")


def test_conv_range_guard_counts_saturation(ops):
    """Beyond 65504 / a_scale = 8188 the split clamps: the output stays finite (never inf/NaN from finite inputs) and the
    range guard REPORTS it; with the guard off nothing is counted.  Same guard on the volume build's operand split."""
    B, H, W = 1, 8, 8
    x = torch.full((B, H, W, 32), 100.0, device="cuda")
    x[0, 3, 4, 7] = 9000.0
    x[0, 5, 1, 30] = -2.0e4
    w = torch.full((128, 32, 1, 1), 0.01, device="cuda")
    pc = ops.PackedConv(w, torch.zeros(128, device="cuda"), [32])
    out = torch.empty(B, H, W, 128, device="cuda")
    ops.saturation_check(True)
    try:
        ops.saturation_count(reset=True)
        ops.conv2d_nhwc(pc, [(x, 0)], (out, 0))
        assert torch.isfinite(out).all()
        c1 = ops.saturation_count()
        assert c1 >= 2                                          # two staged quads (per column tile) held an out-of-range element
        assert ops.saturation_count() == 0                      # reading resets
        x[0, 0, 0, 0] = float("nan")
        ops.conv2d_nhwc(pc, [(x, 0)], (out, 0))
        assert ops.saturation_count() == c1 * 3 // 2            # non-finite inputs are reported too
        f = torch.randn(1, 64, 8, 8, device="cuda")
        f2 = f.clone()
        f2[0, 5, 3, 3] = 3.0e4
        ops.corr_pyramid(f, f2, 2, precision="f16x3")
        assert ops.saturation_count() == 1
        ops.corr_pyramid(f, f, 2, precision="f16x3")
        assert ops.saturation_count() == 0
    finally:
        ops.saturation_check(False)
    ops.conv2d_nhwc(pc, [(x, 0)], (out, 0))
    ops.saturation_check(True)                                  # (back to the default: on)
    assert ops.saturation_count() == 0                          # launches made while the guard was off count nothing


def test_conv_range_guard_counts_in_every_k_split_share(ops):
    """with a K split every workgroup of a tile stages a different range of channel blocks; an out-of-range value in
    ANY of them must be counted, whichever workgroup arrives last (r03 lost the counts of all but the last arrival).  Same
    through the strip kernels' register path."""
    B, H, W, cin, cout = 1, 30, 30, 256, 192
    w = D(syn.normal("rg.w", (cout, cin, 3, 3), 7, std=0.01))
    pc = ops.PackedConv(w, torch.zeros(cout, device="cuda"), [cin])
    out = torch.empty(B, H, W, cout, device="cuda")
    ws = ops.conv_ksplit_workspace("cuda")
    ops.saturation_count(reset=True)
    for ch in (3, 70, 130, 250):                              # one clamped element in each quarter of the channel blocks
        x = torch.full((B, H, W, cin), 1.0, device="cuda")
        x[0, 11, 17, ch] = 2.0e4
        ops.conv2d_nhwc(pc, [(x, 0)], (out, 0), ksplit_ws=ws)
        torch.cuda.synchronize()
        assert ops.saturation_count() >= 1, f"clamp in channel {ch} was not counted"
        ops.conv2d_nhwc(pc, [(x, 0)], (out, 0))
        assert ops.saturation_count() >= 1
    xs = torch.full((2, 60, 80, cin), 1.0, device="cuda")     # strips (fp32 sources: the register path checks while it splits)
    xs[1, 31, 47, 200] = -3.0e4
    outs = torch.empty(2, 60, 80, cout, device="cuda")
    ops.conv2d_nhwc(pc, [(xs, 0)], (outs, 0), tile=5)
    assert torch.isfinite(outs).all() and ops.saturation_count() >= 1
    xs[1, 31, 47, 200] = 1.0
    ops.conv2d_nhwc(pc, [(xs, 0)], (outs, 0), tile=5)
    assert ops.saturation_count() == 0


def test_range_guard_is_visible_in_refiner_output(ops):
    """the clamp event is sticky, on by default and part of the PUBLIC output.  Update-block weights scaled
    by 1e3 drive activations past +-8188: PoseRefiner's "f16x3_range_events" must be nonzero (and zero for sane weights)."""
    from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config
    from rnnpose_amd.transformation import SE3Sequence
    d = syn.make_inputs(1, 128, 160, seed=4)
    z3 = torch.zeros(1, 3, 128, 160, device="cuda")
    rend = SyntheticRenderer(syn_img=z3, image_crop=z3, cfea=D(d["ctx"]), geofea1=D(d["g1"]), geofea2_crop=D(d["g2"]), syn_depth=D(d["depth"]),
                             intrinsics_crop=D(d["K"]), fmap1=D(d["fmap1"]), fmap2=D(d["fmap2"]))
    cfg = default_config(RENDER_ITER_COUNT=1, ITER_COUNT=2, OPTIM_ITER_COUNT=1)
    ref = PoseRefiner(cfg, renderer=rend).cuda().eval()
    ops.saturation_count(reset=True)
    out = ref(None, SE3Sequence(matrix=D(d["G0"])), D(d["K"]))
    n0 = int(out["f16x3_range_events"].item())
    assert n0 == 0, f"{n0} range events with ordinary weights"
    with torch.no_grad():
        for prm in ref.cf_net.update_block.parameters():
            prm.mul_(1.0e3)
    out = ref(None, SE3Sequence(matrix=D(d["G0"])), D(d["K"]))
    n = int(out["f16x3_range_events"].item())
    print("range events with weights x 1e3:", n, "host counter:", ops.saturation_count(reset=False))
    assert n > 0, "activations beyond the fp16x3 range were clamped silently"
    ops.saturation_count(reset=True)


@pytest.mark.parametrize("B,h,w,cout", [(2, 9, 23, 128), (1, 30, 40, 128), (1, 5, 3, 96)])
def test_flow_conv7x7_direct(ops, B, h, w, cout):
    """BasicMotionEncoder.convf1 + ReLU as the direct fp32 kernel (update.py:84,91): borders, ragged row tiles."""
    wt = D(syn.normal("f1.w", (cout, 2, 7, 7), 5, std=0.2))
    bias = D(syn.normal("f1.b", (cout,), 5, std=0.1))
    flow = D(syn.normal("f1.x", (B, 2, h, w), 5, std=3.0))
    flow4 = torch.zeros(B, h, w, 4, device="cuda")
    flow4[..., :2] = flow.permute(0, 2, 3, 1)
    out = torch.full((B, h, w, cout + 4), -7.0, device="cuda")
    ops.flow_conv7x7_relu(flow4, wt.reshape(cout, 98).t().contiguous(), bias, out, out_c_offset=4)
    ref = F.relu(F.conv2d(flow.double(), wt.double(), bias.double(), padding=3))
    y = nchw(out[..., 4:])
    err = float((y.double() - ref).abs().max())
    print(f"flow conv7x7 err {err:.3e} (max {float(ref.max()):.2f})")
    assert err <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert float(out[..., :4].min()) == -7.0 and float(out[..., :4].max()) == -7.0      # channel window respected


@pytest.mark.parametrize("B,H,W,cin,cout,k,stride", [(2, 16, 24, 64, 64, 3, 1), (3, 32, 16, 64, 96, 3, 2), (2, 16, 16, 96, 128, 1, 1)])
def test_conv_tile_stats_feed_instnorm(ops, B, H, W, cin, cout, k, stride):
    """The epilogue's per-128-row-tile column statistics (sum, sum of squares) and the instance norm built on them
    (thirdparty/raft/extractor.py:28-31,48-58) against torch in fp64."""
    x = syn.normal("ts.x", (B, cin, H, W), 9, std=1.0)
    w = syn.normal("ts.w", (cout, cin, k, k), 9, std=float(np.sqrt(2.0 / (cin * k * k))))
    b = syn.uniform("ts.b", (cout,), 9, -0.5, 0.5)
    pc = ops.PackedConv(D(w), D(b), [cin])
    Ho, Wo = -(-H // stride), -(-W // stride)
    assert (Ho * Wo) % 128 == 0
    out = torch.empty(B, Ho, Wo, cout, device="cuda")
    tpi = ops.conv_tiles_per_image(H, W, k, k, stride, cout, 0, B, src_counts=[cin])   # 3x3 stride 1: 8 x 16 patches (strips: 10 x 16 / 2 x 16); else runs of 128 output pixels
    ts = torch.full((B * tpi, cout, 2), -1.0, device="cuda", dtype=torch.float64)      # fp64 tile statistics
    ops.conv2d_nhwc(pc, [(nhwc(D(x)), 0)], (out, 0), ops.EPI_LINEAR, stride=stride, tile_stats=ts)
    y64 = F.conv2d(D(x).double(), D(w).double(), D(b).double(), stride=stride, padding=k // 2)
    if k == 3 and stride == 1:                                                  # tiles = 8 x 16 patches (ragged at the border): compare per image
        got_s = ts.view(B, tpi, cout, 2).sum(1)
        want = torch.stack([y64.sum((2, 3)), (y64 * y64).sum((2, 3))], -1)
        err = float((got_s - want).abs().max())
    else:
        rows = y64.permute(0, 2, 3, 1).reshape(-1, 128, cout)                   # tiles of 128 consecutive pixels
        want = torch.stack([rows.sum(1), (rows * rows).sum(1)], -1)
        err = float((ts.double() - want).abs().max())
    print(f"tile stats err {err:.3e} (max {float(want.abs().max()):.1f})")
    assert err <= 2e-5 * float(want.abs().max())
    got = ops.instnorm_tiles_nhwc(out, ts, relu=True)
    ref = F.relu(F.instance_norm(y64, eps=1e-5))
    assert float((nchw(got).double() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 15, 20, 64, 64), (2, 60, 80, 96, 128), (2, 9, 7, 64, 96)])
def test_conv_per_image_tiles_and_fused_input_norm(ops, B, H, W, cin, cout):
    """(1) tile_stats with H*W not a multiple of 128: rows are tiled PER IMAGE (ragged last tile), the instance norm built on
    them is exact.  (2) conv2d_nhwc(in_norm=...): relu(instance_norm(x)) applied in the load == the materialised sequence
    norm -> ReLU -> convolution (thirdparty/raft/extractor.py:48-52), bit for bit, and within fp32-class error of fp64."""
    x = syn.normal("pn.x", (B, cin, H, W), 4, std=2.0) + 0.7
    w1 = syn.normal("pn.w1", (cin, cin, 3, 3), 4, std=float(np.sqrt(2.0 / (cin * 9))))
    w2 = syn.normal("pn.w2", (cout, cin, 3, 3), 5, std=float(np.sqrt(2.0 / (cin * 9))))
    b1, b2 = syn.uniform("pn.b1", (cin,), 4, -0.5, 0.5), syn.uniform("pn.b2", (cout,), 5, -0.5, 0.5)
    p1, p2 = ops.PackedConv(D(w1), D(b1), [cin]), ops.PackedConv(D(w2), D(b2), [cout if False else cin])
    tpi = ops.conv_tiles_per_image(H, W, 3, 3, 1, cin, 0, B)    # (60 x 80 x 96: the automatic choice takes the strip kernels)
    c1 = torch.empty(B, H, W, cin, device="cuda")
    ts = torch.full((B * tpi, cin, 2), -1.0, device="cuda", dtype=torch.float64)
    ops.conv2d_nhwc(p1, [(nhwc(D(x)), 0)], (c1, 0), ops.EPI_LINEAR, tile_stats=ts)
    y64 = F.conv2d(D(x).double(), D(w1).double(), D(b1).double(), padding=1)
    n64 = F.relu(F.instance_norm(y64, eps=1e-5))
    t = ts.view(B, tpi, cin, 2).double().sum(1)
    assert float((t[..., 0] - y64.sum((2, 3))).abs().max()) <= 2e-5 * float(y64.abs().sum((2, 3)).max())
    got = ops.instnorm_tiles_nhwc(c1, ts, relu=True)
    assert float((nchw(got).double() - n64).abs().max()) < 2e-5
    mr = ops.instnorm_tiles_nhwc(c1, ts, stats_only=True)
    assert mr.shape == (B, cin, 2)
    fused = torch.empty(B, H, W, cout, device="cuda")
    ops.conv2d_nhwc(p2, [(c1, 0)], (fused, 0), ops.EPI_LINEAR, in_norm=mr)
    plain = torch.empty(B, H, W, cout, device="cuda")
    ops.conv2d_nhwc(p2, [(got, 0)], (plain, 0), ops.EPI_LINEAR)
    assert torch.equal(fused, plain)
    z64 = F.conv2d(n64, D(w2).double(), D(b2).double(), padding=1)
    z32 = F.conv2d(n64.float(), D(w2), D(b2), padding=1)
    check(nchw(fused), z64, z32, "fused norm + conv")


@pytest.mark.parametrize("B,h,w,cin,coff", [(2, 9, 23, 256, 0), (1, 16, 20, 128, 256), (1, 5, 3, 64, 4), (1, 6, 9, 320, 0)])
def test_flow_head_out_direct(ops, B, h, w, cin, coff):
    """FlowHead.conv2 (3x3, cin -> 2, update.py:10,14) + coords1 += delta (CFNet.py:157): ragged widths, channel
    windows, the register-resident fast path (cin <= 256) and the general path."""
    cs = coff + cin + 4
    x = D(syn.normal("fh.x", (B, h, w, cs), 13, std=1.0))
    wt = D(syn.normal("fh.w", (2, cin, 3, 3), 13, std=0.05))
    bias = D(syn.normal("fh.b", (2,), 13, std=0.1))
    from rnnpose_amd.corr import coords_grid
    coords1 = coords_grid(B, h, w, device="cuda") + D(syn.normal("fh.c", (B, 2, h, w), 13, std=2.0))
    delta = torch.empty(B, h, w, 2, device="cuda")
    c1o = torch.empty(B, 2, h, w, device="cuda")
    flr = torch.empty(B, h, w, 2, device="cuda")
    ops.flow_head_out(x, coff, cin, wt, bias, coords1, delta, c1o, flr)
    xin = x[..., coff:coff + cin].permute(0, 3, 1, 2).double()
    ref = F.conv2d(xin, wt.double(), bias.double(), padding=1)
    assert float((delta.permute(0, 3, 1, 2).double() - ref).abs().max()) < 2e-5
    assert float((c1o.double() - (coords1.double() + ref)).abs().max()) < 5e-5
    grid = coords_grid(B, h, w, device="cuda").double()
    assert float((flr.permute(0, 3, 1, 2).double() - (coords1.double() + ref - grid)).abs().max()) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,w", [(1, 16, 20), (3, 17, 19), (2, 9, 80), (1, 5, 13)])
def test_mask_upsample_fused_equals_conv_plus_upsample(ops, B, h, w):
    """rnnpose_mask_upsample_f16x3 (mask.2 + convex up-sampling, no mask tensor) against the 1x1 convolution kernel followed
    by the up-sampling kernel (same logits; online instead of two-pass softmax: fp32 round-off apart), for pixel counts
    that are not multiples of its 64-pixel tile, and against an fp64 evaluation of update.py:183-187 + CFNet.py:95-106."""
    x = syn.normal("mu_x", (B, 512, h, w), 11).clip(min=0)                     # relu(heads): mask.0 output in channels 256..511
    wt = syn.normal("mu_w", (576, 256, 1, 1), 11, std=float(np.sqrt(2.0 / 256)))
    bs = syn.uniform("mu_b", (576,), 11, -0.5, 0.5)
    flow = syn.normal("mu_f", (B, 2, h, w), 11, std=2.0)
    xN = nhwc(D(x))
    fl = nhwc(D(flow))
    pc = ops.PackedConv(D(wt), D(bs), [256], post_scale=0.25)
    mask = torch.empty(B, h, w, 576, device="cuda")
    ops.conv2d_nhwc(pc, [(xN, 256)], (mask, 0), ops.EPI_LINEAR)
    want = ops.convex_upsample_nhwc(fl, mask)
    got = ops.mask_upsample(ops.PackedMaskHead(D(wt), D(bs), post_scale=0.25), xN, 256, fl)
    assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    m64 = 0.25 * F.conv2d(D(x)[:, 256:].double(), D(wt).double(), D(bs).double())
    sm = torch.softmax(m64.view(B, 1, 9, 8, 8, h, w), dim=2)
    upf = F.unfold(8 * D(flow).double(), [3, 3], padding=1).view(B, 2, 9, 1, 1, h, w)
    ref = torch.sum(sm * upf, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * h, 8 * w)
    assert float((got.double() - ref).abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,w,cin", [(1, 16, 20, 324), (2, 9, 13, 324), (1, 5, 7, 128), (1, 4, 8, 352)])
def test_conv1x1_resident_vs_fp64_and_igemm(ops, B, h, w, cin):
    """rnnpose_conv1x1_resident_f16x3 (activation tile split once, resident in LDS) against fp64 and against the implicit-GEMM
    kernel on the same layer, incl. the ragged K of convc1 (324 = 10 x 32 + 4) and pixel counts that are not multiples of 32."""
    x = syn.normal("r1_x", (B, cin, h, w), 13)
    wt = syn.normal("r1_w", (256, cin, 1, 1), 13, std=float(np.sqrt(2.0 / cin)))
    bs = syn.uniform("r1_b", (256,), 13, -0.5, 0.5)
    xN = nhwc(D(x))
    y64 = F.relu(F.conv2d(D(x).double(), D(wt).double(), D(bs).double()))
    y32 = F.relu(F.conv2d(D(x), D(wt), D(bs)))
    out = torch.full((B, h, w, 260), 7.0, device="cuda")                      # channel slice [4, 260) of a wider tensor
    ops.conv1x1_resident(ops.PackedConv1x1(D(wt), D(bs)), (xN, 0), (out, 4), relu=True)
    assert float((out[..., :4] - 7.0).abs().max()) == 0.0
    check(nchw(out[..., 4:].contiguous()), y64, y32, "resident 1x1")
    ref = torch.empty(B, h, w, 256, device="cuda")
    ops.conv2d_nhwc(ops.PackedConv(D(wt), D(bs), [cin]), [(xN, 0)], (ref, 0), ops.EPI_RELU)
    assert float((out[..., 4:] - ref).abs().max()) < 2e-5


# ---- split tensors (round 3): activations pre-split into fp16 hi|lo by their producer ----------------------------------------
def test_split_hl_roundtrip_and_layout(ops):
    """rnnpose_split_hl_f32: layout ([hi x 8 | lo x 8] per 8-channel group) and value (hi + lo = x * a_scale to 2^-21)."""
    x = torch.randn(2, 5, 7, 48, device="cuda") * 3
    x[0, 0, 0, :8] = torch.tensor([0.0, 1.0, -1.0, 1e-3, 123.456, -8000.0, 2.0 ** -20, 0.5], device="cuda")
    wide = torch.full((2, 5, 7, 64), 9.0, device="cuda")
    ops.split_hl(x, wide, src_c_offset=8, c_count=32, dst_c_offset=16)
    assert float((wide[..., :16] - 9).abs().max()) == 0 and float((wide[..., 48:] - 9).abs().max()) == 0
    back = ops.unsplit_hl(wide[..., 16:48].contiguous())
    ref = x[..., 8:40]
    assert float(((back - ref).abs() / ref.abs().clamp(min=1e-2)).max()) < 2.0 ** -20
    s = ops.split_hl(x)
    v = s.view(torch.float16).view(2, 5, 7, 6, 2, 8)
    hi = v[..., 0, :].float().reshape(2, 5, 7, 48)
    t = (x * 8.0).half().float()                                                 # rounded to nearest fp16
    assert torch.equal(hi, t)
    assert torch.equal(v[..., 1, :].float().reshape(2, 5, 7, 48), (x * 8.0 - t).half().float())


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("B,H,W,segs,cout,kh,kw", [
    (2, 12, 20, [192, 64], 126, 3, 3),     # ragged output tail (2 valid columns of the last quad)
    (1, 16, 16, [128, 128], 256, 1, 5),
    (3, 7, 11, [128, 128], 128, 5, 1),
    (1, 20, 24, [128], 512, 3, 3),
    (2, 9, 13, [256], 192, 3, 3),
    (1, 8, 8, [128], 64, 1, 1),
])
def test_conv_split_sources_and_outputs(ops, tile, B, H, W, segs, cout, kh, kw):
    """Split-tensor sources (src_hl) through every tile shape, fp32 and split-form outputs: same accuracy class as the
    on-the-fly split (the fp16 operands are bit-identical), and split-form outputs decode to the fp32 ones to 2^-21."""
    cin = sum(segs)
    x = syn.normal("x", (B, cin, H, W), 11, std=1.5)
    w = syn.normal("w", (cout, cin, kh, kw), 11, std=float(np.sqrt(2.0 / (cin * kh * kw))))
    b = syn.uniform("b", (cout,), 11, -0.5, 0.5)
    xd, wd, bd = D(x), D(w), D(b)
    y64 = F.conv2d(xd.double(), wd.double(), bd.double(), padding=(kh // 2, kw // 2))
    y32 = F.conv2d(xd, wd, bd, padding=(kh // 2, kw // 2))
    pc = ops.PackedConv(wd, bd, segs)
    xs, xf, off = [], [], 0
    for c in segs:                       # sources inside wider split tensors at a channel offset of 8
        t = torch.zeros(B, H, W, c + 16, device="cuda")
        t[..., 8:8 + c] = nhwc(xd[:, off:off + c])
        xf.append((t, 8))
        xs.append((ops.split_hl(t), 8))
        off += c
    cs = (cout + 23) // 8 * 8
    out = torch.full((B, H, W, cs), 7.0, device="cuda")
    ops.conv2d_nhwc(pc, xs, (out, 8), ops.EPI_RELU, src_hl=True, tile=tile)
    check(nchw(out[..., 8:8 + cout]), y64.clamp(min=0), y32.clamp(min=0), f"{kh}x{kw} split sources, tile {tile}")
    assert float((out[..., :8] - 7).abs().max()) == 0 and float((out[..., 8 + cout:] - 7).abs().max()) == 0
    ref = torch.full((B, H, W, cs), 7.0, device="cuda")
    ops.conv2d_nhwc(pc, xf, (ref, 8), ops.EPI_RELU)                 # on-the-fly split of the same values
    assert float((out - ref).abs().max()) <= 2e-6 * float(y64.abs().max())
    # split-form destination (+ the extra split copy next to an fp32 destination)
    outs = torch.zeros(B, H, W, cs, device="cuda")
    extra = torch.zeros(B, H, W, cs, device="cuda")
    ops.conv2d_nhwc(pc, xs, (outs, 8), ops.EPI_RELU, src_hl=True, dst_hl=True, tile=tile)
    ops.conv2d_nhwc(pc, xs, (out, 8), ops.EPI_RELU, src_hl=True, dst_split=(extra, 8), tile=tile)
    n8 = cout // 8 * 8                                              # whole groups decode; the ragged group is checked per channel
    for sp in (outs, extra):
        dec = ops.unsplit_hl(sp)
        full = out[..., 8:8 + cout]
        got = dec[..., 8:8 + cout]
        assert float(((got - full).abs() / full.abs().clamp(min=1e-2)).max()) < 2.0 ** -20
        if cout % 8:                                                # channels past cout in the last group were left untouched (zeros)
            assert float(dec[..., 8 + cout:8 + n8 + 8].abs().max()) == 0


@pytest.mark.parametrize("kh,kw", [(1, 5), (5, 1)])
@pytest.mark.parametrize("tile", [0, 3, 4])
def test_conv_gru_epilogues_split(ops, kh, kw, tile):
    """The GRU gate / state-update epilogues on split tensors: h from its split copy, r*h written split, h' fp32 + split."""
    B, H, W, C = 2, 10, 14, 128
    h = np.tanh(syn.normal("h", (B, C, H, W), 2))
    x = syn.normal("x", (B, C, H, W), 2)
    wz, wr, wq = (syn.normal(n, (C, 2 * C, kh, kw), 2, std=0.03) for n in ("wz", "wr", "wq"))
    bz, br, bq = (syn.uniform(n, (C,), 2, -0.2, 0.2) for n in ("bz", "br", "bq"))
    hd, xd = D(h), D(x)
    pad = (kh // 2, kw // 2)
    hx = torch.cat([hd, xd], 1).double()
    z64 = torch.sigmoid(F.conv2d(hx, D(wz).double(), D(bz).double(), padding=pad))
    r64 = torch.sigmoid(F.conv2d(hx, D(wr).double(), D(br).double(), padding=pad))
    q64 = torch.tanh(F.conv2d(torch.cat([r64 * hd.double(), xd.double()], 1), D(wq).double(), D(bq).double(), padding=pad))
    h64 = (1 - z64) * hd.double() + z64 * q64
    pzr = ops.PackedConv(torch.cat([D(wz), D(wr)], 0), torch.cat([D(bz), D(br)], 0), [C, C])
    pq = ops.PackedConv(D(wq), D(bq), [C, C])
    hN, xN = nhwc(hd), nhwc(xd)
    hS, xS = ops.split_hl(hN), ops.split_hl(xN)
    z = torch.empty(B, H, W, C, device="cuda")
    rh = torch.empty(B, H, W, C, device="cuda")
    hnew = torch.empty(B, H, W, C, device="cuda")
    hnew_s = torch.empty(B, H, W, C, device="cuda")
    ops.conv2d_nhwc(pzr, [(hS, 0), (xS, 0)], (z, 0), ops.EPI_GRU_ZR, aux0=(hN, 0), dst2=(rh, 0), dst2_hl=True, gru_c=C, src_hl=True,
                    tile=tile)
    assert float((nchw(z).double() - z64).abs().max()) < 2e-6
    assert float((nchw(ops.unsplit_hl(rh)).double() - r64 * hd.double()).abs().max()) < 2e-6
    ops.conv2d_nhwc(pq, [(rh, 0), (xS, 0)], (hnew, 0), ops.EPI_GRU_Q, aux0=(hN, 0), aux1=(z, 0), src_hl=True, dst_split=(hnew_s, 0),
                    tile=tile)
    assert float((nchw(hnew).double() - h64).abs().max()) < 3e-6
    assert float((ops.unsplit_hl(hnew_s) - hnew).abs().max()) < 2.0 ** -20


def test_split_producers(ops):
    """The two non-GEMM producers of split tensors: the LDS-resident 1x1 convolution and the 7x7 flow-feature kernel."""
    B, h, w = 2, 11, 23
    x = torch.randn(B, h, w, 324, device="cuda")
    wt = torch.randn(256, 324, 1, 1, device="cuda") * 0.08
    bs = torch.randn(256, device="cuda") * 0.1
    pr = ops.PackedConv1x1(wt, bs)
    a = torch.empty(B, h, w, 256, device="cuda")
    s = torch.empty(B, h, w, 256, device="cuda")
    ops.conv1x1_resident(pr, (x, 0), (a, 0), relu=True)
    ops.conv1x1_resident(pr, (x, 0), (s, 0), relu=True, dst_split=True)
    assert float(((ops.unsplit_hl(s) - a).abs() / a.abs().clamp(min=1e-2)).max()) < 2.0 ** -20
    coords = torch.randn(B, 2, h, w, device="cuda") * 3 + 10
    w_t = torch.randn(98, 128, device="cuda") * 0.1
    fb = torch.randn(128, device="cuda") * 0.1
    f32o, m32 = torch.zeros(B, h, w, 128, device="cuda"), torch.zeros(B, h, w, 128, device="cuda")
    fso, mso = torch.zeros(B, h, w, 128, device="cuda"), torch.zeros(B, h, w, 128, device="cuda")
    ops.flow_features(coords, w_t, fb, f32o, m32, 126)
    ops.flow_features(coords, w_t, fb, fso, mso, 126, out_split=True, motion_split=True)
    assert float(((ops.unsplit_hl(fso) - f32o).abs() / f32o.abs().clamp(min=1e-2)).max()) < 2.0 ** -20
    dm = ops.unsplit_hl(mso)
    assert float(((dm[..., 126:] - m32[..., 126:]).abs() / m32[..., 126:].abs().clamp(min=1e-2)).max()) < 2.0 ** -20
    assert float(dm[..., :126].abs().max()) == 0          # the other 6 channels of the group belong to the 126-channel convolution


@pytest.mark.parametrize("B,H,W,cin,cout,hl", [(2, 60, 80, 256, 192, False), (1, 30, 30, 128, 512, False), (3, 13, 21, 64, 64, False),
                                               (2, 60, 80, 256, 126, True), (1, 17, 33, 128, 512, True)])
def test_conv3x3_patch_tiling_equals_row_major(ops, B, H, W, cin, cout, hl):
    """3x3 stride-1 layers run on 8 x 16 image patches (nine taps on one staged halo tile, csrc/conv_igemm.hip SPATIAL) by
    default; rnnpose_conv_spatial_tiles(0) restores the row-major tiling.  Same products, same fp16 operands: the outputs agree
    to fp32 summation order, the per-image statistics too, at sizes whose patches are ragged in both directions."""
    x = syn.normal("sp.x", (B, cin, H, W), 21, std=1.5)
    w = syn.normal("sp.w", (cout, cin, 3, 3), 21, std=float(np.sqrt(2.0 / (cin * 9))))
    b = syn.uniform("sp.b", (cout,), 21, -0.5, 0.5)
    pc = ops.PackedConv(D(w), D(b), [cin])
    xn = nhwc(D(x))
    src = [(ops.split_hl(xn) if hl else xn, 0)]
    cs = (cout + 7) // 8 * 8
    outs, stats = [], []
    try:
        ops.conv_strip(0)                      # (this is about the two tilings of the 128-row kernels)
        for sp in (True, False):
            ops.conv_spatial_tiles(sp)
            tpi = ops.conv_tiles_per_image(H, W, 3, 3, 1)
            assert tpi == ((-(-W // 16)) * (-(-H // 8)) if sp else -(-(H * W) // 128))
            out = torch.full((B, H, W, cs), 3.0, device="cuda")
            ts = torch.zeros(B * tpi, cout, 2, device="cuda", dtype=torch.float64) if not hl else None
            ops.conv2d_nhwc(pc, src, (out, 0), ops.EPI_LINEAR, src_hl=hl, tile_stats=ts)
            outs.append(out)
            stats.append(None if ts is None else ts.view(B, tpi, cout, 2).sum(1))
    finally:
        ops.conv_spatial_tiles(True)
        ops.conv_strip(1)
    y64 = F.conv2d(D(x).double(), D(w).double(), D(b).double(), padding=1)
    y32 = F.conv2d(D(x), D(w), D(b), padding=1)
    check(nchw(outs[0][..., :cout]), y64, y32, "3x3 on patches")
    assert float((outs[0] - outs[1]).abs().max()) <= 3e-6 * float(y64.abs().max())
    assert float((outs[0][..., cout:] - 3).abs().max()) == 0 if cs > cout else True
    if stats[0] is not None:
        want = torch.stack([y64.sum((2, 3)), (y64 * y64).sum((2, 3))], -1)
        scale = torch.stack([y64.abs().sum((2, 3)), (y64 * y64).sum((2, 3))], -1)      # (a sum of signed values can be ~0: compare to the sum of magnitudes)
        for st in stats:
            assert float(((st - want).abs() / scale).max()) < 1e-5
        assert float(((stats[0] - stats[1]).abs() / scale).max()) < 1e-6               # patches vs runs: same values, regrouped


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,segs,cout,kh,kw", [
    (1, 30, 30, [256], 192, 3, 3),         # convc2 at the 240 x 240 crop: 8 patches x 3 column tiles, 8 channel blocks -> 4 splits
    (1, 30, 30, [128, 128, 128], 256, 1, 5),
    (1, 30, 30, [128, 128, 128], 256, 5, 1),
    (1, 30, 30, [128], 512, 3, 3),         # 64 tiles: 2-3 splits
    (1, 16, 20, [256], 126, 3, 3),         # ragged columns
    (2, 9, 13, [324], 256, 1, 1),          # 1x1, K tail (11 blocks)
    (1, 7, 9, [96], 64, 3, 3),             # 3 channel blocks: too few to split (>= 2 per split needs 4)
])
def test_conv_k_split_matches_single_pass(ops, B, H, W, segs, cout, kh, kw):
    """conv2d_nhwc(ksplit_ws=...): launches of few tiles split their K loop over several workgroups per tile; the last arrival
    sums the partial accumulators in split order.  Same result as the single pass to fp32 round-off (the K sum is grouped per
    split), fp32-class against fp64, bit-identical from run to run, counters left at zero, outputs outside the slice untouched."""
    cin = sum(segs)
    x = syn.normal("ks.x", (B, cin, H, W), 3, std=1.5)
    w = syn.normal("ks.w", (cout, cin, kh, kw), 3, std=float(np.sqrt(2.0 / (cin * kh * kw))))
    b = syn.uniform("ks.b", (cout,), 3, -0.5, 0.5)
    xd, wd, bd = D(x), D(w), D(b)
    y64 = F.conv2d(xd.double(), wd.double(), bd.double(), padding=(kh // 2, kw // 2))
    y32 = F.conv2d(xd, wd, bd, padding=(kh // 2, kw // 2))
    pc = ops.PackedConv(wd, bd, segs)
    xs, off = [], 0
    for c in segs:
        xs.append((nhwc(xd[:, off:off + c]), 0))
        off += c
    cs = (cout + 15) // 4 * 4
    single = torch.full((B, H, W, cs), 7.0, device="cuda")
    ops.conv2d_nhwc(pc, xs, (single, 8), ops.EPI_RELU)
    ws = ops.conv_ksplit_workspace("cuda")
    outs = []
    for _ in range(3):
        o = torch.full((B, H, W, cs), 7.0, device="cuda")
        ops.conv2d_nhwc(pc, xs, (o, 8), ops.EPI_RELU, ksplit_ws=ws)
        outs.append(o)
    torch.cuda.synchronize()
    assert int(ws[:256].abs().max()) == 0, "arrival counters not restored"
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "K-split result differs between runs"
    check(nchw(outs[0][..., 8:8 + cout]), y64.clamp(min=0), y32.clamp(min=0), f"{kh}x{kw} K split")
    scale = float(y64.abs().max())
    assert float((outs[0] - single).abs().max()) <= 3e-6 * scale
    assert float((outs[0][..., :8] - 7).abs().max()) == 0 and float((outs[0][..., 8 + cout:] - 7).abs().max()) == 0
    # the measurement switch and a workspace that is too small both give the single pass, bit for bit
    ops.conv_ksplit(False)
    try:
        o2 = torch.full((B, H, W, cs), 7.0, device="cuda")
        ops.conv2d_nhwc(pc, xs, (o2, 8), ops.EPI_RELU, ksplit_ws=ws)
    finally:
        ops.conv_ksplit(True)
    assert torch.equal(o2, single)
    o3 = torch.full((B, H, W, cs), 7.0, device="cuda")
    ops.conv2d_nhwc(pc, xs, (o3, 8), ops.EPI_RELU, ksplit_ws=ws[:512])
    assert torch.equal(o3, single)


@pytest.mark.gpu
def test_conv_k_split_with_statistics_norm_and_gru_epilogues(ops):
    """The epilogue variants behind a K split: per-tile statistics + fused input normalisation (encoder layers at 30 x 30) and
    the GRU gate epilogues (1 x 5) -- all computed by the last-arriving workgroup from the summed accumulators.  (The 128-row
    kernels: tile=1 -- the automatic choice takes 32-row strips for a single 30 x 30 crop since r04.)"""
    B, H, W, cin, cout = 1, 30, 30, 128, 128
    x = syn.normal("ks2.x", (B, cin, H, W), 4, std=2.0) + 0.7
    w1 = syn.normal("ks2.w1", (cin, cin, 3, 3), 4, std=float(np.sqrt(2.0 / (cin * 9))))
    w2 = syn.normal("ks2.w2", (cout, cin, 3, 3), 5, std=float(np.sqrt(2.0 / (cin * 9))))
    b1, b2 = syn.uniform("ks2.b1", (cin,), 4, -0.5, 0.5), syn.uniform("ks2.b2", (cout,), 5, -0.5, 0.5)
    p1, p2 = ops.PackedConv(D(w1), D(b1), [cin]), ops.PackedConv(D(w2), D(b2), [cin])
    tpi = ops.conv_tiles_per_image(H, W, 3, 3, 1, cin, 1, B)
    ws = ops.conv_ksplit_workspace("cuda")
    res = []
    for k in (None, ws):
        c1 = torch.empty(B, H, W, cin, device="cuda")
        ts = torch.full((B * tpi, cin, 2), -1.0, device="cuda", dtype=torch.float64)
        ops.conv2d_nhwc(p1, [(nhwc(D(x)), 0)], (c1, 0), ops.EPI_LINEAR, tile_stats=ts, ksplit_ws=k, tile=1)
        mr = ops.instnorm_tiles_nhwc(c1, ts, stats_only=True)
        c2 = torch.empty(B, H, W, cout, device="cuda")
        ts2 = torch.full((B * tpi, cout, 2), -1.0, device="cuda", dtype=torch.float64)
        ops.conv2d_nhwc(p2, [(c1, 0)], (c2, 0), ops.EPI_LINEAR, in_norm=mr, tile_stats=ts2, ksplit_ws=k, tile=1)
        res.append((c1, ts, c2, ts2))
    for a, bb in zip(res[0], res[1]):
        sc = float(a.abs().max())
        assert float((a.double() - bb.double()).abs().max()) <= 1e-5 * sc
    y64 = F.conv2d(D(x).double(), D(w1).double(), D(b1).double(), padding=1)
    z64 = F.conv2d(F.relu(F.instance_norm(y64, eps=1e-5)), D(w2).double(), D(b2).double(), padding=1)
    assert float((nchw(res[1][2]).double() - z64).abs().max()) < 3e-5 * float(z64.abs().max())
    # GRU epilogues
    C, kh, kw = 128, 1, 5
    h = np.tanh(syn.normal("ks2.h", (B, C, H, W), 2))
    xx = syn.normal("ks2.xx", (B, 2 * C, H, W), 2)
    wz, wr, wq = (syn.normal("ks2." + n, (C, 3 * C, kh, kw), 2, std=0.03) for n in ("wz", "wr", "wq"))
    bz, br, bq = (syn.uniform("ks2." + n, (C,), 2, -0.2, 0.2) for n in ("bz", "br", "bq"))
    hd, xd = D(h), D(xx)
    pad = (kh // 2, kw // 2)
    hx = torch.cat([hd, xd], 1).double()
    z64 = torch.sigmoid(F.conv2d(hx, D(wz).double(), D(bz).double(), padding=pad))
    r64 = torch.sigmoid(F.conv2d(hx, D(wr).double(), D(br).double(), padding=pad))
    q64 = torch.tanh(F.conv2d(torch.cat([r64 * hd.double(), xd.double()], 1), D(wq).double(), D(bq).double(), padding=pad))
    h64 = (1 - z64) * hd.double() + z64 * q64
    pzr = ops.PackedConv(torch.cat([D(wz), D(wr)], 0), torch.cat([D(bz), D(br)], 0), [C, C, C])
    pq = ops.PackedConv(D(wq), D(bq), [C, C, C])
    hN, xN = nhwc(hd), nhwc(xd)
    z = torch.empty(B, H, W, C, device="cuda")
    rh = torch.empty(B, H, W, C, device="cuda")
    hnew = torch.empty(B, H, W, C, device="cuda")
    ops.conv2d_nhwc(pzr, [(hN, 0), (xN, 0), (xN, C)], (z, 0), ops.EPI_GRU_ZR, aux0=(hN, 0), dst2=(rh, 0), gru_c=C, ksplit_ws=ws, tile=1)
    assert float((nchw(z).double() - z64).abs().max()) < 2e-6
    assert float((nchw(rh).double() - r64 * hd.double()).abs().max()) < 2e-6
    ops.conv2d_nhwc(pq, [(rh, 0), (xN, 0), (xN, C)], (hnew, 0), ops.EPI_GRU_Q, aux0=(hN, 0), aux1=(z, 0), ksplit_ws=ws, tile=1)
    assert float((nchw(hnew).double() - h64).abs().max()) < 3e-6
    torch.cuda.synchronize()
    assert int(ws[:256].abs().max()) == 0


# ---- the strip kernels (csrc/conv_strip.hip: tile=5, or the automatic choice on maps that fill the chip with strips) ----
STRIP_SHAPES = [
    (2, 12, 20, [192, 64], 126, 3, 3),     # two sources, ragged output tail, ragged 10 x 16 patches
    (1, 16, 16, [128, 128], 256, 1, 5),    # two column tiles of 128
    (3, 7, 11, [128, 128], 128, 5, 1),     # vertical taps, strips straddle image lines and images (231 rows)
    (1, 20, 32, [128], 512, 3, 3),         # exact patches, four column tiles
    (2, 9, 13, [256], 192, 3, 3),          # three-wave workgroups (96 columns)
    (1, 23, 37, [96], 96, 3, 3),           # three channel blocks, three waves, ragged patches
    (2, 13, 29, [64, 32], 320, 1, 5),      # a 64-channel and a 32-channel source, Cout = 2.5 column tiles
    (2, 40, 48, [64], 64, 3, 3),           # two-wave workgroups (the encoder's 64-channel layers)
    (1, 21, 33, [64], 48, 3, 3),           # ... with a ragged column tile
]


@pytest.fixture(params=[5, 6, 7], ids=["rows160", "rows32", "rows96"])
def strip_tile(request):
    """tile code of the strip height under test: 5 = 160-row strips (10 x 16 patches), 6 = 32 rows (2 x 16), 7 = 96 rows (6 x 16; r06)."""
    return request.param


@pytest.fixture(params=[1, 3], ids=["auto", "two-tile-waves"])
def strip_mode(ops, request):
    """The strip kernels in their automatic shape (one 32-column tile per wave, two workgroups per CU) and with two tiles per
    wave forced (ops.conv_strip(3): the one-workgroup-per-CU measurement form; its epilogue is the general one)."""
    ops.conv_strip(request.param)
    yield request.param
    ops.conv_strip(1)


@pytest.mark.parametrize("hl", [False, True])
@pytest.mark.parametrize("B,H,W,segs,cout,kh,kw", STRIP_SHAPES)
def test_conv_strip_vs_fp64(ops, strip_mode, strip_tile, hl, B, H, W, segs, cout, kh, kw):
    """160-row strips with both operand paths (split-tensor sources by LDS-DMA, fp32 sources through registers) against fp64 and
    against the 128-row kernel on the same operands; fp32 and split-form destinations; bytes around the slice untouched."""
    if strip_mode == 3 and (cout <= 64 or strip_tile != 5):
        pytest.skip("two tiles per wave: 160-row strips, more than 64 output channels")
    cin = sum(segs)
    x = syn.normal("sx", (B, cin, H, W), 21, std=1.5)
    w = syn.normal("sw", (cout, cin, kh, kw), 21, std=float(np.sqrt(2.0 / (cin * kh * kw))))
    b = syn.uniform("sb", (cout,), 21, -0.5, 0.5)
    xd, wd, bd = D(x), D(w), D(b)
    y64 = F.conv2d(xd.double(), wd.double(), bd.double(), padding=(kh // 2, kw // 2))
    y32 = F.conv2d(xd, wd, bd, padding=(kh // 2, kw // 2))
    pc = ops.PackedConv(wd, bd, segs)
    xs, off = [], 0
    for c in segs:                       # sources inside wider tensors at a channel offset of 8
        t = torch.zeros(B, H, W, c + 16, device="cuda")
        t[..., 8:8 + c] = nhwc(xd[:, off:off + c])
        xs.append((ops.split_hl(t) if hl else t, 8))
        off += c
    cs = (cout + 23) // 8 * 8
    out = torch.full((B, H, W, cs), 7.0, device="cuda")
    ops.conv2d_nhwc(pc, xs, (out, 8), ops.EPI_RELU, src_hl=hl, tile=strip_tile)
    check(nchw(out[..., 8:8 + cout]), y64.clamp(min=0), y32.clamp(min=0), f"strip {kh}x{kw} hl={hl}")
    assert float((out[..., :8] - 7).abs().max()) == 0 and float((out[..., 8 + cout:] - 7).abs().max()) == 0
    ref = torch.full((B, H, W, cs), 7.0, device="cuda")
    ops.conv2d_nhwc(pc, xs, (ref, 8), ops.EPI_RELU, src_hl=hl, tile=1)           # same fp16 operands, other summation order
    assert float((out - ref).abs().max()) <= 2e-6 * float(y64.abs().max())
    again = torch.full((B, H, W, cs), 7.0, device="cuda")
    ops.conv2d_nhwc(pc, xs, (again, 8), ops.EPI_RELU, src_hl=hl, tile=strip_tile)
    assert torch.equal(out, again)                                               # deterministic
    outs = torch.zeros(B, H, W, cs, device="cuda")
    extra = torch.zeros(B, H, W, cs, device="cuda")
    ops.conv2d_nhwc(pc, xs, (outs, 8), ops.EPI_RELU, src_hl=hl, dst_hl=True, tile=strip_tile)
    ops.conv2d_nhwc(pc, xs, (again, 8), ops.EPI_LINEAR, src_hl=hl, dst_split=(extra, 8), tile=strip_tile)
    check(nchw(again[..., 8:8 + cout]), y64, y32, f"strip {kh}x{kw} linear")
    full = out[..., 8:8 + cout]
    assert float(((ops.unsplit_hl(outs)[..., 8:8 + cout] - full).abs() / full.abs().clamp(min=1e-2)).max()) < 2.0 ** -20
    lin = again[..., 8:8 + cout]
    assert float(((ops.unsplit_hl(extra)[..., 8:8 + cout] - lin).abs() / lin.abs().clamp(min=1e-2)).max()) < 2.0 ** -20


@pytest.mark.parametrize("hl", [False, True])
@pytest.mark.parametrize("kh,kw", [(1, 5), (5, 1)])
def test_conv_strip_gru_epilogues(ops, strip_mode, strip_tile, kh, kw, hl):
    """GRU gate / state-update epilogues + additive map in the strip kernels (update.py:47-58, hoisted context share)."""
    if strip_mode == 3 and strip_tile != 5:
        pytest.skip("two tiles per wave: 160-row strips only")
    B, H, W, C = 2, 11, 19, 128
    h = np.tanh(syn.normal("h", (B, C, H, W), 2))
    x = syn.normal("x", (B, C, H, W), 2)
    am = syn.normal("am", (B, 3 * C, H, W), 2, std=0.3)
    wz, wr, wq = (syn.normal(n, (C, 2 * C, kh, kw), 2, std=0.03) for n in ("wz", "wr", "wq"))
    bz, br, bq = (syn.uniform(n, (C,), 2, -0.2, 0.2) for n in ("bz", "br", "bq"))
    hd, xd, amd = D(h), D(x), D(am)
    pad = (kh // 2, kw // 2)
    hx = torch.cat([hd, xd], 1).double()
    z64 = torch.sigmoid(F.conv2d(hx, D(wz).double(), D(bz).double(), padding=pad) + amd[:, :C].double())
    r64 = torch.sigmoid(F.conv2d(hx, D(wr).double(), D(br).double(), padding=pad) + amd[:, C:2 * C].double())
    q64 = torch.tanh(F.conv2d(torch.cat([r64 * hd.double(), xd.double()], 1), D(wq).double(), D(bq).double(), padding=pad) + amd[:, 2 * C:].double())
    h64 = (1 - z64) * hd.double() + z64 * q64
    pzr = ops.PackedConv(torch.cat([D(wz), D(wr)], 0), torch.cat([D(bz), D(br)], 0), [C, C])
    pq = ops.PackedConv(D(wq), D(bq), [C, C])
    hN, xN, aN = nhwc(hd), nhwc(xd), nhwc(amd)
    hS, xS = (ops.split_hl(hN), ops.split_hl(xN)) if hl else (hN, xN)
    z = torch.empty(B, H, W, C, device="cuda")
    rh = torch.empty(B, H, W, C, device="cuda")
    hnew = torch.empty(B, H, W, C, device="cuda")
    hnew_s = torch.empty(B, H, W, C, device="cuda")
    ops.conv2d_nhwc(pzr, [(hS, 0), (xS, 0)], (z, 0), ops.EPI_GRU_ZR, aux0=(hN, 0), dst2=(rh, 0), dst2_hl=hl, gru_c=C, src_hl=hl,
                    add_map=(aN, 0), tile=strip_tile)
    assert float((nchw(z).double() - z64).abs().max()) < 2e-6
    rhd = ops.unsplit_hl(rh) if hl else rh
    assert float((nchw(rhd).double() - r64 * hd.double()).abs().max()) < 2e-6
    ops.conv2d_nhwc(pq, [(rh, 0), (xS, 0)], (hnew, 0), ops.EPI_GRU_Q, aux0=(hN, 0), aux1=(z, 0), src_hl=hl, dst_split=(hnew_s, 0),
                    add_map=(aN, 2 * C), tile=strip_tile)
    assert float((nchw(hnew).double() - h64).abs().max()) < 3e-6
    assert float((ops.unsplit_hl(hnew_s) - hnew).abs().max()) < 2.0 ** -20


@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 15, 20, 96, 96), (2, 60, 80, 96, 128), (2, 20, 32, 128, 128), (2, 40, 48, 64, 64)])
def test_conv_strip_tile_stats_and_fused_input_norm(ops, strip_mode, strip_tile, B, H, W, cin, cout):
    """The encoder's pair (extractor.py:48-58) on strips: conv1 with fp64 tile statistics per 10 x 16 patch, conv2 reading
    relu(norm1(conv1 x)) in its load == the materialised sequence, bit for bit."""
    if strip_mode == 3 and (min(cin, cout) <= 64 or strip_tile != 5):
        pytest.skip("two tiles per wave: 160-row strips, more than 64 output channels")
    x = syn.normal("pn.x", (B, cin, H, W), 4, std=2.0) + 0.7
    w1 = syn.normal("pn.w1", (cin, cin, 3, 3), 4, std=float(np.sqrt(2.0 / (cin * 9))))
    w2 = syn.normal("pn.w2", (cout, cin, 3, 3), 5, std=float(np.sqrt(2.0 / (cin * 9))))
    b1, b2 = syn.uniform("pn.b1", (cin,), 4, -0.5, 0.5), syn.uniform("pn.b2", (cout,), 5, -0.5, 0.5)
    p1, p2 = ops.PackedConv(D(w1), D(b1), [cin]), ops.PackedConv(D(w2), D(b2), [cin])
    tpi = ops.conv_tiles_per_image(H, W, 3, 3, 1, cin, strip_tile, B)
    assert tpi == -(-H // {5: 10, 6: 2, 7: 6}[strip_tile]) * -(-W // 16)
    c1 = torch.empty(B, H, W, cin, device="cuda")
    ts = torch.full((B * tpi, cin, 2), -1.0, device="cuda", dtype=torch.float64)
    ops.conv2d_nhwc(p1, [(nhwc(D(x)), 0)], (c1, 0), ops.EPI_LINEAR, tile_stats=ts, tile=strip_tile)
    y64 = F.conv2d(D(x).double(), D(w1).double(), D(b1).double(), padding=1)
    n64 = F.relu(F.instance_norm(y64, eps=1e-5))
    t = ts.view(B, tpi, cin, 2).double().sum(1)
    assert float((t[..., 0] - y64.sum((2, 3))).abs().max()) <= 2e-5 * float(y64.abs().sum((2, 3)).max())
    assert float((t[..., 1] - (y64 * y64).sum((2, 3))).abs().max()) <= 2e-5 * float((y64 * y64).sum((2, 3)).max())
    got = ops.instnorm_tiles_nhwc(c1, ts, relu=True)
    assert float((nchw(got).double() - n64).abs().max()) < 2e-5
    mr = ops.instnorm_tiles_nhwc(c1, ts, stats_only=True)
    fused = torch.empty(B, H, W, cout, device="cuda")
    ops.conv2d_nhwc(p2, [(c1, 0)], (fused, 0), ops.EPI_LINEAR, in_norm=mr, tile=strip_tile)
    plain = torch.empty(B, H, W, cout, device="cuda")
    ops.conv2d_nhwc(p2, [(got, 0)], (plain, 0), ops.EPI_LINEAR, tile=strip_tile)
    assert torch.equal(fused, plain)
    z64 = F.conv2d(n64, D(w2).double(), D(b2).double(), padding=1)
    z32 = F.conv2d(n64.float(), D(w2), D(b2), padding=1)
    check(nchw(fused), z64, z32, "strip: fused norm + conv")

@pytest.mark.gpu
@pytest.mark.parametrize("hl", [False, True])
@pytest.mark.parametrize("B,H,W,segs,cout,kh,kw", [(2, 20, 32, [128, 128], 256, 1, 5), (2, 20, 32, [256], 192, 3, 3), (1, 40, 48, [64], 64, 3, 3)])
def test_conv_strip_single_product_is_plain_fp16(ops, hl, B, H, W, segs, cout, kh, kw):
    """cfg.raft.mixed_precision (single_product=True, 160-row strips): ONE product per multiply-add -- the result is the fp16-rounded
    operands' convolution in fp32: exact against fp64 on the ROUNDED operands to fp32 round-off, ~2^-11 per product against the
    unrounded ones; the three-product default on the same inputs is three orders closer.  Other tile requests ignore the flag."""
    cin = sum(segs)
    x = syn.normal("sp.x", (B, cin, H, W), 3, std=1.5)
    w = syn.normal("sp.w", (cout, cin, kh, kw), 3, std=float(np.sqrt(2.0 / (cin * kh * kw))))
    b = syn.uniform("sp.b", (cout,), 3, -0.5, 0.5)
    pc = ops.PackedConv(D(w), D(b), segs)
    xs, o = [], 0
    for c in segs:
        t = nhwc(D(x[:, o:o + c]))
        xs.append((ops.split_hl(t) if hl else t, 0))
        o += c
    out1 = torch.empty(B, H, W, cout, device="cuda")
    out3 = torch.empty(B, H, W, cout, device="cuda")
    ops.conv2d_nhwc(pc, xs, (out1, 0), ops.EPI_LINEAR, src_hl=hl, tile=5, single_product=True)
    ops.conv2d_nhwc(pc, xs, (out3, 0), ops.EPI_LINEAR, src_hl=hl, tile=5)
    pad = (kh // 2, kw // 2)
    y64 = F.conv2d(D(x).double(), D(w).double(), D(b).double(), padding=pad)
    # the operands as the kernel rounds them: fp16 of value * scale (activations ops.A_SCALE, weights the packer's power of two)
    r16 = lambda t, s: (t * s).half().double() / s
    yr = F.conv2d(r16(D(x), ops.A_SCALE), r16(D(w), pc.w_scale), D(b).double(), padding=pad)
    scale = float(y64.abs().max())
    e1, e3 = float((nchw(out1).double() - y64).abs().max()), float((nchw(out3).double() - y64).abs().max())
    assert e3 < 2e-5 * scale and 20 * e3 < e1 < 4e-3 * scale
    assert float((nchw(out1).double() - yr).abs().max()) < 1e-2 * e1 + 3e-6 * scale      # (it IS the product of the rounded operands)
    ref = torch.empty(B, H, W, cout, device="cuda")
    ops.conv2d_nhwc(pc, xs, (ref, 0), ops.EPI_LINEAR, src_hl=hl, tile=1, single_product=True)        # the 128-row kernels keep three products
    assert float((nchw(ref).double() - y64).abs().max()) < 2e-5 * scale



def test_conv_strip_is_the_automatic_choice_at_the_update_block_shape(ops):
    """60 x 80 maps (the headline's 1/8 resolution), batch 8: the automatic tile choice takes 160-row strips (30 patches per image instead
    of 40); launches of at most 8192 pixels take 32-row strips (2 x 16 patches) -- a single 60 x 80 map, a single 30 x 30 crop; RNNPOSE_STRIP=0 / ops.conv_strip(False) restores the 128-row kernels; same result bit for bit (the order of
    the K sum of an output element does not depend on the tile)."""
    assert ops.conv_tiles_per_image(60, 80, 3, 3, 1, 192, 0, 8) == 30 and ops.conv_tiles_per_image(60, 80, 3, 3, 1, 192, 1, 8) == 40
    assert ops.conv_tiles_per_image(60, 80, 3, 3, 1, 192, 0, 1) == 150 and ops.conv_tiles_per_image(60, 80, 3, 3, 1, 192, 6, 8) == 150
    assert ops.conv_tiles_per_image(60, 80, 3, 3, 1, 192, 0, 2) == 50      # (two images: 9600 pixels; 160-row strips would be 120 three-wave workgroups = 360 waves: r06 takes 96-row strips, 6 x 16 patches)
    assert ops.conv_tiles_per_image(60, 80, 3, 3, 1, 192, 7, 8) == 50 and ops.conv_tiles_per_image(60, 80, 3, 3, 1, 64, 0, 4) == 150      # (convf2 of a half-batch chain: 240 waves -> 32-row strips)
    assert ops.conv_tiles_per_image(60, 80, 3, 3, 1, 126, 0, 4) == 50 and ops.conv_tiles_per_image(60, 80, 3, 3, 1, 126, 0, 8) == 30        # (128-column layers: 480 waves at B = 4 -> 96 rows; 960 at B = 8 -> 160)
    assert ops.conv_tiles_per_image(30, 30, 3, 3, 1, 192, 0, 1) == 30      # LINEMOD crops, one image: 15 x 2 patches of 2 x 16
    B, H, W = 2, 60, 80
    x = D(syn.normal("ax", (B, H, W, 256), 31, std=1.2))
    w = D(syn.normal("aw", (192, 256, 3, 3), 31, std=0.02))
    pc = ops.PackedConv(w, D(syn.uniform("ab", (192,), 31, -0.5, 0.5)), [256])
    a, b = torch.empty(B, H, W, 192, device="cuda"), torch.empty(B, H, W, 192, device="cuda")
    ops.conv2d_nhwc(pc, [(x, 0)], (a, 0), ops.EPI_RELU)
    ops.conv2d_nhwc(pc, [(x, 0)], (b, 0), ops.EPI_RELU, tile=5)
    assert torch.equal(a, b)
    try:
        ops.conv_strip(False)
        assert ops.conv_tiles_per_image(60, 80, 3, 3, 1, 192, 0, 8) == 40
        ops.conv2d_nhwc(pc, [(x, 0)], (a, 0), ops.EPI_RELU)
    finally:
        ops.conv_strip(True)
    assert not torch.equal(a, b) and float((a - b).abs().max()) < 2e-6 * float(b.abs().max())


@pytest.mark.parametrize("B,cin,cout,H,W,norm", [(8, 64, 64, 240, 320, True), (8, 64, 64, 240, 320, False), (8, 96, 96, 120, 160, True), (8, 128, 128, 96, 160, False),
                                                 (2, 64, 64, 40, 48, True), (3, 96, 96, 30, 48, False), (2, 128, 128, 50, 33, True)])
def test_conv_strip_persistent_launch_is_bit_identical(ops, B, cin, cout, H, W, norm):
    """r06: the PERSISTENT form of the fp32-source strip kernels (ops.conv_strip(7): as many workgroups as the
    chip holds, each walking its share of the tile list and pulling the next tile's first activations towards the CU in front of its
    epilogue) at the encoder's launch sizes -- same tiles, same arithmetic, same order: outputs and tile statistics equal the one-tile-
    per-workgroup launch bit for bit, with and without the fused input normalisation.  (Measured slower: off by default.)  The small
    shapes are persistent launches on the host-execution tier (a 4-CU "device": 8 / 16 resident workgroups, tests/host_exec), where
    every workgroup walks 2-3 tiles incl. ragged ones; on the GPU they take the ordinary launch."""
    x = torch.randn(B, H, W, cin, generator=torch.Generator().manual_seed(5)).mul_(1.3).add_(0.2).to("cuda")
    w = D(syn.normal("pw", (cout, cin, 3, 3), 9, std=float(np.sqrt(2.0 / (cin * 9)))))
    pc = ops.PackedConv(w, D(syn.uniform("pb", (cout,), 9, -0.5, 0.5)), [cin])
    mr = None
    if norm:
        mean = x.mean((1, 2))
        rstd = 1.0 / torch.sqrt(x.var((1, 2), unbiased=False) + 1e-5)
        mr = torch.stack([mean, rstd], -1).contiguous()
    outs, stats = [], []
    try:
        for mode in (1, 7):
            ops.conv_strip(mode)
            tpi = ops.conv_tiles_per_image(H, W, 3, 3, 1, cout, 0, B, src_counts=[cin], fused_norm=norm)
            ts = torch.full((B * tpi, cout, 2), -1.0, device="cuda", dtype=torch.float64)
            out = torch.empty(B, H, W, cout, device="cuda")
            ops.conv2d_nhwc(pc, [(x, 0)], (out, 0), ops.EPI_LINEAR, tile_stats=ts, in_norm=mr, src_bounded=True)
            outs.append(out)
            stats.append(ts)
    finally:
        ops.conv_strip(1)
    assert torch.equal(outs[0], outs[1]) and torch.equal(stats[0], stats[1])
    y64 = F.conv2d((F.relu((x - mr[:, None, None, :, 0]) * mr[:, None, None, :, 1]) if norm else x).permute(0, 3, 1, 2).double(), w.double(), pc.bias.double(), padding=1)
    assert float((outs[1].permute(0, 3, 1, 2).double() - y64).abs().max()) < 2e-5 * float(y64.abs().max())


@pytest.mark.parametrize("cin", [16, 48])
def test_conv_tile_stats_with_narrow_sources_fall_back_to_the_128_row_kernels(ops, cin):
    """a launch whose shape would get strips (3x3, 96 columns, a map that fills the chip) but whose source is not whole
    32-channel blocks.  r04 raised an error as soon as tile statistics were asked for (the strip path was taken because the caller had
    "sized its buffer for strips"); since ABI 3 the choice is made in ONE place, from the descriptor, the query
    rnnpose_conv_tiles_per_image_desc answers for exactly that launch, and the launch checks the record count it is given."""
    B, H, W, cout = 2, 60, 80, 96
    x = syn.normal("nf.x", (B, cin, H, W), 3, std=1.5)
    w = syn.normal("nf.w", (cout, cin, 3, 3), 3, std=float(np.sqrt(2.0 / (cin * 9))))
    b = syn.uniform("nf.b", (cout,), 3, -0.5, 0.5)
    pc = ops.PackedConv(D(w), D(b), [cin])
    tpi_strips = ops.conv_tiles_per_image(H, W, 3, 3, 1, cout, 0, B)                       # shape-only rule: strips (r06: 60 three-wave workgroups of 160 rows = 180 waves -> 32-row strips, 2 x 16 patches)
    tpi = ops.conv_tiles_per_image(H, W, 3, 3, 1, cout, 0, B, src_counts=[cin])           # this launch: 8 x 16 patches of the 128-row kernel
    assert tpi_strips == 30 * 5 and ops.conv_tiles_per_image(H, W, 3, 3, 1, cout, 5, B) == 6 * 5 and tpi == 8 * 5
    out = torch.empty(B, H, W, cout, device="cuda")
    ts = torch.full((B * tpi, cout, 2), -1.0, device="cuda", dtype=torch.float64)
    ops.conv2d_nhwc(pc, [(nhwc(D(x)), 0)], (out, 0), ops.EPI_LINEAR, tile_stats=ts)
    y64 = F.conv2d(D(x).double(), D(w).double(), D(b).double(), padding=1)
    assert float((nchw(out).double() - y64).abs().max()) < 2e-5 * float(y64.abs().max())
    want = torch.stack([y64.sum((2, 3)), (y64 * y64).sum((2, 3))], -1)
    assert float((ts.view(B, tpi, cout, 2).sum(1) - want).abs().max()) <= 2e-5 * float(want.abs().max())
    with pytest.raises(ValueError):                                                       # a buffer sized by the shape-only rule is refused
        ops.conv2d_nhwc(pc, [(nhwc(D(x)), 0)], (out, 0), ops.EPI_LINEAR, tile_stats=torch.zeros(B * tpi_strips, cout, 2, device="cuda", dtype=torch.float64))
    # ... and by the C entry point itself when a caller bypasses the wrapper's check (tile_stats_records too small)
    from rnnpose_amd import _lib
    import ctypes as C
    d = _lib.ConvDesc()
    xs = nhwc(D(x))
    d.src[0] = _lib.ConvSrc(xs.data_ptr(), cin, 0, cin)
    d.n_src, d.B, d.H, d.W, d.kh, d.kw, d.stride = 1, B, H, W, 3, 3, 1
    d.w_packed, d.bias, d.c_out, d.a_scale, d.w_scale, d.epilogue = pc.w_packed.data_ptr(), pc.bias.data_ptr(), cout, pc.a_scale, pc.w_scale, 0
    d.dst, d.dst_c_stride, d.dst_c_offset = out.data_ptr(), cout, 0
    d.tile_stats, d.tile_stats_records = ts.data_ptr(), B * tpi - 1
    lib = _lib.load()
    assert lib.rnnpose_conv_tiles_per_image_desc(C.byref(d)) == tpi
    assert lib.rnnpose_conv2d_nhwc_f16x3(C.byref(d), None) != 0 and b"tile_stats_records" in lib.rnnpose_last_error()
    d.tile_stats_records = B * tpi + 1                              # r06: an OVERSIZED record count is refused too -- a consumer takes the
    assert lib.rnnpose_conv2d_nhwc_f16x3(C.byref(d), None) != 0 and b"tile_stats_records" in lib.rnnpose_last_error()      # tiling from the count
    d.tile_stats_records = B * tpi
    assert lib.rnnpose_conv2d_nhwc_f16x3(C.byref(d), None) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("k", [3, 1])
@pytest.mark.parametrize("B,H,W,cin,cout,stats", [(8, 240, 320, 64, 96, True), (8, 120, 160, 96, 128, True), (2, 240, 320, 64, 96, False),
                                                     (3, 96, 160, 32, 64, True), (2, 100, 112, 64, 96, True), (2, 120, 120, 64, 96, True), (1, 60, 60, 96, 128, True)])
def test_conv_stride2_strips_over_parity_planes(ops, B, H, W, cin, cout, stats, k):
    """r05: a stride-2 3x3 layer as a stride-1 layer with 2 x 2 taps over the four parity planes of its input --
    strided views of the NHWC source -- on 160-row strips (csrc/conv_strip.hip).  The encoder's l2.0.c1 / l3.0.c1 shapes at the headline
    resolution, a smaller batch, a two-wave shape, a ragged patch grid and the single-image crop's shapes (32-row strips): against torch in fp64, against the 128-row kernel's stride-2
    mode (same products, another summation order), with the fp64 tile statistics the following instance norm consumes.  k = 1: the 1x1
    stride-2 down-sampling branches (extractor.py:36-39) in the same form -- plane (0, 0) only, one tap in four."""
    x = syn.normal("s2.x", (B, cin, H, W), 6, std=1.5) + 0.3
    w = syn.normal("s2.w", (cout, cin, k, k), 6, std=float(np.sqrt(2.0 / (cin * k * k))))
    b = syn.uniform("s2.b", (cout,), 6, -0.5, 0.5)
    pc = ops.PackedConv(D(w), D(b), [cin])
    xs = nhwc(D(x))
    Ho, Wo = H // 2, W // 2
    y64 = F.conv2d(D(x).double(), D(w).double(), D(b).double(), stride=2, padding=k // 2)
    outs = {}
    try:
        for mode in (1, 5):                                   # 1: automatic (the strip form where it applies), 5: without the stride-2 form
            ops.conv_strip(mode)
            tpi = ops.conv_tiles_per_image(H, W, k, k, 2, cout, 0, B, src_counts=[cin])
            out = torch.full((B, Ho, Wo, cout), 7.0, device="cuda")
            ts = torch.full((B * tpi, cout, 2), -1.0, device="cuda", dtype=torch.float64) if stats else None
            ops.conv2d_nhwc(pc, [(xs, 0)], (out, 0), ops.EPI_LINEAR, stride=2, tile_stats=ts, src_bounded=True)
            outs[mode] = (out, ts, tpi)
    finally:
        ops.conv_strip(1)
    if H >= 120 and W >= 160:       # the encoder's shapes: the strip form must actually be taken (10 x 16 patches of the OUTPUT grid)
        assert outs[1][2] == -(-Ho // 10) * -(-Wo // 16) and outs[5][2] == -(-(Ho * Wo) // 128), (outs[1][2], outs[5][2])
    scale = float(y64.abs().max())
    for mode, (out, ts, tpi) in outs.items():
        err = float((nchw(out).double() - y64).abs().max())
        assert err <= 3e-6 * scale + 1e-6, (mode, err, scale)
        if stats:
            want = torch.stack([y64.sum((2, 3)), (y64 * y64).sum((2, 3))], -1)
            assert float((ts.view(B, tpi, cout, 2).sum(1) - want).abs().max()) <= 2e-5 * float(want.abs().max()), mode
    assert float((outs[1][0] - outs[5][0]).abs().max()) <= 2e-6 * scale + 1e-6
