"""Bit-reproducibility of the refinement loop and of kernels that run next to other kernels (r05).

r04 found fresh refiner instances on identical inputs differing in 64-byte runs of a weight map whenever two streams were active and
took it for a kernel-to-kernel visibility problem.  r05 (profiles/r05_determinism.txt) reduced it to ONE kernel on constant inputs:
corr_weight, compiled by plain -O3 into packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32), returned other values for groups of
16 lanes while mask_upsample or conv1x1_resident -- the two kernels built on v_mfma_f32_16x16x32_f16 -- ran on the other stream; a
plain-HIP probe (tools/probes/pk_f32_vs_mfma.hip) shows the same between a packed-fp32 kernel and a 16x16x32 MFMA loop, and never for
scalar fp32.  The library is built with -fno-slp-vectorize since (tests/test_isa_guard.py guards the ISA); these tests guard the
behaviour: "one deterministic sequence per image" (model/PoseRefiner.py:315-362) whatever else the chip is doing."""
import numpy as np
import pytest
import torch

from oracle import rnnpose_oracle as orc          # (weight generator only)
from rnnpose_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from rnnpose_amd import build, ops as _ops
    build.build()
    return _ops


_INPUTS = {}


def _inputs(B, H, W):
    """Fixed inputs of one shape, generated ONCE on the device (the hash generator of rnnpose_amd.synthetic takes minutes per call at
    B = 8; this test needs identical inputs for every instance, not inputs the CPU oracle can regenerate)."""
    key = (B, H, W)
    if key not in _INPUTS:
        dev = "cuda"
        g = torch.Generator(device=dev)
        g.manual_seed(21)
        r = lambda *s: torch.randn(*s, device=dev, generator=g)
        h, w = H // 8, W // 8
        g1, g2 = r(B, 32, H, W), r(B, 32, H, W)
        g1 /= g1.norm(dim=1, keepdim=True)
        g2 /= g2.norm(dim=1, keepdim=True)
        depth = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.3 + 0.9
        depth[:, :, : H // 4] = 0
        K = torch.tensor([[572.4114, 0, W / 2], [0, 573.57043, H / 2], [0, 0, 1]], device=dev).repeat(B, 1, 1)
        G0 = torch.from_numpy(syn.se3_exp_np(syn.normal("g0", (B, 6), 21, std=0.02)).astype(np.float32)).to(dev)[:, None]
        _INPUTS[key] = dict(cfea=0.1 * r(B, 256, H, W), geofea1=g1, geofea2_crop=g2, syn_depth=depth, intrinsics_crop=K,
                            fmap1=r(B, 256, h, w), fmap2=r(B, 256, h, w), G0=G0, K=K,
                            img1=torch.rand(B, 3, H, W, device=dev, generator=g) * 255, img2=torch.rand(B, 3, H, W, device=dev, generator=g) * 255)
    return _INPUTS[key]


def _instance_outputs(ops, B, H, W, iters, use_graph, encoder=False):
    """One FRESH PoseRefiner on fixed inputs -> {name: tensor} of every per-iteration output (flow map, weight map, H, b, xi, G) plus
    the number of pixels whose recorded weight is not the weight of the recorded flow map."""
    from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config
    from rnnpose_amd.transformation import SE3Sequence
    d = _inputs(B, H, W)
    z3 = torch.zeros(B, 3, H, W, device="cuda")
    kw = {k: d[k] for k in ("cfea", "geofea1", "geofea2_crop", "syn_depth", "intrinsics_crop", "fmap1", "fmap2")}
    kw.update(syn_img=z3, image_crop=z3)
    if encoder:
        kw.update(syn_img=d["img1"], image_crop=d["img2"], fmap1=None, fmap2=None)
    torch.manual_seed(0)
    ref = PoseRefiner(default_config(RENDER_ITER_COUNT=1, ITER_COUNT=iters, OPTIM_ITER_COUNT=1), renderer=SyntheticRenderer(**kw),
                      use_graph=use_graph).cuda().eval()
    ref.cf_net.update_block.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0).items()})
    out = ref(None, SE3Sequence(matrix=d["G0"].clone()), d["K"])
    if use_graph:                      # the first call captured (and ran eagerly while warming up): the replayed result is the second one
        out = ref(None, SE3Sequence(matrix=d["G0"].clone()), d["K"])
    torch.cuda.synchronize()
    res = {"flow_last": out["flow_last"].clone(), "Ti": out["Ti_pred"].G.clone(), "weight": out["weight"].clone()}
    for i, T in enumerate(ref.residual_pose_history):
        res[f"it{i}.G"] = T.G.clone()
        Hm, bv, xi = T.last_system
        res[f"it{i}.H"], res[f"it{i}.b"], res[f"it{i}.xi"] = Hm.clone(), bv.clone(), xi.clone()
    for i, fl in enumerate(ref.flow_history):
        res[f"it{i}.flow"] = fl[0].clone()
    # the weight map the loop recorded for the last iteration must be the weight of that iteration's flow map
    chk = ops.corr_weight(kw["geofea1"], kw["geofea2_crop"], out["flow_last"], kw["syn_depth"], ref.sigma[0])
    res["_weight_mismatch"] = int((chk != out["weight"].reshape(chk.shape)).sum())
    res["_chains"] = ref.cf_net.engine().halves(B)
    return res


ONE = {"RNNPOSE_SPLIT_BATCH": "0", "RNNPOSE_ENCODER_MERGE": "1"}
TWO = {"RNNPOSE_SPLIT_BATCH": "1", "RNNPOSE_ENCODER_MERGE": "0"}
SHAPES = {
    # name: (B, H, W, inner iterations, use_graph, encoder in the loop, environment)
    "headline_default_eager": (8, 480, 640, 2, False, False, {}),
    "headline_default_graph_encoder": (8, 480, 640, 2, True, True, {}),
    "headline_two_chains_two_encoder_streams_graph": (8, 480, 640, 2, True, True, TWO),
    "headline_one_chain_one_stream_graph": (8, 480, 640, 2, True, True, ONE),
    "S1_default_graph": (1, 240, 240, 4, True, True, {}),
    "S1_default_eager": (1, 240, 240, 4, False, False, {}),
    "S1_helper_stream_graph": (1, 240, 240, 4, True, False, {"RNNPOSE_SIDE_STREAM": "1"}),
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_default_schedule_is_bit_reproducible(ops, monkeypatch, name):
    """8 fresh instances after the first on identical inputs: every output of every iteration bit-identical to the first instance's --
    for the default schedule (two half-batch chains + one encoder stream per image set at the headline shape; one chain at the
    reference's own working size, B = 1, 240 x 240), eager and replayed, for the one-stream schedule, and for the single-image helper
    stream."""
    B, H, W, iters, graph, enc, env = SHAPES[name]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 8
    first = _instance_outputs(ops, B, H, W, iters, graph, enc)
    assert first["_weight_mismatch"] == 0
    keep = []                                   # (keeps the allocator from handing every instance the same blocks)
    for t in range(n):
        keep.append(torch.full((1 << 22,), float(t), device="cuda"))
        cur = _instance_outputs(ops, B, H, W, iters, graph, enc)
        assert cur["_weight_mismatch"] == 0, f"{name}: instance {t}: recorded weight is not the weight of the recorded flow map"
        bad = [k for k in first if not k.startswith("_") and not torch.equal(first[k], cur[k])]
        assert not bad, f"{name}: instance {t} differs from the first in {bad}"


def test_two_chains_are_bit_identical_to_one_chain(ops, monkeypatch):
    """Images are independent and at this size both schedules launch the same kernel families (160-row strips from 8192 pixels per
    launch): two half-batch loops on two streams must give the SAME BITS as one full-batch loop (the relaxed comparison of
    test_two_chains_equal_one_chain, which crosses kernel families at its small shape, would let a real ordering race pass)."""
    outs = []
    for env in (ONE, TWO):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs.append(_instance_outputs(ops, 4, 480, 640, 2, False))
    bad = [k for k in outs[0] if not k.startswith("_") and not torch.equal(outs[0][k], outs[1][k])]
    assert not bad, bad


@pytest.mark.parametrize("graph", [False, True])
def test_uneven_chains_are_bit_identical_to_one_chain(ops, monkeypatch, graph):
    """An ODD batch (the partial last batch of an evaluation epoch: utils/distributed_utils.py:150-169 pads the shard, not the batch)
    is cut 2 + 3 by the two-chain schedule: two loops of DIFFERENT size on two streams, each with its own workspaces and graph.  Every
    launch of either chain still has >= 8192 pixels at 1/8 resolution, i.e. the kernel families of the full-batch launch: the SAME
    BITS as one 5-image loop, eager and replayed."""
    outs = []
    for env, chains in ((ONE, [(0, 5)]), (TWO, [(0, 2), (2, 5)])):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o = _instance_outputs(ops, 5, 480, 640, 2, graph)
        assert o["_chains"] == chains, o["_chains"]
        assert o["_weight_mismatch"] == 0
        outs.append(o)
    bad = [k for k in outs[0] if not k.startswith("_") and not torch.equal(outs[0][k], outs[1][k])]
    assert not bad, bad


@pytest.mark.parametrize("shape", [(8, 480, 640, 2, True), (1, 240, 240, 4, True), (3, 480, 640, 2, False)])
def test_coordinates_formed_inside_their_consumers_change_no_bit(ops, monkeypatch, shape):
    """r06: with RNNPOSE_FUSED_INDUCED=1 (an option: measured equal) the lookup / flow-feature launches of an iteration form the pose-induced coordinates
    themselves, with 0 a launch of its own writes them first: every output of every iteration (flow, weight, H, b, xi, G) is the same
    BITS either way -- headline shape (two chains), the single-image crop, an uneven eager batch."""
    B, H, W, iters, graph = shape
    outs = []
    for v in ("0", "1"):
        monkeypatch.setenv("RNNPOSE_FUSED_INDUCED", v)
        o = _instance_outputs(ops, B, H, W, iters, graph, encoder=True)
        assert o["_weight_mismatch"] == 0
        outs.append(o)
    bad = [k for k in outs[0] if not k.startswith("_") and not torch.equal(outs[0][k], outs[1][k])]
    assert not bad, bad


def test_pointwise_kernels_are_exact_next_to_16x16x32_mfma_kernels(ops):
    """The reduced form of r04's finding: a VALU kernel on stream A, mask_upsample / conv1x1_resident (v_mfma_f32_16x16x32_f16) on
    stream B, constant inputs -- every launch must reproduce the kernel's solo output bit for bit.  (With pointwise.hip built by plain
    -O3 -- packed fp32 -- about half of the corr_weight launches differed in one or more 16-pixel runs.)"""
    from rnnpose_amd.streams import reserve
    dev = "cuda"
    B, H, W = 4, 480, 640
    h, w = H // 8, W // 8
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    g1, g2 = r(B, 32, H, W), r(B, 32, H, W)
    g1 /= g1.norm(dim=1, keepdim=True)
    g2 /= g2.norm(dim=1, keepdim=True)
    depth = torch.rand(B, 1, H, W, device=dev, generator=g) * 0.3 + 0.9
    depth[:, :, : H // 4] = 0
    sigma = torch.ones(1, device=dev)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    flow = torch.stack([3.0 * torch.sin(xs / 70.0) + 1.5 * torch.cos(ys / 45.0), 2.0 * torch.cos(xs / 90.0) - 2.5 * torch.sin(ys / 60.0)])[None].repeat(B, 1, 1, 1).float().contiguous()
    K = torch.tensor([[572.4, 0, W / 2], [0, 573.6, H / 2], [0, 0, 1]], device=dev).repeat(B, 1, 1)
    G = ops.se3_exp(r(B, 6) * 0.02)
    mask = r(B, h, w, 576)
    flow_lr = r(B, h, w, 2)
    mhead = ops.PackedMaskHead(r(576, 256, 1, 1) * 0.09, r(576) * 0.1)
    heads = r(B, h, w, 512).clamp_(min=0)
    c1r = ops.PackedConv1x1(r(256, 324, 1, 1) * 0.08, r(256) * 0.1)
    corr, cor1 = r(B, h, w, 324), torch.empty(B, h, w, 256, device=dev)
    up_b = torch.empty(B, 2, H, W, device=dev)
    wm = ops.corr_weight(g1, g2, flow, depth, sigma)
    cases = {
        "corr_weight": (lambda out: ops.corr_weight(g1, g2, flow, depth, sigma, out=out), torch.empty_like(wm)),
        "convex_upsample": (lambda out: ops.convex_upsample_nhwc(flow_lr, mask, out=out), torch.empty(B, 2, H, W, device=dev)),
        "lm_step": (lambda out: out.copy_(ops.lm_step(flow, wm, depth, K, G)[1].reshape(-1)[: out.numel()]), torch.empty(B * 36, device=dev, dtype=torch.float64)),
    }
    loads = {
        "mask_upsample": lambda: ops.mask_upsample(mhead, heads, 256, flow_lr, out=up_b),
        "conv1x1_resident": lambda: ops.conv1x1_resident(c1r, (corr, 0), (cor1, 0)),
    }
    sB = reserve(torch.device(dev)).chain[0]
    for cname, (fn, out) in cases.items():
        fn(out)
        torch.cuda.synchronize()
        solo = out.clone()
        for lname, load in loads.items():
            bad = 0
            for k in range(40):
                out.zero_()
                with torch.cuda.stream(sB):
                    load()
                    load()
                fn(out)
                torch.cuda.synchronize()
                bad += int(not torch.equal(out, solo))
            assert bad == 0, f"{cname} next to {lname}: {bad} of 40 launches differ from the solo result"


def test_packed_fp32_neighbour_next_to_the_16x16_tile_kernels_and_the_loop(ops):
    """r06: the exposure BEYOND this library's own code objects.  A co-tenant kernel built the way an integrator's
    code is built -- plain -O3, packed fp32 (tests/probes/pk_neighbour.hip) -- runs on its own stream while library kernels run on
    theirs; every launch of the neighbour is compared on the device with the neighbour's solo output.
      * mask_upsample / conv1x1_resident: on v_mfma_f32_16x16x32_f16 (r02-r05) 1599 of 1600 / 4 of 1600 neighbour launches differed
        next to them; on the 16x16x16 shape (r06 default, csrc/f16x3.cuh mfma16_c32) NONE may -- asserted.
      * the whole refinement loop: the count is RECORDED, not asserted.  367 of 1600 with the K = 32 shape, 12-75 of 800 with the
        K = 16 shape: the stem and the volume kernel (v_mfma_f32_32x32x16_f16 under full load) still disturb a packed-fp32 neighbour in
        ~20 % of the launches that overlap them, although a bare 32x32x16 MFMA loop on constant operands never does
        (profiles/r06_pk_neighbour.txt).  That residue is the chip's, not this library's to fix: INTEGRATION.md tells integrators to build
        co-tenant code with -fno-slp-vectorize.  The library's own kernels contain no packed fp32 (tests/test_isa_guard.py) and are
        bit-reproducible under every schedule (the tests above)."""
    import os
    import sys
    import warnings
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
    import pk_neighbour
    from rnnpose_amd.pose_refiner import PoseRefiner, SyntheticRenderer, default_config
    from rnnpose_amd.transformation import SE3Sequence
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this box: the neighbour kernel cannot be built")
    nb = pk_neighbour.Neighbour()
    assert "v_pk_" in nb.isa and "_f32" in nb.isa, "the neighbour is meant to be made of packed fp32 instructions"
    bad, n, _ = pk_neighbour.next_to(nb, lambda: None, 200)
    assert bad == 0 and n == 200, "the neighbour differs from itself with nothing else on the chip"
    B, H, W = 8, 480, 640
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    h, w = H // 8, W // 8
    mhead = ops.PackedMaskHead(r(576, 256, 1, 1) * 0.09, r(576) * 0.1)
    heads, flow_lr, up = r(4, h, w, 512).clamp_(min=0), r(4, h, w, 2), torch.empty(4, 2, H, W, device=dev)
    c1r = ops.PackedConv1x1(r(256, 324, 1, 1) * 0.08, r(256) * 0.1)
    corr, cor1 = r(4, h, w, 324), torch.empty(4, h, w, 256, device=dev)
    for name, load in (("mask_upsample", lambda: [ops.mask_upsample(mhead, heads, 256, flow_lr, out=up) for _ in range(4)]),
                       ("conv1x1_resident", lambda: [ops.conv1x1_resident(c1r, (corr, 0), (cor1, 0)) for _ in range(4)])):
        bad, n, vals = pk_neighbour.next_to(nb, load, 800, per=4)
        assert bad == 0, f"{bad} of {n} launches of a packed-fp32 neighbour differ next to {name} ({vals} values)"
    d = _inputs(B, H, W)
    kw = {k: d[k] for k in ("cfea", "geofea1", "geofea2_crop", "syn_depth", "intrinsics_crop")}
    kw.update(syn_img=d["img1"], image_crop=d["img2"], fmap1=None, fmap2=None)            # (the encoder runs inside the loop)
    torch.manual_seed(0)
    ref = PoseRefiner(default_config(RENDER_ITER_COUNT=2, ITER_COUNT=8, OPTIM_ITER_COUNT=1), renderer=SyntheticRenderer(**kw)).cuda().eval()
    step = lambda: ref(None, SE3Sequence(matrix=d["G0"].clone()), d["K"])
    step()
    torch.cuda.synchronize()
    bad, n, vals = pk_neighbour.next_to(nb, step, 1200, per=300)
    msg = f"packed-fp32 neighbour next to the refinement loop: {bad} of {n} launches differ from the solo output ({vals} values)"
    print(msg)
    if bad:
        warnings.warn(msg + " -- the chip's packed-fp32 defect under MFMA load (INTEGRATION.md); recorded, not a failure of this library's results")
