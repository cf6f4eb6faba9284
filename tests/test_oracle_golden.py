"""Pin the CPU oracle (oracle/rnnpose_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/gen_golden.py from /root/reference).
Tolerances: north_star's 1e-4 on correlation/flow quantities, 1e-5 on pose quantities,
1e-7 relative on the fp64 normal equations (their inputs are fp32)."""
import numpy as np
import pytest
import torch

from oracle import rnnpose_oracle as orc
from rnnpose_amd import synthetic as syn


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def maxdiff(a, b):
    a = a.numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def excess(a, b, atol, rtol):
    """max(|a-b| - atol - rtol*|b|): <= 0 means within mixed tolerance (fp32 round-off scales with
    magnitude: re-projections of near-zero depths reach thousands of pixels)."""
    a = a.numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.max(np.abs(a - b) - atol - rtol * np.abs(b)))


def upd_weights(seed=0):
    return syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=seed)


def test_corr_pyramid_and_lookup(golden):
    g = golden("corr")
    B, C, h, w = 2, 256, 16, 24
    f1 = syn.normal("fmap1", (B, C, h, w), 11)
    f2 = syn.normal("fmap2", (B, C, h, w), 11)
    pyr = orc.corr_pyramid(f1, f2)
    assert [tuple(p.shape) for p in pyr] == [(768, 16, 24), (768, 8, 12), (768, 4, 6), (768, 2, 3)]
    assert maxdiff(pyr[0].reshape(B * h * w, h * w)[::5], g["level0_rows"]) < 1e-5
    for l in (1, 2, 3):
        assert maxdiff(pyr[l][:, None], g[f"level{l}"]) < 1e-5
    assert abs(float(pyr[0].double().sum()) - float(g["level0_sum"])) < 1e-2
    grid = orc.coords_grid_lowres(B, h, w)
    cases = {
        "int": grid.clone(),
        "sub": grid + T(syn.uniform("lk_sub", (B, 2, h, w), 11, -3.0, 3.0)),
        "oob": grid + T(syn.uniform("lk_oob", (B, 2, h, w), 11, -30.0, 30.0)),
    }
    for k, c in cases.items():
        out = orc.corr_lookup(pyr, c)
        assert out.shape == (B, 324, h, w)
        assert maxdiff(out[:, :, ::2, ::3], g[f"lookup_{k}"]) < 1e-4, k


def test_lookup_integer_coords_are_dot_products():
    """KAT without the reference: at integer coords level-0 taps equal direct dot products / 16."""
    B, C, h, w = 1, 256, 16, 16
    f1 = syn.normal("a", (B, C, h, w), 1)
    f2 = syn.normal("b", (B, C, h, w), 1)
    pyr = orc.corr_pyramid(f1, f2)
    out = orc.corr_lookup(pyr, orc.coords_grid_lowres(B, h, w))
    Y, X = 7, 9
    for i, j in ((4, 4), (0, 8), (8, 2)):   # channel i*9+j -> x offset i-4, y offset j-4
        x2, y2 = X + i - 4, Y + j - 4
        ref = float((f1[0, :, Y, X].astype(np.float64) * f2[0, :, y2, x2]).sum() / 16)
        assert abs(float(out[0, i * 9 + j, Y, X]) - ref) < 1e-5


def test_update_block(golden):
    g = golden("update_block")
    B, h, w = 1, 16, 20
    hid = np.tanh(syn.normal("u_net", (B, 128, h, w), 3))
    inp = np.maximum(syn.normal("u_inp", (B, 128, h, w), 3), 0)
    corr = syn.normal("u_corr", (B, 324, h, w), 3)
    flow = syn.normal("u_flow", (B, 2, h, w), 3, std=2.0)
    net, mask, df = orc.update_block(upd_weights(), hid, inp, corr, flow)
    assert maxdiff(net, g["net"]) < 1e-5
    assert maxdiff(mask, g["mask"]) < 1e-4
    assert maxdiff(df, g["dflow"]) < 1e-4


def test_upsample_ctx_flowinit(golden):
    g = golden("upsample_ctx")
    B, h, w = 2, 16, 12
    flow = syn.normal("up_flow", (B, 2, h, w), 5, std=3.0)
    mask = syn.normal("up_mask", (B, 576, h, w), 5, std=2.0)
    assert maxdiff(orc.convex_upsample(flow, mask), g["flow_up"]) < 1e-4
    ctx = syn.normal("ctx", (1, 256, 64, 96), 5, std=0.1)
    net, inp = orc.context_prep(ctx)
    assert maxdiff(net, g["net"]) < 1e-6 and maxdiff(inp, g["inp"]) < 1e-6
    finit = syn.normal("finit", (2, 2, 64, 96), 5, std=4.0)
    keep = finit.copy()
    assert maxdiff(orc.flow_init_to_coords1(finit), g["coords1"]) < 1e-5
    assert np.array_equal(keep, finit), "oracle must not mutate flow_init"


def test_induced_flow_and_weight(golden):
    g = golden("geometry")
    d = syn.make_inputs(2, 64, 96, seed=7, pose_sigma=0.03)
    flow, vmask = orc.induced_flow(g["depth"], d["K"], g["G"])
    assert excess(flow[:, None], g["flow_init"], 1e-4, 1e-6) <= 0
    assert maxdiff(vmask[:, None, :, :, None], g["vmask"]) == 0
    w = orc.corr_weight(d["g1"], d["g2"], T(g["target"])[:, 0], g["depth"], g["sigma"])
    assert maxdiff(w[:, None, :, :, None], g["weight"]) < 1e-4


@pytest.mark.parametrize("pat", ["desc", "ones", "sparse", "zero"])
def test_lm_step(golden, pat):
    g = golden("geometry")
    d = syn.make_inputs(2, 64, 96, seed=7, pose_sigma=0.03)
    B, H, W = 2, 64, 96
    wgt = {
        "desc": T(g["weight"])[:, 0, :, :, 0],
        "ones": torch.ones(B, H, W),
        "sparse": (T(syn.uniform("wsp", (B, 1, H, W, 1), 7)) > 0.97).float()[:, 0, :, :, 0] * 2.5,
        "zero": torch.zeros(B, H, W),
    }[pat]
    tgt = T(g["target"])[:, 0]
    Hm, b = orc.lm_normal_eq(tgt, wgt, g["depth"], d["K"], g["G"])
    Hd = Hm + orc.EP_LMBDA * torch.eye(6, dtype=torch.float64) + orc.LM_LMBDA * Hm * torch.eye(6, dtype=torch.float64)
    ref_H, ref_b = g[f"lm_{pat}_Hd0"][:, 0], g[f"lm_{pat}_b0"][:, 0]
    # H,b are fp64 sums of Jacobians built from fp32 points: the fp32 part (G*X summation order inside the
    # reference's einsum/bmm) bounds agreement at fp32-epsilon level, not fp64.
    assert maxdiff(Hd, ref_H) <= 1e-7 * max(1.0, float(np.abs(ref_H).max()))
    assert maxdiff(b, ref_b) <= 1e-7 * max(1.0, float(np.abs(ref_b).max()))
    G1, _ = orc.lm_step(tgt, wgt, g["depth"], d["K"], g["G"], 1)
    assert maxdiff(G1, g[f"lm_{pat}_G1"]) < 1e-5
    G2, _ = orc.lm_step(tgt, wgt, g["depth"], d["K"], g["G"], 2)
    assert maxdiff(G2, g[f"lm_{pat}_G"]) < 1e-5
    if pat == "zero":                      # H = 100 I -> xi = 0 -> pose unchanged
        assert maxdiff(G1, g["G"]) < 1e-7


def test_exact_target_recovery(golden):
    """KAT: identity start, exact projected targets, unit weights -> converges to G*."""
    g = golden("geometry")
    d = syn.make_inputs(2, 64, 96, seed=7)
    G = torch.eye(4).repeat(2, 1, 1, 1)
    tgt = T(g["rec_target"])[:, 0]
    errs = []
    for k in range(4):
        G, _ = orc.lm_step(tgt, torch.ones(2, 64, 96), g["depth"], d["K"], G, 1)
        assert maxdiff(G, g["rec_G"][k]) < 1e-5
        errs.append(maxdiff(G, g["rec_Gstar"]))
    assert errs[-1] < 1e-5 and errs[0] > errs[1] > errs[2]


def test_solve_and_exp(golden):
    g = golden("geometry")
    # cholesky.solve has no damping: undo it by passing zero lambdas
    x = orc.lm_solve(g["solve_H"], g["solve_b"], ep_lmbda=0.0, lm_lmbda=0.0)
    assert maxdiff(x[:, None], g["solve_x"]) < 1e-6
    assert np.all(np.abs(x) <= 1.0) and np.any(np.abs(x) == 1.0)      # clamp is exercised
    assert maxdiff(orc.se3_exp(g["exp_xi"]), g["exp_G"]) < 1e-6
    G0 = T(g["G"])[:1, 0].repeat(len(g["exp_xi"]), 1, 1)
    assert maxdiff(orc.se3_increment(G0, g["exp_xi"]), g["inc_G"]) < 1e-6
    # non-SPD -> NaN -> 0 (geometry/cholesky.py:43-44)
    bad = -np.eye(6)[None]
    assert np.all(orc.lm_solve(bad, np.ones((1, 6)), 0.0, 0.0) == 0)


def test_fast_variants_equal_explicit_restatement():
    """The library-call variants used ONLY by bench.py's cpu_baseline leg equal the explicit restatements."""
    B, C, h, w = 2, 64, 16, 24
    f1, f2 = syn.normal("fmap1", (B, C, h, w), 9), syn.normal("fmap2", (B, C, h, w), 9)
    pyr = orc.corr_pyramid(f1, f2)
    c = orc.coords_grid_lowres(B, h, w) + T(syn.uniform("c", (B, 2, h, w), 9, -20.0, 20.0))
    assert maxdiff(orc.corr_lookup_fast(pyr, c), orc.corr_lookup(pyr, c)) < 1e-5
    flow, mask = syn.normal("f", (B, 2, h, w), 9, std=3.0), syn.normal("m", (B, 576, h, w), 9, std=2.0)
    assert maxdiff(orc.convex_upsample_fast(flow, mask), orc.convex_upsample(flow, mask)) < 1e-5
    d = syn.make_inputs(2, 64, 96, seed=9)
    tgt = torch.stack(orc._pix_grid(64, 96), -1)[None] + T(syn.normal("t", (2, 64, 96, 2), 9, std=5.0))
    a = orc.corr_weight_fast(d["g1"], d["g2"], tgt, d["depth"], d["sigma"])
    assert maxdiff(a, orc.corr_weight(d["g1"], d["g2"], tgt, d["depth"], d["sigma"])) < 1e-5


def test_encoder(golden):
    g = golden("encoder")
    W = syn.make_module_weights(orc.encoder_shapes(), seed=2)
    assert sorted(W) == [k for k in g["keys"]]
    img1 = syn.uniform("img_render", (2, 3, 64, 96), 2)
    img2 = syn.uniform("img_target", (2, 3, 64, 96), 2)
    f1, f2 = orc.image_encoder(W, img1, img2)
    assert maxdiff(f1, g["fmap1"]) < 1e-4 and maxdiff(f2, g["fmap2"]) < 1e-4


@pytest.mark.parametrize("name,shape,outer,inner,opt", [("loop_128", (2, 128, 128, 21), 1, 3, 1),
                                                         ("loop_2x2", (2, 128, 160, 22), 2, 2, 2),
                                                         ("loop_S1", (1, 240, 240, 23), 1, 3, 1)])
def test_loop(golden, name, shape, outer, inner, opt):
    g = golden(name)
    B, H, W, seed = shape
    d = syn.make_inputs(B, H, W, seed=seed)
    res = orc.refine(d, {"upd": upd_weights()}, outer=outer, inner=inner, optim_iters=opt, capture=True)
    Gi = torch.stack([t["Tij"] for t in res["trace"]])
    assert maxdiff(Gi, g["G_iters"]) < 1e-5
    assert maxdiff(res["G"], g["G_final"]) < 1e-5
    fl = res["flow_up"]
    if name == "loop_128":
        assert maxdiff(fl, g["flow_last"]) < 1e-4
        assert maxdiff(res["trace"][0]["flow_up"], g["flow_first"]) < 1e-4
        assert maxdiff(res["weight"], g["w_last"]) < 1e-4
    elif name == "loop_2x2":
        assert maxdiff(fl[:, :, ::2, ::2], g["flow_last"]) < 1e-4
    else:
        assert maxdiff(fl[:, :, ::3, ::3], g["flow_last"]) < 1e-4
        assert maxdiff(res["weight"][:, ::3, ::3], g["w_last"]) < 1e-4


LOOP480 = dict(B=2, H=480, W=640, seed=51, inner=2)


def loop480_inputs(device="cpu"):
    """The inputs tests/golden/gen_golden.py g_loop480 fed the reference: closed-form, bit-reproducible on any device."""
    c = LOOP480
    return syn.make_inputs_t(c["B"], c["H"], c["W"], seed=c["seed"], device=device, with_images=True)


@pytest.mark.parametrize("fixture", ["loop_480", "loop_480_g25", "loop_480_g50"])
def test_loop_480_oracle_pinned_at_the_headline_resolution(golden, fixture):
    """the oracle against the REFERENCE ITSELF at 480 x 640 with the encoder in the loop (B = 2, 1 outer x 2
    inner iterations, reference BasicEncoder + GRU_CFUpdator + reprojction_optim, its literal legacy start pose Ti * Ti.inv()):
    N = 4800 correlation columns, K = 2304-term convolutions -- where fp32 summation order could let a restatement drift unseen.
    Pose 1e-5; first correspondence field 1e-4, reported for the literal legacy start pose AND for the exact identity the oracle
     and the product use by default."""
    import os
    g = golden(fixture)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    dt = loop480_inputs()
    d = {k: v.numpy() for k, v in dt.items()}
    d.pop("fmap1"), d.pop("fmap2")
    W = {"upd": upd_weights(), "enc": syn.make_module_weights(orc.encoder_shapes(), seed=3, gain=float(g["enc_gain"]))}
    f1, f2 = orc.image_encoder(W["enc"], d["img_render"], d["img_target"])
    fmax = float(g["max_abs_fmap"].max())
    assert maxdiff(f1[:, ::16, ::5, ::5], g["fmap1_sub"]) < 1e-5 * max(fmax, 1.0) and maxdiff(f2[:, ::16, ::5, ::5], g["fmap2_sub"]) < 1e-5 * max(fmax, 1.0)
    dist = {}
    for lit in (True, False):
        res = orc.refine(d, W, outer=1, inner=LOOP480["inner"], optim_iters=1, capture=True, fast=True, literal_legacy_pose=lit)
        Gi = torch.stack([t["Tij"] for t in res["trace"]])
        dist[lit] = dict(pose=max(maxdiff(Gi, g["G_iters"]), maxdiff(res["G"], g["G_final"])),
                         flow_first=maxdiff(res["trace"][0]["flow_up"][:, :, ::8, ::8], g["flow_first"]),
                         w_first=maxdiff(res["trace"][0]["weight"][:, ::8, ::8], g["w_first"]),
                         flow_last=maxdiff(res["flow_up"][:, :, ::8, ::8], g["flow_last"]))
    print(fixture, "oracle vs reference (max |feature map| %.1f, max |flow| %.2f):" % (fmax, float(g["max_abs_flow"])),
          "literal Ti*Ti^-1:", dist[True], " exact identity:", dist[False])
    # the LITERAL restatement is the pin: pose 1e-5, first field 1e-4 (measured: 1.1e-8, 3.7e-5)
    assert dist[True]["pose"] < 1e-5 and dist[True]["flow_first"] < 1e-4 and dist[True]["w_first"] < 1e-4, dist[True]
    assert dist[True]["flow_last"] < 5e-4, dist[True]                   # second iteration: the pose has been fed back
    # exact identity instead of the product (encoder gain 1): the pose still agrees to 1e-7, the first field moves by 1.6e-4 px -- the ~1e-7 rounding
    # noise of the reference's OWN Ti * Ti.inv() shifts the first lookup off the integer grid, and un-normalised correlation
    # features (|f| ~ 31, |corr| ~ 900) turn that into 1e-4-level field differences at this resolution.  This
    # leg is a sensitivity record with a documented bound, not the parity gate.
    assert dist[False]["pose"] < 1e-5 and dist[False]["w_first"] < 1e-4 and dist[False]["flow_first"] < 5e-4 and dist[False]["flow_last"] < 1e-3, dist[False]


LOOP960 = dict(B=1, H=960, W=1280, seed=61, inner=2)


def loop960_inputs(device="cpu"):
    """The inputs tests/golden/gen_golden.py g_loop960 fed the reference."""
    c = LOOP960
    return syn.make_inputs_t(c["B"], c["H"], c["W"], seed=c["seed"], device=device, with_images=True)


def test_loop_960_oracle_pinned_at_config5_image_size(golden):
    """BASELINE config 5's per-GPU image size (960 x 1280: 120 x 160 feature maps, N = 19 200 correlation
    columns, a 1.47-GB volume) against the REFERENCE ITSELF -- its BasicEncoder (kaiming gain 0.25) + GRU_CFUpdator +
    reprojction_optim on one image, legacy start pose.  Until r05 the oracle was pinned up to 480 x 640 only, while
    test_full_shape_short_horizon_vs_oracle[1-960-1280] used it at this size.  First iteration only here (the CPU suite's time budget);
    the GPU test compares both iterations with the fixture directly."""
    import os
    g = golden("loop_960")
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    dt = loop960_inputs()
    d = {k: v.numpy() for k, v in dt.items()}
    d.pop("fmap1"), d.pop("fmap2")
    W = {"upd": upd_weights(), "enc": syn.make_module_weights(orc.encoder_shapes(), seed=3, gain=float(g["enc_gain"]))}
    f1, f2 = orc.image_encoder(W["enc"], d["img_render"], d["img_target"])
    fmax = float(g["max_abs_fmap"].max())
    assert maxdiff(f1[:, ::16, ::10, ::10], g["fmap1_sub"]) < 1e-5 * max(fmax, 1.0) and maxdiff(f2[:, ::16, ::10, ::10], g["fmap2_sub"]) < 1e-5 * max(fmax, 1.0)
    d["fmap1"], d["fmap2"] = f1.numpy(), f2.numpy()              # (the encoder is checked above: the loop takes its maps as given)
    d.pop("img_render"), d.pop("img_target")
    res = orc.refine(d, {"upd": W["upd"]}, outer=1, inner=1, optim_iters=1, capture=True, fast=True, literal_legacy_pose=True)
    dist = dict(pose=maxdiff(res["trace"][0]["Tij"], g["G_iters"][0]),
                flow_first=maxdiff(res["trace"][0]["flow_up"][:, :, ::16, ::16], g["flow_first"]),
                w_first=maxdiff(res["trace"][0]["weight"][:, ::16, ::16], g["w_first"]))
    print("loop_960 oracle vs reference (max |feature map| %.1f, max |flow| %.2f):" % (fmax, float(g["max_abs_flow"])), dist)
    assert dist["pose"] < 1e-5 and dist["flow_first"] < 1e-4 and dist["w_first"] < 1e-4, dist


LOOPS1ENC = dict(B=1, H=240, W=240, seed=71, outer=3, inner=4)


def loopS1enc_inputs(device="cpu"):
    """The inputs tests/golden/gen_golden.py g_loopS1enc fed the reference."""
    c = LOOPS1ENC
    return syn.make_inputs_t(c["B"], c["H"], c["W"], seed=c["seed"], device=device, with_images=True)


def test_loop_S1_shipped_schedule_oracle_pinned(golden):
    """BASELINE configs[0]'s shape at the SHIPPED schedule (one 240 x 240 crop, 3 outer x 4 inner iterations,
    config/linemod/template_fw0.5.yml:76-81) against the REFERENCE ITSELF with its encoder in the loop: pins the multi-outer
    accumulation Ti <- Tij * Ti with the literal start Tij = Ti * Ti.inv() of every outer iteration (model/PoseRefiner.py:241-244,365)
    end to end -- all 12 relative poses and the final pose at 1e-5, the first field of the first outer iteration at 1e-4 (identical
    inputs), the later fields (free-running: poses fed back) at the drift bound 5e-4."""
    import os
    g = golden("loop_S1_enc")
    c = LOOPS1ENC
    assert tuple(g["schedule"]) == (c["outer"], c["inner"])
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    d = {k: v.numpy() for k, v in loopS1enc_inputs().items()}
    d.pop("fmap1"), d.pop("fmap2")
    W = {"upd": upd_weights(), "enc": syn.make_module_weights(orc.encoder_shapes(), seed=3, gain=float(g["enc_gain"]))}
    f1, f2 = orc.image_encoder(W["enc"], d["img_render"], d["img_target"])
    fmax = float(g["max_abs_fmap"].max())
    assert maxdiff(f1[:, ::16, ::3, ::3], g["fmap1_sub"]) < 1e-5 * max(fmax, 1.0) and maxdiff(f2[:, ::16, ::3, ::3], g["fmap2_sub"]) < 1e-5 * max(fmax, 1.0)
    res = orc.refine(d, W, outer=c["outer"], inner=c["inner"], optim_iters=1, capture=True, fast=True)        # (default: the literal start)
    Gi = torch.stack([t["Tij"] for t in res["trace"]])
    dist = dict(pose=max(maxdiff(Gi, g["G_iters"]), maxdiff(res["G"], g["G_final"])),
                flow_outer_first=[maxdiff(res["trace"][o * c["inner"]]["flow_up"][:, :, ::4, ::4], g["flow_outer_first"][o]) for o in range(c["outer"])],
                flow_last=maxdiff(res["flow_up"][:, :, ::4, ::4], g["flow_last"]), w_last=maxdiff(res["weight"][:, ::4, ::4], g["w_last"]))
    print("loop_S1_enc oracle vs reference (max |feature map| %.1f, max |flow| %.2f):" % (fmax, float(g["max_abs_flow"])), dist)
    assert dist["pose"] < 1e-5 and dist["flow_outer_first"][0] < 1e-4, dist
    assert max(dist["flow_outer_first"]) < 5e-4 and dist["flow_last"] < 5e-4 and dist["w_last"] < 5e-4, dist


# ---- row f2: evaluator arithmetic pinned to the reference's own utils/eval_metric.py (tests/golden/gen_golden_eval.py) ----
EVAL_SETS = [("cat_", False), ("driller_", False), ("sym_eggbox_", True)]


def check_eval_against_reference(m, g, pre, sym):
    """m (n,5) [ADD, ADD-S, proj2d px, trans cm, rot deg] vs the distances the reference computed.
    The reference does the model transform, the rotation product and np.mean in fp32 (float32 model / poses), so its own
    values carry ~1e-6 relative noise; the angle comes from arccos((trace-1)/2) of an fp32 trace, whose resolution near
    0 is sqrt(eps32) ~ 0.03 deg: that is the absolute floor of the rotation comparison (thresholds sit at 5 deg)."""
    d = m[:, 1] if sym else m[:, 0]
    assert np.allclose(d, g[pre + "add"], rtol=2e-5, atol=1e-9)
    assert np.allclose(m[:, 2], g[pre + "proj2d"], rtol=1e-4, atol=1e-5)
    assert np.allclose(m[:, 3], g[pre + "trans_cm"], rtol=1e-6, atol=1e-5)
    assert np.allclose(m[:, 4], g[pre + "rot_deg"], rtol=1e-4, atol=0.05)


def eval_flags(m, diameter, sym):
    d = m[:, 1] if sym else m[:, 0]
    return np.stack([d < 0.1 * diameter, d < 0.02 * diameter, d < 0.05 * diameter, m[:, 2] < 5.0,
                     (m[:, 3] < 5.0) & (m[:, 4] < 5.0)], 1)


EVAL_DIAMETERS = {"cat_": 15.2633 / 100, "driller_": 25.9425 / 100, "sym_eggbox_": 17.6364 / 100}   # linemod_config.py:2-19


@pytest.mark.parametrize("pre,sym", EVAL_SETS)
def test_eval_oracle_matches_reference_eval_metric(golden, pre, sym):
    from oracle import eval_oracle as eo
    g = golden("eval_metric")
    m = eo.pose_metrics(g[pre + "model"], g[pre + "pred"], g[pre + "gt"], g["linemod_K"], sym)
    check_eval_against_reference(m, g, pre, sym)
    flags = eval_flags(m, EVAL_DIAMETERS[pre], sym)
    assert np.array_equal(flags, g[pre + "flags"])               # every threshold decision of the reference reproduced
    assert 0 < flags.sum() < flags.size                          # ... and the set crosses the thresholds
    if not sym:
        s = g[pre + "summary"]                                   # LineMODEvaluator.summarize(): proj2d, add, add2, add5, cmd5, n
        assert np.allclose(flags.mean(0)[[3, 0, 1, 2, 4]], s[:5]) and s[5] == len(flags)
