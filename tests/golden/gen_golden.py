#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python modules (build container only).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

/root/reference is imported read-only with the two shims of SURVEY.md Appendix B (yacs.CfgNode and
an attr-dict standing in for EasyDict).  Inputs come from rnnpose_amd.synthetic (closed-form hash,
bit-reproducible anywhere), so the fixtures hold reference OUTPUTS (plus the few tiny inputs that are
not formula-generated).  The reference never travels to the GPU box; these .npz files do.

Harness-composed pieces (PoseRefiner.forward itself cannot be imported: cv2/easydict/pytorch3d) quote
the reference lines they reproduce.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path[:0] = [ROOT, REF, os.path.join(REF, "thirdparty")]
warnings.filterwarnings("ignore")


class AttrDict(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


yacs = types.ModuleType("yacs")
yacs.config = types.ModuleType("yacs.config")
yacs.config.CfgNode = AttrDict
sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yacs.config

import geometry.transformation as tr  # noqa: E402
import geometry.cholesky as chol  # noqa: E402
import geometry.se3 as se3  # noqa: E402
from geometry.projective_ops import coords_grid, normalize_coords_grid  # noqa: E402
import model.CFNet as cf  # noqa: E402
from thirdparty.raft.corr import CorrBlock  # noqa: E402
from thirdparty.raft.extractor import BasicEncoder  # noqa: E402

from rnnpose_amd import synthetic as syn  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)
EPS = 1e-5  # model/PoseRefiner.py:21


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  {name}.npz  {os.path.getsize(path)/1024:.0f} KiB")


def make_cfnet(seed=0):
    net = cf.GRU_CFUpdator(AttrDict(pretrained_model=None, mixed_precision=True, fea_net="default")).eval()
    shapes = {k: tuple(v.shape) for k, v in net.update_block.state_dict().items()}
    W = syn.make_module_weights(shapes, seed=seed)
    net.update_block.load_state_dict({k: T(v) for k, v in W.items()}, strict=True)
    return net


# ------------------------------------------------------------------------------------------------
def g_corr():
    """G1/G2: CorrBlock build + lookup (thirdparty/raft/corr.py:13-57)."""
    B, C, h, w = 2, 256, 16, 24
    f1 = syn.normal("fmap1", (B, C, h, w), 11)
    f2 = syn.normal("fmap2", (B, C, h, w), 11)
    blk = CorrBlock(T(f1), T(f2), num_levels=4, radius=4)
    pyr = blk.corr_pyramid
    grid = cf.coords_grid(B, h, w)
    cases = {
        "int": grid.clone(),
        "sub": grid + T(syn.uniform("lk_sub", (B, 2, h, w), 11, -3.0, 3.0)),
        "oob": grid + T(syn.uniform("lk_oob", (B, 2, h, w), 11, -30.0, 30.0)),
    }
    out = {f"lookup_{k}": blk(v)[:, :, ::2, ::3] for k, v in cases.items()}   # pixel subset keeps the fixture small
    save("corr", level0_rows=pyr[0].reshape(B * h * w, h * w)[::5], level1=pyr[1], level2=pyr[2], level3=pyr[3],
         level0_sum=pyr[0].double().sum(), level0_abs=pyr[0].double().abs().sum(), **out)


def g_update():
    """G3: BasicUpdateBlock.forward one step (thirdparty/raft/update.py:178-188)."""
    B, h, w = 1, 16, 20
    net = make_cfnet()
    hid = np.tanh(syn.normal("u_net", (B, 128, h, w), 3))
    inp = np.maximum(syn.normal("u_inp", (B, 128, h, w), 3), 0)
    corr = syn.normal("u_corr", (B, 324, h, w), 3)
    flow = syn.normal("u_flow", (B, 2, h, w), 3, std=2.0)
    n2, mask, df = net.update_block(T(hid), T(inp), T(corr), T(flow))
    save("update_block", net=n2, mask=mask, dflow=df)


def g_upsample_ctx():
    """G4: upsample_flow (model/CFNet.py:95-106); ctx prep (:124-133); flow_init downsample (:136-144)."""
    net = make_cfnet()
    B, h, w = 2, 16, 12
    flow = syn.normal("up_flow", (B, 2, h, w), 5, std=3.0)
    mask = syn.normal("up_mask", (B, 576, h, w), 5, std=2.0)
    up = net.upsample_flow(T(flow), T(mask))
    ctx = syn.normal("ctx", (1, 256, 64, 96), 5, std=0.1)
    cnet = F.interpolate(T(ctx), scale_factor=1 / 8, mode="bilinear", align_corners=True)
    hid, inp = torch.split(cnet, [128, 128], dim=1)
    finit = syn.normal("finit", (2, 2, 64, 96), 5, std=4.0)
    fi = T(finit.copy())
    ds = 8
    fi /= ds
    fi = F.interpolate(fi, scale_factor=1 / ds, mode="bilinear", align_corners=True)
    coords1 = cf.coords_grid(2, 8, 12) + fi
    save("upsample_ctx", flow_up=up, net=torch.tanh(hid), inp=torch.relu(inp), coords1=coords1)


def synth_pose_case(B, H, W, seed, sigma=0.02):
    d = syn.make_inputs(B, H, W, seed=seed, pose_sigma=sigma)
    return d


def g_geometry():
    """G5: SE3Sequence.transform + PoseRefiner.py:324-328; G6: weight (:342-345); G7: reprojction_optim;
    G8: se3 exp."""
    B, H, W = 2, 64, 96
    d = synth_pose_case(B, H, W, 7, sigma=0.03)
    depth = T(d["depth"])
    # straddle the 0.1 validity threshold and the 0.02 Jacobian cut-off
    depth[:, :, 20:24] = 0.05
    depth[:, :, 24:26] = 0.012
    K = T(d["K"])
    G = T(d["G0"])
    Tij = tr.SE3Sequence(matrix=G.clone())
    depths = depth + EPS
    reproj, vmask = Tij.transform(depths, K, valid_mask=True)
    grids = coords_grid(depths)
    # PoseRefiner.py:327 with the per-sample mask fix for B>1
    flow_init = torch.einsum("...ijk->...kij", reproj - grids[..., :2]) * (depths > EPS)[:, :, None]
    # ---- weight ----
    target = grids[..., :2] + T(syn.normal("tgt_noise", (B, 1, H, W, 2), 7, std=3.0))
    g1, g2, sigma = T(d["g1"]), T(d["g2"]), torch.tensor([0.7])
    warped = F.grid_sample(g2, normalize_coords_grid(target).squeeze(1))                     # :343
    cw = torch.sum(g1 * warped, dim=1, keepdim=True).permute(0, 2, 3, 1)[:, None]           # :344
    cw = torch.exp(-torch.abs(1 - cw) / sigma[0]) * (depth > 0)[..., None].float()          # :345
    # ---- LM ----
    captured = {}
    orig = tr.cholesky_solve

    def spy(Hm, b):
        captured.setdefault("H", []).append(Hm.clone())
        captured.setdefault("b", []).append(b.clone())
        return orig(Hm, b)

    tr.cholesky_solve = spy
    lm = {}
    wpat = {
        "desc": cw,
        "ones": torch.ones(B, 1, H, W, 1),
        "sparse": (T(syn.uniform("wsp", (B, 1, H, W, 1), 7)) > 0.97).float() * 2.5,
        "zero": torch.zeros(B, 1, H, W, 1),
    }
    for nm, wgt in wpat.items():
        captured.clear()
        T0 = tr.SE3Sequence(matrix=G.clone())
        T1 = T0.reprojction_optim(target, wgt, depths, K, num_iters=2)
        lm[f"lm_{nm}_Hd0"] = captured["H"][0]      # damped H of iteration 0
        lm[f"lm_{nm}_b0"] = captured["b"][0]
        lm[f"lm_{nm}_Hd1"] = captured["H"][1]
        lm[f"lm_{nm}_b1"] = captured["b"][1]
        lm[f"lm_{nm}_G"] = T1.G
        Ts = tr.SE3Sequence(matrix=G.clone()).reprojction_optim(target, wgt, depths, K, num_iters=1)
        lm[f"lm_{nm}_G1"] = Ts.G
    tr.cholesky_solve = orig
    # exact-target recovery KAT: targets = projection under G*, unit weights, start at identity
    Gstar = T(syn.se3_exp_np(np.array([[0.01, -0.02, 0.015, 0.02, -0.01, 0.03]] * B)).astype(np.float32)).reshape(B, 1, 4, 4)
    tstar = tr.SE3Sequence(matrix=Gstar).transform(depths, K)
    Tk = tr.SE3Sequence(matrix=torch.eye(4).repeat(B, 1, 1, 1))
    rec = []
    for _ in range(4):
        Tk = Tk.reprojction_optim(tstar, torch.ones(B, 1, H, W, 1), depths, K, num_iters=1)
        rec.append(Tk.G.clone())
    # ---- solve + exp ----
    rng = np.random.RandomState(0)
    A = rng.randn(5, 6, 6)
    Hs = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(6)
    bs = rng.randn(5, 6) * np.array([0.1, 1, 5, 20, 0.01])[:, None]
    xs = chol.solve(T(Hs)[:, None], T(bs)[:, None])
    thetas = np.array([0.0, 5e-5, 0.99e-4, 1.01e-4, 1e-3, 0.5, np.pi - 1e-3, 2.0])
    axis = np.array([0.3, -0.5, 0.81])
    axis = axis / np.linalg.norm(axis)
    xi = np.concatenate([np.tile([[0.2, -0.1, 0.4]], (len(thetas), 1)), thetas[:, None] * axis[None]], 1).astype(np.float32)
    Gexp = se3._se3_matrix_expm(T(xi))
    Ginc = se3.se3_matrix_increment(G[:, 0][:1].repeat(len(thetas), 1, 1), T(xi))
    Ginv = se3.se3_matrix_inverse(G)
    save("geometry", depth=depth, G=G, flow_init=flow_init, vmask=vmask, reproj=reproj, target=target, sigma=sigma,
         weight=cw, solve_H=Hs, solve_b=bs, solve_x=xs, exp_xi=xi, exp_G=Gexp, inc_G=Ginc, inv_G=Ginv,
         rec_Gstar=Gstar, rec_target=tstar, rec_G=torch.stack(rec), **lm)


def g_encoder():
    """ImageFeaEncoder.forward (model/CFNet.py:41-49) over BasicEncoder(instance norm)."""
    enc = object.__new__(cf.ImageFeaEncoder)
    torch.nn.Module.__init__(enc)
    enc.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=False, input_dim=3)
    enc.eval()
    shapes = {k: tuple(v.shape) for k, v in enc.fnet.state_dict().items()}
    W = syn.make_module_weights(shapes, seed=2)
    enc.fnet.load_state_dict({k: T(v) for k, v in W.items()}, strict=True)
    img1 = syn.uniform("img_render", (2, 3, 64, 96), 2)
    img2 = syn.uniform("img_target", (2, 3, 64, 96), 2)
    f1, f2 = enc(T(img1), T(img2))
    save("encoder", fmap1=f1.float(), fmap2=f2.float(), keys=np.array(sorted(shapes)))


def run_loop(d, net, outer, inner, optim_iters, sigma):
    """Harness-composed PoseRefiner inner/outer loop (model/PoseRefiner.py:239-365) on static synthetic
    renderings, with the per-sample fixes of SURVEY.md Appendix B for B>1 (:328 and :336)."""
    depth, K = T(d["depth"]), T(d["K"])
    f1, f2, ctx, g1, g2 = (T(d[k]) for k in ("fmap1", "fmap2", "ctx", "g1", "g2"))
    B = depth.shape[0]
    Ti = tr.SE3Sequence(matrix=T(d["G0"]).clone())
    Tij = Ti.copy().identity()
    per_iter = []
    first_flow = None
    for ren in range(outer):
        Ti = Tij * Ti
        Tij.identity_()
        Tij = Ti * Ti.inv()                                             # legacy branch :243-244
        depths = depth + EPS
        for i in range(inner):
            Tij = Tij.copy(stop_gradients=True)
            reproj, vmask = Tij.transform(depths, K, valid_mask=True)
            grids = coords_grid(depths)
            flow_init = (reproj - grids[..., :2]).permute(0, 1, 4, 2, 3) * (depths > EPS)[:, :, None]
            flow = net(f1, f2, flow_init=flow_init.squeeze(1), context_fea=ctx, update_corr_fn=i == 0)
            if first_flow is None:
                first_flow = flow[-1].clone()
            target = flow[-1].permute(0, 2, 3, 1)[:, None] + grids[..., :2]
            warped = F.grid_sample(g2, normalize_coords_grid(target).squeeze(1))
            cw = torch.sum(g1 * warped, dim=1, keepdim=True).permute(0, 2, 3, 1)[:, None]
            cw = torch.exp(-torch.abs(1 - cw) / sigma) * (depth > 0)[..., None].float()
            Tij = Tij.reprojction_optim(target, cw, depths, K, num_iters=optim_iters)
            per_iter.append(dict(G=Tij.G.clone(), flow=flow[-1].clone(), w=cw.clone()))
    Ti = Tij * Ti
    return Ti.G, per_iter, first_flow


def g_loop():
    """G9: end-to-end loops (short horizons)."""
    net = make_cfnet()
    d = synth_pose_case(2, 128, 128, 21)
    Gf, it, _ = run_loop(d, net, outer=1, inner=3, optim_iters=1, sigma=1.0)
    save("loop_128", G_final=Gf, G_iters=torch.stack([x["G"] for x in it]), flow_last=it[-1]["flow"],
         flow_first=it[0]["flow"], w_last=it[-1]["w"][:, 0, ..., 0])
    d = synth_pose_case(2, 128, 160, 22)
    Gf, it, _ = run_loop(d, net, outer=2, inner=2, optim_iters=2, sigma=1.0)
    save("loop_2x2", G_final=Gf, G_iters=torch.stack([x["G"] for x in it]), flow_last=it[-1]["flow"][:, :, ::2, ::2])
    d = synth_pose_case(1, 240, 240, 23)
    Gf, it, _ = run_loop(d, net, outer=1, inner=3, optim_iters=1, sigma=1.0)
    save("loop_S1", G_final=Gf, G_iters=torch.stack([x["G"] for x in it]), flow_last=it[-1]["flow"][:, :, ::3, ::3],
         w_last=it[-1]["w"][:, 0, ::3, ::3, 0])


def g_loop480():
    """The HEADLINE resolution against the reference itself: B = 2 images of 480 x 640, the reference's own
    ImageFeaEncoder (model/CFNet.py:26-49 over thirdparty/raft/extractor.py:118-232) in the loop, GRU_CFUpdator, reprojction_optim,
    1 outer x 2 inner iterations, its legacy start pose Ti * Ti.inv(), hash weights (encoder seed 3, update block seed 0).
    Poses in full; the fields sub-sampled ::8 (reading every pixel of the 1/8-resolution grid once).
    Two fixtures: encoder weights at kaiming gain 1 (feature maps of magnitude ~31, correlation values ~900: the harshest fp32
    conditioning this path sees -- the reference's CPU result is itself ~2e-4 px away from its own fp64 evaluation there) and at
    gain 0.25 (|f| ~ 8, |corr| ~ 60); r05: a third one at gain 0.5, between the two (where does the literal 1e-4 stop holding?)."""
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    net = make_cfnet()
    dt = syn.make_inputs_t(2, 480, 640, seed=51, device="cpu", with_images=True)
    for name, gain in (("loop_480", 1.0), ("loop_480_g25", 0.25), ("loop_480_g50", 0.5)):
        enc = object.__new__(cf.ImageFeaEncoder)
        torch.nn.Module.__init__(enc)
        enc.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=False, input_dim=3)
        enc.eval()
        shapes = {k: tuple(v.shape) for k, v in enc.fnet.state_dict().items()}
        enc.fnet.load_state_dict({k: T(v) for k, v in syn.make_module_weights(shapes, seed=3, gain=gain).items()}, strict=True)
        d = {k: v.numpy() for k, v in dt.items()}
        f1, f2 = enc(dt["img_render"], dt["img_target"])              # (PoseRefiner.py:311: image_fea_enc(syn_img, image_crop))
        d["fmap1"], d["fmap2"] = f1.float().numpy(), f2.float().numpy()
        Gf, it, first = run_loop(d, net, outer=1, inner=2, optim_iters=1, sigma=1.0)
        save(name, G_final=Gf, G_iters=torch.stack([x["G"] for x in it]), flow_first=it[0]["flow"][:, :, ::8, ::8],
             flow_last=it[-1]["flow"][:, :, ::8, ::8], w_first=it[0]["w"][:, 0, ::8, ::8, 0], w_last=it[-1]["w"][:, 0, ::8, ::8, 0],
             fmap1_sub=f1.float()[:, ::16, ::5, ::5], fmap2_sub=f2.float()[:, ::16, ::5, ::5], enc_gain=np.array(gain),
             max_abs_fmap=np.array([float(f1.abs().max()), float(f2.abs().max())]), max_abs_flow=np.array(float(it[0]["flow"].abs().max())))


def g_loop960():
    """BASELINE config 5's per-GPU image size against the reference itself: ONE image of 960 x 1280 (N = 19 200
    correlation columns, a 1.47-GB volume), the reference's BasicEncoder at kaiming gain 0.25 + GRU_CFUpdator + reprojction_optim, 1 outer
    x 2 inner iterations, legacy start pose.  Fields sub-sampled ::16."""
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    net = make_cfnet()
    dt = syn.make_inputs_t(1, 960, 1280, seed=61, device="cpu", with_images=True)
    gain = 0.25
    enc = object.__new__(cf.ImageFeaEncoder)
    torch.nn.Module.__init__(enc)
    enc.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=False, input_dim=3)
    enc.eval()
    shapes = {k: tuple(v.shape) for k, v in enc.fnet.state_dict().items()}
    enc.fnet.load_state_dict({k: T(v) for k, v in syn.make_module_weights(shapes, seed=3, gain=gain).items()}, strict=True)
    d = {k: v.numpy() for k, v in dt.items()}
    f1, f2 = enc(dt["img_render"], dt["img_target"])
    d["fmap1"], d["fmap2"] = f1.float().numpy(), f2.float().numpy()
    Gf, it, first = run_loop(d, net, outer=1, inner=2, optim_iters=1, sigma=1.0)
    save("loop_960", G_final=Gf, G_iters=torch.stack([x["G"] for x in it]), flow_first=it[0]["flow"][:, :, ::16, ::16],
         flow_last=it[-1]["flow"][:, :, ::16, ::16], w_first=it[0]["w"][:, 0, ::16, ::16, 0], w_last=it[-1]["w"][:, 0, ::16, ::16, 0],
         fmap1_sub=f1.float()[:, ::16, ::10, ::10], fmap2_sub=f2.float()[:, ::16, ::10, ::10], enc_gain=np.array(gain),
         max_abs_fmap=np.array([float(f1.abs().max()), float(f2.abs().max())]), max_abs_flow=np.array(float(it[0]["flow"].abs().max())))


def g_loopS1enc():
    """BASELINE configs[0]'s shape at the SHIPPED schedule against the reference itself: ONE 240 x 240 crop
    (30 x 30 feature maps), the reference's BasicEncoder (kaiming gain 0.5) + GRU_CFUpdator + reprojction_optim, RENDER_ITER_COUNT = 3
    outer x ITER_COUNT = 4 inner iterations (config/linemod/template_fw0.5.yml:76-81), every outer iteration started from the legacy
    product Ti * Ti.inv() and accumulated Ti <- Tij * Ti (model/PoseRefiner.py:241-244,365).  All 12 relative poses, the final pose,
    the first field of every outer iteration and the last field / weight (::4)."""
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    net = make_cfnet()
    dt = syn.make_inputs_t(1, 240, 240, seed=71, device="cpu", with_images=True)
    gain = 0.5
    enc = object.__new__(cf.ImageFeaEncoder)
    torch.nn.Module.__init__(enc)
    enc.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=False, input_dim=3)
    enc.eval()
    shapes = {k: tuple(v.shape) for k, v in enc.fnet.state_dict().items()}
    enc.fnet.load_state_dict({k: T(v) for k, v in syn.make_module_weights(shapes, seed=3, gain=gain).items()}, strict=True)
    d = {k: v.numpy() for k, v in dt.items()}
    f1, f2 = enc(dt["img_render"], dt["img_target"])
    d["fmap1"], d["fmap2"] = f1.float().numpy(), f2.float().numpy()
    outer, inner = 3, 4
    Gf, it, first = run_loop(d, net, outer=outer, inner=inner, optim_iters=1, sigma=1.0)
    save("loop_S1_enc", G_final=Gf, G_iters=torch.stack([x["G"] for x in it]),
         flow_outer_first=torch.stack([it[o * inner]["flow"][:, :, ::4, ::4] for o in range(outer)]),
         flow_last=it[-1]["flow"][:, :, ::4, ::4], w_last=it[-1]["w"][:, 0, ::4, ::4, 0],
         fmap1_sub=f1.float()[:, ::16, ::3, ::3], fmap2_sub=f2.float()[:, ::16, ::3, ::3], enc_gain=np.array(gain),
         schedule=np.array([outer, inner]), max_abs_fmap=np.array([float(f1.abs().max()), float(f2.abs().max())]),
         max_abs_flow=np.array(float(it[0]["flow"].abs().max())))


if __name__ == "__main__":
    which = sys.argv[1:] or ["corr", "update", "upsample_ctx", "geometry", "encoder", "loop", "loop480", "loop960", "loopS1enc"]
    for nm in which:
        print("generating", nm)
        globals()["g_" + nm]()
