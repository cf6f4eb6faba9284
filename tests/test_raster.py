"""GPU tests of the mesh rasteriser (second half: csrc/raster.hip, rnnpose_amd/rasterizer.py).
PARITY UNPINNED against PyTorch3D; checked against oracle/raster_oracle.py and through properties that any correct
rasteriser satisfies: depth ordering, exact reproduction of attributes that are linear in the object coordinates
(perspective-correct interpolation), agreement of the coverage with the vertex splat, and the refiner running on it."""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from oracle import rnnpose_oracle as orc
from rnnpose_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from rnnpose_amd import build, ops as _ops
    build.build()
    return _ops


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def icosphere(sub=3, scale=(0.09, 0.06, 0.05)):
    """Closed triangle mesh of an ellipsoid: subdivided icosahedron (consistent outward winding)."""
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(sub):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * np.array(scale)).astype(np.float32), np.array(f, np.int32)


def scene(B, seed=1):
    verts, faces = icosphere()
    K = np.tile(np.array([[572.4114, 0, 80.0], [0, 573.57043, 64.0], [0, 0, 1]], np.float32), (B, 1, 1))
    G = syn.se3_exp_np(syn.normal("g", (B, 6), seed, std=0.4))
    G[:, :3, 3] = syn.uniform("t", (B, 3), seed, -0.02, 0.02) + np.array([0, 0, 0.75])
    return verts, faces, K, G.astype(np.float32)


def test_raster_matches_oracle(ops):
    from rnnpose_amd.rasterizer import MeshRenderer
    B, H, W = 2, 128, 160
    verts, faces, K, G = scene(B)
    cols = syn.uniform("col", (verts.shape[0], 3), 2)
    attr = syn.normal("attr", (B, verts.shape[0], 8), 3)
    ren = MeshRenderer({"obj": dict(verts=verts, faces=faces, colors=cols)}, shade=False)
    out, depth = ren(["obj"] * B, T(attr).cuda(), T=T(G).cuda(), K=T(K).cuda(), render_image_size=(H, W), render_tex=True)
    vdepth = ren.render_depth(["obj"] * B, T=T(G).cuda(), K=T(K).cuda(), render_image_size=(H, W))
    assert out.shape == (B, 3 + 8, H, W) and depth.shape == (B, 1, H, W) and vdepth.shape == (B, 1, H, W)
    for b in range(B):
        f, z, w, _ = ro.rasterize(verts, faces, G[b], K[b], H, W, perspective=True)
        hit_o, hit_g = f >= 0, depth[b, 0].cpu().numpy() > 0
        assert hit_o.mean() > 0.15                                             # the object fills a good part of the crop
        assert (hit_o != hit_g).mean() < 2e-3                                   # only edge pixels may flip (fp32 vs fp64 edge tests)
        both = hit_o & hit_g
        assert np.abs(depth[b, 0].cpu().numpy() - z)[both].max() < 2e-5
        assert np.all(depth[b, 0].cpu().numpy()[~hit_g] == -1.0)                # PoseRefiner.py:139 expects -1 for "no surface"
        want = ro.interpolate(f, w, faces, np.concatenate([cols, attr[b]], 1))
        err = np.abs(out[b].cpu().numpy() - want)[:, both]
        assert np.quantile(err, 0.999) < 1e-4                                   # (a face switch at an edge changes a few pixels)
        f2, _, w2, vz = ro.rasterize(verts, faces, G[b], K[b], H, W, perspective=True)      # (render_depth: perspective-correct weights)
        vd = vdepth[b, 0].cpu().numpy()
        both2 = (f2 >= 0) & (vd > 0)
        assert (np.abs(vd - vz)[both2] < 1e-5).mean() > 0.995                   # nearest-vertex ties at w_i == w_j may flip
        assert np.all(vd[~(vd > 0)] == 0.0)


def test_depth_ordering_and_linear_attributes(ops):
    """(1) Two parallel quads at different depths: the nearer one wins everywhere they overlap.  (2) An attribute that is
    linear in the OBJECT coordinates must be reproduced exactly at the surface point every pixel's ray hits (this is what
    perspective-correct interpolation means), and the z-buffer must equal that point's camera z."""
    from rnnpose_amd.rasterizer import MeshRenderer
    quad = lambda z, s: np.array([[-s, -s, z], [s, -s, z], [s, s, z], [-s, s, z]], np.float32)
    verts = np.concatenate([quad(0.0, 0.05), quad(0.1, 0.1)])                   # small near quad (z=0) in front of a big one
    faces = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]], np.int32)
    H, W = 96, 128
    K = np.array([[[500.0, 0, 64.0], [0, 500.0, 48.0], [0, 0, 1]]], np.float32)
    G = syn.se3_exp_np(np.array([[0, 0, 0, 0.3, -0.2, 0.1]]))
    G[:, :3, 3] = [0.005, -0.004, 0.6]
    G = G.astype(np.float32)
    A = np.array([[2.0, -1.0, 0.5], [0.3, 0.7, -1.1]], np.float32)              # attr = A @ X_obj + c
    cvec = np.array([0.2, -0.4], np.float32)
    attr = verts @ A.T + cvec
    ren = MeshRenderer({"q": dict(verts=verts, faces=faces, colors=None)})
    out, depth = ren(["q"], T(attr)[None].cuda(), T=T(G).cuda(), K=T(K).cuda(), render_image_size=(H, W), render_tex=False)
    z = depth[0, 0].cpu().numpy()
    hit = z > 0
    ys, xs = np.mgrid[0:H, 0:W]
    ray = np.stack([(xs + 0.5 - K[0, 0, 2]) / K[0, 0, 0], (ys + 0.5 - K[0, 1, 2]) / K[0, 1, 1], np.ones_like(xs, float)], -1)
    Xc = ray * z[..., None]
    Xo = (Xc - G[0, :3, 3]) @ G[0, :3, :3]                                       # R^T (Xc - t)
    want = Xo @ A.T + cvec
    got = out[0].cpu().numpy().transpose(1, 2, 0)
    assert hit.mean() > 0.2 and np.abs(got - want)[hit].max() < 2e-4
    # every hit pixel lies on one of the two planes (object z = 0 or 0.1) ...
    zo = Xo[..., 2]
    on_near, on_far = np.abs(zo) < 1e-4, np.abs(zo - 0.1) < 1e-4
    assert np.all((on_near | on_far)[hit])
    # ... and wherever the ray hits the near quad inside its extent, the near quad was chosen
    inside_near = on_near & (np.abs(Xo[..., 0]) < 0.049) & (np.abs(Xo[..., 1]) < 0.049)
    assert inside_near.sum() > 500
    # project the near quad analytically: pixels whose ray meets z_obj = 0 inside the quad must not show the far plane
    n = G[0, :3, 2]
    tt = (G[0, :3, 3] @ n) / (ray @ n)
    Xn = (ray * tt[..., None] - G[0, :3, 3]) @ G[0, :3, :3]
    must_near = (np.abs(Xn[..., 0]) < 0.048) & (np.abs(Xn[..., 1]) < 0.048) & (tt > 0)
    assert np.all(on_near[must_near & hit]) and np.all(hit[must_near])


def test_coverage_agrees_with_vertex_splat_and_refiner_runs(ops):
    """The mask of the rendered mesh and the reference's vertex splat (render_pointcloud, foreground = depth > 0) describe
    the same silhouette (a dense closed mesh: every splatted vertex pixel is covered up to the half-pixel convention), and
    PoseRefiner accepts the renderer as is (model/RNNPose.py:76-79) with views that move with the pose."""
    from rnnpose_amd.pose_refiner import PoseRefiner, default_config
    from rnnpose_amd.rasterizer import MeshRenderer
    from rnnpose_amd.transformation import SE3Sequence
    B, H, W = 2, 240, 320
    verts, faces = icosphere(sub=4)
    P = verts.shape[0]
    ren = MeshRenderer({"cat": dict(verts=verts, faces=faces, colors=syn.uniform("c", (P, 3), 1))})
    K = np.tile(np.array([[572.4114, 0, 160.0], [0, 573.57043, 120.0], [0, 0, 1]], np.float32), (B, 1, 1))
    G = syn.se3_exp_np(syn.normal("g", (B, 6), 4, std=0.3))
    G[:, :3, 3] = [0.01, -0.01, 0.8]
    G = G.astype(np.float32)
    names = ["cat"] * B
    pc = ren.render_pointcloud(names, T=T(G).cuda(), K=T(K).cuda(), render_image_size=(H, W))
    _, depth = ren(names, torch.zeros(1, P, 4, device="cuda"), T=T(G).cuda(), K=T(K).cuda(), render_image_size=(H, W))
    mesh = (depth > 0).float()
    grown = torch.nn.functional.max_pool2d(mesh, 3, 1, 1)                        # 1-px tolerance for the half-pixel convention
    fg = pc > 0
    assert float(fg.float().mean()) > 0.01
    assert float((grown[fg] > 0).float().mean()) > 0.999                        # splatted vertices lie on the rendered silhouette
    # a splatted vertex is never in FRONT of the rendered surface (it may be behind: back-facing vertices splat too), up to
    # the depth slope across one pixel -- unbounded at the limb, hence a quantile instead of the minimum
    both = fg & (depth > 0)
    assert float(torch.quantile((pc[both] - depth[both]).float(), 0.02)) > -5e-3      # (object depth extent: 0.1)
    # the refiner on top of it
    cfg = default_config(RENDER_ITER_COUNT=2, ITER_COUNT=2, OPTIM_ITER_COUNT=1, render_image_size=(H, W), zoom_crop_size=(128, 160))
    ref = PoseRefiner(cfg, renderer=ren).cuda().eval()
    ref.cf_net.update_block.load_state_dict({k: T(v) for k, v in syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0).items()})
    ref.image_fea_enc.fnet.load_state_dict({k: T(v) for k, v in syn.make_module_weights(orc.encoder_shapes(), seed=2).items()})
    t = lambda n, s, sd: T(syn.normal(n, s, sd, std=0.2)).cuda()
    out = ref(Ts=SE3Sequence(matrix=T(G).cuda()[:, None]), intrinsics=T(K).cuda(), image=T(syn.uniform("img", (B, 3, H, W), 5)).cuda(),
              fea_3d=t("f3", (1, P, 256), 6), Tj_gt=None, obj_cls=names, geofea_2d=t("g2", (B, 32, H, W), 7), geofea_3d=t("g3", (1, P, 32), 8))
    assert torch.isfinite(out["Ti_pred"].G).all() and out["syn_depth"][0].shape == (B, 1, 128, 160)
    assert float((out["syn_depth"][0] > 0).float().mean()) > 0.1               # the zoom window is filled by the object
    assert float((out["syn_depth"][0] - out["syn_depth"][2]).abs().max()) > 1e-5   # views moved with the pose
