"""Zoom-crop: oracle vs torch-generated golden vectors on CPU, HIP kernels vs oracle / golden /
torch on the GPU.  Reference: model/PoseRefiner.py:145-218,286-291."""
import os

import numpy as np
import pytest
import torch

from oracle import zoom_oracle as zo

GOLD = os.path.join(os.path.dirname(__file__), "golden", "zoom_small.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


# ---------------------------------------------------------------- CPU: oracle pinned to torch's affine_grid / grid_sample
def test_oracle_affine_grid_and_sample_match_torch(gold):
    hc, wc = [int(v) for v in gold["crop_size"]]
    grid = zo.affine_grid(gold["theta_oracle"], hc, wc)
    np.testing.assert_allclose(grid, gold["grid_torch"], atol=2e-6, rtol=0)
    crop = zo.grid_sample(gold["x"][:2], gold["grid_torch"])
    np.testing.assert_allclose(crop, gold["crop_torch"], atol=2e-5, rtol=0)


def test_oracle_bbox(gold):
    bb = zo.mask_bbox(gold["depth"])
    assert bb.tolist() == [[20, 10, 49, 29], [5, 23, 19, 37], [0, 0, 0, 0]]
    assert (bb == gold["bbox"]).all()


def test_axis_aligned_affine_solves_the_three_points():
    rng = np.random.default_rng(0)
    for _ in range(20):
        a, b = rng.normal(size=2)
        c, d = a + rng.uniform(0.5, 3), b + rng.uniform(0.5, 3)
        x1, y1 = rng.normal(size=2) * 50
        x2, y2 = x1 + rng.uniform(1, 200), y1 + rng.uniform(1, 200)
        src = np.array([[a, b], [a, d], [c, b]])
        dst = np.array([[x1, y1], [x1, y2], [x2, y1]])
        M = zo.affine_from_axis_aligned(src, dst)
        np.testing.assert_allclose(M[:, :2] @ src.T + M[:, 2:], dst.T, atol=1e-9)      # definition of getAffineTransform


def test_oracle_window_properties(gold):
    """The crop window is centred on the projected model origin, keeps the aspect ratio, covers the mask with the
    margin, and K_crop maps the same 3-D point to the crop pixel the window transform predicts."""
    H, W = gold["depth"].shape[-2:]
    hc, wc = [int(v) for v in gold["crop_size"]]
    theta, K_crop = gold["theta_oracle"], gold["K_crop_oracle"]
    K, T, bbox = gold["K"], gold["T"], gold["bbox"]
    for b in range(2):
        c = K[b] @ T[b, :3, 3]
        cx, cy = c[0] / c[2], c[1] / c[2]
        # theta: centre of the normalised window = crop centre; half extents keep H/W ratio in pixels
        np.testing.assert_allclose((theta[b, 0, 2] + 1) * W / 2, cx, atol=1e-3)
        np.testing.assert_allclose((theta[b, 1, 2] + 1) * H / 2, cy, atol=1e-3)
        cw, ch = theta[b, 0, 0] * W, theta[b, 1, 1] * H
        np.testing.assert_allclose(ch / cw, H / W, rtol=1e-5)
        x0, y0, x1, y1 = bbox[b]
        assert cx - cw / 2 <= x0 and cx + cw / 2 >= x1 and cy - ch / 2 <= y0 and cy + ch / 2 >= y1
        # the model origin projects to the centre of the crop
        p = K_crop[b] @ T[b, :3, 3]
        np.testing.assert_allclose([p[0] / p[2], p[1] / p[2]], [(wc - 1) / 2, (hc - 1) / 2], atol=2e-3)


# ---------------------------------------------------------------- GPU: HIP kernels
def D(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from rnnpose_amd import build, ops as _ops
    build.build()
    return _ops


@pytest.mark.gpu
def test_gpu_bbox_and_params_match_oracle(ops, gold):
    depth, K, T = gold["depth"], gold["K"], gold["T"]
    H, W = depth.shape[-2:]
    hc, wc = [int(v) for v in gold["crop_size"]]
    bb = ops.mask_bbox(D(depth)).cpu().numpy()
    assert bb[:2].tolist() == gold["bbox"][:2].tolist()
    assert bb[2].tolist() == [2**31 - 1, 2**31 - 1, -1, -1]                    # empty mask sentinel
    theta, K_crop = ops.zoom_crop_params(D(bb.astype(np.int32)), D(K), D(T), (H, W), (hc, wc), 0.4)
    np.testing.assert_allclose(theta.cpu().numpy()[:2], gold["theta_oracle"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(K_crop.cpu().numpy()[:2], gold["K_crop_oracle"], rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_gpu_zoom_crop_matches_torch_golden(ops, gold):
    hc, wc = [int(v) for v in gold["crop_size"]]
    out, grid = ops.zoom_crop(D(gold["x"][:2]), D(gold["theta_oracle"]), (hc, wc), want_grid=True)
    np.testing.assert_allclose(grid.cpu().numpy(), gold["grid_torch"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out.cpu().numpy(), gold["crop_torch"], atol=2e-5, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,hc,wc", [(8, 3, 480, 640, 240, 240), (2, 32, 480, 640, 320, 320), (1, 1, 7, 5, 3, 9)])
def test_gpu_zoom_pipeline_full_size_vs_torch(ops, B, C, H, W, hc, wc):
    """BASELINE sizes: the fused kernel against F.grid_sample(F.affine_grid) run by torch on the same device, with the
    window taken from a disc-shaped depth mask; plus the facade (no host synchronisation inside)."""
    import torch.nn.functional as F
    from rnnpose_amd import synthetic as syn, zoom
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, C, H, W, device="cuda", generator=g)
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    depth = torch.zeros(B, 1, H, W, device="cuda")
    for b in range(B):
        cx, cy, r = W * (0.3 + 0.05 * b), H * (0.6 - 0.04 * b), min(H, W) * (0.1 + 0.02 * b)
        depth[b, 0][((xx - cx) ** 2 + (yy - cy) ** 2) < r * r] = 1.0 + b
    K = D(syn.intrinsics(B, H, W).astype(np.float32))
    T = torch.eye(4, device="cuda").repeat(B, 1, 1)
    T[:, 2, 3] = 0.8
    T[:, 0, 3] = torch.linspace(-0.05, 0.05, B, device="cuda")
    grids, K_crop, theta = zoom.gen_zoom_crop_grids(depth, K, T, [B, C, hc, wc])
    ref_grid = F.affine_grid(theta, [B, C, hc, wc], align_corners=False)
    ref = F.grid_sample(x, ref_grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    out = zoom.zoom_crop(x, theta, (hc, wc))
    assert float((grids - ref_grid).abs().max()) < 5e-6
    # coordinates agree to ~1e-6 of the normalised range = ~3e-4 px: values within a few 1e-4 of unit-variance noise
    assert float((out - ref).abs().max()) < 2e-3
    assert float((out - ref).abs().mean()) < 2e-5
    bb = zo.mask_bbox(depth.cpu().numpy())
    th_o, Kc_o = zo.zoom_params(bb, K.cpu().numpy(), T.cpu().numpy(), H, W, hc, wc)
    np.testing.assert_allclose(theta.cpu().numpy(), th_o, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(K_crop.cpu().numpy(), Kc_o, rtol=1e-5, atol=1e-3)


# ---------------------------------------------------------------- point-cloud depth splat (foreground source of the crop)
PC_GOLD = os.path.join(os.path.dirname(__file__), "golden", "pointcloud_small.npz")


def test_oracle_pointcloud_matches_torch_statements():
    """oracle vs the reference's own torch statements (diff_render_optim.py:381-401) run on CPU: the SAME foreground mask
    (all the zoom crop consumes).  Which of several vertices landing on one pixel leaves its depth there is unspecified in
    the reference (torch's indexed assignment with repeated indices); the oracle and the kernel keep the highest vertex
    index -- so every torch depth must be the depth of SOME vertex of that pixel, and ours that of the last one."""
    g = dict(np.load(PC_GOLD))
    H, W = [int(v) for v in g["size"]]
    d = zo.render_pointcloud(g["verts"], g["T"], g["K"], H, W)
    np.testing.assert_array_equal(d > 0, g["depth_torch"] > 0)
    # per-pixel candidate depths, by brute force
    R, t = g["T"][:3, :3].astype(np.float32), g["T"][:3, 3].astype(np.float32)
    Xc = g["verts"] @ R.T + t
    x = Xc @ g["K"].T
    px = np.clip(np.rint(x[:, 0] / x[:, 2]), 0, W - 1).astype(int)
    py = np.clip(np.rint(x[:, 1] / x[:, 2]), 0, H - 1).astype(int)
    for yy, xx in zip(*np.nonzero(d > 0)):
        cand = x[(px == xx) & (py == yy), 2]
        assert np.abs(cand - g["depth_torch"][yy, xx]).min() < 1e-5
        assert abs(cand[-1] - d[yy, xx]) < 1e-5


@pytest.mark.gpu
def test_gpu_pointcloud_depth_matches_oracle_and_feeds_the_crop(ops):
    from rnnpose_amd import zoom
    g = dict(np.load(PC_GOLD))
    H, W = [int(v) for v in g["size"]]
    want = zo.render_pointcloud(g["verts"], g["T"], g["K"], H, W)
    T2 = np.stack([g["T"], g["T"]])
    T2[1, :3, 3] += np.float32([0.05, 0.02, 0.1])
    K2 = np.stack([g["K"], g["K"]])
    got = zoom.render_pointcloud([D(g["verts"]), D(g["verts"][:1500])], D(T2), D(K2), (H, W)).cpu().numpy()
    np.testing.assert_array_equal(got[0, 0], want)                                  # bit-exact incl. collisions
    np.testing.assert_array_equal(got[1, 0], zo.render_pointcloud(g["verts"][:1500], T2[1], g["K"], H, W))
    bb = ops.mask_bbox(D(got)).cpu().numpy()
    assert bb.tolist() == zo.mask_bbox(got).tolist()
    # off-screen and behind-the-camera vertices: clamped to the border / negative depth, never out of bounds
    far = np.float32([[5.0, 0, 0], [-5.0, 0, 0], [0, 5.0, 0], [0, 0, -2.0], [0, 0, 0]])
    Tn = np.eye(4, dtype=np.float32)
    Tn[2, 3] = 0.5
    g2 = zoom.render_pointcloud([D(far)], D(Tn[None]), D(g["K"][None]), (H, W)).cpu().numpy()
    np.testing.assert_array_equal(g2[0, 0], zo.render_pointcloud(far, Tn, g["K"], H, W))
