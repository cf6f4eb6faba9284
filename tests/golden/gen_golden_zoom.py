#!/usr/bin/env python3
"""Golden vectors for the zoom-crop: outputs of torch's F.affine_grid / F.grid_sample -- the
functions model/PoseRefiner.py:214,286-291 calls -- on small seeded inputs.  Run from the repo root:
    python tests/golden/gen_golden_zoom.py
(cv2.getAffineTransform, used by the reference for the window arithmetic, is not installed here; the window parameters
stored in the fixture come from oracle/zoom_oracle.py and are labelled as such.)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import zoom_oracle as zo          # noqa: E402
from rnnpose_amd import synthetic as syn      # noqa: E402

B, C, H, W, hc, wc = 3, 5, 48, 64, 24, 32
depth = np.zeros((B, 1, H, W), np.float32)
depth[0, 0, 10:30, 20:50] = 1.5                       # box
yy, xx = np.mgrid[0:H, 0:W]
depth[1, 0][(yy - 30) ** 2 + (xx - 12) ** 2 < 64] = 0.7    # disc near the left border (window leaves the image)
# depth[2]: empty mask
K = syn.intrinsics(B, H, W).astype(np.float32)
T = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
T[:, :3, 3] = np.float32([[0.02, -0.01, 0.9], [-0.12, 0.05, 0.8], [0.0, 0.0, 1.0]])
x = syn.normal("zoom.x", (B, C, H, W), 7, std=1.0).astype(np.float32)
bbox = zo.mask_bbox(depth)
theta, K_crop = zo.zoom_params(bbox[:2], K[:2], T[:2], H, W, hc, wc)      # the empty-mask sample has no finite window
theta_t = torch.from_numpy(theta)
grid = F.affine_grid(theta_t, [2, C, hc, wc], align_corners=False)
crop = F.grid_sample(torch.from_numpy(x[:2]), grid, mode="bilinear", padding_mode="zeros", align_corners=False)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "zoom_small.npz"), depth=depth, K=K, T=T, x=x, bbox=bbox,
                    theta_oracle=theta, K_crop_oracle=K_crop, grid_torch=grid.numpy(), crop_torch=crop.numpy(),
                    crop_size=np.int64([hc, wc]))
print("wrote zoom_small.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "zoom_small.npz")), "bytes")

# ---- point-cloud depth splat: the torch statements of DiffRender.render_pointcloud (geometry/diff_render_optim.py:381-401),
# executed on CPU on a seeded vertex cloud (the class itself imports PyTorch3D and cannot be instantiated here)
P, Hs, Ws = 4000, 60, 80
verts = syn.normal("pc.v", (P, 3), 11, std=0.05).astype(np.float32)
Tp = np.eye(4, dtype=np.float32)
Tp[:3, :3] = syn.se3_exp_np(np.float32([0, 0, 0, 0.3, -0.2, 0.1]))[0, :3, :3]
Tp[:3, 3] = np.float32([0.02, -0.03, 0.6])
Kp = np.float32([[70.0, 0, 40.0], [0, 70.0, 30.0], [0, 0, 1]])
Tt, Kt, vt = torch.from_numpy(Tp)[None], torch.from_numpy(Kp)[None], torch.from_numpy(verts)
R = Tt[..., :3, :3].transpose(-1, -2)
t = Tt[..., :3, 3]
X_cam = (vt @ R + t)
xx = X_cam @ Kt.transpose(-1, -2)
dep = xx[..., -1]
xx = xx / xx[..., -1:]
outp = torch.zeros([1, 1, Hs, Ws])
outp[:, :, torch.round(xx[0, :, 1]).long().clamp(0, Hs - 1), torch.round(xx[0, :, 0]).long().clamp(0, Ws - 1)] = dep
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pointcloud_small.npz"), verts=verts, T=Tp, K=Kp,
                    depth_torch=outp[0, 0].numpy(), size=np.int64([Hs, Ws]))
print("wrote pointcloud_small.npz")
