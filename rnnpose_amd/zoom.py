"""Zoom-crop of every outer refinement iteration, on device.

Mirrors `PoseRefiner.get_affine_transformation` / `gen_zoom_crop_grids` (model/PoseRefiner.py:145-218) and the
`F.grid_sample(image, grids)` / `F.grid_sample(geofea_2d, grids)` that consume the grids (:286-291) -- without the two
device->host synchronisations per outer iteration the reference pays (mask -> numpy at :154, crop centre at :213)."""
from __future__ import annotations

import torch

from . import ops


def gen_zoom_crop_grids(fg_depth, K, T, output_size, margin_ratio=0.4, want_grids=True):
    """fg_depth (B,1,H,W): the point-cloud depth whose `> 0` pixels are the foreground mask (PoseRefiner.py:259);
    K (B,3,3), T (B,4,4); output_size = [B, C, hc, wc] like the reference.
    -> (grids (B,hc,wc,2) or None, intrinsics_crop (B,3,3), theta (B,2,3))."""
    _, _, H, W = fg_depth.shape
    hc, wc = int(output_size[-2]), int(output_size[-1])
    bbox = ops.mask_bbox(fg_depth.float())
    theta, K_crop = ops.zoom_crop_params(bbox, K.float().contiguous(), T.float().contiguous(), (H, W), (hc, wc), margin_ratio)
    grids = ops.zoom_crop(None, theta, (hc, wc), want_grid=True)[1] if want_grids else None
    return grids, K_crop, theta


def zoom_crop(x, theta, crop_size):
    """F.grid_sample(x, F.affine_grid(theta, ...)) in one kernel (PoseRefiner.py:286-291)."""
    return ops.zoom_crop(x.float().contiguous(), theta, crop_size)


def render_pointcloud(verts_per_image, T, K, render_image_size):
    """`DiffRendererWrapper.render_pointcloud` (geometry/diff_render_optim.py:474-480) for a batch: verts_per_image is a
    list of (P_b,3) vertex tensors (one model per image) -> (B,1,H,W) depth splat, foreground = depth > 0."""
    verts = torch.cat([v.float() for v in verts_per_image], 0).contiguous()
    counts = torch.tensor([0] + [int(v.shape[0]) for v in verts_per_image], dtype=torch.int32)
    offs = torch.cumsum(counts, 0).to(torch.int32).to(verts.device)
    return ops.pointcloud_depth(verts, offs, T.float().contiguous(), K.float().contiguous(), render_image_size)
