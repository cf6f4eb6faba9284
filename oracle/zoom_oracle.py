"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's zoom-crop.

Follows model/PoseRefiner.py:145-218 (get_affine_transformation, gen_zoom_crop_grids) and the torch functions it feeds
(F.affine_grid / F.grid_sample with their defaults: bilinear, zero padding, align_corners=False).

Pinning: `affine_grid` / `grid_sample` below are checked against torch's own CPU implementations (the functions the
reference calls) through tests/golden/zoom_*.npz.  The window arithmetic calls cv2.getAffineTransform in the reference;
OpenCV is not installed in this image, so the 3-point solve is restated in closed form (the point triples are axis
aligned) and checked through the point correspondences that define it -- that part is "parity unpinned" against cv2."""
import numpy as np


def mask_bbox(depth):
    """depth (B,1,H,W) -> (B,4) int [xmin,ymin,xmax,ymax]; empty -> zeros   (PoseRefiner.py:154-164)"""
    out = np.zeros((depth.shape[0], 4), np.int64)
    for b in range(depth.shape[0]):
        ys, xs = np.nonzero(depth[b, 0] > 0)
        if len(ys) > 0:
            out[b] = [xs.min(), ys.min(), xs.max(), ys.max()]
    return out


def affine_from_axis_aligned(src, dst):
    """cv2.getAffineTransform(src, dst) for src = [[a,b],[a,d],[c,b]], dst = [[x1,y1],[x1,y2],[x2,y1]] (float64)."""
    (a, b), (_, d), (c, _) = np.float64(src)
    (x1, y1), (_, y2), (x2, _) = np.float64(dst)
    sx, sy = (x2 - x1) / (c - a), (y2 - y1) / (d - b)
    return np.array([[sx, 0.0, x1 - sx * a], [0.0, sy, y1 - sy * b]])


def zoom_params(bbox, K, T, H, W, hc, wc, margin_ratio=0.4):
    """-> theta (B,2,3) fp32, K_crop (B,3,3) fp32   (PoseRefiner.py:145-218)"""
    B = K.shape[0]
    K = K.astype(np.float32)
    c = np.einsum("bij,bj->bi", K, T[:, :3, 3].astype(np.float32)).astype(np.float32)
    center = (c[:, :2] / c[:, 2:3]).astype(np.float32)
    ratio = float(H) / float(W)
    theta = np.zeros((B, 2, 3), np.float32)
    K_crop = np.zeros((B, 3, 3), np.float32)
    for b in range(B):
        cx, cy = float(center[b, 0]), float(center[b, 1])
        x0, y0, x1, y1 = [float(v) for v in bbox[b]]
        crop_h = max(ratio * (x1 - cx), ratio * (cx - x0), cy - y0, y1 - cy) * 2 * (1 + margin_ratio)
        crop_w = crop_h / ratio
        nx1, nx2 = (cx - crop_w / 2) * 2 / W - 1, (cx + crop_w / 2) * 2 / W - 1
        ny1, ny2 = (cy - crop_h / 2) * 2 / H - 1, (cy + crop_h / 2) * 2 / H - 1
        pts1 = np.float32([[nx1, ny1], [nx1, ny2], [nx2, ny1]])
        pts2 = np.float32([[-1, -1], [-1, 1], [1, -1]])
        theta[b] = affine_from_axis_aligned(pts2, pts1).astype(np.float32)
        wx1, wx2, wy1, wy2 = cx - crop_w / 2, cx + crop_w / 2, cy - crop_h / 2, cy + crop_h / 2
        pts1 = np.float32([[wx1, wy1], [wx1, wy2], [wx2, wy1]])
        pts2 = np.float32([[0, 0], [0, hc - 1], [wc - 1, 0]])
        M = np.eye(3, dtype=np.float32)
        M[:2] = affine_from_axis_aligned(pts2, pts1).astype(np.float32)
        K_crop[b] = (np.linalg.inv(M.astype(np.float64)) @ K[b].astype(np.float64)).astype(np.float32)
    return theta, K_crop


def affine_grid(theta, hc, wc):
    """F.affine_grid(theta, (B,C,hc,wc), align_corners=False) -> (B,hc,wc,2)"""
    xs = (2.0 * np.arange(wc) + 1.0) / wc - 1.0
    ys = (2.0 * np.arange(hc) + 1.0) / hc - 1.0
    bx, by = np.meshgrid(xs, ys)
    base = np.stack([bx, by, np.ones_like(bx)], -1)                       # (hc,wc,3)
    return np.einsum("hwk,bik->bhwi", base, theta.astype(np.float64)).astype(np.float32)


def grid_sample(x, grid):
    """F.grid_sample(x, grid) bilinear, zeros, align_corners=False.  x (B,C,H,W), grid (B,hc,wc,2) -> (B,C,hc,wc)"""
    B, C, H, W = x.shape
    ix = ((grid[..., 0].astype(np.float64) + 1) * W - 1) / 2
    iy = ((grid[..., 1].astype(np.float64) + 1) * H - 1) / 2
    x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
    out = np.zeros((B, C) + grid.shape[1:3], np.float64)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            w = (1 - np.abs(ix - xx)) * (1 - np.abs(iy - yy))
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            xc, yc = np.clip(xx, 0, W - 1), np.clip(yy, 0, H - 1)
            for b in range(B):
                out[b] += x[b][:, yc[b], xc[b]] * (w[b] * ok[b])[None]
    return out.astype(np.float32)


def render_pointcloud(verts, T, K, H, W):
    """One image: geometry/diff_render_optim.py:369-401 (fp32).  verts (P,3), T (4,4), K (3,3) -> (H,W) depth splat.
    Several vertices on one pixel: the reference's torch indexed assignment leaves the winner unspecified; numpy keeps the
    LAST of repeated indices, and so does the HIP kernel (highest vertex index).  Only `> 0` is consumed downstream."""
    R = T[:3, :3].T.astype(np.float32)
    t = T[:3, 3].astype(np.float32)
    v = verts.astype(np.float32)
    Xc = ((v[:, 0:1] * R[0] + v[:, 1:2] * R[1]) + v[:, 2:3] * R[2] + t).astype(np.float32)
    Kt = K.T.astype(np.float32)
    x = ((Xc[:, 0:1] * Kt[0] + Xc[:, 1:2] * Kt[1]) + Xc[:, 2:3] * Kt[2]).astype(np.float32)
    depth = x[:, 2].copy()
    with np.errstate(all="ignore"):
        u = np.rint(x[:, 0] / x[:, 2])
        w = np.rint(x[:, 1] / x[:, 2])
    px = np.where(np.isfinite(u) & (np.abs(u) < 1e9), np.clip(u, 0, W - 1), 0).astype(np.int64)
    py = np.where(np.isfinite(w) & (np.abs(w) < 1e9), np.clip(w, 0, H - 1), 0).astype(np.int64)
    out = np.zeros((H, W), np.float32)
    out[py, px] = depth
    return out
