"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64, loop over faces) of the mesh rasteriser of
rnnpose_amd/csrc/raster.hip: nearest face per pixel centre by interpolated camera z, perspective-correct barycentrics
(PyTorch3D `BarycentricPerspectiveCorrection`), nearest-vertex depth, interpolated vertex attributes.
PARITY UNPINNED: the reference renders through PyTorch3D (geometry/diff_render_optim.py:283-367), which is not in this image
and for which the reference holds no fixture; this oracle states the documented semantics, not PyTorch3D's code."""
import numpy as np


def rasterize(verts, faces, T, K, H, W, near=0.1, pixel_center=0.5, perspective=True):
    """verts (P,3), faces (F,3), T (4,4), K (3,3) -> face index (H,W) (-1 empty), zbuf (H,W), bary (H,W,3), vz (H,W)"""
    v = np.asarray(verts, np.float64)
    Xc = v @ np.asarray(T, np.float64)[:3, :3].T + np.asarray(T, np.float64)[:3, 3]
    K = np.asarray(K, np.float64)
    z = Xc[:, 2]
    zc = np.where(z > near, z, near)
    sx = K[0, 0] * Xc[:, 0] / zc + K[0, 2]
    sy = K[1, 1] * Xc[:, 1] / zc + K[1, 2]
    best_z = np.full((H, W), np.inf)
    best_f = np.full((H, W), -1, np.int64)
    best_w = np.zeros((H, W, 3))
    ys, xs = np.mgrid[0:H, 0:W]
    px, py = xs + pixel_center, ys + pixel_center
    for fi, (a, b, c) in enumerate(np.asarray(faces)):
        if not (z[a] > near and z[b] > near and z[c] > near):
            continue
        x0, y0, x1, y1, x2, y2 = sx[a], sy[a], sx[b], sy[b], sx[c], sy[c]
        area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
        if abs(area) <= 1e-8:
            continue
        lo_x, hi_x = int(max(0, np.floor(min(x0, x1, x2) - 1))), int(min(W - 1, np.ceil(max(x0, x1, x2) + 1)))
        lo_y, hi_y = int(max(0, np.floor(min(y0, y1, y2) - 1))), int(min(H - 1, np.ceil(max(y0, y1, y2) + 1)))
        if lo_x > hi_x or lo_y > hi_y:
            continue
        sl = (slice(lo_y, hi_y + 1), slice(lo_x, hi_x + 1))
        qx, qy = px[sl], py[sl]
        w0 = ((x2 - x1) * (qy - y1) - (y2 - y1) * (qx - x1)) / area
        w1 = ((x0 - x2) * (qy - y2) - (y0 - y2) * (qx - x2)) / area
        w2 = ((x1 - x0) * (qy - y0) - (y1 - y0) * (qx - x0)) / area
        inside = (w0 >= 0) & (w1 >= 0) & (w2 >= 0)
        if perspective:
            q0, q1, q2 = w0 / z[a], w1 / z[b], w2 / z[c]
            s = q0 + q1 + q2
            with np.errstate(invalid="ignore", divide="ignore"):
                w0, w1, w2 = q0 / s, q1 / s, q2 / s
        zz = w0 * z[a] + w1 * z[b] + w2 * z[c]
        better = inside & (zz > 0) & (zz < best_z[sl])
        bz, bf, bw = best_z[sl], best_f[sl], best_w[sl]
        bz[better] = zz[better]
        bf[better] = fi
        bw[better] = np.stack([w0, w1, w2], -1)[better]
    hit = best_f >= 0
    fz = z[np.asarray(faces)[np.clip(best_f, 0, None)]]                      # (H,W,3)
    vz = np.where(hit, np.take_along_axis(fz, best_w.argmax(-1)[..., None], -1)[..., 0], 0.0)
    return best_f, np.where(hit, best_z, -1.0), best_w, vz


def interpolate(best_f, best_w, faces, attr):
    """-> (C,H,W) attribute map, 0 where empty"""
    f = np.asarray(faces)[np.clip(best_f, 0, None)]                          # (H,W,3)
    a = np.asarray(attr, np.float64)[f]                                      # (H,W,3,C)
    out = (a * best_w[..., None]).sum(2) * (best_f >= 0)[..., None]
    return np.moveaxis(out, -1, 0)
