"""A stand-in for the reference's `DiffRendererWrapper` (geometry/diff_render_optim.py:404-494) with exactly its call shape,
written in plain torch: z-buffered vertex splats instead of PyTorch3D's mesh rasteriser.  Test infrastructure only -- it
exists so that the render hand-off (rnnpose_amd/render_adapter.py) and the hipGraph paths of PoseRefiner see views that
CHANGE with the pose between outer iterations, as they do behind the real renderer."""
import torch


class FakeDiffRenderer:
    def __init__(self, verts_by_class, colors_by_class, splat=2):
        self.verts = verts_by_class            # {name: (P,3)}
        self.colors = colors_by_class          # {name: (P,3)}
        self.splat = splat
        self.calls = []

    def _project(self, v, T, K):
        X = v @ T[:3, :3].t() + T[:3, 3]
        x = X @ K.t()
        return x[:, 0] / x[:, 2], x[:, 1] / x[:, 2], x[:, 2]

    def _zbuffer(self, name, T, K, size, splat):
        """-> depth (H,W) (inf where empty), winner vertex index (H,W) (-1 where empty)"""
        H, W = size
        v = self.verts[name]
        px, py, z = self._project(v, T, K)
        ix0, iy0 = torch.round(px).long(), torch.round(py).long()
        idx_all, z_all, vid_all = [], [], []
        r = range(-(splat // 2), splat - splat // 2)
        for dy in r:
            for dx in r:
                ix, iy = ix0 + dx, iy0 + dy
                ok = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H) & (z > 0.05)
                idx_all.append((iy * W + ix)[ok])
                z_all.append(z[ok])
                vid_all.append(torch.nonzero(ok)[:, 0])
        idx, zz, vid = torch.cat(idx_all), torch.cat(z_all), torch.cat(vid_all)
        depth = torch.full((H * W,), float("inf"), device=v.device)
        depth.scatter_reduce_(0, idx, zz, reduce="amin")
        win = zz == depth[idx]
        winner = torch.full((H * W,), v.shape[0], device=v.device, dtype=torch.long)
        winner.scatter_reduce_(0, idx[win], vid[win], reduce="amin")
        winner[winner == v.shape[0]] = -1
        return depth.view(H, W), winner.view(H, W)

    def render_pointcloud(self, model_names, T, K, render_image_size, near=0.1, far=6):
        self.calls.append("render_pointcloud")
        out = []
        for b, name in enumerate(model_names):
            d, _ = self._zbuffer(name, T[b], K[b], render_image_size, 1)
            out.append(torch.where(torch.isinf(d), torch.zeros_like(d), d)[None, None])
        return torch.cat(out, 0)

    def render_depth(self, model_names, T, K, render_image_size, near=0.1, far=6):
        self.calls.append("render_depth")
        out = []
        for b, name in enumerate(model_names):
            d, _ = self._zbuffer(name, T[b], K[b], render_image_size, self.splat)
            out.append(torch.where(torch.isinf(d), torch.zeros_like(d), d)[None, None])
        return torch.cat(out, 0)

    def __call__(self, model_names, vert_attribute, T, K, render_image_size, near=0.1, far=6, render_tex=False):
        self.calls.append("forward")
        maps, depths = [], []
        for b, name in enumerate(model_names):
            d, win = self._zbuffer(name, T[b], K[b], render_image_size, self.splat)
            attr = vert_attribute[b if vert_attribute.shape[0] > 1 else 0]
            if render_tex:
                attr = torch.cat([self.colors[name], attr], dim=-1)
            fm = attr[win.clamp(min=0)] * (win >= 0)[..., None]            # (H,W,C)
            maps.append(fm.permute(2, 0, 1)[None])
            depths.append(torch.where(win >= 0, d, torch.full_like(d, -1.0))[None, None])   # -1 = empty (PoseRefiner.py:139)
        return torch.cat(maps, 0), torch.cat(depths, 0)
