"""Render hand-off of every outer refinement iteration (model/PoseRefiner.py:246-304) and checkpoint loading
(tools/eval.py:386-413) -- the two pieces of glue that make `PoseRefiner(cfg, renderer=diff_renderer)` and a reference
`.tckpt` usable with this package unchanged.

RendererAdapter turns ANY object with the call shape of the reference's `DiffRendererWrapper`
(geometry/diff_render_optim.py:404-494)

    renderer.render_pointcloud(model_names, T=, K=, render_image_size=)                       -> (B,1,H,W) vertex depth splat
    renderer(model_names, vert_attribute, T=, K=, render_image_size=, near=, far=, render_tex=True)
                                                                                              -> (B,3+C,h,w) colour|features, (B,1,h,w) depth (-1 = empty)
    renderer.render_depth(model_names, T=, K=, render_image_size=, near=, far=)               -> (B,1,h,w)

into the `render_views` protocol of rnnpose_amd.PoseRefiner.  The zoom window is computed on the device
(csrc/zoom_crop.hip: mask bounding box -> affine -> cropped intrinsics -> fused affine_grid+grid_sample); the reference
does the same arithmetic on the host behind two device->host synchronisations per outer iteration
(model/PoseRefiner.py:154,213).  `rnnpose_amd.rasterizer.MeshRenderer` is a HIP implementation of that call shape; the
reference's PyTorch3D renderer plugs in the same way.
"""
from __future__ import annotations

import re

import torch

from . import zoom


class RendererAdapter:
    def __init__(self, renderer, render_image_size=(480, 640), zoom_crop_size=(240, 240), legacy=True, margin_ratio=0.4,
                 near=0.1, far=6):
        """render_image_size / zoom_crop_size: `BASIC.render_image_size` / `BASIC.zoom_crop_size` of the reference config
        (config/default.py:48-49, config/linemod/template_fw0.5.yml:14-15)."""
        for name in ("render_pointcloud", "render_depth"):
            if not callable(getattr(renderer, name, None)):
                raise TypeError(f"renderer must provide {name}() (geometry/diff_render_optim.py:404-494)")
        if not callable(renderer):
            raise TypeError("renderer must be callable: renderer(model_names, vert_attribute, T=, K=, render_image_size=, ...)")
        self.renderer = renderer
        self.render_image_size = tuple(int(v) for v in render_image_size)
        self.zoom_crop_size = tuple(int(v) for v in zoom_crop_size)
        self.legacy = legacy
        self.margin_ratio = float(margin_ratio)
        self.near, self.far = near, far

    @torch.no_grad()
    def render_views(self, Ti, intrinsics, obj_cls=None, image=None, fea_3d=None, geofea_3d=None, geofea_2d=None):
        """Ti (B,4,4) current absolute pose, intrinsics (B,3,3) of the full image -> views dict of one outer iteration."""
        r, zs = self.renderer, self.zoom_crop_size
        pc_depth = r.render_pointcloud(obj_cls, T=Ti, K=intrinsics, render_image_size=self.render_image_size)       # :253-254
        B = pc_depth.shape[0]
        # foreground mask = pc_depth > 0 (:259); window, grids and cropped intrinsics on the device (:145-218)
        _, K_crop, theta = zoom.gen_zoom_crop_grids(pc_depth, intrinsics, Ti, [B, 1, *zs], margin_ratio=self.margin_ratio,
                                                    want_grids=False)
        fea_cat = torch.cat([fea_3d, geofea_3d], dim=-1) if geofea_3d is not None else fea_3d                        # :269-272
        color, depth = r(obj_cls, fea_cat, T=Ti, K=K_crop, render_image_size=zs, near=self.near, far=self.far,
                         render_tex=True)                                                                            # :135-137
        depth = depth.detach().masked_fill(depth == -1, 0.0)                                                         # :139 (no host sync)
        c3 = fea_3d.shape[-1]
        if geofea_3d is not None:
            syn_img, cfea, geofea1 = torch.split(color, [3, c3, geofea_3d.shape[-1]], dim=1)                         # :277
        else:
            syn_img, cfea = torch.split(color, [3, c3], dim=1)
            geofea1 = None
        cfea = (cfea * 0.1).contiguous()                                                                             # :283
        image_crop = zoom.zoom_crop(image, theta, zs)                                                                # :287
        geofea2_crop = None
        if geofea1 is not None and geofea_2d is not None:
            geofea1 = geofea1.contiguous()
            geofea2_crop = zoom.zoom_crop(geofea_2d, theta, zs)                                                      # :291
        syn_depth = depth
        if self.legacy:                                                                                              # :295-304
            syn_depth = r.render_depth(obj_cls, T=Ti, K=K_crop, render_image_size=zs, near=self.near, far=self.far)
        return dict(syn_img=syn_img.contiguous(), image_crop=image_crop, cfea=cfea, geofea1=geofea1,
                    geofea2_crop=geofea2_crop, syn_depth=syn_depth.contiguous(), intrinsics_crop=K_crop,
                    fmap1=None, fmap2=None, theta=theta, pc_depth=pc_depth)


def filter_param_dict(state_dict, include=None, exclude=None):
    """tools/eval.py:110-127: keep keys matching `include` (re.match) and not matching `exclude`."""
    inc = re.compile(include) if include is not None else None
    exc = re.compile(exclude) if exclude is not None else None
    return {k: p for k, p in state_dict.items()
            if (inc is None or inc.match(k) is not None) and (exc is None or exc.match(k) is None)}


def load_motion_net_checkpoint(refiner, checkpoint, prefix="motion_net.", include=None, exclude=None, strict=False):
    """Load the `motion_net.*` sub-tree of a reference checkpoint (`.tckpt` = torch.save(RNNPose.state_dict()),
    torchplus/train/checkpoint.py:92) into a rnnpose_amd.PoseRefiner, with the selection rule of tools/eval.py:386-413:
    include/exclude regular expressions on the FULL key, then only keys that exist in the model with the same shape
    are taken; the others are reported.  -> (loaded_keys, skipped_keys).  `checkpoint`: path or state dict."""
    sd = torch.load(checkpoint, map_location="cpu") if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__") else checkpoint
    sd = filter_param_dict(dict(sd), include, exclude)
    model = refiner.state_dict()
    loaded, skipped = {}, []
    for k, v in sd.items():
        if not k.startswith(prefix):
            continue
        kk = k[len(prefix):]
        if kk in model and tuple(v.shape) == tuple(model[kk].shape):
            loaded[kk] = v
        else:
            skipped.append(k)
    missing = [k for k in model if k not in loaded]
    if strict and (missing or skipped):
        raise RuntimeError(f"checkpoint does not cover the model: missing {missing[:5]}..., skipped {skipped[:5]}...")
    model.update(loaded)
    refiner.load_state_dict(model)
    return sorted(loaded), skipped
