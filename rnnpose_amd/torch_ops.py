"""`torch.ops.rnnpose.*`: the SURVEY.md section 8(b) operator set registered with the PyTorch dispatcher on top of the C ABI.

    import rnnpose_amd.torch_ops            # registers the library (idempotent)
    pyr  = torch.ops.rnnpose.corr_pyramid(fmap1, fmap2, 4)        # Tensor[] of (B*N,1,h_l,w_l) levels (one buffer)
    corr = torch.ops.rnnpose.corr_lookup(pyr, coords, 4)           # (B, L*81, h, w)

Schemas are the ones SURVEY 8(b) lists.  Kernels are registered for the CUDA (= HIP on ROCm) dispatch key only: a CPU
tensor fails in the dispatcher ("could not run ... with arguments from the 'CPU' backend") -- there is no CPU product path.
Every op also has a fake (meta) implementation, so FakeTensorMode / torch.compile tracing see correct shapes and dtypes
without touching the GPU.  Inference only: no autograd formulas (tools/eval.py:526 runs under no_grad).

Reference call sites replaced: thirdparty/raft/corr.py:13-57 (CorrBlock), model/CFNet.py:95-106 (upsample_flow),
geometry/transformation.py:184-198,265-316 (SE3.transform, reprojction_optim), model/PoseRefiner.py:342-345 (weight),
geometry/cholesky.py:32-50 + geometry/se3.py:303-306 (solve + increment).
"""
from __future__ import annotations

import torch

from . import ops

_lib = None
_SCHEMAS = {
    "corr_pyramid": "(Tensor fmap1, Tensor fmap2, int levels=4) -> Tensor[]",
    "corr_lookup": "(Tensor[] pyramid, Tensor coords, int radius=4) -> Tensor",
    "convex_upsample": "(Tensor flow, Tensor mask, int scale=8) -> Tensor",
    "induced_flow": "(Tensor depth, Tensor K, Tensor G, float eps=1e-5) -> (Tensor flow, Tensor vmask)",
    "corr_weight": "(Tensor g1, Tensor g2, Tensor target, Tensor depth, Tensor sigma) -> Tensor",
    "lm_normal_eq": "(Tensor target, Tensor weight, Tensor depth, Tensor K, Tensor G) -> (Tensor H, Tensor b)",
    "lm_solve_update": "(Tensor H, Tensor b, Tensor G, float ep_lambda=100.0, float lm_lambda=1e-4, float max_update=1.0) -> (Tensor G_new, Tensor xi)",
    "lm_step": "(Tensor target, Tensor weight, Tensor depth, Tensor K, Tensor G, int num_iters=1, float ep_lambda=100.0, float lm_lambda=1e-4, float max_update=1.0) -> (Tensor G_new, Tensor xi)",
}


def _level_sizes(h, w, levels):
    return [(h >> l, w >> l) for l in range(levels)]


def _flat_pyramid(pyramid):
    """Tensor[] levels in the reference's shapes -> the flat buffer the lookup kernel reads (level 0 j-patch-major:
    include/rnnpose_hip.h).  The dispatcher-level operators carry the pyramid as the reference does, a list of dense levels, so
    this packs once per call; the refinement path (CorrBlock / the engine) keeps the buffer form and never comes through here."""
    return ops.pyramid_from_levels(list(pyramid))


# ---- CUDA (HIP) implementations ----------------------------------------------------------------------------------------
def _corr_pyramid(fmap1, fmap2, levels=4):
    _, views = ops.corr_pyramid(fmap1, fmap2, levels)
    return list(views)


def _corr_lookup(pyramid, coords, radius=4):
    return ops.corr_lookup(_flat_pyramid(pyramid), coords, len(pyramid), radius)


def _convex_upsample(flow, mask, scale=8):
    return ops.convex_upsample(flow, mask, scale)


def _induced_flow(depth, K, G, eps=1e-5):
    flow, vmask = ops.induced_flow(depth, K, G, eps, want_vmask=True)
    return flow, vmask


def _corr_weight(g1, g2, target, depth, sigma):
    return ops.corr_weight(g1, g2, target, depth, sigma)


def _lm_normal_eq(target, weight, depth, K, G):
    return ops.lm_normal_eq(target, weight, depth, K, G)


def _lm_solve_update(H, b, G, ep_lambda=100.0, lm_lambda=1e-4, max_update=1.0):
    Gn, xi, _ = ops.lm_solve_update(H, b, G.reshape(-1, 4, 4), ep_lambda, lm_lambda, max_update)
    return Gn.reshape(G.shape), xi


def _lm_step(target, weight, depth, K, G, num_iters=1, ep_lambda=100.0, lm_lambda=1e-4, max_update=1.0):
    Gn, _, _, xi, _ = ops.lm_step(target, weight, depth, K, G, num_iters, ep_lambda, lm_lambda, max_update)
    return Gn.reshape(G.shape), xi


# ---- fake (meta) implementations: shapes / dtypes only -----------------------------------------------------------------
def _f_corr_pyramid(fmap1, fmap2, levels=4):
    B, _, h, w = fmap1.shape
    return [fmap1.new_empty((B * h * w, 1, hl, wl), dtype=torch.float32) for hl, wl in _level_sizes(h, w, levels)]


def _f_corr_lookup(pyramid, coords, radius=4):
    B, _, h, w = coords.shape
    return coords.new_empty((B, len(pyramid) * (2 * radius + 1) ** 2, h, w), dtype=torch.float32)


def _f_convex_upsample(flow, mask, scale=8):
    B, _, h, w = flow.shape
    return flow.new_empty((B, 2, scale * h, scale * w), dtype=torch.float32)


def _f_induced_flow(depth, K, G, eps=1e-5):
    B, H, W = depth.shape[0], depth.shape[-2], depth.shape[-1]
    return depth.new_empty((B, 2, H, W), dtype=torch.float32), depth.new_empty((B, H, W), dtype=torch.float32)


def _f_corr_weight(g1, g2, target, depth, sigma):
    B, _, H, W = g1.shape
    return g1.new_empty((B, H, W), dtype=torch.float32)


def _f_lm_normal_eq(target, weight, depth, K, G):
    B = depth.shape[0]
    return depth.new_empty((B, 6, 6), dtype=torch.float64), depth.new_empty((B, 6), dtype=torch.float64)


def _f_lm_solve_update(H, b, G, ep_lambda=100.0, lm_lambda=1e-4, max_update=1.0):
    return G.new_empty(G.shape, dtype=torch.float32), G.new_empty((H.shape[0], 6), dtype=torch.float32)


def _f_lm_step(target, weight, depth, K, G, num_iters=1, ep_lambda=100.0, lm_lambda=1e-4, max_update=1.0):
    return G.new_empty(G.shape, dtype=torch.float32), G.new_empty((depth.shape[0], 6), dtype=torch.float32)


def register():
    """Define the `rnnpose` operator library and attach the HIP kernels (CUDA dispatch key) and fake implementations."""
    global _lib
    if _lib is not None:
        return _lib
    lib = torch.library.Library("rnnpose", "DEF")
    impls = {"corr_pyramid": (_corr_pyramid, _f_corr_pyramid), "corr_lookup": (_corr_lookup, _f_corr_lookup),
             "convex_upsample": (_convex_upsample, _f_convex_upsample), "induced_flow": (_induced_flow, _f_induced_flow),
             "corr_weight": (_corr_weight, _f_corr_weight), "lm_normal_eq": (_lm_normal_eq, _f_lm_normal_eq),
             "lm_solve_update": (_lm_solve_update, _f_lm_solve_update), "lm_step": (_lm_step, _f_lm_step)}
    for name, schema in _SCHEMAS.items():
        lib.define(name + schema)
        real, fake = impls[name]
        lib.impl(name, real, "CUDA")
        torch.library.register_fake("rnnpose::" + name, fake, lib=lib)
    _lib = lib
    return lib


register()
