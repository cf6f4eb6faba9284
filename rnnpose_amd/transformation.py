"""SE3 / SE3Sequence with the reference's interface (geometry/transformation.py:65-320), plus the module-level
helpers of geometry/cholesky.py and geometry/se3.py that the hot path uses.  G is (B,1,4,4) fp32 on the GPU;
all arithmetic runs in the HIP kernels of csrc/lm.hip and csrc/pointwise.hip.

Deliberate differences from the reference (documented in DESIGN.md):
  * batched per-sample semantics are defined for B>1 (the reference loop is batch-1 only);
  * only one frame per pose sequence (K=1, what PoseRefiner uses) is supported;
  * a non-positive-definite damped system makes torch.cholesky raise in the reference; here the update of
    that sample becomes 0 (the NaN->0 guard of cholesky.py:43-44) and `last_info` flags it without a host sync.
"""
from __future__ import annotations

import torch

from . import ops

MIN_DEPTH = 0.1          # geometry/transformation.py:16
LM_LMBDA = 1e-4          # config/default.py:54
EP_LMBDA = 100.0         # config/default.py:55


# ---- geometry/cholesky.py:32-50 and geometry/se3.py:194-209,228-306 as free functions ---------------------
def cholesky_solve_update(H, b, G, max_update=1.0, ep_lambda=0.0, lm_lambda=0.0):
    """(H (..,6,6), b (..,6), G (..,4,4)) -> (G_new, xi) with xi = clamp(nan_to_zero(H^-1 b)) in fp32."""
    sh = G.shape
    Gn, xi, _ = ops.lm_solve_update(H.reshape(-1, 6, 6), b.reshape(-1, 6), G.reshape(-1, 4, 4), ep_lambda, lm_lambda,
                                    max_update)
    return Gn.reshape(sh), xi.reshape(*sh[:-2], 6)


def solve(H, b, max_update=1.0):
    """cholesky.solve: x = clamp(nan_to_zero(chol_solve(H, b)), +-max_update) as fp32."""
    eye = torch.eye(4, device=H.device).expand(H.reshape(-1, 6, 6).shape[0], 4, 4).contiguous()
    _, xi, _ = ops.lm_solve_update(H.reshape(-1, 6, 6), b.reshape(-1, 6), eye, 0.0, 0.0, max_update)
    return xi.reshape(b.shape)


def se3_matrix_expm(upsilon_omega):
    return ops.se3_exp(upsilon_omega)


def se3_matrix_increment(G, upsilon_omega):
    """G <- expm(xi) G   (left increment, se3.py:303-306)"""
    return ops.se3_compose(ops.se3_exp(upsilon_omega).reshape(G.shape), G)


def se3_matrix_inverse(G):
    return ops.se3_inverse(G)


def coords_grid(ref, homogeneous=True):
    """(.., H, W, 3|2) grid of pixel coordinates shaped like the reference's (projective_ops.py:25-44)."""
    H, W = ref.shape[-2:]
    ys, xs = torch.meshgrid(torch.arange(H, device=ref.device, dtype=torch.float32),
                            torch.arange(W, device=ref.device, dtype=torch.float32), indexing="ij")
    parts = [xs, ys, torch.ones_like(xs)] if homogeneous else [xs, ys]
    g = torch.stack(parts, dim=-1)
    return g.reshape([1] * (ref.dim() - 2) + [H, W, -1]).repeat(list(ref.shape[:-2]) + [1, 1, 1])


class SE3:
    def __init__(self, upsilon=None, matrix=None, so3=None, translation=None, eq=None, internal="matrix"):
        if internal != "matrix":
            raise NotImplementedError("only the matrix representation is used by RNNPose")
        self.eq = eq
        self.internal = internal
        if upsilon is not None:
            self.G = se3_matrix_expm(upsilon)
        elif matrix is not None:
            self.G = matrix
        self.last_info = None

    # -- algebra -------------------------------------------------------------------------------------------
    def __mul__(self, other):
        return self.__class__(matrix=ops.se3_compose(self.G, other.G), internal=self.internal)

    def identity_(self):
        shape = self.G.shape
        self.G = torch.eye(4, device=self.G.device).repeat([*shape[:-2], 1, 1])

    def identity(self):
        batch = self.G.shape[0]
        I = torch.eye(4, dtype=self.G.dtype, device=self.G.device).repeat([batch, 1, 1, 1])
        return self.__class__(matrix=I, internal=self.internal, eq=self.eq)

    def increment(self, upsilon):
        return self.__class__(matrix=se3_matrix_increment(self.G, upsilon), internal=self.internal)

    def copy(self, stop_gradients=False):
        return self.__class__(matrix=self.G.detach() if stop_gradients else self.G, internal=self.internal)

    def inv(self):
        return self.__class__(matrix=se3_matrix_inverse(self.G), internal=self.internal)

    def matrix(self, fill=True):
        return self.G

    def shape(self):
        return (self.G.shape[0], self.G.shape[1])

    # -- geometry ------------------------------------------------------------------------------------------
    def _check_frames(self, depth):
        if self.G.dim() != 4 or self.G.shape[1] != 1 or depth.dim() != 4 or depth.shape[1] != 1:
            raise NotImplementedError("rnnpose_amd supports one frame per pose (G (B,1,4,4), depth (B,1,H,W))")

    def transform(self, depth, intrinsics, valid_mask=False, return3d=False):
        """depth (B,1,H,W) (callers pass rendered depth + EPS) -> coords (B,1,H,W,2) [, vmask (B,1,H,W,1)]."""
        if return3d:
            raise NotImplementedError("return3d is unused by the refinement loop")
        self._check_frames(depth)
        uv, vm = ops.induced_flow(depth, intrinsics, self.G, eps=0.0, want_vmask=valid_mask, absolute=True)
        coords = uv.permute(0, 2, 3, 1)[:, None]
        if valid_mask:
            return coords, vm[:, None, :, :, None]
        return coords

    def induced_flow(self, depth, intrinsics, valid_mask=False):
        coords0 = coords_grid(depth, homogeneous=False)
        if valid_mask:
            coords1, vmask = self.transform(depth, intrinsics, valid_mask=True)
            return coords1 - coords0, vmask
        return self.transform(depth, intrinsics) - coords0


class SE3Sequence(SE3):
    """Collection of SE3 objects, one frame per batch element (geometry/transformation.py:217-320)."""

    def __init__(self, upsilon=None, matrix=None, so3=None, translation=None, eq="aijk,ai...k->ai...j",
                 internal="matrix"):
        super().__init__(upsilon, matrix, so3, translation, internal=internal, eq=eq)

    def reprojction_optim(self, target, weight, depth, intrinsics, num_iters=2, depth_img_coords=None,
                          lm_lmbda=LM_LMBDA, ep_lmbda=EP_LMBDA):
        """Damped Gauss-Newton on the weighted re-projection error (transformation.py:265-316).
        target (B,1,H,W,2), weight (B,1,H,W,1), depth (B,1,H,W) [already + EPS] -> new SE3Sequence.
        Like the reference it also overwrites self.G (:310)."""
        if depth_img_coords is not None:
            raise NotImplementedError("depth_img_coords is unused by the refinement loop")
        self._check_frames(depth)
        B, _, H, W = depth.shape
        G, Hm, bv, xi, info = ops.lm_step(target.reshape(B, H, W, 2), weight.reshape(B, H, W), depth, intrinsics,
                                          self.G, num_iters=num_iters, ep_lambda=ep_lmbda, lm_lambda=lm_lmbda,
                                          max_update=1.0, eps=0.0)
        G = G.reshape(B, 1, 4, 4)
        self.G = G
        out = SE3Sequence(matrix=G, internal=self.internal)
        out.last_info = info
        out.last_system = (Hm, bv, xi)
        return out
