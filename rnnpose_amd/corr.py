"""CorrBlock with the reference's interface (thirdparty/raft/corr.py:12-67), backed by the HIP kernels
corr_pyramid (build, once per outer iteration) and corr_lookup (per GRU step)."""
from __future__ import annotations

import torch

from . import ops


def coords_grid(batch, ht, wd, device=None):
    """(B,2,ht,wd), channel 0 = x, channel 1 = y   (thirdparty/raft/utils/utils.py:74-77)"""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device, dtype=torch.float32),
                            torch.arange(wd, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack([xs, ys], dim=0)[None].repeat(batch, 1, 1, 1)


class CorrBlock:
    """corr_fn = CorrBlock(fmap1, fmap2, num_levels=4, radius=4); corr = corr_fn(coords)

    `corr_pyramid` is the list of (B*h*w, 1, h_l, w_l) levels like the reference's attribute (ops.PyramidLevels: levels 1.. are
    views into the one device buffer a single kernel launch fills; level 0, which the buffer holds j-patch-major for the store and
    lookup kernels, is un-blocked into a copy when it is asked for)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, downsample_rate=1, reuse=None, precision="f32"):
        if downsample_rate != 1:
            # the reference branch is broken (nn.MaxPool2d called with a tensor, corr.py:20-24) and never taken
            raise NotImplementedError("downsample_rate != 1 is not supported (dead code in the reference)")
        if radius != 4:
            raise NotImplementedError("only radius=4 (the reference's value) is implemented")
        self.num_levels = num_levels
        self.radius = radius
        # reuse: a previous CorrBlock whose device buffer may be overwritten (same shapes) -- keeps addresses stable
        # precision: "f32" = fp32 MFMA kernel; "f16x3" = fp16 hi/lo split on the fp16 matrix cores (fp32-class accuracy,
        # ~3x faster build: HBM-write-bound instead of MFMA-bound)
        out = None if reuse is None else reuse._buf
        if isinstance(fmap1, ops.SplitTensor):      # operands already split by their producer (encoder output convolution)
            if precision != "f16x3":
                raise ValueError("split feature maps feed the fp16x3 volume kernel only")
            self._buf, self.corr_pyramid = ops.corr_pyramid_split(fmap1, fmap2, num_levels, out=out)
        else:
            self._buf, self.corr_pyramid = ops.corr_pyramid(fmap1.float(), fmap2.float(), num_levels, out=out, precision=precision)

    def __call__(self, coords):
        return ops.corr_lookup(self._buf, coords, self.num_levels, self.radius)

    @staticmethod
    def corr(fmap1, fmap2):
        """Level 0 only, shaped (B,h,w,1,h,w) like CorrBlock.corr (corr.py:59-67)."""
        B, _, h, w = fmap1.shape
        _, views = ops.corr_pyramid(fmap1.float(), fmap2.float(), 1)
        return views[0].view(B, h, w, 1, h, w)


class AlternateCorrBlock:
    """The reference's volume-free variant (thirdparty/raft/corr.py:70-98): same constructor and call as CorrBlock, but no
    all-pairs volume is built -- every call computes the 9x9 windows of the 4 levels on the fly from fmap1 and the pooled
    fmap2 pyramid (csrc/corr_alt.hip).  Equal to CorrBlock up to fp32 summation order.  Kept as a MEASURED alternative
    (tools/corr_alt_bench.py, profiles/r03_corr_alt.txt): per iteration it is slower than amortised build + lookup at the bench
    shape, so the fused engine uses the materialised volume (GRU_CFUpdator rejects alternate_corr=True, like before)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        if radius != 4:
            raise NotImplementedError("only radius=4 (the reference's value) is implemented")
        self.num_levels = num_levels
        self.radius = radius
        self.f1 = ops.nchw_to_nhwc(fmap1.float())
        self.f2 = ops.nchw_to_nhwc(fmap2.float())
        self.pooled = ops.fmap_pyramid(self.f2, num_levels)

    def __call__(self, coords):
        out = ops.corr_alt_lookup(self.f1, self.f2, self.pooled, coords, self.num_levels, self.radius)
        return ops.nhwc_to_nchw(out)
