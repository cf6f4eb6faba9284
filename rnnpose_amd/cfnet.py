"""GRU_CFUpdator / ImageFeaEncoder with the reference's interface (model/CFNet.py:26-173).

GRU_CFUpdator.forward keeps the reference signature
    forward(fmap1, fmap2, iters=1, flow_init=None, upsample=True, test_mode=False, context_fea=None,
            update_corr_fn=True) -> [flow_up (B,2,H,W)] * iters
and its statefulness: the correlation pyramid, the hidden state `net` and the context input `inp` persist
on the module between calls and are rebuilt only when update_corr_fn is True (CFNet.py:115-133).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .corr import CorrBlock, coords_grid
from .extractor import BasicEncoder
from .update import BasicUpdateBlock


class AttrDict(dict):
    """Minimal stand-in for EasyDict (the reference passes `cfg.raft`, an EasyDict)."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _as_args(args):
    if args is None:
        args = {}
    if isinstance(args, dict) and not isinstance(args, AttrDict):
        args = AttrDict(args)
    return args


class ImageFeaEncoder(nn.Module):
    """model/CFNet.py:26-49.  `pretrained` = path of img_fea_enc.pth (the reference loads
    <repo>/weights/img_fea_enc.pth unconditionally); None keeps the random initialisation.
    forward(image1, image2) -> (fmap1, fmap2): both images go through the NHWC encoder engine as one batch on two streams;
    the reference's input normalisation `2*(x/255)-1` (CFNet.py:42-43, applied even to inputs already in [0,1]) is fused
    into the stem kernel's load."""

    def __init__(self, input_dim=3, output_dim=256, pretrained=None):
        super().__init__()
        self.fnet = BasicEncoder(output_dim=output_dim, norm_fn="instance", dropout=False, input_dim=input_dim)
        if pretrained is not None:
            self.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=True)

    def engine(self):
        return self.fnet.engine()

    @torch.no_grad()
    def forward(self, image1, image2):
        if not image1.is_cuda:
            raise RuntimeError("ImageFeaEncoder runs on the GPU only (no CPU path in rnnpose_amd)")
        B = image1.shape[0]
        f = self.engine()([image1, image2], normalize=True)
        return f[:B], f[B:]

    @torch.no_grad()
    def forward_split(self, image1, image2):
        """Same features as ops.SplitTensor pairs (pixel-major fp16 hi|lo written by the output convolution): the operand
        format of the fp16x3 volume build (GRU_CFUpdator.prepare accepts them in place of the NCHW maps)."""
        if not image1.is_cuda:
            raise RuntimeError("ImageFeaEncoder runs on the GPU only (no CPU path in rnnpose_amd)")
        B = image1.shape[0]
        f = self.engine()([image1, image2], normalize=True, split_out=True)
        return f[:B], f[B:]


class GRU_CFUpdator(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = args = _as_args(args)
        self.hidden_dim = 128
        self.context_dim = 128
        args.corr_levels = 4
        args.corr_radius = 4
        if "alternate_corr" not in args:
            args.alternate_corr = False
        if args.alternate_corr:
            # (thirdparty/raft/corr.py:70-98; the reference never enables it, model/CFNet.py:63-64.)  The class exists on its own --
            # rnnpose_amd.corr.AlternateCorrBlock, window features computed on the fly by csrc/corr_alt.hip -- and is measured
            # against the materialised volume (tools/corr_alt_bench.py, profiles/r03_corr_alt.txt); the fused update engine reads
            # the volume, which is faster per iteration at every measured shape.
            raise NotImplementedError("alternate_corr is not wired into the fused update engine; "
                                      "rnnpose_amd.corr.AlternateCorrBlock provides the volume-free lookup on its own")
        self.update_block = BasicUpdateBlock(args, hidden_dim=self.hidden_dim)
        pre = args.get("pretrained_model", None)
        if pre is not None:
            path = pre if isinstance(pre, str) and os.path.exists(pre) else None
            if path is None:
                raise FileNotFoundError(f"pretrained_model={pre!r} not found (expected gru_update.pth)")
            self.load_state_dict(torch.load(path, map_location="cpu"), strict=True)
        self.corr_fn = None
        self._net = None
        self.inp = None
        self.fmap1 = self.fmap2 = None
        # volume build: "f16x3" = fp16 hi/lo split on the fp16 matrix cores (fp32-class accuracy, default);
        # "f32" = exact fp32 MFMA kernel
        self.corr_precision = args.get("corr_precision", "f16x3")
        self._net_in_engine = False

    @property
    def net(self):
        """Hidden state (B,128,h,w); converted from the engine's NHWC buffer on demand."""
        if self._net_in_engine:
            return self.engine().hidden_nchw()
        return self._net

    @net.setter
    def net(self, v):
        self._net = v
        self._net_in_engine = False

    def engine(self):
        """The fused NHWC execution engine of the update block (rnnpose_amd/engine.py)."""
        return self.update_block.engine()

    def initialize_flow(self, img, downsample_rate=8):
        N, _, H, W = img.shape
        c0 = coords_grid(N, H // downsample_rate, W // downsample_rate, device=img.device)
        return c0, c0.clone()

    def upsample_flow(self, flow, mask, upsample_scale=8):
        return ops.convex_upsample(flow, mask, upsample_scale)

    def prepare(self, fmap1, fmap2, context_fea):
        """The update_corr_fn=True branch (CFNet.py:115-133): volume + pyramid, hidden state, context input."""
        self.fmap1 = fmap1.float()           # (NCHW tensors, or ops.SplitTensor pairs from ImageFeaEncoder.forward_split)
        self.fmap2 = fmap2.float()
        self.corr_fn = CorrBlock(self.fmap1, self.fmap2, radius=self.args.corr_radius, reuse=self.corr_fn,
                                 precision=self.corr_precision)
        assert context_fea is not None
        h, w = self.fmap1.shape[-2:]
        self.net, self.inp = ops.context_prep(context_fea, h, w, self.hidden_dim)
        self.engine().load_state(self._net, self.inp)
        self._net_in_engine = True

    def step(self, coords0, coords1, tail=None, need_coords=True, flow_up_out=None):
        """One GRU iteration given low-res coords (CFNet.py:147-168) -> (coords1_new, flow_up).
        tail(b0, b1, flow_up[b0:b1]): optional consumer of the up-sampled flow of images [b0, b1), issued on the stream
        that produced it (the HIP engine runs the two batch halves as two staggered chains)."""
        if not self._net_in_engine:                      # hidden state was assigned from outside
            self.engine().load_state(self._net, self.inp)
            self._net_in_engine = True
        coords1_new, flow_up = self.engine().step(self.corr_fn, coords1, tail=tail, flow_up=flow_up_out)
        return (coords1_new.clone() if need_coords else None), flow_up      # (engine buffer: overwritten by the next step)

    @torch.no_grad()
    def forward(self, fmap1, fmap2, iters=1, flow_init=None, upsample=True, test_mode=False, context_fea=None,
                update_corr_fn=True):
        if update_corr_fn:
            self.prepare(fmap1, fmap2, context_fea)
        if self.corr_fn is None:
            raise RuntimeError("GRU_CFUpdator.forward called with update_corr_fn=False before any volume was built")
        B, _, h, w = self.fmap1.shape
        coords0 = coords_grid(B, h, w, device=self.fmap1.device)
        if flow_init is not None:
            coords1 = ops.flow_to_coords(flow_init, h, w)      # grid + resize(flow_init/ds); input left untouched
        else:
            coords1 = coords0.clone()
        flow_predictions = []
        flow_up = None
        for _ in range(iters):
            coords1, flow_up = self.step(coords0, coords1)
            flow_predictions.append(flow_up)
        if test_mode:
            return coords1 - coords0, flow_up
        return flow_predictions
