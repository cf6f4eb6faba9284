"""RAFT BasicEncoder (instance-norm variant) with the reference's module tree / state_dict keys
(thirdparty/raft/extractor.py:6-58,118-232), so `img_fea_enc.pth` loads unchanged.  The modules only hold the
parameters: `BasicEncoder.forward` runs the NHWC encoder engine (rnnpose_amd/engine.py: stem kernel, implicit-GEMM
convolutions, fused instance-norm passes -- all hand-written HIP), SURVEY.md section 8(f1)."""
from __future__ import annotations

import torch
import torch.nn as nn


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn="instance", stride=1):
        super().__init__()
        if norm_fn != "instance":
            raise NotImplementedError("only norm_fn='instance' (the RNNPose configuration) is implemented")
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = nn.InstanceNorm2d(planes)
        self.norm2 = nn.InstanceNorm2d(planes)
        if stride == 1:
            self.downsample = None
        else:
            self.norm3 = nn.InstanceNorm2d(planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        """(B,C,H,W) NCHW -> NCHW (extractor.py:48-58) through the NHWC kernels of the encoder engine."""
        from . import ops
        from .engine import EncoderEngine
        if not x.is_cuda:
            raise RuntimeError("ResidualBlock runs on the GPU only (no CPU path in rnnpose_amd)")
        pc = lambda m: ops.PackedConv(m.weight, m.bias, [m.weight.shape[1]])
        W = {"b.c1": pc(self.conv1), "b.c2": pc(self.conv2)}
        if self.downsample is not None:
            W["b.down"] = pc(self.downsample[0])
        y = EncoderEngine._block(W, "b", self, ops.nchw_to_nhwc(x))
        return ops.nhwc_to_nchw(y)


class BasicEncoder(nn.Module):
    def __init__(self, output_dim=128, norm_fn="instance", dropout=0.0, input_dim=3):
        super().__init__()
        if norm_fn != "instance":
            raise NotImplementedError("only norm_fn='instance' (the RNNPose configuration) is implemented")
        self.norm_fn = norm_fn
        self.norm1 = nn.InstanceNorm2d(64)
        self.conv1 = nn.Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=2)
        self.layer3 = self._make_layer(128, stride=2)
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        self._engine = None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, dim, stride=1):
        layers = (ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride),
                  ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def engine(self):
        if self._engine is None:
            from .engine import EncoderEngine
            self._engine = EncoderEngine(self)
        return self._engine

    @torch.no_grad()
    def forward(self, x):
        """x: (N,3,H,W) or a list/tuple of two such tensors (already normalised, extractor.py:187-232)."""
        is_list = isinstance(x, (tuple, list))
        xs = list(x) if is_list else x
        if not (xs[0] if is_list else xs).is_cuda:
            raise RuntimeError("BasicEncoder runs on the GPU only (no CPU path in rnnpose_amd)")
        out = self.engine()(xs, normalize=False)
        if is_list:
            out = torch.split(out, [t.shape[0] for t in xs], dim=0)
        return out
