"""BasicUpdateBlock with the reference's module tree and state_dict keys
(thirdparty/raft/update.py:6-14,33-60,79-97,164-188), so `gru_update.pth` / `.tckpt` files load unchanged.

The modules only HOLD the parameters; every forward runs on the hand-written HIP kernels:
  * `BasicUpdateBlock.forward(net, inp, corr, flow)` (the reference's boundary, NCHW in / NCHW out) executes the fused
    NHWC schedule of rnnpose_amd/engine.py (one implicit-GEMM launch per convolution, GRU gates / ReLU / bias in the
    epilogues, virtual concats) between layout transposes;
  * the sub-module forwards (`FlowHead`, `SepConvGRU`, `BasicMotionEncoder`) replay the reference's literal call
    sequence with `ops.conv2d_nchw` (the same implicit-GEMM kernel behind an NCHW facade) and the two GRU gate kernels.
There is no torch/MIOpen convolution in this package (the A/B comparison against MIOpen lives in tools/conv_bench.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def _conv(m: nn.Conv2d, x, relu=False):
    return ops.conv2d_nchw(m.weight, m.bias, x, relu=relu)


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)

    def forward(self, x):
        return _conv(self.conv2, _conv(self.conv1, x, relu=True))                     # update.py:13-14


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.hidden_dim = hidden_dim
        c = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))

    def forward(self, h, x):
        """update.py:46-60, literal order: z, r from [h|x]; q from [r*h|x]; h = (1-z)h + zq; horizontal then vertical."""
        C = self.hidden_dim
        hx = torch.cat([h, x], dim=1).contiguous()
        rhx = hx.clone()
        z = torch.empty_like(hx[:, :C]).contiguous()
        for sfx in ("1", "2"):
            cz, cr, cq = (getattr(self, n + sfx) for n in ("convz", "convr", "convq"))
            zr = torch.cat([_conv(cz, hx), _conv(cr, hx)], dim=1)                     # pre-activations of z | r
            ops.gru_gate(zr, hx, z, rhx, C)                                           # z = sig(.), rhx[:, :C] = sig(r) * h
            ops.gru_update(z, _conv(cq, rhx), hx, hx, C)                              # h = (1-z) h + z tanh(q)
        return hx[:, :C].contiguous()


class BasicMotionEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        cor = _conv(self.convc2, _conv(self.convc1, corr, relu=True), relu=True)      # update.py:89-90
        flo = _conv(self.convf2, _conv(self.convf1, flow, relu=True), relu=True)      # :91-92
        out = _conv(self.conv, torch.cat([cor, flo], dim=1), relu=True)               # :94-96
        return torch.cat([out, flow], dim=1)                                          # :97


class BasicUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dim=128, input_dim=128, downsample_scale=8):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(
            nn.Conv2d(128, 256, 3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(256, downsample_scale * downsample_scale * 9, 1, padding=0))
        self._engine = None

    def engine(self):
        if self._engine is None:
            from .engine import UpdateEngine
            self._engine = UpdateEngine(self)
        return self._engine

    @torch.no_grad()
    def forward(self, net, inp, corr, flow, upsample=True):
        """-> (net, mask, delta_flow)   (update.py:178-188).  GPU tensors only."""
        if not net.is_cuda:
            raise RuntimeError("BasicUpdateBlock runs on the GPU only (no CPU path in rnnpose_amd)")
        return self.engine().forward_nchw(net, inp, corr, flow)
