"""BasicUpdateBlock with the reference's module tree and state_dict keys
(thirdparty/raft/update.py:6-14,33-60,79-97,164-188), so `gru_update.pth` / `.tckpt` files load unchanged.

Dense convolutions go to MIOpen through torch (north_star: "MFMA only if the feature-extraction convs
prove the bottleneck"); what is hand-written here is everything around them: the z|r convolutions of each
GRU half are issued as ONE conv over concatenated weights, the hidden/input concat lives in a persistent
(B,384,h,w) buffer that the gate kernels read and update in place, and sigmoid / r*h / tanh /
(1-z)h+zq are two fused HIP kernels per half instead of ~8 ATen launches.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu_(self.conv1(x)))


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
        self._zr_cache = {}

    def _zr(self, sfx):
        """Concatenated z|r weights/bias of one half, rebuilt only when the parameters change."""
        cz, cr = getattr(self, "convz" + sfx), getattr(self, "convr" + sfx)
        key = (cz.weight._version, cr.weight._version, cz.bias._version, cr.bias._version,
               cz.weight.data_ptr(), cr.weight.data_ptr())
        hit = self._zr_cache.get(sfx)
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([cz.weight, cr.weight], 0).detach(), torch.cat([cz.bias, cr.bias], 0).detach())
            self._zr_cache[sfx] = hit
        return hit[1], hit[2]

    def step_inplace(self, hx, rhx, z):
        """hx (B,384,h,w) = [h | x]; updates hx[:, :128] in place.  rhx: same shape with rhx[:,128:] == x."""
        C = self.hidden_dim
        for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
            wzr, bzr = self._zr(sfx)
            zr = F.conv2d(hx, wzr, bzr, padding=pad)                      # update.py:48-49 / :55-56
            ops.gru_gate(zr, hx, z, rhx, C)                               # z = sig(.), rhx[:, :C] = sig(r)*h
            cq = getattr(self, "convq" + sfx)
            q = F.conv2d(rhx, cq.weight, cq.bias, padding=pad)            # :50 / :57
            ops.gru_update(z, q, hx, hx, C)                               # h = (1-z)h + z tanh(q)   :51 / :58

    def forward(self, h, x):
        hx = torch.cat([h, x], dim=1).contiguous()
        rhx = hx.clone()
        z = torch.empty_like(h)
        self.step_inplace(hx, rhx, z)
        return hx[:, :self.hidden_dim].contiguous()


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
        cor = F.relu_(self.convc1(corr))
        cor = F.relu_(self.convc2(cor))
        flo = F.relu_(self.convf1(flow))
        flo = F.relu_(self.convf2(flo))
        out = F.relu_(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


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

    def forward(self, net, inp, corr, flow, upsample=True):
        """-> (net, mask, delta_flow)   (update.py:178-188).  GPU tensors only."""
        if not net.is_cuda:
            raise RuntimeError("BasicUpdateBlock runs on the GPU only (no CPU path in rnnpose_amd)")
        motion = self.encoder(flow, corr)                               # (B,128,h,w)
        hx = torch.cat([net, inp, motion], dim=1)                       # [h | x], x = inp | motion  (:181)
        rhx = hx.clone()
        z = torch.empty_like(net)
        self.gru.step_inplace(hx, rhx, z)
        net = hx[:, :net.shape[1]].contiguous()
        delta_flow = self.flow_head(net)
        mask = self.mask(net)
        mask.mul_(0.25)                                                 # scale mask to balance gradients (:187)
        return net, mask, delta_flow
