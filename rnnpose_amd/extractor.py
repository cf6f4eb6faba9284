"""RAFT BasicEncoder (instance-norm variant) with the reference's module tree / state_dict keys
(thirdparty/raft/extractor.py:6-58,118-232).  SURVEY.md section 8(f1): adjacent to the hot path -- it runs
once per outer iteration -- and stays on MIOpen; kept here so `img_fea_enc.pth` loads unchanged and the
benchmark's timed region matches SURVEY.md section 8(d)."""
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
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


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
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, dim, stride=1):
        layers = (ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride),
                  ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return nn.Sequential(*layers)

    def forward(self, x):
        is_list = isinstance(x, (tuple, list))
        if is_list:
            batch_dim = x[0].shape[0]
            x = torch.cat(x, dim=0)
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        x = self.conv2(x)
        if is_list:
            x = torch.split(x, [batch_dim, batch_dim], dim=0)
        return x
