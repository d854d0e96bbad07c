"""P3D-ResNet backbone on HIP kernels -- drop-in for the reference's ``backbone`` module.

Same constructor (``P3D19(config=cfg)``), same ``stages()`` surface (three callables on NCDHW tensors)
and the same state-dict keys / OIDHW shapes as backbone.py:117-164, but every op is a fused NDHWC HIP
kernel: conv + folded BatchNorm(eval) + ReLU (+ residual) in one launch; 1x1x1, 1x3x3 and 3x1x1 convs
run on the fp32 MFMA implicit-GEMM path, the C_in = 1 stem on the direct VALU path.
"""
import math

import torch.nn as nn

from . import ops
from .layers import Conv3dParams, frozen_bn
from .ops import ACT_NONE, ACT_RELU


class _Stage(nn.Sequential):
    """nn.Sequential naming ("0", "1", ...) with an NDHWC fast path; NCDHW views at the boundary."""

    def forward_ndhwc(self, x):
        for m in self:
            x = m.forward_ndhwc(x)
        return x

    def forward(self, x):
        return ops.to_ncdhw(self.forward_ndhwc(ops.to_ndhwc(x)))


class Bottleneck(nn.Module):
    """backbone.py:26-114.  ``block`` is 1-based; ST structure cycles A, B, C."""
    expansion = 4

    def __init__(self, inplanes, planes, block, expand=False, stride=1, ST_structure=("A", "B", "C")):
        super().__init__()
        self.stride, self.expand = stride, expand
        self.ST = list(ST_structure)[(block - 1) % len(ST_structure)]
        out_planes = planes * 4 if expand else inplanes
        self.conv1 = Conv3dParams(inplanes, planes, 1, stride=stride)
        self.bn1 = frozen_bn(planes)
        self.conv2 = Conv3dParams(planes, planes, (1, 3, 3), padding=(0, 1, 1))   # conv_S
        self.bn2 = frozen_bn(planes)
        self.conv3 = Conv3dParams(planes, planes, (3, 1, 1), padding=(1, 0, 0))   # conv_T
        self.bn3 = frozen_bn(planes)
        self.conv4 = Conv3dParams(planes, out_planes, 1)
        self.bn4 = frozen_bn(out_planes)
        if expand:
            self.downsample = nn.Sequential(Conv3dParams(inplanes, planes * 4, 1, stride=2), frozen_bn(planes * 4))

    def forward_ndhwc(self, x):
        out = self.conv1(x, ACT_RELU, bn=self.bn1)
        if self.ST == "A":        # S then T
            out = self.conv3(self.conv2(out, ACT_RELU, bn=self.bn2), ACT_RELU, bn=self.bn3)
        elif self.ST == "B":      # T(x) + S(x)
            out = ops.add(self.conv3(out, ACT_RELU, bn=self.bn3), self.conv2(out, ACT_RELU, bn=self.bn2))
        else:                     # y = S(x); y + T(y)
            y = self.conv2(out, ACT_RELU, bn=self.bn2)
            out = ops.add(y, self.conv3(y, ACT_RELU, bn=self.bn3))
        res = x
        if self.expand:
            res = self.downsample[0](x, ACT_NONE, bn=self.downsample[1])
        return self.conv4(out, ACT_RELU, bn=self.bn4, res=res)   # relu(bn4(conv4) + residual)

    def forward(self, x):
        return ops.to_ncdhw(self.forward_ndhwc(ops.to_ndhwc(x)))


class _Stem(_Stage):
    """C1: Conv3d k(kd,7,7) s2 + BN + ReLU + MaxPool3d(2,2) (backbone.py:123-128); children "0" and "1" carry the
    state ("2"/"3" of the reference Sequential are parameter-free)."""

    def __init__(self, input_channel, channels, kd):
        super().__init__(Conv3dParams(input_channel, channels, (kd, 7, 7), stride=2, padding=(kd // 2, 3, 3)),
                         frozen_bn(channels))

    def forward_ndhwc(self, x):
        return ops.maxpool2(self[0](x, ACT_RELU, bn=self[1]))


class P3D(nn.Module):
    def __init__(self, block, layers, input_channel=1, config=None, stem_kd=3):
        super().__init__()
        ch = config.BACKBONE_CHANNELS
        self.inplanes = ch[0]
        self.C1 = _Stem(input_channel, ch[0], stem_kd)
        self.C2 = self._make_layer(block, ch[0], layers[0], stride=2)
        self.C3 = self._make_layer(block, ch[1], layers[1], stride=2)
        for m in self.modules():   # backbone.py:133-139 (overridden later by MaskRCNN.initialize_weights)
            if isinstance(m, Conv3dParams):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        layers = [block(self.inplanes, planes, 1, True, stride)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, i, False) for i in range(2, blocks + 1)]
        return _Stage(*layers)

    def forward(self, x):
        return ops.to_ncdhw(self.C3.forward_ndhwc(self.C2.forward_ndhwc(self.C1.forward_ndhwc(ops.to_ndhwc(x)))))

    def stages(self):
        return [self.C1, self.C2, self.C3]


def P3D19(**kwargs):
    """Two stages of [2, 3] bottlenecks (backbone.py:161-164)."""
    return P3D(Bottleneck, [2, 3], **kwargs)


def P3D35(**kwargs):
    """LiTS fork: [4, 5] bottlenecks and a (5,7,7) stem (LiTS_2017/backbone.py:124,172-176)."""
    kwargs.setdefault("stem_kd", 5)
    return P3D(Bottleneck, [4, 5], **kwargs)
