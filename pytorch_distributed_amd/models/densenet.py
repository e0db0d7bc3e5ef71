"""DenseNet-121/161/169/201 on the NHWC device ops (an ``--arch`` choice of the reference through
``torchvision.models``, /root/reference/1.dataparallel.py:23-24,97-102).

``state_dict()`` keys are torchvision's (``features.conv0.weight``,
``features.denseblock1.denselayer1.norm1.weight``, ``features.transition1.conv.weight``,
``features.norm5.weight``, ``classifier.weight``).  In channels-last memory the concatenation of a dense
block's feature maps joins the contiguous dimension.  Per layer: BatchNorm+ReLU of the joined map in one
pass, then the 1x1 conv fused with the second BatchNorm+ReLU, then the 3x3 conv.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ._base import NHWCModel, make_ctor


class _DenseLayer(tnn.Module):
    def __init__(self, cin: int, growth_rate: int, bn_size: int, drop_rate: float):
        super().__init__()
        self.norm1 = pnn.BatchNorm2d(cin)
        self.relu1 = pnn.ReLU()
        self.conv1 = pnn.Conv2d(cin, bn_size * growth_rate, 1, 1, 0, bias=False)
        self.norm2 = pnn.BatchNorm2d(bn_size * growth_rate)
        self.relu2 = pnn.ReLU()
        self.conv2 = pnn.Conv2d(bn_size * growth_rate, growth_rate, 3, 1, 1, bias=False)
        self.drop_rate = float(drop_rate)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.norm1(x, None, True)
        y = pnn.ConvBnAct.run(self.conv1, self.norm2, y, relu=True)
        y = self.conv2(y)
        if self.drop_rate > 0:
            y = ops.dropout(y, self.drop_rate, self.training)
        return y


class _DenseBlock(tnn.ModuleDict):
    def __init__(self, num_layers: int, cin: int, bn_size: int, growth_rate: int, drop_rate: float):
        super().__init__()
        for i in range(num_layers):
            self[f"denselayer{i + 1}"] = _DenseLayer(cin + i * growth_rate, growth_rate, bn_size, drop_rate)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer in self.values():
            x = torch.cat([x, layer(x)], dim=3)
        return x


class _Transition(tnn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.norm = pnn.BatchNorm2d(cin)
        self.relu = pnn.ReLU()
        self.conv = pnn.Conv2d(cin, cout, 1, 1, 0, bias=False)
        self.pool = pnn.AvgPool2d(2, 2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.pool(self.conv(self.norm(x, None, True)))


class DenseNet(NHWCModel):
    def __init__(self, growth_rate: int = 32, block_config: Sequence[int] = (6, 12, 24, 16),
                 num_init_features: int = 64, bn_size: int = 4, drop_rate: float = 0.0, num_classes: int = 1000):
        super().__init__()
        f = OrderedDict()
        f["conv0"] = pnn.Conv2d(3, num_init_features, 7, 2, 3, bias=False)
        f["norm0"] = pnn.BatchNorm2d(num_init_features)
        f["relu0"] = pnn.ReLU()
        f["pool0"] = pnn.MaxPool2d(3, 2, 1)
        c = num_init_features
        for i, n in enumerate(block_config):
            f[f"denseblock{i + 1}"] = _DenseBlock(n, c, bn_size, growth_rate, drop_rate)
            c += n * growth_rate
            if i != len(block_config) - 1:
                f[f"transition{i + 1}"] = _Transition(c, c // 2)
                c //= 2
        f["norm5"] = pnn.BatchNorm2d(c)
        self.features = tnn.Sequential(f)
        self.classifier = pnn.Linear(c, num_classes)
        for m in self.modules():
            if isinstance(m, pnn.Conv2d):
                tnn.init.kaiming_normal_(m.weight)
            elif isinstance(m, pnn.BatchNorm2d):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)
            elif isinstance(m, tnn.Linear):
                tnn.init.constant_(m.bias, 0)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        f = self.features
        if f.conv0.out_channels == 64:
            # same stem as the ResNets: layout change, conv, BatchNorm, ReLU and pool as one fused op
            x = ops.stem_conv_bn_relu_pool(images, f.conv0.weight, f.norm0.weight, f.norm0.bias, f.conv0.stride,
                                           f.conv0.padding, f.norm0.state(), self.compute_dtype,
                                           pool=(f.pool0.kernel_size, f.pool0.stride, f.pool0.padding))
        else:
            x = f.pool0(pnn.ConvBnAct.run(f.conv0, f.norm0, self.enter(images), relu=True))
        for name, m in f.named_children():
            if name.startswith(("denseblock", "transition")):
                x = m(x)
        x = f.norm5(x, None, True)
        return self.classifier(ops.global_avg_pool(x))


def _dn(name: str, growth: int, cfg, init: int):
    return make_ctor(name, lambda **kw: DenseNet(growth, cfg, init, **kw),
                     f"{name} on the NHWC device ops (torchvision-compatible state_dict).")


densenet121 = _dn("densenet121", 32, (6, 12, 24, 16), 64)
densenet161 = _dn("densenet161", 48, (6, 12, 36, 24), 96)
densenet169 = _dn("densenet169", 32, (6, 12, 32, 32), 64)
densenet201 = _dn("densenet201", 32, (6, 12, 48, 32), 64)

__all__ = ["DenseNet", "densenet121", "densenet161", "densenet169", "densenet201"]
