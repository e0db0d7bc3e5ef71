"""MobileNetV2 on the NHWC device ops (an ``--arch`` choice of the reference through ``torchvision.models``,
/root/reference/1.dataparallel.py:23-24,97-102).

``state_dict()`` keys are torchvision's (``features.0.0.weight``, ``features.2.conv.1.0.weight`` - the
depthwise ``[C, 1, 3, 3]`` filter -, ``features.18.1.running_mean``, ``classifier.1.weight``).  The 1x1
expand / project convolutions run as fused conv + BatchNorm on the MFMA kernels, the 3x3 depthwise
convolution on the vector-memory depthwise kernels, ReLU6 folded into the BatchNorm passes as a clipped ReLU.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ._base import NHWCModel, make_ctor


def _make_divisible(v: float, divisor: int = 8) -> int:
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class ConvBNReLU(tnn.Sequential):
    """``conv -> BatchNorm -> ReLU6`` with torchvision's child names 0 / 1 / 2."""

    def __init__(self, cin: int, cout: int, kernel_size: int = 3, stride: int = 1, groups: int = 1):
        super().__init__(pnn.Conv2d(cin, cout, kernel_size, stride, (kernel_size - 1) // 2, bias=False, groups=groups),
                         pnn.BatchNorm2d(cout), pnn.ReLU6())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # ReLU6 rides in the BatchNorm passes (forward clip, backward mask): no activation kernel of its own
        return pnn.ConvBnAct.run(self[0], self[1], x, relu=6.0)


class InvertedResidual(tnn.Module):
    def __init__(self, inp: int, oup: int, stride: int, expand_ratio: int):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU(inp, hidden, kernel_size=1))
        layers += [ConvBNReLU(hidden, hidden, stride=stride, groups=hidden),
                   pnn.Conv2d(hidden, oup, 1, 1, 0, bias=False), pnn.BatchNorm2d(oup)]
        self.conv = tnn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = x
        for m in list(self.conv)[:-2]:
            y = m(y)
        # linear bottleneck: project conv + BatchNorm, the skip connection added in the same pass
        return pnn.ConvBnAct.run(self.conv[-2], self.conv[-1], y, relu=False,
                                 residual=x if self.use_res_connect else None)


class MobileNetV2(NHWCModel):
    def __init__(self, num_classes: int = 1000, width_mult: float = 1.0, dropout: float = 0.2):
        super().__init__()
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
                   [6, 320, 1, 1]]
        cin = _make_divisible(32 * width_mult)
        self.last_channel = _make_divisible(1280 * max(1.0, width_mult))
        feats = [ConvBNReLU(3, cin, stride=2)]
        for t, c, n, s in setting:
            cout = _make_divisible(c * width_mult)
            for i in range(n):
                feats.append(InvertedResidual(cin, cout, s if i == 0 else 1, t))
                cin = cout
        feats.append(ConvBNReLU(cin, self.last_channel, kernel_size=1))
        self.features = tnn.Sequential(*feats)
        self.classifier = tnn.Sequential(pnn.Dropout(dropout), pnn.Linear(self.last_channel, num_classes))
        for m in self.modules():
            if isinstance(m, pnn.Conv2d):
                tnn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, pnn.BatchNorm2d):
                tnn.init.ones_(m.weight)
                tnn.init.zeros_(m.bias)
            elif isinstance(m, tnn.Linear):
                tnn.init.normal_(m.weight, 0, 0.01)
                tnn.init.zeros_(m.bias)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = self.features(self.enter(images))
        return self.classifier(ops.global_avg_pool(x))


mobilenet_v2 = make_ctor("mobilenet_v2", lambda **kw: MobileNetV2(**kw),
                         "MobileNetV2 on the NHWC device ops (torchvision-compatible state_dict).")

__all__ = ["MobileNetV2", "InvertedResidual", "ConvBNReLU", "mobilenet_v2"]
