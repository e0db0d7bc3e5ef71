"""SqueezeNet 1.0 / 1.1 on the NHWC device ops (an ``--arch`` choice of the reference through
``torchvision.models``, /root/reference/1.dataparallel.py:23-24,97-102).

``state_dict()`` keys are torchvision's (``features.3.squeeze.weight``, ``features.3.expand3x3.bias``,
``classifier.1.weight``).  Channels-last makes the Fire module's concatenation of its two expand
branches a concatenation along the contiguous dimension; the pools use ``ceil_mode`` as upstream.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ._base import NHWCModel, make_ctor, run_layers


class Fire(tnn.Module):
    def __init__(self, inplanes: int, squeeze_planes: int, expand1x1_planes: int, expand3x3_planes: int):
        super().__init__()
        self.inplanes = inplanes
        self.squeeze = pnn.Conv2d(inplanes, squeeze_planes, 1)
        self.squeeze_activation = pnn.ReLU()
        self.expand1x1 = pnn.Conv2d(squeeze_planes, expand1x1_planes, 1)
        self.expand1x1_activation = pnn.ReLU()
        self.expand3x3 = pnn.Conv2d(squeeze_planes, expand3x3_planes, 3, 1, 1)
        self.expand3x3_activation = pnn.ReLU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.squeeze(x, relu=True)
        # relu(cat(a, b)) == cat(relu(a), relu(b)): one activation pass over the joined tensor
        return ops.relu(torch.cat([self.expand1x1(x), self.expand3x3(x)], dim=3))


class SqueezeNet(NHWCModel):
    def __init__(self, version: str = "1_0", num_classes: int = 1000, dropout: float = 0.5):
        super().__init__()
        self.num_classes = num_classes
        pool = lambda: pnn.MaxPool2d(3, 2, ceil_mode=True)  # noqa: E731
        if version == "1_0":
            self.features = tnn.Sequential(
                pnn.Conv2d(3, 96, 7, 2), pnn.ReLU(), pool(),
                Fire(96, 16, 64, 64), Fire(128, 16, 64, 64), Fire(128, 32, 128, 128), pool(),
                Fire(256, 32, 128, 128), Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256),
                pool(), Fire(512, 64, 256, 256))
        elif version == "1_1":
            self.features = tnn.Sequential(
                pnn.Conv2d(3, 64, 3, 2), pnn.ReLU(), pool(),
                Fire(64, 16, 64, 64), Fire(128, 16, 64, 64), pool(),
                Fire(128, 32, 128, 128), Fire(256, 32, 128, 128), pool(),
                Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256), Fire(512, 64, 256, 256))
        else:
            raise ValueError(f"unsupported SqueezeNet version {version!r}: 1_0 or 1_1 expected")
        final_conv = pnn.Conv2d(512, num_classes, 1)
        self.classifier = tnn.Sequential(pnn.Dropout(dropout), final_conv, pnn.ReLU(), pnn.AdaptiveAvgPool2d((1, 1)))
        for m in self.modules():
            if isinstance(m, pnn.Conv2d):
                if m is final_conv:
                    tnn.init.normal_(m.weight, mean=0.0, std=0.01)
                else:
                    tnn.init.kaiming_uniform_(m.weight)
                if m.bias is not None:
                    tnn.init.constant_(m.bias, 0)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = run_layers(self.features, self.enter(images))
        x = run_layers(self.classifier, x)
        return x.reshape(x.shape[0], -1)


squeezenet1_0 = make_ctor("squeezenet1_0", lambda **kw: SqueezeNet("1_0", **kw),
                          "SqueezeNet 1.0 on the NHWC device ops (torchvision-compatible state_dict).")
squeezenet1_1 = make_ctor("squeezenet1_1", lambda **kw: SqueezeNet("1_1", **kw),
                          "SqueezeNet 1.1 on the NHWC device ops (torchvision-compatible state_dict).")

__all__ = ["SqueezeNet", "Fire", "squeezenet1_0", "squeezenet1_1"]
