"""VGG-11/13/16/19 (plain and ``_bn``) and AlexNet on the NHWC device ops.

Two of the families the reference's ``--arch`` switch reaches through ``torchvision.models``
(/root/reference/1.dataparallel.py:23-24,97-102).  ``state_dict()`` keys and shapes are torchvision's:
``features.<i>.weight`` with ``<i>`` counting the ReLU and pooling modules in between, and
``classifier.0/3/6`` (VGG) or ``classifier.1/4/6`` (AlexNet).  The ``_bn`` variants run every
``conv, bn, relu`` triple as the fused op.
"""
from __future__ import annotations

from typing import List, Union

import torch
import torch.nn as tnn

from .. import nn as pnn
from ._base import NHWCModel, flatten_chw, init_like_torchvision_vgg, make_ctor, run_layers

_CFGS = {
    "A": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "B": [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "D": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    "E": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"],
}


def _vgg_features(cfg: List[Union[int, str]], batch_norm: bool) -> tnn.Sequential:
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(pnn.MaxPool2d(2, 2))
            continue
        layers.append(pnn.Conv2d(cin, v, 3, 1, 1))
        if batch_norm:
            layers.append(pnn.BatchNorm2d(v))
        layers.append(pnn.ReLU())
        cin = v
    return tnn.Sequential(*layers)


class VGG(NHWCModel):
    def __init__(self, cfg: str, batch_norm: bool = False, num_classes: int = 1000, dropout: float = 0.5):
        super().__init__()
        self.features = _vgg_features(_CFGS[cfg], batch_norm)
        self.avgpool = pnn.AdaptiveAvgPool2d((7, 7))
        self.classifier = tnn.Sequential(
            pnn.Linear(512 * 7 * 7, 4096), pnn.ReLU(), pnn.Dropout(dropout),
            pnn.Linear(4096, 4096), pnn.ReLU(), pnn.Dropout(dropout),
            pnn.Linear(4096, num_classes))
        init_like_torchvision_vgg(self)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = run_layers(self.features, self.enter(images))
        x = flatten_chw(self.avgpool(x))
        return self.classifier(x)


class AlexNet(NHWCModel):
    def __init__(self, num_classes: int = 1000, dropout: float = 0.5):
        super().__init__()
        self.features = tnn.Sequential(
            pnn.Conv2d(3, 64, 11, 4, 2), pnn.ReLU(), pnn.MaxPool2d(3, 2),
            pnn.Conv2d(64, 192, 5, 1, 2), pnn.ReLU(), pnn.MaxPool2d(3, 2),
            pnn.Conv2d(192, 384, 3, 1, 1), pnn.ReLU(),
            pnn.Conv2d(384, 256, 3, 1, 1), pnn.ReLU(),
            pnn.Conv2d(256, 256, 3, 1, 1), pnn.ReLU(), pnn.MaxPool2d(3, 2))
        self.avgpool = pnn.AdaptiveAvgPool2d((6, 6))
        self.classifier = tnn.Sequential(
            pnn.Dropout(dropout), pnn.Linear(256 * 6 * 6, 4096), pnn.ReLU(),
            pnn.Dropout(dropout), pnn.Linear(4096, 4096), pnn.ReLU(),
            pnn.Linear(4096, num_classes))

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = run_layers(self.features, self.enter(images))
        x = flatten_chw(self.avgpool(x))
        return self.classifier(x)


def _vgg(name: str, cfg: str, bn: bool):
    return make_ctor(name, lambda **kw: VGG(cfg, bn, **kw),
                     f"{name} on the NHWC device ops (torchvision-compatible state_dict).")


vgg11, vgg11_bn = _vgg("vgg11", "A", False), _vgg("vgg11_bn", "A", True)
vgg13, vgg13_bn = _vgg("vgg13", "B", False), _vgg("vgg13_bn", "B", True)
vgg16, vgg16_bn = _vgg("vgg16", "D", False), _vgg("vgg16_bn", "D", True)
vgg19, vgg19_bn = _vgg("vgg19", "E", False), _vgg("vgg19_bn", "E", True)
alexnet = make_ctor("alexnet", lambda **kw: AlexNet(**kw),
                    "AlexNet on the NHWC device ops (torchvision-compatible state_dict).")

__all__ = ["VGG", "AlexNet", "alexnet", "vgg11", "vgg11_bn", "vgg13", "vgg13_bn", "vgg16", "vgg16_bn", "vgg19",
           "vgg19_bn"]
