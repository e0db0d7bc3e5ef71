"""Inception v3 on the NHWC device ops (an ``--arch`` choice of the reference through
``torchvision.models``, /root/reference/1.dataparallel.py:23-24,97-102).  299x299 inputs upstream; any
input of 75x75 or more works without the auxiliary head.

``state_dict()`` keys are torchvision's (``Conv2d_1a_3x3.conv.weight``, ``Mixed_6b.branch7x7_2.conv.weight``
- a 1x7 filter -, ``AuxLogits.conv1.bn.running_var``, ``fc.weight``).  Every conv is conv + BatchNorm(eps=1e-3)
+ ReLU and runs as the fused op; the factorised 1x7 / 7x1 / 1x3 / 3x1 filters go through the same implicit-GEMM
kernels with rectangular windows.  ``aux_logits`` defaults to False for the reason given in
:mod:`.googlenet`; with ``aux_logits=True`` training mode returns ``(logits, aux_logits)`` as upstream.
"""
from __future__ import annotations

from collections import namedtuple

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ._base import NHWCModel, make_ctor
from .googlenet import BasicConv2d as BC

InceptionOutputs = namedtuple("InceptionOutputs", ["logits", "aux_logits"])


def _cat(ts):
    return torch.cat(ts, dim=3)


class InceptionA(tnn.Module):
    def __init__(self, cin: int, pool_features: int):
        super().__init__()
        self.branch1x1 = BC(cin, 64, 1)
        self.branch5x5_1 = BC(cin, 48, 1)
        self.branch5x5_2 = BC(48, 64, 5, padding=2)
        self.branch3x3dbl_1 = BC(cin, 64, 1)
        self.branch3x3dbl_2 = BC(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BC(96, 96, 3, padding=1)
        self.branch_pool = BC(cin, pool_features, 1)

    def forward(self, x):
        return _cat([self.branch1x1(x), self.branch5x5_2(self.branch5x5_1(x)),
                     self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                     self.branch_pool(ops.avg_pool2d(x, 3, 1, 1))])


class InceptionB(tnn.Module):
    def __init__(self, cin: int):
        super().__init__()
        self.branch3x3 = BC(cin, 384, 3, stride=2)
        self.branch3x3dbl_1 = BC(cin, 64, 1)
        self.branch3x3dbl_2 = BC(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BC(96, 96, 3, stride=2)

    def forward(self, x):
        return _cat([self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                     ops.max_pool2d(x, 3, 2)])


class InceptionC(tnn.Module):
    def __init__(self, cin: int, channels_7x7: int):
        super().__init__()
        c7 = channels_7x7
        self.branch1x1 = BC(cin, 192, 1)
        self.branch7x7_1 = BC(cin, c7, 1)
        self.branch7x7_2 = BC(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7_3 = BC(c7, 192, (7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BC(cin, c7, 1)
        self.branch7x7dbl_2 = BC(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BC(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BC(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BC(c7, 192, (1, 7), padding=(0, 3))
        self.branch_pool = BC(cin, 192, 1)

    def forward(self, x):
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        d = self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))
        d = self.branch7x7dbl_5(self.branch7x7dbl_4(d))
        return _cat([self.branch1x1(x), b7, d, self.branch_pool(ops.avg_pool2d(x, 3, 1, 1))])


class InceptionD(tnn.Module):
    def __init__(self, cin: int):
        super().__init__()
        self.branch3x3_1 = BC(cin, 192, 1)
        self.branch3x3_2 = BC(192, 320, 3, stride=2)
        self.branch7x7x3_1 = BC(cin, 192, 1)
        self.branch7x7x3_2 = BC(192, 192, (1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BC(192, 192, (7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BC(192, 192, 3, stride=2)

    def forward(self, x):
        b7 = self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x)))
        return _cat([self.branch3x3_2(self.branch3x3_1(x)), self.branch7x7x3_4(b7), ops.max_pool2d(x, 3, 2)])


class InceptionE(tnn.Module):
    def __init__(self, cin: int):
        super().__init__()
        self.branch1x1 = BC(cin, 320, 1)
        self.branch3x3_1 = BC(cin, 384, 1)
        self.branch3x3_2a = BC(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3_2b = BC(384, 384, (3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BC(cin, 448, 1)
        self.branch3x3dbl_2 = BC(448, 384, 3, padding=1)
        self.branch3x3dbl_3a = BC(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BC(384, 384, (3, 1), padding=(1, 0))
        self.branch_pool = BC(cin, 192, 1)

    def forward(self, x):
        b3 = self.branch3x3_1(x)
        d = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        return _cat([self.branch1x1(x), self.branch3x3_2a(b3), self.branch3x3_2b(b3), self.branch3x3dbl_3a(d),
                     self.branch3x3dbl_3b(d), self.branch_pool(ops.avg_pool2d(x, 3, 1, 1))])


class InceptionAux(tnn.Module):
    def __init__(self, cin: int, num_classes: int):
        super().__init__()
        self.conv0 = BC(cin, 128, 1)
        self.conv1 = BC(128, 768, 5)
        self.conv1.stddev = 0.01
        self.fc = pnn.Linear(768, num_classes)
        self.fc.stddev = 0.001

    def forward(self, x):
        if x.shape[1] < 17 or x.shape[2] < 17:
            raise ValueError("the auxiliary head needs a 17x17 feature map (299x299 images)")
        x = self.conv1(self.conv0(ops.avg_pool2d(x, 5, 3)))
        return self.fc(ops.global_avg_pool(x))


class Inception3(NHWCModel):
    def __init__(self, num_classes: int = 1000, aux_logits: bool = False, dropout: float = 0.5):
        super().__init__()
        self.aux_logits = aux_logits
        self.Conv2d_1a_3x3 = BC(3, 32, 3, stride=2)
        self.Conv2d_2a_3x3 = BC(32, 32, 3)
        self.Conv2d_2b_3x3 = BC(32, 64, 3, padding=1)
        self.maxpool1 = pnn.MaxPool2d(3, 2)
        self.Conv2d_3b_1x1 = BC(64, 80, 1)
        self.Conv2d_4a_3x3 = BC(80, 192, 3)
        self.maxpool2 = pnn.MaxPool2d(3, 2)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.AuxLogits = InceptionAux(768, num_classes) if aux_logits else None
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.avgpool = pnn.GlobalAvgPool2d()
        self.dropout = pnn.Dropout(dropout)
        self.fc = pnn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, (pnn.Conv2d, tnn.Linear)):
                tnn.init.trunc_normal_(m.weight, mean=0.0, std=getattr(m, "stddev", 0.1), a=-2, b=2)
            elif isinstance(m, pnn.BatchNorm2d):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)

    def forward(self, images: torch.Tensor):
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(self.enter(images))))
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(self.maxpool1(x)))
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(self.maxpool2(x))))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))
        aux = self.AuxLogits(x) if self.AuxLogits is not None and self.training else None
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        x = self.fc(self.dropout(self.avgpool(x)))
        if self.training and self.aux_logits:
            return InceptionOutputs(x, aux)
        return x


inception_v3 = make_ctor("inception_v3", lambda **kw: Inception3(**kw),
                         "Inception v3 on the NHWC device ops (torchvision-compatible state_dict).")

__all__ = ["Inception3", "InceptionOutputs", "inception_v3"]
