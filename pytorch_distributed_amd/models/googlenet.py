"""GoogLeNet (Inception v1) on the NHWC device ops (an ``--arch`` choice of the reference through
``torchvision.models``, /root/reference/1.dataparallel.py:23-24,97-102).

``state_dict()`` keys are torchvision's (``conv1.conv.weight``, ``inception3a.branch2.1.bn.running_mean``,
``aux1.fc1.weight``, ``fc.weight``), including upstream's 3x3 kernel in the "5x5" branch and BatchNorm
``eps=0.001``.  Channels-last makes the concatenation of the four branches a join of the contiguous
dimension.  ``aux_logits`` defaults to False here: the reference's training loop feeds the model output
straight into the loss (/root/reference/2.distributed.py:213-214), which only works with a single tensor;
with ``aux_logits=True`` training mode returns ``(logits, aux2, aux1)`` as upstream.
"""
from __future__ import annotations

from collections import namedtuple

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ._base import NHWCModel, flatten_chw, make_ctor

GoogLeNetOutputs = namedtuple("GoogLeNetOutputs", ["logits", "aux_logits2", "aux_logits1"])


class BasicConv2d(tnn.Module):
    """conv (no bias) -> BatchNorm(eps=1e-3) -> ReLU as one fused op where the MFMA kernels tile the layer."""

    def __init__(self, cin: int, cout: int, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = pnn.Conv2d(cin, cout, kernel_size, stride, padding, bias=False)
        self.bn = pnn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return pnn.ConvBnAct.run(self.conv, self.bn, x, relu=True)


class Inception(tnn.Module):
    def __init__(self, cin, ch1x1, ch3x3red, ch3x3, ch5x5red, ch5x5, pool_proj):
        super().__init__()
        self.branch1 = BasicConv2d(cin, ch1x1, 1)
        self.branch2 = tnn.Sequential(BasicConv2d(cin, ch3x3red, 1), BasicConv2d(ch3x3red, ch3x3, 3, padding=1))
        self.branch3 = tnn.Sequential(BasicConv2d(cin, ch5x5red, 1), BasicConv2d(ch5x5red, ch5x5, 3, padding=1))
        self.branch4 = tnn.Sequential(pnn.MaxPool2d(3, 1, 1, ceil_mode=True), BasicConv2d(cin, pool_proj, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.cat([self.branch1(x), self.branch2(x), self.branch3(x), self.branch4(x)], dim=3)


class InceptionAux(tnn.Module):
    def __init__(self, cin: int, num_classes: int, dropout: float = 0.7):
        super().__init__()
        self.conv = BasicConv2d(cin, 128, 1)
        self.fc1 = pnn.Linear(2048, 1024)
        self.fc2 = pnn.Linear(1024, num_classes)
        self.dropout = pnn.Dropout(dropout)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.conv(ops.adaptive_avg_pool2d(x, (4, 4)))
        x = ops.relu(self.fc1(flatten_chw(x)))
        return self.fc2(self.dropout(x))


class GoogLeNet(NHWCModel):
    def __init__(self, num_classes: int = 1000, aux_logits: bool = False, dropout: float = 0.2,
                 dropout_aux: float = 0.7):
        super().__init__()
        self.aux_logits = aux_logits
        pool = lambda k=3: pnn.MaxPool2d(k, 2, ceil_mode=True)  # noqa: E731
        self.conv1 = BasicConv2d(3, 64, 7, 2, 3)
        self.maxpool1 = pool()
        self.conv2 = BasicConv2d(64, 64, 1)
        self.conv3 = BasicConv2d(64, 192, 3, padding=1)
        self.maxpool2 = pool()
        self.inception3a = Inception(192, 64, 96, 128, 16, 32, 32)
        self.inception3b = Inception(256, 128, 128, 192, 32, 96, 64)
        self.maxpool3 = pool()
        self.inception4a = Inception(480, 192, 96, 208, 16, 48, 64)
        self.inception4b = Inception(512, 160, 112, 224, 24, 64, 64)
        self.inception4c = Inception(512, 128, 128, 256, 24, 64, 64)
        self.inception4d = Inception(512, 112, 144, 288, 32, 64, 64)
        self.inception4e = Inception(528, 256, 160, 320, 32, 128, 128)
        self.maxpool4 = pool(2)
        self.inception5a = Inception(832, 256, 160, 320, 32, 128, 128)
        self.inception5b = Inception(832, 384, 192, 384, 48, 128, 128)
        if aux_logits:
            self.aux1 = InceptionAux(512, num_classes, dropout_aux)
            self.aux2 = InceptionAux(528, num_classes, dropout_aux)
        else:
            self.aux1 = self.aux2 = None
        self.avgpool = pnn.GlobalAvgPool2d()
        self.dropout = pnn.Dropout(dropout)
        self.fc = pnn.Linear(1024, num_classes)
        for m in self.modules():
            if isinstance(m, (pnn.Conv2d, tnn.Linear)):
                tnn.init.trunc_normal_(m.weight, mean=0.0, std=0.01, a=-2, b=2)
            elif isinstance(m, pnn.BatchNorm2d):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)

    def forward(self, images: torch.Tensor):
        x = self.maxpool1(self.conv1(self.enter(images)))
        x = self.maxpool2(self.conv3(self.conv2(x)))
        x = self.maxpool3(self.inception3b(self.inception3a(x)))
        x = self.inception4a(x)
        aux1 = self.aux1(x) if self.aux1 is not None and self.training else None
        x = self.inception4d(self.inception4c(self.inception4b(x)))
        aux2 = self.aux2(x) if self.aux2 is not None and self.training else None
        x = self.maxpool4(self.inception4e(x))
        x = self.inception5b(self.inception5a(x))
        x = self.fc(self.dropout(self.avgpool(x)))
        if self.training and self.aux_logits:
            return GoogLeNetOutputs(x, aux2, aux1)
        return x


googlenet = make_ctor("googlenet", lambda **kw: GoogLeNet(**kw),
                      "GoogLeNet on the NHWC device ops (torchvision-compatible state_dict).")

__all__ = ["GoogLeNet", "GoogLeNetOutputs", "Inception", "InceptionAux", "BasicConv2d", "googlenet"]
