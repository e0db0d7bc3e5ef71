"""MNASNet 0.5 / 0.75 / 1.0 / 1.3 on the NHWC device ops (an ``--arch`` choice of the reference through
``torchvision.models``, /root/reference/1.dataparallel.py:23-24,97-102).

``state_dict()`` keys are torchvision's (``layers.0.weight``, ``layers.8.0.layers.3.weight`` - a depthwise
filter -, ``layers.14.weight``, ``classifier.1.weight``); BatchNorm momentum is upstream's ``1 - 0.9997``.
Expand / project 1x1 convs run fused with their BatchNorm on the MFMA kernels, the 3x3 / 5x5 depthwise
convs on the depthwise kernels.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ._base import NHWCModel, make_ctor, run_layers

_BN_MOMENTUM = 1 - 0.9997


def _bn(c: int) -> pnn.BatchNorm2d:
    return pnn.BatchNorm2d(c, momentum=_BN_MOMENTUM)


class _InvertedResidual(tnn.Module):
    def __init__(self, cin: int, cout: int, kernel_size: int, stride: int, expansion: int):
        super().__init__()
        if stride not in (1, 2) or kernel_size not in (3, 5):
            raise ValueError("stride must be 1 or 2 and kernel_size 3 or 5")
        mid = cin * expansion
        self.apply_residual = cin == cout and stride == 1
        self.layers = tnn.Sequential(
            pnn.Conv2d(cin, mid, 1, bias=False), _bn(mid), pnn.ReLU(),
            pnn.Conv2d(mid, mid, kernel_size, stride, kernel_size // 2, bias=False, groups=mid), _bn(mid), pnn.ReLU(),
            pnn.Conv2d(mid, cout, 1, bias=False), _bn(cout))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = run_layers(list(self.layers)[:6], x)
        return pnn.ConvBnAct.run(self.layers[6], self.layers[7], y, relu=False,
                                 residual=x if self.apply_residual else None)


def _stack(cin, cout, kernel_size, stride, expansion, repeats) -> tnn.Sequential:
    first = _InvertedResidual(cin, cout, kernel_size, stride, expansion)
    rest = [_InvertedResidual(cout, cout, kernel_size, 1, expansion) for _ in range(1, repeats)]
    return tnn.Sequential(first, *rest)


def _round_to_multiple_of(val: float, divisor: int, round_up_bias: float = 0.9) -> int:
    new_val = max(divisor, int(val + divisor / 2) // divisor * divisor)
    return new_val if new_val >= round_up_bias * val else new_val + divisor


def _get_depths(alpha: float):
    return [_round_to_multiple_of(d * alpha, 8) for d in [32, 16, 24, 40, 80, 96, 192, 320]]


class MNASNet(NHWCModel):
    def __init__(self, alpha: float, num_classes: int = 1000, dropout: float = 0.2):
        super().__init__()
        if alpha <= 0.0:
            raise ValueError("alpha should be greater than 0.0")
        d = _get_depths(alpha)
        self.layers = tnn.Sequential(
            pnn.Conv2d(3, d[0], 3, 2, 1, bias=False), _bn(d[0]), pnn.ReLU(),
            pnn.Conv2d(d[0], d[0], 3, 1, 1, bias=False, groups=d[0]), _bn(d[0]), pnn.ReLU(),
            pnn.Conv2d(d[0], d[1], 1, 1, 0, bias=False), _bn(d[1]),
            _stack(d[1], d[2], 3, 2, 3, 3), _stack(d[2], d[3], 5, 2, 3, 3), _stack(d[3], d[4], 5, 2, 6, 3),
            _stack(d[4], d[5], 3, 1, 6, 2), _stack(d[5], d[6], 5, 2, 6, 4), _stack(d[6], d[7], 3, 1, 6, 1),
            pnn.Conv2d(d[7], 1280, 1, 1, 0, bias=False), _bn(1280), pnn.ReLU())
        self.classifier = tnn.Sequential(pnn.Dropout(dropout), pnn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, pnn.Conv2d):
                tnn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, pnn.BatchNorm2d):
                tnn.init.ones_(m.weight)
                tnn.init.zeros_(m.bias)
            elif isinstance(m, tnn.Linear):
                tnn.init.kaiming_uniform_(m.weight, mode="fan_out", nonlinearity="sigmoid")
                tnn.init.zeros_(m.bias)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = run_layers(self.layers, self.enter(images))
        return self.classifier(ops.global_avg_pool(x))


def _mn(name, alpha):
    return make_ctor(name, lambda **kw: MNASNet(alpha, **kw),
                     f"{name} on the NHWC device ops (torchvision-compatible state_dict).")


mnasnet0_5, mnasnet0_75 = _mn("mnasnet0_5", 0.5), _mn("mnasnet0_75", 0.75)
mnasnet1_0, mnasnet1_3 = _mn("mnasnet1_0", 1.0), _mn("mnasnet1_3", 1.3)

__all__ = ["MNASNet", "mnasnet0_5", "mnasnet0_75", "mnasnet1_0", "mnasnet1_3"]
