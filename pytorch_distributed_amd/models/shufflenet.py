"""ShuffleNetV2 x0.5 / x1.0 / x1.5 / x2.0 on the NHWC device ops (an ``--arch`` choice of the reference
through ``torchvision.models``, /root/reference/1.dataparallel.py:23-24,97-102).

``state_dict()`` keys are torchvision's (``conv1.0.weight``, ``stage2.0.branch1.0.weight``,
``stage3.1.branch2.3.weight``, ``conv5.1.running_mean``, ``fc.weight``).  With channels as the contiguous
dimension the channel split is a slice of the last dimension and the channel shuffle a transpose of its
two factors.  Depthwise 3x3 on the depthwise kernels; pointwise convs on the MFMA kernels where their
channel counts allow it, else on the library GEMM.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ._base import NHWCModel, make_ctor, run_layers


def channel_shuffle(x: torch.Tensor, groups: int) -> torch.Tensor:
    """NHWC channel shuffle: channel ``g * (C/groups) + i`` moves to ``i * groups + g``."""
    n, h, w, c = x.shape
    return x.reshape(n, h, w, groups, c // groups).transpose(3, 4).reshape(n, h, w, c)


def _dw(c: int, stride: int) -> pnn.Conv2d:
    return pnn.Conv2d(c, c, 3, stride, 1, bias=False, groups=c)


class InvertedResidual(tnn.Module):
    def __init__(self, inp: int, oup: int, stride: int):
        super().__init__()
        if not 1 <= stride <= 3:
            raise ValueError("illegal stride value")
        self.stride = stride
        bf = oup // 2
        if stride == 1 and inp != bf << 1:
            raise ValueError(f"stride-1 unit needs inp == oup, got {inp} and {oup}")
        if stride > 1:
            self.branch1 = tnn.Sequential(_dw(inp, stride), pnn.BatchNorm2d(inp),
                                          pnn.Conv2d(inp, bf, 1, 1, 0, bias=False), pnn.BatchNorm2d(bf), pnn.ReLU())
        else:
            self.branch1 = tnn.Sequential()
        self.branch2 = tnn.Sequential(
            pnn.Conv2d(inp if stride > 1 else bf, bf, 1, 1, 0, bias=False), pnn.BatchNorm2d(bf), pnn.ReLU(),
            _dw(bf, stride), pnn.BatchNorm2d(bf),
            pnn.Conv2d(bf, bf, 1, 1, 0, bias=False), pnn.BatchNorm2d(bf), pnn.ReLU())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.stride == 1:
            half = x.shape[3] // 2
            out = torch.cat((x[..., :half], run_layers(self.branch2, x[..., half:].contiguous())), dim=3)
        else:
            out = torch.cat((run_layers(self.branch1, x), run_layers(self.branch2, x)), dim=3)
        return channel_shuffle(out, 2)


class ShuffleNetV2(NHWCModel):
    def __init__(self, stages_repeats, stages_out_channels, num_classes: int = 1000):
        super().__init__()
        if len(stages_repeats) != 3 or len(stages_out_channels) != 5:
            raise ValueError("expected 3 stage repeat counts and 5 channel counts")
        cin, cout = 3, stages_out_channels[0]
        self.conv1 = tnn.Sequential(pnn.Conv2d(cin, cout, 3, 2, 1, bias=False), pnn.BatchNorm2d(cout), pnn.ReLU())
        cin = cout
        self.maxpool = pnn.MaxPool2d(3, 2, 1)
        for name, repeats, cout in zip(("stage2", "stage3", "stage4"), stages_repeats, stages_out_channels[1:]):
            seq = [InvertedResidual(cin, cout, 2)] + [InvertedResidual(cout, cout, 1) for _ in range(repeats - 1)]
            setattr(self, name, tnn.Sequential(*seq))
            cin = cout
        cout = stages_out_channels[-1]
        self.conv5 = tnn.Sequential(pnn.Conv2d(cin, cout, 1, 1, 0, bias=False), pnn.BatchNorm2d(cout), pnn.ReLU())
        self.fc = pnn.Linear(cout, num_classes)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = self.maxpool(run_layers(self.conv1, self.enter(images)))
        x = self.stage4(self.stage3(self.stage2(x)))
        x = run_layers(self.conv5, x)
        return self.fc(ops.global_avg_pool(x))


def _sn(name, repeats, channels):
    return make_ctor(name, lambda **kw: ShuffleNetV2(repeats, channels, **kw),
                     f"{name} on the NHWC device ops (torchvision-compatible state_dict).")


shufflenet_v2_x0_5 = _sn("shufflenet_v2_x0_5", [4, 8, 4], [24, 48, 96, 192, 1024])
shufflenet_v2_x1_0 = _sn("shufflenet_v2_x1_0", [4, 8, 4], [24, 116, 232, 464, 1024])
shufflenet_v2_x1_5 = _sn("shufflenet_v2_x1_5", [4, 8, 4], [24, 176, 352, 704, 1024])
shufflenet_v2_x2_0 = _sn("shufflenet_v2_x2_0", [4, 8, 4], [24, 244, 488, 976, 2048])

__all__ = ["ShuffleNetV2", "channel_shuffle", "shufflenet_v2_x0_5", "shufflenet_v2_x1_0", "shufflenet_v2_x1_5",
           "shufflenet_v2_x2_0"]
