"""Shared plumbing of the non-residual model families (VGG, AlexNet, SqueezeNet, DenseNet).

The reference accepts every ``torchvision.models`` constructor as ``--arch``
(/root/reference/1.dataparallel.py:23-24,29-33); these families are the ones that are built from dense
convolutions, so they run on the same NHWC device ops as the ResNets.  Parameter names and shapes are
torchvision's, activations are physically ``[N, H, W, C]`` in ``compute_dtype`` and parameters stay
fp32 masters (``amp.initialize`` switches ``compute_dtype`` through :meth:`set_compute_dtype`).
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops


class NHWCModel(tnn.Module):
    """``compute_dtype`` handling and the NCHW loader batch -> NHWC activation entry."""

    def __init__(self):
        super().__init__()
        self.compute_dtype = torch.float32

    def set_compute_dtype(self, dtype: torch.dtype):
        """Activation / MFMA input dtype (fp32, fp16 or bf16); parameters stay fp32 masters."""
        assert dtype in (torch.float32, torch.float16, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    def enter(self, images: torch.Tensor) -> torch.Tensor:
        return images.permute(0, 2, 3, 1).contiguous().to(self.compute_dtype)


def run_layers(layers, x: torch.Tensor) -> torch.Tensor:
    """``nn.Sequential`` semantics with one peephole: ``Conv2d, BatchNorm2d[, ReLU]`` runs as the fused
    conv + BatchNorm (+ ReLU) op, ``BatchNorm2d, ReLU`` as one normalise-and-activate pass and
    ``Conv2d, ReLU`` with the activation inside the conv's bias pass."""
    mods = list(layers)
    i, n = 0, len(mods)
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        if isinstance(m, pnn.Conv2d) and isinstance(nxt, pnn.BatchNorm2d):
            relu = i + 2 < n and isinstance(mods[i + 2], pnn.ReLU)
            x = pnn.ConvBnAct.run(m, nxt, x, relu=relu)
            i += 3 if relu else 2
        elif isinstance(m, pnn.BatchNorm2d) and isinstance(nxt, pnn.ReLU):
            x = m(x, None, True)
            i += 2
        elif isinstance(m, pnn.Conv2d) and isinstance(nxt, pnn.ReLU) and not m._forward_hooks:
            x = m(x, relu=True)   # the activation rides in the conv's bias pass
            i += 2
        else:
            x = m(x)
            i += 1
    return x


def flatten_chw(x: torch.Tensor) -> torch.Tensor:
    """[N,H,W,C] -> [N, C*H*W] in torchvision's (C, H, W) feature order, so that the first classifier
    weight means the same thing as in the stock model."""
    if x.shape[1] == 1 and x.shape[2] == 1:
        return x.reshape(x.shape[0], -1)
    return x.permute(0, 3, 1, 2).reshape(x.shape[0], -1)


def init_like_torchvision_vgg(model: tnn.Module) -> None:
    for m in model.modules():
        if isinstance(m, pnn.Conv2d):
            tnn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                tnn.init.constant_(m.bias, 0)
        elif isinstance(m, pnn.BatchNorm2d):
            tnn.init.constant_(m.weight, 1)
            tnn.init.constant_(m.bias, 0)
        elif isinstance(m, tnn.Linear):
            tnn.init.normal_(m.weight, 0, 0.01)
            tnn.init.constant_(m.bias, 0)


def make_ctor(name: str, build, doc: str):
    def ctor(pretrained: bool = False, **kw):
        model = build(**kw)
        if pretrained:
            from . import load_pretrained
            load_pretrained(model, name)
        return model
    ctor.__name__ = name
    ctor.__doc__ = doc
    return ctor


__all__ = ["NHWCModel", "run_layers", "flatten_chw", "init_like_torchvision_vgg", "make_ctor", "ops"]
