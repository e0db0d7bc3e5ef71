"""Model zoo.  ``models.__dict__[arch]()`` / ``models.get_model(arch)`` mirror the
reference's ``torchvision.models.__dict__[args.arch](pretrained=...)`` factory and
``model_names`` its ``--arch`` choices (/root/reference/1.dataparallel.py:23-24,97-102).
"""
from __future__ import annotations

import os

import torch

from .resnet import (ResNet, BasicBlock, Bottleneck, resnet18, resnet34, resnet50, resnet101, resnet152,
                     wide_resnet50_2, wide_resnet101_2, resnext50_32x4d, resnext101_32x8d, resnext101_64x4d)
from .vgg import VGG, AlexNet, alexnet, vgg11, vgg11_bn, vgg13, vgg13_bn, vgg16, vgg16_bn, vgg19, vgg19_bn
from .squeezenet import SqueezeNet, squeezenet1_0, squeezenet1_1
from .densenet import DenseNet, densenet121, densenet161, densenet169, densenet201
from .mobilenet import MobileNetV2, mobilenet_v2
from .shufflenet import (ShuffleNetV2, shufflenet_v2_x0_5, shufflenet_v2_x1_0, shufflenet_v2_x1_5,
                         shufflenet_v2_x2_0)
from .mnasnet import MNASNet, mnasnet0_5, mnasnet0_75, mnasnet1_0, mnasnet1_3
from .googlenet import GoogLeNet, googlenet
from .inception import Inception3, inception_v3
from .mnist import Net, mnist_net
from .reference import torch_resnet, TorchResNet

# Every classification architecture of the torchvision release the reference pins (0.4.0:
# /root/reference/requirements.txt:2), plus resnext101_64x4d.
model_names = sorted(["resnet18", "resnet34", "resnet50", "resnet101", "resnet152", "wide_resnet50_2",
                      "wide_resnet101_2", "resnext50_32x4d", "resnext101_32x8d", "resnext101_64x4d", "alexnet", "vgg11", "vgg11_bn", "vgg13", "vgg13_bn", "vgg16", "vgg16_bn",
                      "vgg19", "vgg19_bn", "squeezenet1_0", "squeezenet1_1", "densenet121", "densenet161",
                      "densenet169", "densenet201", "mobilenet_v2", "shufflenet_v2_x0_5", "shufflenet_v2_x1_0",
                      "shufflenet_v2_x1_5", "shufflenet_v2_x2_0", "mnasnet0_5", "mnasnet0_75", "mnasnet1_0",
                      "mnasnet1_3", "googlenet", "inception_v3", "mnist_net"])


def get_model(arch: str, pretrained: bool = False, **kw):
    if arch not in model_names:
        raise ValueError(f"unknown arch {arch!r}; choices: {', '.join(model_names)}")
    return globals()[arch](pretrained=pretrained, **kw)


def load_pretrained(model, arch: str) -> None:
    """There is no network here: ``--pretrained`` loads ``$PDA_PRETRAINED_DIR/<arch>.pth``
    (a torchvision-format state_dict) if present and raises otherwise."""
    root = os.environ.get("PDA_PRETRAINED_DIR", "")
    path = os.path.join(root, f"{arch}.pth")
    if not root or not os.path.exists(path):
        raise FileNotFoundError(
            f"pretrained weights for {arch} requested but {path!r} does not exist "
            "(set PDA_PRETRAINED_DIR to a directory of torchvision state_dicts; nothing is downloaded)")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(sd.get("state_dict", sd))


__all__ = ["ResNet", "BasicBlock", "Bottleneck", "VGG", "AlexNet", "SqueezeNet", "DenseNet", "MobileNetV2", "ShuffleNetV2", "MNASNet", "GoogLeNet", "Inception3", *model_names,
           "Net", "mnist_net", "torch_resnet", "TorchResNet", "model_names", "get_model", "load_pretrained"]
