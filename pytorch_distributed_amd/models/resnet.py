"""ResNet-18/34/50/101/152, Wide-ResNet-50-2/101-2 and ResNeXt-50 32x4d / 101 32x8d / 101 64x4d
(torchvision v1.5 topology: stride on the 3x3 of a bottleneck) built on the fused NHWC ops.

``state_dict()`` keys, shapes and dtypes are those of torchvision's ResNet
(``conv1.weight``, ``layer1.0.bn1.running_mean``, ``layer2.0.downsample.0.weight``,
``fc.bias`` ...), so checkpoints move freely between this model and the stock one.
The reference builds these through ``models.__dict__[args.arch]()`` with the
default 1000-class head even for CIFAR-10 (/root/reference/1.dataparallel.py:97-102,
/root/reference/2.distributed.py:100-105).

Forward takes the loader's NCHW batch (fp32, fp16 or uint8).  On the GPU the stem
op does the layout change, normalisation and cast in one kernel; everything after
it runs NHWC in ``compute_dtype`` (fp32 by default; ``model.set_compute_dtype(torch.float16)``
or ``amp.initialize`` for mixed precision) with fp32 master parameters.
"""
from __future__ import annotations

import os
from typing import List, Optional, Type, Union

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops
from ..ops.fused import BNBackwardLink, ShortcutLink


# bit i = the i-th inner hand-off of a block may skip its normalisation pass.  Default: only the hand-off into a
# block's last 1x1 conv (bit 1).  A 3x3 conv that normalises on load is held to the register-staged tiles (padding
# has to stay zero, so the transform needs per-chunk validity) and measured 0.65 ms per ResNet-50 step slower than
# storing the activation and running the LDS-DMA ring kernels on it.
_DEFER_MASK = int(os.environ.get("PDA_DEFER_INNER_MASK", "2"))


def _conv3x3(cin, cout, stride=1, groups=1):
    return pnn.Conv2d(cin, cout, 3, stride, 1, bias=False, groups=groups)


def _conv1x1(cin, cout, stride=1):
    return pnn.Conv2d(cin, cout, 1, stride, 0, bias=False)


def _premasked_out(block) -> bool:
    # a forward hook can hand the block's output to a consumer the model does not know about, whose
    # gradient would arrive unmasked: keep the block's own ReLU mask then
    return block.premasked_out and not block._forward_hooks


def _init_fusion_state(block) -> None:
    block.premask_in = False     # see ResNet.fuse_relu_backward
    block.premasked_out = False
    block.fuse_inner = True      # backward hand-off between the convs inside the block
    block.defer_inner = (False, False)   # per inner hand-off: consumer normalises on load (set by the block)
    block._link_in = None        # per-forward: set by ResNet.features for the first conv
    block._link_out = None       # per-forward: read (and cleared) by ResNet.features
    block.defer_out = False      # the next block's first conv can finish this block (see ResNet.fuse_relu_backward)
    block._defer_out_now = False  # per-forward: set by ResNet.features when that next block really runs next


def _links(block, n_inner: int):
    """Per-forward backward links of a block: (link from the previous block, inner links, link to the
    next block, whether the block's output gradient arrives masked)."""
    in_link, block._link_in = (block._link_in if block.premask_in else None), None
    # inner hand-offs also skip the producer's normalisation pass where the consuming conv can apply it on load
    inner = tuple(BNBackwardLink(defer=block.defer_inner[i] and bool((_DEFER_MASK >> i) & 1)) if block.fuse_inner else None
                  for i in range(n_inner))
    pm = _premasked_out(block)
    block._link_out = BNBackwardLink(defer_out=block.defer_out and block._defer_out_now) if pm else None
    return in_link, inner, block._link_out, pm


class BasicBlock(tnn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample: Optional[tnn.Module] = None, base_width: int = 64,
                 groups: int = 1):
        super().__init__()
        if base_width != 64 or groups != 1:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = pnn.BatchNorm2d(planes)
        self.relu = pnn.ReLU()
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = pnn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        _init_fusion_state(self)
        self.defer_inner = (pnn.ConvBnAct.takes_fused_path(self.conv2, torch.float16),)

    def forward(self, x):
        # `identity` is an alias of x handed back by conv1's op, so the gradient of the skip branch
        # is added inside conv1's dgrad epilogue rather than by a separate kernel
        in_link, l1, lo, pm = _links(self, 1)
        out, identity = pnn.ConvBnAct.run(self.conv1, self.bn1, x, relu=True, passthrough=True,
                                          mask_dx=self.premask_in, in_link=in_link,
                                          dout_premasked=l1[0] is not None, out_link=l1[0])
        sl = None
        if self.downsample is not None:
            # the shortcut's BatchNorm is applied by the block's last op, in the pass that adds it
            sl = ShortcutLink()
            identity = pnn.ConvBnAct.run(self.downsample[0], self.downsample[1], identity, relu=False, res_out=sl)
        return pnn.ConvBnAct.run(self.conv2, self.bn2, out, relu=True, residual=identity, in_link=l1[0],
                                 dout_premasked=pm, out_link=lo, res_in=sl)


class Bottleneck(tnn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample: Optional[tnn.Module] = None, base_width: int = 64,
                 groups: int = 1):
        super().__init__()
        # wide_resnet*_2: twice the inner (3x3) width, same block output; resnext: `groups` groups of
        # `base_width * planes / 64` channels in the 3x3
        width = planes * base_width // 64 * groups
        self.conv1 = _conv1x1(inplanes, width)
        self.bn1 = pnn.BatchNorm2d(width)
        self.conv2 = _conv3x3(width, width, stride, groups)
        self.bn2 = pnn.BatchNorm2d(width)
        self.conv3 = _conv1x1(width, planes * self.expansion)
        self.bn3 = pnn.BatchNorm2d(planes * self.expansion)
        self.relu = pnn.ReLU()
        self.downsample = downsample
        self.stride = stride
        _init_fusion_state(self)
        self.defer_inner = (pnn.ConvBnAct.takes_fused_path(self.conv2, torch.float16),
                            pnn.ConvBnAct.takes_fused_path(self.conv3, torch.float16))

    def forward(self, x):
        in_link, (l1, l2), lo, pm = _links(self, 2)
        sl = ShortcutLink() if self.downsample is not None else None
        out, identity = pnn.ConvBnAct.run(self.conv1, self.bn1, x, relu=True, passthrough=True,
                                          mask_dx=self.premask_in, in_link=in_link,
                                          dout_premasked=l1 is not None, out_link=l1, alias_link=sl)
        out = pnn.ConvBnAct.run(self.conv2, self.bn2, out, relu=True, in_link=l1,
                                dout_premasked=l2 is not None, out_link=l2)
        if self.downsample is not None:
            # the shortcut's BatchNorm is applied by the block's last op, in the pass that adds it; a strided
            # shortcut's input gradient travels back to conv1's dgrad epilogue in compact form
            identity = pnn.ConvBnAct.run(self.downsample[0], self.downsample[1], identity, relu=False, res_out=sl)
        return pnn.ConvBnAct.run(self.conv3, self.bn3, out, relu=True, residual=identity, in_link=l2,
                                 dout_premasked=pm, out_link=lo, res_in=sl)


class ResNet(tnn.Module):
    def __init__(self, block: Type[Union[BasicBlock, Bottleneck]], layers: List[int], num_classes: int = 1000,
                 zero_init_residual: bool = False, input_mean=None, input_std=None, input_scale: float = 1.0,
                 width_per_group: int = 64, groups: int = 1):
        super().__init__()
        self.inplanes = 64
        self.base_width = width_per_group
        self.groups = groups
        self.conv1 = pnn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = pnn.BatchNorm2d(64)
        self.relu = pnn.ReLU()
        self.maxpool = pnn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = pnn.GlobalAvgPool2d()
        self.fc = pnn.Linear(512 * block.expansion, num_classes)
        self.compute_dtype = torch.float32
        # optional on-device input normalisation (what the reference's data_prefetcher does
        # on its side stream, /root/reference/4.apex_distributed.py:85-86,121-122)
        if input_mean is not None:
            self.register_buffer("_in_mean", torch.tensor(input_mean, dtype=torch.float32), persistent=False)
            self.register_buffer("_in_inv_std", 1.0 / torch.tensor(input_std, dtype=torch.float32), persistent=False)
        else:
            self._in_mean = None
            self._in_inv_std = None
        self.input_scale = input_scale

        for m in self.modules():
            if isinstance(m, pnn.Conv2d):
                tnn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, pnn.BatchNorm2d):
                tnn.init.constant_(m.weight, 1)
                tnn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    tnn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    tnn.init.constant_(m.bn2.weight, 0)
        self.fuse_relu_backward(True)

    def fuse_relu_backward(self, enable: bool = True) -> "ResNet":
        """Move the backward of every block-output ReLU into the next block's first conv.

        Inside ``layer1..layer4`` the output of one block is consumed only by the next block (its first
        conv and its skip branch, whose gradients meet in that conv's dgrad epilogue).  With the fusion on,
        that epilogue also applies the ``output > 0`` mask, and the block that owns the ReLU skips it: its
        BatchNorm backward stops re-reading the block output and stops writing a separate masked copy of
        the gradient for the skip branch (about 4 bytes per activation element less HBM traffic).  The
        same switch covers the hand-off between consecutive convs (inside a block and across blocks): the
        dgrad epilogue that writes a gradient tile also takes the two per-channel sums that the BatchNorm
        backward of the producing layer needs, which removes that layer's separate reduction pass over
        the gradient and the activation.  The arithmetic is the same either way.  Turn it off (``model.fuse_relu_backward(False)``) when code
        outside the model consumes a block's output directly, e.g. a feature tap between two blocks."""
        blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]
        for b in blocks:
            b.premask_in = b.premasked_out = False
            b.fuse_inner = bool(enable)
            b.defer_out = False
        if enable:
            for prev, nxt in zip(blocks[:-1], blocks[1:]):
                prev.premasked_out = True
                nxt.premask_in = True
                # ... and where the next block starts with a fused 1x1 conv on 256 / 512 channels, that conv's kernel
                # also runs the forward of this block's last pass (BatchNorm + shortcut + ReLU) on its operand rows
                c1 = getattr(nxt, "conv1", None)
                prev.defer_out = (isinstance(nxt, Bottleneck) and c1.kernel_size == (1, 1) and c1.stride == (1, 1)
                                  and c1.in_channels in (256, 512) and c1.out_channels <= 256
                                  and pnn.ConvBnAct.takes_fused_path(c1, torch.float16))
        return self

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = tnn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride),
                                        pnn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self.base_width, self.groups)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, base_width=self.base_width, groups=self.groups))
        return tnn.Sequential(*layers)

    def set_input_normalization(self, mean, std, scale: float = 1.0) -> "ResNet":
        """Normalise inside the stem kernel: input * ``scale`` (1/255 for raw ``uint8`` batches), minus
        ``mean``, divided by ``std`` per channel - the loader then ships bytes, not floats."""
        dev = self.conv1.weight.device
        self._in_mean = torch.tensor(mean, dtype=torch.float32, device=dev)
        self._in_inv_std = 1.0 / torch.tensor(std, dtype=torch.float32, device=dev)
        self.input_scale = float(scale)
        return self

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        for n in ("_in_mean", "_in_inv_std"):       # plain attributes when set after construction
            t = self.__dict__.get(n)
            if torch.is_tensor(t):
                self.__dict__[n] = fn(t)
        return r

    def set_compute_dtype(self, dtype: torch.dtype) -> "ResNet":
        """Activation / MFMA input dtype (fp32, fp16 or bf16); parameters stay fp32 masters."""
        assert dtype in (torch.float32, torch.float16, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    def features(self, images: torch.Tensor) -> torch.Tensor:
        x = ops.stem_conv_bn_relu_pool(images, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv1.stride,
                                       self.conv1.padding, self.bn1.state(), self.compute_dtype,
                                       pool=(self.maxpool.kernel_size, self.maxpool.stride, self.maxpool.padding),
                                       mean=self._in_mean, inv_std=self._in_inv_std, pre_scale=self.input_scale)
        # block by block (what nn.Sequential would do) so that each block's backward link reaches the
        # block that consumes its output; the link is dropped from the modules right away - only the
        # autograd nodes of this forward keep it
        link = None
        blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]
        for i, block in enumerate(blocks):
            block._link_in = link
            # (only here is the consumer of the link known to run next; a hook on it would see the unfinished tensor)
            block._defer_out_now = (i + 1 < len(blocks) and not blocks[i + 1]._forward_pre_hooks
                                    and not blocks[i + 1]._forward_hooks)
            try:
                x = block(x)
            finally:
                block._defer_out_now = False
            link, block._link_out = block._link_out, None
        return x

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = self.features(images)
        x = self.avgpool(x)
        return self.fc(x)


_CFG = {
    "resnet18": (BasicBlock, [2, 2, 2, 2]),
    "resnet34": (BasicBlock, [3, 4, 6, 3]),
    "resnet50": (Bottleneck, [3, 4, 6, 3]),
    "resnet101": (Bottleneck, [3, 4, 23, 3]),
    "resnet152": (Bottleneck, [3, 8, 36, 3]),
    "wide_resnet50_2": (Bottleneck, [3, 4, 6, 3]),
    "wide_resnet101_2": (Bottleneck, [3, 4, 23, 3]),
    "resnext50_32x4d": (Bottleneck, [3, 4, 6, 3]),
    "resnext101_32x8d": (Bottleneck, [3, 4, 23, 3]),
    "resnext101_64x4d": (Bottleneck, [3, 4, 23, 3]),
}
_WIDTHS = {"wide_resnet50_2": (1, 128), "wide_resnet101_2": (1, 128), "resnext50_32x4d": (32, 4),
           "resnext101_32x8d": (32, 8), "resnext101_64x4d": (64, 4)}


def _make(name):
    def ctor(pretrained: bool = False, **kw) -> ResNet:
        block, layers = _CFG[name]
        if name in _WIDTHS:
            kw.setdefault("groups", _WIDTHS[name][0])
            kw.setdefault("width_per_group", _WIDTHS[name][1])
        model = ResNet(block, layers, **kw)
        if pretrained:
            from . import load_pretrained
            load_pretrained(model, name)
        return model
    ctor.__name__ = name
    ctor.__doc__ = f"{name} on the fused NHWC ops (torchvision-compatible state_dict)."
    return ctor


resnet18 = _make("resnet18")
resnet34 = _make("resnet34")
resnet50 = _make("resnet50")
resnet101 = _make("resnet101")
resnet152 = _make("resnet152")
wide_resnet50_2 = _make("wide_resnet50_2")
wide_resnet101_2 = _make("wide_resnet101_2")
resnext50_32x4d = _make("resnext50_32x4d")
resnext101_32x8d = _make("resnext101_32x8d")
resnext101_64x4d = _make("resnext101_64x4d")
