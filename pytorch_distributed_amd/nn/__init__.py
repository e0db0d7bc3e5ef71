"""Layer modules over the NHWC device ops.

Parameter / buffer names and logical shapes match ``torch.nn`` (and therefore
torchvision checkpoints): ``Conv2d.weight`` is logically ``[K, C, R, S]`` but is
kept in channels-last memory (physically KRSC, the GEMM B-operand layout);
``BatchNorm2d`` carries ``weight, bias, running_mean, running_var,
num_batches_tracked``.  Activations flowing between these modules are physically
``[N, H, W, C]``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as tnn

from .. import ops
from ..ops import conv as _conv
from ..ops.norm import BNState


class Conv2d(tnn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, bias: bool = True,
                 groups: int = 1):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        if in_channels % groups or out_channels % groups:
            raise ValueError("in_channels and out_channels must be divisible by groups")
        self.in_channels, self.out_channels, self.kernel_size, self.groups = in_channels, out_channels, ks, groups
        self.stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.weight = tnn.Parameter(_conv.make_weight(out_channels, in_channels // groups, ks[0], ks[1]))
        self.bias = tnn.Parameter(torch.empty(out_channels)) if bias else None
        self._cache = _conv.WeightCache()
        self.reset_parameters()

    def reset_parameters(self) -> None:
        tnn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels // self.groups * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            tnn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, self._cache, self.groups, relu)

    def extra_repr(self) -> str:
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, groups={self.groups}, bias={self.bias is not None}, layout=NHWC")


class BatchNorm2d(tnn.Module):
    """BatchNorm over the channel (last) dimension of an NHWC tensor."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: Optional[float] = 0.1, affine: bool = True,
                 track_running_stats: bool = True, process_group=None, sync: bool = False):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        self.process_group, self.sync = process_group, sync
        if affine:
            self.weight = tnn.Parameter(torch.ones(num_features))
            self.bias = tnn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        if track_running_stats:
            self.register_buffer("running_mean", torch.zeros(num_features))
            self.register_buffer("running_var", torch.ones(num_features))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        else:
            self.register_buffer("running_mean", None)
            self.register_buffer("running_var", None)
            self.register_buffer("num_batches_tracked", None)

    def state(self) -> BNState:
        group = None
        if self.sync and self.training:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
                group = self.process_group if self.process_group is not None else dist.group.WORLD
        return BNState(self.eps, self.momentum, self.training, self.running_mean, self.running_var,
                       self.num_batches_tracked, group)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
        return ops.batch_norm_act(x, self.weight, self.bias, residual, relu, self.state())

    def extra_repr(self) -> str:
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, sync={self.sync}, layout=NHWC"


class SyncBatchNorm(BatchNorm2d):
    """BatchNorm2d whose batch statistics are all-reduced over ``process_group``."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats, process_group, sync=True)

    @classmethod
    def convert_sync_batchnorm(cls, module: tnn.Module, process_group=None) -> tnn.Module:
        """Switch every BatchNorm2d of this package inside ``module`` to cross-rank statistics (in place)."""
        for m in module.modules():
            if isinstance(m, BatchNorm2d):
                m.sync = True
                m.process_group = process_group
        return module


class _ZeroGradFor(torch.autograd.Function):
    """Identity on ``out`` that also hands ``param`` an all-zero gradient (so that optimizers and gradient
    reducers see every parameter of the step)."""

    @staticmethod
    def forward(ctx, out, param):
        ctx.meta = (param.shape, param.dtype, param.device)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        shape, dtype, device = ctx.meta
        return dout, torch.zeros(shape, dtype=dtype, device=device)


class ConvBnAct(tnn.Module):
    """Functional fusion helper: runs ``conv`` + ``bn`` (+ residual) (+ ReLU) as one op while the two
    sub-modules keep owning their parameters (so names stay ``conv1.weight`` / ``bn1.weight``)."""

    @staticmethod
    def takes_fused_path(conv: Conv2d, dtype: torch.dtype) -> bool:
        """Whether :meth:`run` sends this conv to the fused op (so a producer may defer its normalisation to it)."""
        pointwise = conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
        return (conv.bias is None and conv.groups == 1 and conv.out_channels % 8 == 0
                and bool(_conv.igemm_tiles(conv.in_channels, dtype))
                and (pointwise or bool(_conv.igemm_tiles(conv.out_channels, dtype))))

    @staticmethod
    def run(conv: Conv2d, bn: BatchNorm2d, x: torch.Tensor, relu: bool = True,
            residual: Optional[torch.Tensor] = None, passthrough: bool = False, mask_dx: bool = False,
            dout_premasked: bool = False, in_link=None, out_link=None, res_in=None, res_out=None,
            alias_link=None):
        """``mask_dx`` / ``dout_premasked`` / ``in_link`` / ``out_link`` / ``res_in`` / ``res_out`` / ``alias_link``: see
        :func:`pytorch_distributed_amd.ops.conv_bn_act`."""
        pointwise = conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
        fused_path = (conv.bias is None and conv.groups == 1 and conv.out_channels % 8 == 0
                      and bool(_conv.igemm_tiles(conv.in_channels, x.dtype))
                      and (pointwise or bool(_conv.igemm_tiles(conv.out_channels, x.dtype))))
        if res_in is not None and res_in.affine is not None and not fused_path:
            # only the fused op adds a shortcut that its producer left un-normalised
            from ..ops.fused import materialize_shortcut
            residual, res_in = materialize_shortcut(residual, res_in), None
        plain = not (passthrough or mask_dx or dout_premasked or in_link is not None or out_link is not None)
        if (conv.bias is not None and plain and x.is_cuda and bn.training and bn.track_running_stats
                and bn.momentum is not None and conv.groups == 1 and conv.out_channels % 8 == 0
                and _conv.igemm_tiles(conv.in_channels, x.dtype)
                and (pointwise or _conv.igemm_tiles(conv.out_channels, x.dtype))):
            # A bias in front of a training-mode BatchNorm only moves the batch mean, which the normalisation
            # removes again: the output is that of the bias-free fused op, the bias gradient is exactly zero,
            # and the one place the bias shows is the running mean (VGG-*_bn keeps such biases).
            o = ops.conv_bn_act(x, conv.weight, bn.weight, bn.bias, residual, conv.stride, conv.padding, relu,
                                bn.state(), conv._cache)
            with torch.no_grad():
                bn.running_mean.add_(conv.bias.detach().to(bn.running_mean.dtype), alpha=bn.momentum)
            return _ZeroGradFor.apply(o, conv.bias)
        if conv.bias is not None or conv.groups != 1 or conv.out_channels % 8 \
                or not _conv.igemm_tiles(conv.in_channels, x.dtype) \
                or not (pointwise or _conv.igemm_tiles(conv.out_channels, x.dtype)):
            # unfused modules (a bias, groups, or a channel count the MFMA kernels do not tile): the BatchNorm op masks its own ReLU gradient (dout_premasked needs nothing, a
            # link given to it stays empty), and the promise made by mask_dx / in_link is kept by an explicit
            # gradient mask on x
            if in_link is not None and in_link.mat is not None:
                from ..ops.fused import materialize_block_output
                x = materialize_block_output(x, in_link)
            if in_link is not None and in_link.deferred:
                from ..ops.fused import materialize_deferred
                x = materialize_deferred(x, in_link)
            if (mask_dx or (in_link is not None and in_link.relu)) and x.is_cuda:
                from ..ops.fused import grad_mask
                x = grad_mask(x)
            if out_link is not None:
                out_link.relu = relu
            o = bn(conv(x), residual, relu)
            return (o, x) if passthrough else o
        return ops.conv_bn_act(x, conv.weight, bn.weight, bn.bias, residual, conv.stride, conv.padding, relu,
                               bn.state(), conv._cache, passthrough, mask_dx, dout_premasked, in_link, out_link,
                               res_in, res_out, alias_link)


class ReLU(tnn.Module):
    def __init__(self, inplace: bool = False):
        super().__init__()

    def forward(self, x):
        return ops.relu(x)


class ReLU6(tnn.Module):
    def __init__(self, inplace: bool = False):
        super().__init__()

    def forward(self, x):
        return ops.relu6(x)


class MaxPool2d(tnn.Module):
    def __init__(self, kernel_size: int, stride: Optional[int] = None, padding: int = 0, ceil_mode: bool = False):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride or kernel_size, padding
        self.ceil_mode = ceil_mode

    def forward(self, x):
        return ops.max_pool2d(x, self.kernel_size, self.stride, self.padding, self.ceil_mode)


class AvgPool2d(tnn.Module):
    def __init__(self, kernel_size: int, stride: Optional[int] = None):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride or kernel_size

    def forward(self, x):
        return ops.avg_pool2d(x, self.kernel_size, self.stride)


class AdaptiveAvgPool2d(tnn.Module):
    """[N,H,W,C] -> [N,oh,ow,C]."""

    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        return ops.adaptive_avg_pool2d(x, self.output_size)


class GlobalAvgPool2d(tnn.Module):
    """AdaptiveAvgPool2d((1,1)) + flatten: [N,H,W,C] -> [N,C]."""

    def forward(self, x):
        return ops.global_avg_pool(x)


class Dropout(tnn.Module):
    def __init__(self, p: float = 0.5):
        super().__init__()
        self.p = p

    def forward(self, x):
        return ops.dropout(x, self.p, self.training)


class Dropout2d(Dropout):
    def forward(self, x):
        return ops.dropout2d(x, self.p, self.training)


class Linear(tnn.Linear):
    """``torch.nn.Linear`` parameters, computed in the activation dtype on this package's MFMA GEMM kernels
    (``ops/linear.py``: forward = 1x1 fprop, split over K when the batch gives too few output tiles; dX = dgrad
    with the class dimension zero-padded to the kernel's K step, or the split-K kernel for small batches;
    dW = split-K wgrad, db = column sums); the library GEMM only for shapes those kernels do not take."""

    def forward(self, x):
        from ..ops.linear import linear, LinearCache
        cache = self.__dict__.get("_pda_cache")
        if cache is None:
            cache = self.__dict__["_pda_cache"] = LinearCache()
        return linear(x, self.weight, self.bias, cache)


class CrossEntropyLoss(tnn.Module):
    """Softmax cross-entropy (mean). Pass ``metrics`` (4 floats on the logits' device) to also accumulate
    {loss_sum, correct@1, correct@5, rows} on the device."""

    def __init__(self, reduction: str = "mean"):
        super().__init__()
        self.reduction = reduction

    def forward(self, logits, target, metrics: Optional[torch.Tensor] = None):
        return ops.cross_entropy(logits, target, metrics, self.reduction)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """NCHW -> physically NHWC tensor of shape [N,H,W,C]."""
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2).contiguous()


__all__ = ["Conv2d", "BatchNorm2d", "SyncBatchNorm", "ConvBnAct", "ReLU", "ReLU6", "MaxPool2d", "AvgPool2d",
           "AdaptiveAvgPool2d", "GlobalAvgPool2d", "Dropout",
           "Dropout2d", "Linear", "CrossEntropyLoss", "to_nhwc", "to_nchw"]
