"""Max pooling and global average pooling on NHWC tensors
(kernels: ``csrc/kernels/pool.hip``; CPU path: stock PyTorch).
Reference: torchvision ResNet ``maxpool``/``avgpool`` and ``F.max_pool2d`` in
/root/reference/5.2.horovod_pytorch_mnist.py:46-47.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib
from .conv import out_size


def pool_out_size(i: int, k: int, s: int, p: int, ceil_mode: bool = False) -> int:
    """Output length of a pooling window sweep; with ``ceil_mode`` a last, partial window is kept as long
    as it starts inside the input or its left padding (the ``torch.nn.MaxPool2d`` rule)."""
    if not ceil_mode:
        return out_size(i, k, s, p)
    o = -((i + 2 * p - k) // -s) + 1
    if (o - 1) * s >= i + p:
        o -= 1
    return o


def maxpool_fwd_gpu(x: torch.Tensor, k: int, s: int, p: int, scale=None, shift=None, ceil_mode: bool = False):
    """-> (pooled, arg-max bytes).  With ``scale``/``shift`` (fp32 [C]) the pooled quantity is
    ``relu(x*scale+shift)``: BatchNorm apply + ReLU + pool in one pass, without the full-size activation."""
    l = _lib.lib()
    N, H, W, Cc = x.shape
    if not 0 <= p <= k // 2 or k < 1 or s < 1 or H + 2 * p < k or W + 2 * p < k:
        raise ValueError(f"max pool: window {k}, stride {s}, padding {p} do not fit a {H}x{W} input")
    Ho, Wo = pool_out_size(H, k, s, p, ceil_mode), pool_out_size(W, k, s, p, ceil_mode)
    out = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
    idx = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=torch.uint8)
    _lib.check(l.pda_maxpool_fwd(_lib.dt(x), x.data_ptr(), out.data_ptr(), idx.data_ptr(), N, H, W, Cc, Ho, Wo, k, s,
                                 p, _lib.ptr(scale), _lib.ptr(shift), _lib.stream()), "pda_maxpool_fwd")
    return out, idx


def maxpool_bwd_gpu(dout: torch.Tensor, idx: torch.Tensor, in_shape, k: int, s: int, p: int) -> torch.Tensor:
    l = _lib.lib()
    N, H, W, Cc = in_shape
    Ho, Wo = dout.shape[1], dout.shape[2]
    dx = torch.empty(tuple(in_shape), device=dout.device, dtype=dout.dtype)
    _lib.check(l.pda_maxpool_bwd(_lib.dt(dout), dout.data_ptr(), idx.data_ptr(), dx.data_ptr(), N, H, W, Cc, Ho, Wo,
                                 k, s, p, _lib.stream()), "pda_maxpool_bwd")
    return dx


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p, ceil_mode=False):
        x = x.contiguous()
        out, idx = maxpool_fwd_gpu(x, k, s, p, ceil_mode=ceil_mode)
        ctx.save_for_backward(idx)
        ctx.cfg = (tuple(x.shape), k, s, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        shape, k, s, p = ctx.cfg
        return maxpool_bwd_gpu(dout.contiguous(), idx, shape, k, s, p), None, None, None, None


def max_pool2d(x: torch.Tensor, k: int, s: int = None, p: int = 0, ceil_mode: bool = False) -> torch.Tensor:
    s = k if s is None else s
    if not x.is_cuda:
        return F.max_pool2d(x.permute(0, 3, 1, 2), k, s, p, ceil_mode=ceil_mode).permute(0, 2, 3, 1).contiguous()
    return _MaxPoolFn.apply(x, k, s, p, ceil_mode)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def _launch(direction, a, b, geom):
        l = _lib.lib()
        _lib.check(l.pda_avgpool(_lib.dt(a), direction, a.data_ptr(), b.data_ptr(), *geom, _lib.stream()),
                   "pda_avgpool")

    @staticmethod
    def forward(ctx, x, k, s, p):
        x = x.contiguous()
        N, H, W, Cc = x.shape
        if not 0 <= 2 * p <= k or H + 2 * p < k or W + 2 * p < k:
            raise ValueError(f"avg pool: window {k}, stride {s}, padding {p} do not fit a {H}x{W} input")
        Ho, Wo = out_size(H, k, s, p), out_size(W, k, s, p)
        out = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
        ctx.geom = (N, H, W, Cc, Ho, Wo, k, s, p)
        _AvgPoolFn._launch(0, x, out, ctx.geom)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        N, H, W, Cc = ctx.geom[:4]
        dx = torch.empty((N, H, W, Cc), device=dout.device, dtype=dout.dtype)
        _AvgPoolFn._launch(1, dout, dx, ctx.geom)
        return dx, None, None, None


def avg_pool2d(x: torch.Tensor, k: int, s: int = None, p: int = 0) -> torch.Tensor:
    """Average pooling of an NHWC tensor, padding counted in the mean as ``F.avg_pool2d`` does by default
    (Inception's 3x3 s1 p1 branch pools and 5x5 s3 auxiliary pool, DenseNet's 2x2 s2 transitions).  Own
    kernels on the GPU: the stock op's backward on a channels-last view returned wrong gradients there."""
    s = k if s is None else s
    N, H, W, Cc = x.shape
    if x.is_cuda:
        return _AvgPoolFn.apply(x, k, s, p)
    if p:
        return F.avg_pool2d(x.permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1).contiguous()
    if s == k:
        Ho, Wo = H // k, W // k
        if (Ho * k, Wo * k) != (H, W):
            x = x[:, :Ho * k, :Wo * k]
        return x.reshape(N, Ho, k, Wo, k, Cc).mean(dim=(2, 4))
    return F.avg_pool2d(x.permute(0, 3, 1, 2), k, s).permute(0, 2, 3, 1).contiguous()


def adaptive_avg_pool2d(x: torch.Tensor, out_hw) -> torch.Tensor:
    """``nn.AdaptiveAvgPool2d(out_hw)`` on an NHWC tensor: nothing to do when the map already has that
    size (VGG / AlexNet at 224x224), the global-average kernel for (1, 1), stock pooling otherwise."""
    oh, ow = (out_hw, out_hw) if isinstance(out_hw, int) else tuple(out_hw)
    N, H, W, Cc = x.shape
    if (H, W) == (oh, ow):
        return x
    if (oh, ow) == (1, 1):
        return global_avg_pool(x).view(N, 1, 1, Cc)
    # stock pooling, on a contiguous NCHW copy: on the GPU the stock pooling backward is not trusted with
    # channels-last views (see avg_pool2d)
    return F.adaptive_avg_pool2d(x.permute(0, 3, 1, 2).contiguous(), (oh, ow)).permute(0, 2, 3, 1).contiguous()


class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        l = _lib.lib()
        x = x.contiguous()
        N, H, W, Cc = x.shape
        out = torch.empty((N, Cc), device=x.device, dtype=x.dtype)
        _lib.check(l.pda_gap_fwd(_lib.dt(x), x.data_ptr(), out.data_ptr(), N, H * W, Cc, _lib.stream()), "pda_gap_fwd")
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        l = _lib.lib()
        dout = dout.contiguous()
        N, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, device=dout.device, dtype=dout.dtype)
        _lib.check(l.pda_gap_bwd(_lib.dt(dout), dout.data_ptr(), dx.data_ptr(), N, H * W, Cc, _lib.stream()),
                   "pda_gap_bwd")
        return dx


def global_avg_pool(x: torch.Tensor) -> torch.Tensor:
    """[N,H,W,C] -> [N,C]."""
    if not x.is_cuda:
        return x.float().mean(dim=(1, 2)).to(x.dtype)
    return _GapFn.apply(x)
