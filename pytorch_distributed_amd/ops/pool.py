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


def maxpool_fwd_gpu(x: torch.Tensor, k: int, s: int, p: int, scale=None, shift=None):
    """-> (pooled, arg-max bytes).  With ``scale``/``shift`` (fp32 [C]) the pooled quantity is
    ``relu(x*scale+shift)``: BatchNorm apply + ReLU + pool in one pass, without the full-size activation."""
    l = _lib.lib()
    N, H, W, Cc = x.shape
    Ho, Wo = out_size(H, k, s, p), out_size(W, k, s, p)
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
    def forward(ctx, x, k, s, p):
        x = x.contiguous()
        out, idx = maxpool_fwd_gpu(x, k, s, p)
        ctx.save_for_backward(idx)
        ctx.cfg = (tuple(x.shape), k, s, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        shape, k, s, p = ctx.cfg
        return maxpool_bwd_gpu(dout.contiguous(), idx, shape, k, s, p), None, None, None


def max_pool2d(x: torch.Tensor, k: int, s: int = None, p: int = 0) -> torch.Tensor:
    s = k if s is None else s
    if not x.is_cuda:
        return F.max_pool2d(x.permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1).contiguous()
    return _MaxPoolFn.apply(x, k, s, p)


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
