"""Grouped convolution: ResNeXt's 32-group 3x3 on the MFMA implicit-GEMM kernels, and depthwise
convolution (one filter per channel) on its own vector-memory kernels (``csrc/kernels/depthwise_conv.hip``).

A group of a ResNeXt layer is 4..32 channels wide: narrower than one MFMA tile, so a per-group GEMM would
leave most of every matrix instruction idle and a scalar kernel would run at a few TFLOP/s.  On this
hardware it is cheaper to spend the FLOPs: the grouped weight ``[K, C/groups, R, S]`` is expanded to its
block-diagonal dense form ``[K, R, S, C]`` (zeros off the diagonal) and fprop / dgrad / wgrad run on the
dense kernels at MFMA rate; the weight gradient is the block diagonal of the dense one.  The expansion
costs ``groups`` times the useful FLOPs of the layer, which for ResNeXt-50 (32x4d) makes the whole network
about 2.8x the MFMA work of ResNet-50 - still several times faster than scalar grouped kernels.  Meant
for ``groups <= 64``.  A depthwise layer (``groups == in_channels == out_channels``) is bound by memory
traffic instead and has dedicated fprop / dgrad / wgrad kernels.

Reference role: ``torchvision.models.resnext*`` behind the reference's ``--arch`` switch
(/root/reference/1.dataparallel.py:23-24,97-102); the CPU path is ``F.conv2d(groups=...)``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib
from . import conv as C

MAX_GROUPS = 64


def densify(w_kcrs: torch.Tensor, groups: int, dtype: torch.dtype) -> torch.Tensor:
    """Logical ``[K, C/groups, R, S]`` -> physical dense ``[K, R, S, C]`` with the groups on the diagonal."""
    K, Cg, R, S = w_kcrs.shape
    Kg = K // groups
    w = w_kcrs.detach().permute(0, 2, 3, 1).to(dtype).reshape(groups, Kg, R, S, Cg)
    dense = torch.zeros((groups, Kg, R, S, groups, Cg), device=w.device, dtype=dtype)
    idx = torch.arange(groups, device=w.device)
    dense[idx, :, :, :, idx, :] = w
    return dense.view(K, R, S, groups * Cg)


def block_diagonal(dw_dense: torch.Tensor, groups: int) -> torch.Tensor:
    """Dense ``[K, R, S, C]`` -> logical ``[K, C/groups, R, S]`` (the part that belongs to the groups)."""
    K, R, S, Cc = dw_dense.shape
    Kg, Cg = K // groups, Cc // groups
    idx = torch.arange(groups, device=dw_dense.device)
    d = dw_dense.view(groups, Kg, R, S, groups, Cg)[idx, :, :, :, idx, :]
    return d.reshape(K, R, S, Cg).permute(0, 3, 1, 2)


class _GroupedConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, groups):
        x = x.contiguous()
        K, Cg, R, S = weight.shape
        wd = densify(weight, groups, x.dtype)
        y = C.fprop_gpu(x, wd, stride, pad)
        if y is None:
            raise RuntimeError(f"grouped conv: no implicit-GEMM kernel for input {tuple(x.shape)}, weight "
                               f"{tuple(weight.shape)}, groups {groups} (channel counts must be multiples of 8)")
        if bias is not None:
            l = _lib.lib()
            _lib.check(l.pda_bias_add(_lib.dt(y), y.data_ptr(), bias.float().data_ptr(), y.data_ptr(), y.numel(), K,
                                      0, _lib.stream()), "pda_bias_add")
        ctx.save_for_backward(x, weight)
        ctx.wd = wd
        ctx.cfg = (stride, pad, groups, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import backends
        x, weight = ctx.saved_tensors
        stride, pad, groups, has_bias = ctx.cfg
        dy = dy.contiguous()
        K, Cg, R, S = weight.shape
        N, Hi, Wi, _ = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = C.dgrad_gpu(dy, C.krsc_to_crsk(ctx.wd, dy.dtype), (Hi, Wi), stride, pad)
            if dx is None:
                raise RuntimeError("grouped conv: no implicit-GEMM dgrad kernel for this shape")
        if ctx.needs_input_grad[1]:
            dwd = C.wgrad_gpu(dy, x, R, S, stride, pad)
            if dwd is None:
                raise RuntimeError("grouped conv: no implicit-GEMM wgrad kernel for this shape")
            dw = block_diagonal(dwd, groups).to(weight.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            l = _lib.lib()
            db = torch.empty(K, device=dy.device, dtype=torch.float32)
            _lib.check(l.pda_colsum(_lib.dt(dy), dy.data_ptr(), db.data_ptr(), dy.numel() // K, K,
                                    int(backends.deterministic), _lib.stream()),
                       "pda_colsum")
        return dx, dw, db, None, None, None


class _DepthwiseConvFn(torch.autograd.Function):
    @staticmethod
    def _launch(mode, a, b, bias, out, geom, slabs=0):
        l = _lib.lib()
        _lib.check(l.pda_depthwise_conv(_lib.dt(a), mode, a.data_ptr(), b.data_ptr(), _lib.ptr(bias), out.data_ptr(),
                                        *geom, slabs, _lib.stream()), "pda_depthwise_conv")

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        x = x.contiguous()
        Cc, _, R, S = weight.shape
        N, Hi, Wi, _ = x.shape
        Ho, Wo = C.out_size(Hi, R, stride[0], pad[0]), C.out_size(Wi, S, stride[1], pad[1])
        if Ho < 1 or Wo < 1:
            raise ValueError(f"depthwise conv: a {R}x{S} window does not fit a {Hi}x{Wi} input with padding {pad}")
        geom = (N, Hi, Wi, Cc, Ho, Wo, R, S, stride[0], stride[1], pad[0], pad[1])
        w_rsc = weight.detach().reshape(Cc, R * S).t().to(x.dtype).contiguous()   # tap-major [R*S, C]
        y = torch.empty((N, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
        _DepthwiseConvFn._launch(0, x, w_rsc, None if bias is None else bias.float().contiguous(), y, geom)
        ctx.save_for_backward(x, weight, w_rsc)
        ctx.cfg = (geom, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import backends
        x, weight, w_rsc = ctx.saved_tensors
        geom, has_bias = ctx.cfg
        dy = dy.contiguous()
        Cc, _, R, S = weight.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _DepthwiseConvFn._launch(1, dy, w_rsc, None, dx, geom)
        if ctx.needs_input_grad[1]:
            d = torch.zeros((R * S, Cc), device=dy.device, dtype=torch.float32)
            _DepthwiseConvFn._launch(2, dy, x, None, d, geom, 1 if backends.deterministic else 0)
            dw = d.t().reshape(Cc, 1, R, S).to(weight.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            l = _lib.lib()
            db = torch.empty(Cc, device=dy.device, dtype=torch.float32)
            _lib.check(l.pda_colsum(_lib.dt(dy), dy.data_ptr(), db.data_ptr(), dy.numel() // Cc, Cc,
                                    int(backends.deterministic), _lib.stream()),
                       "pda_colsum")
        return dx, dw, db, None, None


def grouped_conv2d(x: torch.Tensor, weight: torch.Tensor, bias, stride, pad, groups: int) -> torch.Tensor:
    """NHWC grouped convolution; ``weight`` is logical ``[K, C/groups, R, S]``."""
    if x.shape[3] != weight.shape[1] * groups or weight.shape[0] % groups:
        raise ValueError(f"grouped conv: input channels {x.shape[3]} / weight {tuple(weight.shape)} / groups {groups}")
    if not x.is_cuda:
        y = F.conv2d(x.permute(0, 3, 1, 2), weight.to(x.dtype), None if bias is None else bias.to(x.dtype),
                     stride=stride, padding=pad, groups=groups)
        return y.permute(0, 2, 3, 1).contiguous()
    if weight.shape[1] == 1 and weight.shape[0] == groups:
        return _DepthwiseConvFn.apply(x, weight, bias, tuple(stride), tuple(pad))
    if groups > MAX_GROUPS:
        raise NotImplementedError(f"grouped conv with {groups} groups of {weight.shape[1]} channels has no kernel "
                                  f"here (supported: depthwise, or at most {MAX_GROUPS} groups)")
    return _GroupedConvFn.apply(x, weight, bias, tuple(stride), tuple(pad), groups)
