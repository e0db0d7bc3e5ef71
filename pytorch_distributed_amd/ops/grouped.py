"""Grouped convolution: ResNeXt's 32-group 3x3 on MFMA kernels, and depthwise convolution (one filter per
channel) on its own vector-memory kernels (``csrc/kernels/depthwise_conv.hip``).

A group of a ResNeXt layer is 4..32 channels wide: narrower than one MFMA tile, so a per-group GEMM would
leave most of every matrix instruction idle and a scalar kernel would run at a few TFLOP/s.  The 3x3 layers
with as many output as input channels per group (every grouped layer of the ResNeXt family) run on
``csrc/kernels/grouped_conv.hip``: the channels are cut into chunks of 32 - whole groups - and each chunk is a
dense 32 -> 32 convolution whose filter holds its groups on the diagonal, ``32 / Cg`` times the useful FLOPs
(8x for 4-channel groups, none for 32-channel ones) on a layer that is bound by memory traffic anyway.

Any other grouped shape (another filter size, fp32, groups wider than 32 channels) falls back to the dense
kernels: the grouped weight ``[K, C/groups, R, S]`` is expanded to its block-diagonal dense form
``[K, R, S, C]`` and the weight gradient is the block diagonal of the dense one, at ``groups`` times the useful
FLOPs.  Meant for ``groups <= 64``.  A depthwise layer (``groups == in_channels == out_channels``) is bound by
memory traffic instead and has dedicated fprop / dgrad / wgrad kernels.

Reference role: ``torchvision.models.resnext*`` behind the reference's ``--arch`` switch
(/root/reference/1.dataparallel.py:23-24,97-102); the CPU path is ``F.conv2d(groups=...)``.
"""
from __future__ import annotations

import os

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


_CHUNK_COLUMNS: dict = {}


def _chunk_columns(Cg: int, device) -> torch.Tensor:
    """``[32, Cg]``: for output channel ``o`` of a 32-channel chunk, the chunk-local input channels of its group."""
    key = (Cg, str(device))
    idx = _CHUNK_COLUMNS.get(key)
    if idx is None:
        o = torch.arange(32, device=device)
        idx = ((o // Cg) * Cg).view(32, 1) + torch.arange(Cg, device=device).view(1, Cg)
        _CHUNK_COLUMNS[key] = idx
    return idx


def chunk_filters(w_kcrs: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Logical ``[K, Cg, R, S]`` (``K`` a multiple of 32, ``Cg`` dividing 32) -> ``[K/32, 32, R*S, 32]``: per
    chunk of 32 channels the dense ``[output][tap][input]`` filter with the chunk's groups on the diagonal."""
    K, Cg, R, S = w_kcrs.shape
    nch = K // 32
    w = w_kcrs.detach().to(dtype).permute(0, 2, 3, 1).reshape(nch, 32, R * S, Cg)
    out = torch.zeros((nch, 32, R * S, 32), device=w.device, dtype=dtype)
    idx = _chunk_columns(Cg, w.device).view(1, 32, 1, Cg).expand(nch, 32, R * S, Cg)
    return out.scatter_(3, idx, w)


def chunk_diagonal(dwc: torch.Tensor, Cg: int, R: int, S: int) -> torch.Tensor:
    """``[K/32, 32, R*S, 32]`` -> logical ``[K, Cg, R, S]`` (the part that belongs to the groups)."""
    nch = dwc.shape[0]
    idx = _chunk_columns(Cg, dwc.device).view(1, 32, 1, Cg).expand(nch, 32, R * S, Cg)
    return dwc.gather(3, idx).reshape(nch * 32, R, S, Cg).permute(0, 3, 1, 2)


def _chunked_ok(x: torch.Tensor, weight: torch.Tensor, stride, pad, groups: int) -> bool:
    K, Cg, R, S = weight.shape
    N, Hi, Wi, Cc = x.shape
    return (x.element_size() == 2 and R == 3 and S == 3 and K == Cc and K % 32 == 0 and 32 % Cg == 0
            and stride[0] == stride[1] and stride[0] in (1, 2) and pad[0] == pad[1] and 0 <= pad[0] <= 2
            and max(Hi, Wi) < 512 and x.numel() * 2 < 2 ** 31 - 1 and os.environ.get("PDA_GROUPED_DENSE", "0") != "1")


def chunk_filters_gpu(weight: torch.Tensor, dtype: torch.dtype, with_dgrad: bool, rotate: bool):
    """One launch for :func:`chunk_filters` and, with ``with_dgrad``, the input-gradient layout
    ``[K/32, 32 input, 9, 32 output]`` (taps rotated by 180 degrees when ``rotate``: stride-1 layers)."""
    K, Cg = weight.shape[0], weight.shape[1]
    w = weight.detach()
    if w.dtype not in (torch.float32, dtype) or not w.is_contiguous():
        w = w.to(dtype).contiguous()
    wf = torch.empty((K // 32, 32, 9, 32), device=w.device, dtype=dtype)
    wd = torch.empty_like(wf) if with_dgrad else None
    _lib.check(_lib.lib().pda_grouped_chunk_filters(_lib.dt(wf), _lib.dt(w), w.data_ptr(), wf.data_ptr(), _lib.ptr(wd), K,
                                                   Cg, int(rotate), _lib.stream()), "pda_grouped_chunk_filters")
    return wf, wd


class _ChunkedGroupedConvFn(torch.autograd.Function):
    """3x3 grouped convolution with ``Cg`` input and output channels per group on the 32-channel-chunk kernels."""

    @staticmethod
    def _conv(mode, src, wc, out_shape, stride, pad):
        l = _lib.lib()
        N, Hs, Ws, Cc = src.shape
        out = torch.empty(out_shape, device=src.device, dtype=src.dtype)
        _lib.check(l.pda_grouped_conv3x3(_lib.dt(src), mode, src.data_ptr(), wc.data_ptr(), out.data_ptr(), N, Hs, Ws,
                                         out_shape[1], out_shape[2], Cc, stride, pad, _lib.stream()),
                   "pda_grouped_conv3x3")
        return out

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        x = x.contiguous()
        K = weight.shape[0]
        N, Hi, Wi, _ = x.shape
        Ho, Wo = C.out_size(Hi, 3, stride[0], pad[0]), C.out_size(Wi, 3, stride[1], pad[1])
        if Ho < 1 or Wo < 1:
            raise ValueError(f"grouped conv: a 3x3 window does not fit a {Hi}x{Wi} input with padding {pad}")
        wc, wd = chunk_filters_gpu(weight, x.dtype, ctx.needs_input_grad[0], stride[0] == 1)
        y = _ChunkedGroupedConvFn._conv(0, x, wc, (N, Ho, Wo, K), stride[0], pad[0])
        if bias is not None:
            l = _lib.lib()
            _lib.check(l.pda_bias_add(_lib.dt(y), y.data_ptr(), bias.float().data_ptr(), y.data_ptr(), y.numel(), K,
                                      0, _lib.stream()), "pda_bias_add")
        ctx.save_for_backward(x, weight)
        ctx.wd = wd
        ctx.cfg = (stride, pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import backends
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        dy = dy.contiguous()
        K, Cg, R, S = weight.shape
        N, Hi, Wi, _ = x.shape
        l = _lib.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if stride[0] == 1:
                # a stride-1 layer's input gradient is the forward kernel on dy with each chunk's filter rotated by
                # 180 degrees and transposed, and padding 2 - pad
                dx = _ChunkedGroupedConvFn._conv(0, dy, ctx.wd, tuple(x.shape), 1, 2 - pad[0])
            else:
                # dx = sum over taps of dy * W^T gathered with the stride's parity test: [input][tap][output] filters
                dx = _ChunkedGroupedConvFn._conv(1, dy, ctx.wd, tuple(x.shape), stride[0], pad[0])
        if ctx.needs_input_grad[1]:
            dwc = torch.zeros((K // 32, 32, 9, 32), device=dy.device, dtype=torch.float32)
            _lib.check(l.pda_grouped_conv3x3_wgrad(_lib.dt(dy), dy.data_ptr(), x.data_ptr(), dwc.data_ptr(), N, Hi, Wi,
                                                   dy.shape[1], dy.shape[2], K, stride[0], pad[0], _lib.stream()),
                       "pda_grouped_conv3x3_wgrad")
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            _lib.check(l.pda_grouped_chunk_diag(_lib.dt(dw), dwc.data_ptr(), dw.data_ptr(), K, Cg, _lib.stream()),
                       "pda_grouped_chunk_diag")
        if has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(K, device=dy.device, dtype=torch.float32)
            _lib.check(l.pda_colsum(_lib.dt(dy), dy.data_ptr(), db.data_ptr(), dy.numel() // K, K,
                                    int(backends.deterministic), _lib.stream()),
                       "pda_colsum")
        return dx, dw, db, None, None


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
    from .. import backends
    if _chunked_ok(x, weight, tuple(stride), tuple(pad), groups) and not backends.deterministic:
        return _ChunkedGroupedConvFn.apply(x, weight, bias, tuple(stride), tuple(pad))
    return _GroupedConvFn.apply(x, weight, bias, tuple(stride), tuple(pad), groups)
