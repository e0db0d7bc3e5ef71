"""Convolution on NHWC tensors: MFMA implicit-GEMM kernels on the GPU, a plain
PyTorch composition on the CPU.

Tensors are physically ``[N, H, W, C]`` (channels innermost).  Weights are
``nn.Parameter`` s of logical shape ``[K, C, R, S]`` stored in channels-last
memory format, i.e. physically ``[K, R, S, C]`` ("KRSC"), which is both what the
GEMM B operand wants (K contiguous) and what keeps ``state_dict()`` shape-compatible
with torchvision checkpoints.

Reference behaviour being reproduced: the torchvision ResNet convolutions that
run under ``model(images)`` / ``loss.backward()`` in
/root/reference/2.distributed.py:213,231 (cuDNN there).
"""
from __future__ import annotations

import os
import weakref
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from .. import backends

_PARAM_GENERATION = 0  # bumped by our optimizers, which update weights behind autograd's back
# BatchNorm statistic partials: one row per tile (default), or PDA_STAT_SLOTS=1 = a few float-atomic
# accumulation rows (shorter finalize, but measured slower end to end on MI355X: the atomics cost more
# in the conv / reduce epilogues than the finalize kernels save)
_STAT_SLOTS_OFF = os.environ.get("PDA_STAT_SLOTS", "0") != "1"


# half-precision model copies an optimizer kernel keeps equal to their fp32 masters (amp O2):
# data_ptr of the copy -> the parameter generation it is current for
_PINNED_SYNCED: dict = {}


def bump_param_generation(synced_copies=None) -> None:
    """Parameters changed behind autograd's back (optimizer kernel, multi-tensor broadcast): cached
    low-precision / transposed weights are stale.  ``synced_copies``: half copies the caller has just
    written together with the masters - those stay current."""
    global _PARAM_GENERATION
    _PARAM_GENERATION += 1
    if synced_copies is not None:
        for t in synced_copies:
            _PINNED_SYNCED[t.data_ptr()] = _PARAM_GENERATION


def param_generation() -> int:
    return _PARAM_GENERATION


_TUNED: dict = {}
# 0 = heuristic; 1/2/3 = 128x128, 128x64, 64x64 tiles; +8 = 32-deep K steps; 12 = 256x128 (128x64 per wave);
# 5/6/7 = LDS-DMA operand ring: 256x128 (8 waves, 3 x 64-deep), 128x128 (4 x 32-deep), 256x256 (8 waves, 4 x 32-deep)
# (64 = slab tile, 256x128 with 8 waves for stride-1 "same" filters such as 3x3 pad 1: all taps of a channel slice
# read one LDS-resident slab of input pixels.  It cuts the L2 traffic of the A operand ~9x but measured 5-40 %
# slower than the tiles below on every ResNet-50 3x3 layer - the step time is set by the per-step wait / barrier /
# fragment-read latency, not by L2 bandwidth (profiles/igemm_stamps_r2.txt) - so the tuner does not try it.)
_FPROP_CFGS = (0, 1, 2, 3, 9, 10, 11, 12, 5, 6, 7, 13)   # 13: 256x128 ring on 2x2 waves (128x64 each), 3 x 32-deep
# 33: the A-stationary pointwise kernel (csrc/kernels/pointwise_astat.hip): unpadded 1x1, 64 / 128 / 256 / 512 input
# channels, 16-bit, plain or normalise-on-load input, statistics epilogue only
_FPROP_PW_CFG = 33
_TUNE_LONG = os.environ.get("PDA_TUNE_LONG", "1") == "1"
_TUNE_MARGIN = float(os.environ.get("PDA_TUNE_MARGIN", "0.97"))   # a candidate replaces the heuristic below this ratio
_FUSE_BIAS_RELU = os.environ.get("PDA_CONV_FUSE_BIAS_RELU", "1") == "1"   # plain convs: bias + ReLU in the GEMM epilogue
_FPROP_C64_CFG = 34   # weight-stationary 3x3 on 64 -> 64 channels (csrc/kernels/conv3x3_c64.hip)
_FPROP_XF_CFGS = (0, 1, 2, 3, 9, 10, 11, 12)  # input normalised on load: the register-staged tiles
_DGRAD_EX_CFGS = (0, 1, 2, 3, 9, 10, 11, 5, 6)  # the fused-epilogue kernels have no 256x128 / 256x256 register tile
_WGRAD_CFGS = (0, 1, 2, 3, 9, 10, 11, 12, 13)  # wgrad: 1/2/3 = 128x128, 64x64, 64x128; 12/13 = 256x128, 128x256
# + 64: the LDS-DMA ring kernel (csrc/kernels/wgrad_ring.hip), 16-bit operands without the normalise-on-load input:
# 65 = 128x128 (4 x 32 pixels), 66 = 64x256 on 1 x 4 waves, 68 / 69 = 256x128 / 128x256 (3 x 32), 70 = 128x128 (3 x 64)
_WGRAD_RING_CFGS = (65, 66, 68, 69, 70)


def _tuned_cfg(kind: str, key: tuple, cands, run) -> int:
    """``backends.benchmark``: the fastest of ``cands`` for this (op, shape), measured once with device
    events on the caller's operands.  A candidate has to beat the heuristic (``cands[0]``) by 3 % to
    replace it, so timing noise does not churn the choice."""
    k = (kind,) + key
    hit = _TUNED.get(k)
    if hit is not None:
        return hit
    if torch.cuda.is_current_stream_capturing():
        return cands[0]
    times = []
    for c in cands:
        if run(c) is None:          # configuration not available for this shape
            times.append(float("inf"))
            continue
        torch.cuda.synchronize()
        reps, t = 4, 0.0
        while True:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(c)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps
            if reps >= 16 or t * reps >= 1.5 or not _TUNE_LONG:      # short kernels: a longer window, so that a few us of jitter do not decide
                break
            reps = 16
        times.append(t)
    best = min(range(len(cands)), key=lambda i: times[i])
    choice = cands[best] if times[best] < _TUNE_MARGIN * times[0] else cands[0]
    _TUNED[k] = choice
    return choice


def out_size(i: int, k: int, s: int, p: int) -> int:
    return (i + 2 * p - k) // s + 1


def krsc(weight: torch.Tensor) -> torch.Tensor:
    """Physical [K,R,S,C] view of a logical [K,C,R,S] channels-last weight (copies if needed)."""
    w = weight.permute(0, 2, 3, 1)
    return w if w.is_contiguous() else w.contiguous()


def make_weight(k: int, c: int, r: int, s: int, device=None, dtype=torch.float32) -> torch.Tensor:
    """Logical [K,C,R,S] tensor whose memory is KRSC."""
    return torch.empty(k, r, s, c, device=device, dtype=dtype).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------- raw GPU launches
def fprop_gpu(x: torch.Tensor, w_krsc: torch.Tensor, stride: Tuple[int, int], pad: Tuple[int, int],
              res: Optional[torch.Tensor] = None, relu: bool = False, stats: bool = False, cfg: int = 0,
              geom: Optional[tuple] = None, stats_buf: Optional[torch.Tensor] = None,
              bias: Optional[torch.Tensor] = None, in_affine: Optional[tuple] = None):
    """y = conv(x, w) [+ bias] [+ res] [relu]; optionally also per-tile column sums for BatchNorm.
    ``bias``: fp32 ``[K]``, added in the GEMM epilogue (BatchNorm folded into the weights for inference).

    ``in_affine = (scale, shift)`` (``[Ci]`` tensors of ``x``'s 16-bit dtype): ``x`` is the producing layer's
    conv output *before* its BatchNorm and the kernel convolves ``relu(x * scale + shift)`` instead, applied to
    the operand tiles on their way into LDS (padding stays exactly zero) - the normalised activation is never
    written to memory.  Not combined with ``res`` / ``relu`` / ``bias``; ``None`` is returned when no kernel
    covers the request (fp32, odd channel counts).

    ``geom`` overrides the gather geometry ``(N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, R, S)`` for inputs
    that are not plain NHWC (the padded 4-channel stem image).  ``stats_buf`` (fp32 ``[2, slots, K]``,
    all zeros on entry) makes the kernel accumulate the statistics into ``slots`` rows with float
    atomics instead of writing one row per tile; whoever consumes them must zero them again
    (``norm.finalize(..., clear=True)``).
    """
    l = _lib.lib()
    K = w_krsc.shape[0]
    if geom is None:
        N, Hi, Wi, Ci = x.shape
        R, S = w_krsc.shape[1], w_krsc.shape[2]
        Ho, Wo = out_size(Hi, R, stride[0], pad[0]), out_size(Wi, S, stride[1], pad[1])
        sn, sh, sw = Hi * Wi * Ci, Wi * Ci, Ci
    else:
        N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, R, S = geom
    M = N * Ho * Wo
    if in_affine is not None:
        if res is not None or relu or bias is not None or x.element_size() != 2:
            return None
        xs, xf = in_affine
        if not (xs.dtype == xf.dtype == x.dtype and xs.numel() == xf.numel() == Ci and xs.is_contiguous()
                and xf.is_contiguous()):
            raise ValueError("fprop_gpu: in_affine must be two contiguous [Ci] tensors of the input's dtype")
    pw_ok = (geom is None and R == 1 and S == 1 and tuple(pad) == (0, 0) and res is None and not relu and bias is None
             and x.element_size() == 2 and Ci in (64, 128, 256, 512) and K % 8 == 0)
    if cfg == _FPROP_PW_CFG and not pw_ok:
        return None
    c64_ok = (geom is None and R == 3 and S == 3 and Ci == 64 and K == 64 and stride[0] == stride[1] and stride[0] in (1, 2)
              and pad[0] == pad[1] and 0 <= pad[0] <= 2 and res is None and not relu and bias is None
              and in_affine is None and x.element_size() == 2 and x.numel() * 2 < 2 ** 31 - 1
              and (not stats or stats_buf is None or True))
    if cfg == _FPROP_C64_CFG and not c64_ok:
        return None
    if cfg == 0 and backends.benchmark and geom is None:
        cands = _FPROP_XF_CFGS if in_affine is not None else (_FPROP_CFGS if res is None else _DGRAD_EX_CFGS)
        if pw_ok:
            cands = cands + (_FPROP_PW_CFG,)
        if c64_ok:
            cands = cands + (_FPROP_C64_CFG,)
        cfg = _tuned_cfg("fprop", (x.dtype, N, Hi, Wi, Ci, K, R, S, tuple(stride), tuple(pad), res is not None, stats,
                                   in_affine is not None), cands,
                         lambda c: fprop_gpu(x, w_krsc, stride, pad, res, relu, stats, c or 1 << 30, geom, None, bias,
                                             in_affine))
    if cfg == 0 and not backends.benchmark and geom is None:
        # without the tuner: the specialised kernels where they win on every measured layer (the weight-stationary
        # 3x3 on 64 -> 64 channels; the A-stationary pointwise kernel up to 256 input channels)
        if c64_ok:
            cfg = _FPROP_C64_CFG
        elif pw_ok and Ci <= 256:
            cfg = _FPROP_PW_CFG
        elif x.element_size() == 2 and K >= 128 and res is None and in_affine is None:
            # ... and the 256x128 LDS-DMA ring tile on 2x2 waves from 128 output channels up: 6.46 against 6.97 ms per
            # step over ResNet-50's layers for the register-staged heuristic (scripts/fprop_dgrad_cfg_totals.py)
            r = fprop_gpu(x, w_krsc, stride, pad, None, relu, stats, 13, None, stats_buf, bias, None)
            if r is not None:
                return r
    cfg &= (1 << 30) - 1   # (1 << 30: "heuristic, and do not tune" - the tuner's own call for candidate 0)
    if cfg == _FPROP_C64_CFG:
        y = torch.empty((N, Ho, Wo, K), device=x.device, dtype=x.dtype)
        part = None
        if stats:
            rows = l.pda_conv3x3_c64_rows(N, Ho, Wo, stride[0])
            if rows <= 0:
                return None
            part = torch.empty((2, rows, K), device=x.device, dtype=torch.float32)
        rc = l.pda_conv3x3_c64(_lib.dt(x), x.data_ptr(), w_krsc.data_ptr(), y.data_ptr(),
                               part[0].data_ptr() if stats else None, part[1].data_ptr() if stats else None, N, Hi, Wi,
                               Ho, Wo, stride[0], pad[0], _lib.stream())
        if rc < 0:
            return None
        _lib.check(rc, "pda_conv3x3_c64")
        return (y, part[0], part[1]) if stats else y
    y = torch.empty((N, Ho, Wo, K), device=x.device, dtype=x.dtype)
    psum = psq = None
    slots = 0
    if stats:
        bm = (l.pda_conv_pw_bm(Ci) if cfg == _FPROP_PW_CFG else
              l.pda_conv_igemm_bm(M, K, cfg, _lib.dt(x), int(res is not None)))
        tiles_m = (M + bm - 1) // bm
        if stats_buf is not None and tiles_m > stats_buf.shape[1]:
            slots = stats_buf.shape[1]
            psum, psq = stats_buf[0], stats_buf[1]
        else:
            part = torch.empty((2, tiles_m, K), device=x.device, dtype=torch.float32)
            psum, psq = part[0], part[1]
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != K or not bias.is_contiguous()):
        raise ValueError("fprop_gpu: bias must be a contiguous fp32 [K] tensor")
    if cfg == _FPROP_PW_CFG:
        rc = l.pda_conv_pw_fprop(_lib.dt(x), x.data_ptr(), w_krsc.data_ptr(), y.data_ptr(),
                                 None if in_affine is None else in_affine[0].data_ptr(),
                                 None if in_affine is None else in_affine[1].data_ptr(), _lib.ptr(psum), _lib.ptr(psq),
                                 N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, K, stride[0], stride[1], slots, _lib.stream())
    elif in_affine is not None:
        rc = l.pda_conv_igemm_xf(_lib.dt(x), x.data_ptr(), w_krsc.data_ptr(), y.data_ptr(), in_affine[0].data_ptr(),
                                 in_affine[1].data_ptr(), _lib.ptr(psum), _lib.ptr(psq), N, Hi, Wi, Ci, sn, sh, sw,
                                 Ho, Wo, K, R, S, stride[0], stride[1], pad[0], pad[1], cfg, slots, _lib.stream())
    else:
        rc = l.pda_conv_igemm(_lib.dt(x), 0, x.data_ptr(), w_krsc.data_ptr(), y.data_ptr(), _lib.ptr(res), 0, 0, 0,
                              _lib.ptr(bias), _lib.ptr(psum), _lib.ptr(psq), N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, K, R,
                              S, stride[0], stride[1], pad[0], pad[1], int(relu), cfg, slots, _lib.stream())
    if rc < 0:
        return None  # shape not covered by the MFMA kernels
    _lib.check(rc, "pda_conv_igemm(fprop)")
    return (y, psum, psq) if stats else y


_ONES: dict = {}


def _ones(n: int, device) -> torch.Tensor:
    key = (n, str(device))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(n, device=device, dtype=torch.float32)
    return t


def pw_fprop_materialize(y3: torch.Tensor, res: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                         res_scale: Optional[torch.Tensor], w_krsc: torch.Tensor, stats: bool = False,
                         stats_buf: Optional[torch.Tensor] = None):
    """The last pass of a residual block and the next block's first 1x1 convolution in one kernel
    (``csrc/kernels/pointwise_astat.hip``, MAT mode): ``out = relu(y3*scale + shift + res*res_scale)`` is formed in
    fp32 on the convolution's operand rows, stored with its sign bits (as ``norm.apply(want_mask=True)`` would) and
    multiplied into ``y = conv1x1(out, w)``.  ``y3`` / ``res``: ``[N,H,W,Ci]`` with ``Ci`` 256 or 512; ``w_krsc``:
    ``[Co,1,1,Ci]``, ``Co <= 256``; ``scale`` / ``shift`` / ``res_scale``: fp32 ``[Ci]`` (``res_scale=None``: an
    identity shortcut).  Returns ``(out, maskbits, y)`` or, with ``stats``, ``(out, maskbits, y, psum, psq)``;
    ``None`` when the kernel does not cover the request."""
    l = _lib.lib()
    N, H, W, Ci = y3.shape
    K = w_krsc.shape[0]
    if (y3.element_size() != 2 or Ci not in (256, 512) or K > 256 or K % 8 or tuple(w_krsc.shape[1:]) != (1, 1, Ci)
            or res.shape != y3.shape or res.dtype != y3.dtype or w_krsc.dtype != y3.dtype
            or not (y3.is_contiguous() and res.is_contiguous() and w_krsc.is_contiguous())
            or scale.dtype != torch.float32 or shift.dtype != torch.float32 or y3.numel() * 2 >= 2 ** 31 - 1):
        return None
    M = N * H * W
    out = torch.empty_like(y3)
    bits = torch.empty(y3.numel() // 8, device=y3.device, dtype=torch.uint8)
    y = torch.empty((N, H, W, K), device=y3.device, dtype=y3.dtype)
    psum = psq = None
    slots = 0
    if stats:
        bm = l.pda_conv_pw_bm(Ci)
        tiles_m = (M + bm - 1) // bm
        if stats_buf is not None and tiles_m > stats_buf.shape[1]:
            slots = stats_buf.shape[1]
            psum, psq = stats_buf[0], stats_buf[1]
        else:
            part = torch.empty((2, tiles_m, K), device=y3.device, dtype=torch.float32)
            psum, psq = part[0], part[1]
    rs = _ones(Ci, y3.device) if res_scale is None else res_scale
    rc = l.pda_conv_pw_fprop_mat(_lib.dt(y3), y3.data_ptr(), res.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                 rs.data_ptr(), out.data_ptr(), bits.data_ptr(), w_krsc.data_ptr(), y.data_ptr(),
                                 _lib.ptr(psum), _lib.ptr(psq), M, Ci, K, slots, _lib.stream())
    if rc < 0:
        return None
    _lib.check(rc, "pda_conv_pw_fprop_mat")
    return (out, bits, y, psum, psq) if stats else (out, bits, y)


def gemm_splitk(x: torch.Tensor, w: torch.Tensor, ksplit: int, cfg: int = 1) -> Optional[torch.Tensor]:
    """fp32 ``x[M, K] @ w[N, K]^T`` with the reduction spread over ``ksplit`` workgroups per output tile (float
    atomics into a zeroed result): for GEMMs whose ``ceil(M/128) * ceil(N/128)`` output tiles alone would leave
    most of the 256 CUs idle.  ``None`` when the kernel does not take the shape."""
    M, K = x.shape
    N = w.shape[0]
    acc = torch.zeros((M, N), device=x.device, dtype=torch.float32)
    rc = _lib.lib().pda_gemm_splitk(_lib.dt(x), x.data_ptr(), w.data_ptr(), acc.data_ptr(), M, K, N, cfg, ksplit,
                                    _lib.stream())
    if rc < 0:
        return None
    _lib.check(rc, "pda_gemm_splitk")
    return acc


def dgrad_gpu(dy: torch.Tensor, w_crsk: torch.Tensor, in_hw: Tuple[int, int], stride, pad,
              res: Optional[torch.Tensor] = None, cfg: int = 0, mask: Optional[torch.Tensor] = None,
              bn_input: Optional[torch.Tensor] = None, bn_scale: Optional[torch.Tensor] = None,
              bn_shift: Optional[torch.Tensor] = None, bn_sums: bool = False, mask_bits: bool = False,
              sums_buf: Optional[torch.Tensor] = None, res_s2: bool = False):
    """dx[N,Hi,Wi,C] from dy[N,Ho,Wo,K]; w_crsk is the [C,R,S,K] transposed weight.

    ``res`` ([N,Hi,Wi,C]) is added in the epilogue; with ``res_s2`` it is the compact
    ``[N, ceil(Hi/2), ceil(Wi/2), C]`` input gradient of a stride-2 1x1 shortcut conv over the same input, added at
    the pixels with even row and column (stride-1 convs in a 16-bit type; ``None`` is returned otherwise); ``mask`` ([N,Hi,Wi,C], the forward input when that
    input was a ReLU output) zeroes dx wherever ``mask <= 0`` - the ReLU backward of the producer, applied
    where the gradient is written instead of by a later pass over it.  With ``mask_bits`` the mask is the
    uint8 sign-bit tensor of that input (``norm.apply(want_mask=True)``: one byte per 16-byte vector).

    ``bn_input`` ([N,Hi,Wi,C]) is the pre-normalisation tensor of the BatchNorm that produced the forward
    input.  With ``bn_scale``/``bn_shift`` (fp32 [C]) that layer's ReLU mask ``bn_input*scale+shift > 0``
    is applied to dx; with ``bn_sums`` the call returns ``(dx, sums)``, ``sums`` = fp32 ``[2, rows, C]``
    per-tile partial sums of dx and of dx*bn_input - the two reductions that BatchNorm's backward
    needs, taken while the gradient tile is still on chip.  ``sums_buf`` (fp32 ``[2, slots, C]``, all zeros on
    entry): when the launch has more tiles than ``slots`` the sums are accumulated into it with float atomics and
    it is returned as ``sums`` - whoever consumes them zeroes it again (``norm.backward(presums_clear=True)``)."""
    l = _lib.lib()
    N, Ho, Wo, K = dy.shape
    C, R, S = w_crsk.shape[0], w_crsk.shape[1], w_crsk.shape[2]
    Hi, Wi = in_hw
    # the weight-stationary 64 -> 64 kernel takes the dgrad of a stride-1 layer with exactly the "producer's BatchNorm
    # + ReLU backward" extras (mask recomputed from bn_input, sums of dx and dx * bn_input)
    c64_ok = (R == 3 and S == 3 and C == 64 and K == 64 and tuple(stride) == (1, 1) and pad[0] == pad[1]
              and 0 <= pad[0] <= 2 and dy.element_size() == 2 and res is None and mask is None and bn_input is not None
              and bn_scale is not None and bn_shift is not None and bn_sums and not res_s2
              and dy.numel() * 2 < 2 ** 31 - 1)
    if cfg == _FPROP_C64_CFG and not c64_ok:
        return None
    if cfg == 0 and backends.benchmark:
        extras = res is not None or mask is not None or bn_input is not None
        cfg = _tuned_cfg("dgrad", (dy.dtype, N, Ho, Wo, K, C, R, S, Hi, Wi, tuple(stride), tuple(pad), res is not None,
                                   mask is not None, bn_input is not None, bn_scale is not None, bn_sums, mask_bits,
                                   res_s2),
                         (_DGRAD_EX_CFGS if extras else _FPROP_CFGS) + ((_FPROP_C64_CFG,) if c64_ok else ()),
                         lambda c: dgrad_gpu(dy, w_crsk, in_hw, stride, pad, res, c or 1 << 30, mask, bn_input, bn_scale,
                                             bn_shift, bn_sums, mask_bits, None, res_s2))
    if cfg == 0 and not backends.benchmark and c64_ok:
        cfg = _FPROP_C64_CFG        # without the tuner: the weight-stationary kernel wins on this layer type
    elif (cfg == 0 and not backends.benchmark and dy.element_size() == 2 and C >= 128 and res is None and mask is None
          and bn_input is None and not res_s2):
        r = dgrad_gpu(dy, w_crsk, in_hw, stride, pad, cfg=13)      # ... and the 256x128 ring tile for plain input gradients
        if r is not None:
            return r
    elif (cfg == 0 and not backends.benchmark and dy.element_size() == 2 and not res_s2
          and (res is not None or mask is not None or bn_input is not None)):
        # ... and the 128x128 ring tile for the fused epilogue: 5.53 against 5.90 ms per step over the ResNet-50 layers
        # that use it, per-layer best 5.49 (scripts/exdgrad_cfg_totals.py)
        r = dgrad_gpu(dy, w_crsk, in_hw, stride, pad, res, 6, mask, bn_input, bn_scale, bn_shift, bn_sums, mask_bits,
                      sums_buf, False)
        if r is not None:
            return r
    cfg &= (1 << 30) - 1
    dx = torch.empty((N, Hi, Wi, C), device=dy.device, dtype=dy.dtype)
    if cfg == _FPROP_C64_CFG:
        if (bn_input.shape != dx.shape or bn_input.dtype != dx.dtype or not bn_input.is_contiguous()
                or bn_scale.dtype != torch.float32 or bn_shift.dtype != torch.float32):
            raise ValueError(f"dgrad_gpu: bn_input must be a contiguous {tuple(dx.shape)} {dx.dtype} tensor")
        rows = l.pda_conv3x3_c64_rows(N, Hi, Wi, 1)
        if rows <= 0:
            return None
        sums = torch.empty((2, rows, C), device=dy.device, dtype=torch.float32)
        rc = l.pda_conv3x3_c64_dgrad_masked(_lib.dt(dy), dy.data_ptr(), w_crsk.data_ptr(), dx.data_ptr(),
                                            bn_input.data_ptr(), bn_scale.data_ptr(), bn_shift.data_ptr(),
                                            sums[0].data_ptr(), sums[1].data_ptr(), N, Ho, Wo, Hi, Wi, pad[0],
                                            _lib.stream())
        if rc < 0:
            return None
        _lib.check(rc, "pda_conv3x3_c64_dgrad_masked")
        return dx, sums
    for name, t in (("res", None if res_s2 else res), ("mask", None if mask_bits else mask), ("bn_input", bn_input)):
        if t is not None and (t.shape != dx.shape or t.dtype != dx.dtype or not t.is_contiguous()):
            raise ValueError(f"dgrad_gpu: {name} must be a contiguous {tuple(dx.shape)} {dx.dtype} tensor")
    if res_s2:
        if res is None or tuple(res.shape) != (N, (Hi + 1) // 2, (Wi + 1) // 2, C) or res.dtype != dx.dtype \
                or not res.is_contiguous():
            raise ValueError("dgrad_gpu: res_s2 wants a contiguous [N, ceil(Hi/2), ceil(Wi/2), C] residual")
        if tuple(stride) != (1, 1) or dx.element_size() != 2 or C % 8:
            return None
    if mask_bits and mask is not None and (mask.dtype != torch.uint8 or not mask.is_contiguous()
                                           or mask.numel() != dx.numel() * dx.element_size() // 16):
        raise ValueError("dgrad_gpu: mask_bits wants one contiguous uint8 per 16 bytes of dx")
    sums = None
    slots = 0
    if bn_sums:
        if bn_input is None:
            raise ValueError("dgrad_gpu: bn_sums needs bn_input")
        rows = l.pda_conv_dgrad_stat_rows(N, Hi, Wi, C, stride[0], stride[1], cfg)
        if (sums_buf is not None and rows > sums_buf.shape[1] and sums_buf.shape[2] == C and sums_buf.is_contiguous()
                and sums_buf.dtype == torch.float32 and not backends.deterministic):
            sums, slots = sums_buf, sums_buf.shape[1]
        else:
            sums = torch.empty((2, rows, C), device=dy.device, dtype=torch.float32)
    flags = (1 if (mask_bits and mask is not None) else 0) | (2 if res_s2 else 0)
    rc = l.pda_conv_igemm_flags(_lib.dt(dy), 1, dy.data_ptr(), w_crsk.data_ptr(), dx.data_ptr(), _lib.ptr(res),
                                _lib.ptr(mask), _lib.ptr(bn_input), _lib.ptr(bn_scale), _lib.ptr(bn_shift),
                                sums[0].data_ptr() if bn_sums else 0, sums[1].data_ptr() if bn_sums else 0, N, Ho, Wo, K,
                                Ho * Wo * K, Wo * K, K, Hi, Wi, C, R, S, stride[0], stride[1], pad[0], pad[1], 0, cfg,
                                slots, flags, _lib.stream())
    if rc == -8:
        # 16-bit dX whose channel count is not a multiple of 8: the epilogue extras only exist in the
        # vector write-out, so run the plain kernel and apply them with tensor ops (no sums: the caller's
        # BatchNorm backward then reduces by itself)
        dx = dgrad_gpu(dy, w_crsk, in_hw, stride, pad, cfg=cfg)
        if dx is None:
            return None
        if res is not None and res_s2:
            dx[:, ::2, ::2] += res
        elif res is not None:
            dx += res
        if mask is not None:
            dx *= unpack_mask_bits(mask, dx) if mask_bits else (mask > 0).to(dx.dtype)
        if bn_input is not None and bn_scale is not None:
            dx *= ((bn_input.float() * bn_scale + bn_shift) > 0).to(dx.dtype)
        return (dx, None) if bn_sums else dx
    if rc < 0:
        return None
    _lib.check(rc, "pda_conv_igemm(dgrad)")
    return (dx, sums) if bn_sums else dx


def unpack_mask_bits(bits: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """0/1 tensor shaped and typed like ``like`` from its sign-bit bytes (one byte per 16-byte vector)."""
    epc = 16 // like.element_size()
    sh = torch.arange(epc, device=bits.device, dtype=torch.uint8)
    return ((bits.view(-1, 1) >> sh) & 1).reshape(like.shape).to(like.dtype)


_WGRAD_WS: dict = {}
_WGRAD_WS_BYTES = int(os.environ.get("PDA_WGRAD_WORKSPACE_MB", "64")) << 20


def _wgrad_workspace(device) -> Optional[torch.Tensor]:
    """Scratch for the split-K partials of wgrad, one per (device, stream): launches on one stream are
    ordered, so they can share it.  ``PDA_WGRAD_WORKSPACE_MB=0`` selects the float-atomic reduction."""
    if _WGRAD_WS_BYTES <= 0:
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = torch.empty(_WGRAD_WS_BYTES // 4, device=device, dtype=torch.float32)
        _WGRAD_WS[key] = ws
    return ws


def wgrad_gpu(dy: torch.Tensor, x: torch.Tensor, R: int, S: int, stride, pad, cfg: int = 0, nsplit: int = 0,
              geom: Optional[tuple] = None, out: Optional[torch.Tensor] = None, workspace: bool = True,
              in_affine: Optional[tuple] = None):
    """fp32 dW in KRSC layout, shape [K,R,S,C] (accumulates into ``out`` when given).
    ``in_affine``: as in :func:`fprop_gpu` - the gradient is taken against ``relu(x * scale + shift)``."""
    l = _lib.lib()
    K = dy.shape[-1]
    if geom is None:
        N, Hi, Wi, Ci = x.shape
        Ho, Wo = dy.shape[1], dy.shape[2]
        sn, sh, sw = Hi * Wi * Ci, Wi * Ci, Ci
    else:
        N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, R, S = geom
    if cfg == 0 and nsplit == 0 and backends.benchmark:
        cfg = _tuned_cfg("wgrad", (dy.dtype, N, Hi, Wi, Ci, Ho, Wo, K, R, S, tuple(stride), tuple(pad), workspace,
                                   backends.deterministic, in_affine is not None, None if geom is None else tuple(geom)),
                         _WGRAD_CFGS + (_WGRAD_RING_CFGS if dy.element_size() == 2 and (
                             in_affine is None or (tuple(pad) == (0, 0) and R == 1 and S == 1)) else ()),
                         lambda c: wgrad_gpu(dy, x, R, S, stride, pad, c or 1 << 30, 0, geom, None, workspace,
                                             in_affine))
    if (cfg == 0 and nsplit == 0 and not backends.benchmark and geom is None and dy.element_size() == 2
            and (in_affine is None or (tuple(pad) == (0, 0) and R == 1 and S == 1))):
        # without the tuner: the LDS-DMA ring kernel, 256x128 tiles for the deep reductions over small images and
        # 128x128 otherwise (the best static choice over ResNet-50's layers: 6.6 against 7.95 ms per step for the
        # register-staged heuristic, 6.05 for the per-layer best; scripts/wgrad_cfg_totals.py)
        deep = Ho * Wo <= 196 and (R * S > 1 or (K >= 512 and R * S * Ci >= 1024))
        r = wgrad_gpu(dy, x, R, S, stride, pad, 68 if deep else 65, 0, geom, out, workspace, in_affine)
        if r is not None:
            return r
    cfg &= (1 << 30) - 1
    if backends.deterministic:
        cfg |= 32
    dw = out if out is not None else torch.zeros((K, R, S, Ci), device=dy.device, dtype=torch.float32)
    ws = _wgrad_workspace(dy.device) if workspace else None
    if in_affine is not None:
        if x.element_size() != 2:
            return None
        fn = l.pda_conv_wgrad_ring_xf if cfg & 64 else l.pda_conv_wgrad_xf
        if cfg & 64:
            cfg &= 39
        rc = fn(_lib.dt(dy), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), in_affine[0].data_ptr(),
                in_affine[1].data_ptr(), N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, K, R, S, stride[0], stride[1], pad[0],
                pad[1], cfg, nsplit, _lib.ptr(ws), 0 if ws is None else ws.numel() * 4, _lib.stream())
    elif cfg & 64:
        rc = l.pda_conv_wgrad_ring(_lib.dt(dy), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, Hi, Wi, Ci, sn, sh, sw,
                                   Ho, Wo, K, R, S, stride[0], stride[1], pad[0], pad[1], cfg & 39, nsplit,
                                   _lib.ptr(ws), 0 if ws is None else ws.numel() * 4, _lib.stream())
    else:
        rc = l.pda_conv_wgrad(_lib.dt(dy), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, Hi, Wi, Ci, sn, sh, sw,
                              Ho, Wo, K, R, S, stride[0], stride[1], pad[0], pad[1], cfg, nsplit,
                              _lib.ptr(ws), 0 if ws is None else ws.numel() * 4, _lib.stream())
    if rc < 0:
        return None
    _lib.check(rc, "pda_conv_wgrad")
    return dw


def bottleneck_tail_bwd(dz: torch.Tensor, y3: torch.Tensor, coef: torch.Tensor, x2: torch.Tensor,
                        in_affine: Optional[tuple], w_crsk: torch.Tensor, dw: Optional[torch.Tensor]):
    """BatchNorm backward apply + dgrad (+ wgrad) of a bottleneck's last 1x1 conv in one pass
    (``csrc/kernels/bottleneck_tail_bwd.hip``): ``dy3 = coef[0]*dz + coef[1]*y3 + coef[2]`` is formed per 64-pixel
    tile on chip, ``dx = (dy3 @ W) * (a2 > 0)`` with ``a2 = relu(x2*scale + shift)``.

    64 -> 256 channels: also ``dw += dy3^T @ a2`` (fp32 ``[256,1,1,64]``, accumulated with atomics); ``dy3`` is never
    stored.  128 -> 512 channels: ``dw`` is ``None`` and ``dy3`` is returned for a separate weight-gradient kernel.
    Returns ``(dx, sums, dy3 or None)`` with ``sums`` = fp32 ``[2, rows, Ci]`` partial sums of ``dx`` and ``dx*x2``
    (the reductions of the producing BatchNorm's backward), or ``None`` for other shapes.

    ``in_affine=None``: ``x2`` is the conv's input activation itself - ``dx = dy3 @ W`` unmasked, ``dw += dy3^T @ x2``,
    ``sums`` is ``None`` (the 1x1 conv + BatchNorm of a shortcut branch)."""
    l = _lib.lib()
    Co, Ci = y3.shape[-1], x2.shape[-1]
    M = y3.numel() // Co
    if ((Ci, Co) not in ((64, 256), (128, 512)) or y3.element_size() != 2 or dz.shape != y3.shape
            or dz.dtype != y3.dtype or x2.dtype != y3.dtype or x2.numel() != M * Ci or coef.dtype != torch.float32
            or coef.shape[-1] != Co or coef.shape[0] < 3 or not coef.is_contiguous()
            or tuple(w_crsk.shape) != (Ci, 1, 1, Co) or w_crsk.dtype != y3.dtype
            or not (dz.is_contiguous() and y3.is_contiguous() and x2.is_contiguous() and w_crsk.is_contiguous())):
        return None
    with_wgrad = Ci == 64
    if with_wgrad and (dw is None or dw.dtype != torch.float32 or dw.numel() != Co * Ci or not dw.is_contiguous()):
        return None
    rows = l.pda_bottleneck_tail_bwd_rows(M)
    dx = torch.empty_like(x2)
    dy = None if with_wgrad else torch.empty_like(y3)
    sums = None if in_affine is None else torch.empty((2, rows, Ci), device=y3.device, dtype=torch.float32)
    rc = l.pda_bottleneck_tail_bwd(_lib.dt(y3), dz.data_ptr(), y3.data_ptr(), coef.data_ptr(), x2.data_ptr(),
                                   None if in_affine is None else in_affine[0].data_ptr(),
                                   None if in_affine is None else in_affine[1].data_ptr(), w_crsk.data_ptr(),
                                   dx.data_ptr(), dw.data_ptr() if with_wgrad else None, _lib.ptr(dy), _lib.ptr(sums), M,
                                   Ci, Co, _lib.stream())
    if rc < 0:
        return None
    _lib.check(rc, "pda_bottleneck_tail_bwd")
    return dx, sums, dy


def stem_bwd_fused(dact: torch.Tensor, y0: torch.Tensor, coef: torch.Tensor, xp: torch.Tensor, geom: tuple,
                   stride_h: int) -> Optional[torch.Tensor]:
    """BatchNorm backward apply + weight gradient of the ResNet stem in one pass (``csrc/kernels/stem_bwd.hip``):
    ``dy0 = coef[0]*g + coef[1]*y0 + coef[2]`` with ``g = dact * (y0*coef[3] + coef[4] > 0)`` is formed per tile on
    chip and multiplied into the gathered operand rows of ``xp`` (``geom``: the stem's implicit-GEMM geometry,
    ``ops/fused.py::_stem_geometry``).  Returns the fp32 weight gradient ``[K, R, 1, 32]`` or ``None`` when the
    geometry is not the 7-row, 64-filter stem in a 16-bit type."""
    l = _lib.lib()
    N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, R, S = geom
    K = y0.shape[-1]
    M = y0.numel() // K
    if (K != 64 or R != 7 or S != 1 or Ci != 32 or y0.element_size() != 2 or dact.shape != y0.shape
            or dact.dtype != y0.dtype or xp.dtype != y0.dtype or coef.dtype != torch.float32
            or tuple(coef.shape) != (5, K) or not (dact.is_contiguous() and y0.is_contiguous() and xp.is_contiguous()
                                                    and coef.is_contiguous()) or M != N * Ho * Wo):
        return None
    dw = torch.zeros((K, R, 1, 32), device=y0.device, dtype=torch.float32)
    rc = l.pda_stem_bwd_fused(_lib.dt(y0), dact.data_ptr(), y0.data_ptr(), coef.data_ptr(), xp.data_ptr(),
                              dw.data_ptr(), M, K, R, Ho, Wo, sn, sh, sw, stride_h, _lib.stream())
    if rc < 0:
        return None
    _lib.check(rc, "pda_stem_bwd_fused")
    return dw


def krsc_to_crsk(w_krsc: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    l = _lib.lib()
    K, R, S, C = w_krsc.shape
    out = torch.empty((C, R, S, K), device=w_krsc.device, dtype=dtype)
    _lib.check(l.pda_krsc_to_crsk(_lib.dt(w_krsc), _lib.dt(dtype), w_krsc.data_ptr(), out.data_ptr(), K, R * S, C,
                                  _lib.stream()), "pda_krsc_to_crsk")
    return out


def direct_conv(mode: int, a: torch.Tensor, b: torch.Tensor, bias, out: torch.Tensor, dims, stride, pad) -> None:
    l = _lib.lib()
    N, Hi, Wi, Ci, Ho, Wo, Co, R, S = dims
    _lib.check(l.pda_direct_conv(_lib.dt(a), mode, a.data_ptr(), b.data_ptr(), _lib.ptr(bias), out.data_ptr(),
                                 N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride[0], stride[1], pad[0], pad[1],
                                 _lib.stream()), "pda_direct_conv")


def grad_sink(weight: torch.Tensor):
    """KRSC fp32 view of ``weight.grad`` that wgrad may accumulate into directly, or None.

    Opt-in per parameter (``_pda_direct_grad``, set by this package's DDP wrapper and fused
    optimizer): it skips the zero-fill + AccumulateGrad add per conv, but bypasses autograd's own
    gradient hooks, so ``_pda_grad_ready`` (if present) is called instead."""
    if not getattr(weight, "_pda_direct_grad", False):
        return None
    g = weight.grad
    if g is None or g.dtype != torch.float32 or not g.is_cuda or g.shape != weight.shape:
        return None
    v = g.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else None


def notify_grad_ready(weight: torch.Tensor) -> None:
    cb = getattr(weight, "_pda_grad_ready", None)
    if cb is not None:
        cb()


# --------------------------------------------------------------------------- weight caches
class WeightCache:
    """Low-precision KRSC copy and CRSK transpose of one fp32 master weight, refreshed
    only when the parameter has changed (torch version counter or our optimizer's
    generation counter)."""

    __slots__ = ("key_lp", "lp", "key_t", "t", "pinned_lp", "t_weight", "stats", "key_fold", "fold", "used_gen",
                 "t_layout", "__weakref__")

    STAT_SLOTS = 64

    def stats_buffer(self, channels: int, device) -> torch.Tensor:
        """Persistent, zero-on-entry ``[2, STAT_SLOTS, channels]`` accumulator for the BatchNorm
        statistics the conv epilogue produces (one per device: DataParallel replicas share this object)."""
        if _STAT_SLOTS_OFF:
            return None
        key = (str(device), channels)
        buf = self.stats.get(key)
        if buf is None:
            buf = torch.zeros((2, self.STAT_SLOTS, channels), device=device, dtype=torch.float32)
            self.stats[key] = buf
        return buf

    def __init__(self):
        self.stats = {}
        self.pinned_lp = None  # (dtype, tensor, torch version) kept current by the optimizer kernel (amp O2)
        self.key_lp = None
        self.lp = None
        self.key_fold = None   # inference: weights with the BatchNorm scale folded in, and the matching bias
        self.fold = None
        self.key_t = None
        self.t = None
        self.t_weight = None   # weakref to the parameter the transpose in ``t`` belongs to
        self.used_gen = 0      # optimizer generation in which the transpose was last asked for
        self.t_layout = None   # (shape, strides) of the parameter when ``t`` was built from its KRSC memory

    @staticmethod
    def _key(weight: torch.Tensor, dtype) -> tuple:
        return (weight.data_ptr(), weight._version, _PARAM_GENERATION, dtype)

    def low_precision(self, weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        w = krsc(weight.detach())
        if w.dtype == dtype:
            return w
        if self.pinned_lp is not None and self.pinned_lp[0] == dtype:
            half = self.pinned_lp[1]
            # stale when the master changed through torch (load_state_dict, init: version counter) or
            # through a raw kernel that is not the optimizer owning this copy (parameter broadcast of a
            # DistributedDataParallel / hvd.broadcast_parameters: generation counter)
            if self.pinned_lp[2] != weight._version or _PINNED_SYNCED.get(half.data_ptr()) != _PARAM_GENERATION:
                half.copy_(w)
                self.pinned_lp = (dtype, half, weight._version)
                _PINNED_SYNCED[half.data_ptr()] = _PARAM_GENERATION
            return half
        key = self._key(weight, dtype)
        if self.key_lp != key:
            self.lp = w.to(dtype)
            self.key_lp = key
        return self.lp

    def folded(self, weight: torch.Tensor, dtype: torch.dtype, versions, coeffs):
        """``(w * scale[k]) in dtype, shift``: eval-mode BatchNorm folded into the conv
        (``y = conv(x, w) * scale + shift == conv(x, w * scale) + shift``).  ``coeffs()`` returns the fp32
        ``(scale, shift)``; ``versions`` identifies the state they depend on - the pair is rebuilt only when
        that or the weight moved."""
        key = self._key(weight, dtype) + tuple(versions)
        if self.key_fold != key:
            scale, shift = coeffs()
            w = krsc(weight.detach()).float()
            self.fold = ((w * scale.view(-1, 1, 1, 1)).to(dtype).contiguous(), shift.float().contiguous())
            self.key_fold = key
        return self.fold

    def transposed(self, weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        self.used_gen = _PARAM_GENERATION
        key = self._key(weight, dtype)
        if self.key_t != key:
            K, C, R, S = weight.shape
            if (self.t is not None and self.t.dtype == dtype and self.t.shape == (C, R, S, K)
                    and self.t.device == weight.device):
                # a stale transpose of the same weight: the optimizer has stepped, so every conv of the
                # model is stale too - refresh all of them with one launch
                _refresh_transposes(dtype, weight.device)
                if self.key_t == key:
                    return self.t
            wk = krsc(weight.detach())
            self.t = krsc_to_crsk(wk, dtype)
            self.key_t = key
            self.t_weight = weakref.ref(weight)
            # the batched refresh reads the parameter's memory as KRSC: only while it keeps this exact layout
            self.t_layout = (tuple(weight.shape), tuple(weight.stride())) if wk.data_ptr() == weight.data_ptr() else None
            _T_CACHES.add(self)
        return self.t


_T_CACHES: "weakref.WeakSet[WeightCache]" = weakref.WeakSet()
_T_TABLES: dict = {}


def _refresh_transposes(dtype: torch.dtype, device) -> None:
    """Recompute, in one launch, every registered CRSK transpose of ``dtype`` on ``device`` that is stale."""
    todo = []
    for c in _T_CACHES:
        if c.used_gen + 8 < _PARAM_GENERATION:
            continue  # not asked for in the last few optimizer steps (a model that is no longer training): it
                      # refreshes on its own when it is used again, instead of costing every step of the others
        w = c.t_weight() if c.t_weight is not None else None
        if w is None or c.t is None or c.t.dtype != dtype or c.t.device != device or w.device != device:
            continue
        key = WeightCache._key(w, dtype)
        if c.key_t != key and c.t_layout == (tuple(w.shape), tuple(w.stride())):
            todo.append((c, w, key))
    if not todo:
        return
    todo.sort(key=lambda e: e[1].data_ptr())
    sig = (dtype, todo[0][1].dtype, tuple((w.data_ptr(), c.t.data_ptr()) for c, w, _ in todo))
    tab = _T_TABLES.get(device)
    if tab is None or tab[0] != sig:
        if any(w.dtype != todo[0][1].dtype for _, w, _ in todo):
            return  # mixed source dtypes: let the per-tensor path handle it
        meta, blk = [], []
        for i, (c, w, _) in enumerate(todo):
            K, C, R, S = w.shape
            meta.append([w.data_ptr(), c.t.data_ptr(), K, R * S, C])
            for rs in range(R * S):
                for kt in range((K + 63) // 64):
                    for ct in range((C + 63) // 64):
                        blk.append([i, rs, kt, ct])
        tab = (sig, torch.tensor(meta, dtype=torch.int64).to(device), torch.tensor(blk, dtype=torch.int32).to(device),
               len(blk))
        _T_TABLES[device] = tab
    l = _lib.lib()
    _lib.check(l.pda_krsc_to_crsk_batched(_lib.dt(todo[0][1]), _lib.dt(dtype), tab[1].data_ptr(), tab[2].data_ptr(),
                                          tab[3], _lib.stream()), "pda_krsc_to_crsk_batched")
    for c, w, key in todo:
        c.key_t = key


# --------------------------------------------------------------------------- autograd op
class _Conv2dFn(torch.autograd.Function):
    """Plain convolution (optional bias, optional ReLU applied in the bias pass) on the GPU kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, cache: WeightCache, relu: bool = False):
        x = x.contiguous()
        dtype = x.dtype
        K, C, R, S = weight.shape
        w_lp = cache.low_precision(weight, dtype)
        N, Hi, Wi, _ = x.shape
        Ho, Wo = out_size(Hi, R, stride[0], pad[0]), out_size(Wi, S, stride[1], pad[1])
        # bias and ReLU ride in the GEMM epilogue (no pass of their own over the output); K a multiple of 4 keeps the
        # epilogue's 16-byte bias loads aligned
        b32 = None if bias is None else bias.detach().float().contiguous()
        fuse = _FUSE_BIAS_RELU and (bias is not None or relu) and K % 8 == 0
        y = fprop_gpu(x, w_lp, stride, pad, relu=relu and fuse, bias=b32 if fuse else None)
        ctx.used_igemm = y is not None
        if y is None:
            y = torch.empty((N, Ho, Wo, K), device=x.device, dtype=dtype)
            direct_conv(0, x, w_lp, b32, y, (N, Hi, Wi, C, Ho, Wo, K, R, S), stride, pad)
            if relu:
                _lib.check(_lib.lib().pda_relu_fwd(_lib.dt(y), y.data_ptr(), y.data_ptr(), y.numel(), _lib.stream()),
                           "pda_relu_fwd")
        elif fuse:
            pass
        elif bias is not None:
            l = _lib.lib()
            _lib.check(l.pda_bias_add(_lib.dt(y), y.data_ptr(), b32.data_ptr(), y.data_ptr(), y.numel(), K,
                                      int(relu), _lib.stream()), "pda_bias_add")
        elif relu:
            _lib.check(_lib.lib().pda_relu_fwd(_lib.dt(y), y.data_ptr(), y.data_ptr(), y.numel(), _lib.stream()),
                       "pda_relu_fwd")
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.w_lp = w_lp
        ctx.cfg = (stride, pad, cache, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y_act = ctx.saved_tensors
        stride, pad, cache, has_bias = ctx.cfg
        dy = dy.contiguous()
        if y_act is not None:   # the ReLU that ran in the bias pass
            masked = torch.empty_like(dy)
            _lib.check(_lib.lib().pda_relu_bwd(_lib.dt(dy), dy.data_ptr(), y_act.data_ptr(), masked.data_ptr(),
                                               dy.numel(), _lib.stream()), "pda_relu_bwd")
            dy = masked
        K, C, R, S = weight.shape
        N, Hi, Wi, _ = x.shape
        Ho, Wo = dy.shape[1], dy.shape[2]
        dims = (N, Hi, Wi, C, Ho, Wo, K, R, S)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dgrad_gpu(dy, cache.transposed(weight, dy.dtype), (Hi, Wi), stride, pad) if ctx.used_igemm else None
            if dx is None:
                dx = torch.empty_like(x)
                direct_conv(1, dy, ctx.w_lp, None, dx, dims, stride, pad)
        if ctx.needs_input_grad[1]:
            sink = grad_sink(weight)
            dwk = wgrad_gpu(dy, x, R, S, stride, pad, out=sink) if ctx.used_igemm else None
            if dwk is None:
                dwk = sink if sink is not None else torch.zeros((K, R, S, C), device=dy.device, dtype=torch.float32)
                direct_conv(2, dy, x, None, dwk, dims, stride, pad)
            if sink is not None:
                notify_grad_ready(weight)
            else:
                dw = dwk.permute(0, 3, 1, 2).to(weight.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            l = _lib.lib()
            db = torch.empty(K, device=dy.device, dtype=torch.float32)
            _lib.check(l.pda_colsum(_lib.dt(dy), dy.data_ptr(), db.data_ptr(), N * Ho * Wo, K,
                                    int(backends.deterministic), _lib.stream()),
                       "pda_colsum")
        return dx, dw, db, None, None, None, None


def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """``a [M, K] @ b [N, K]^T (+ bias)`` on this package's any-shape MFMA GEMM (``csrc/kernels/gemm_anyk.hip``):
    no alignment or divisibility demands on K and N.  fp16 / bf16 / fp32 device matrices with unit inner stride."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1] and a.dtype == b.dtype
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    bf = None if bias is None else bias.detach().float().contiguous()
    _lib.check(_lib.lib().pda_gemm_anyk(_lib.dt(a), a.data_ptr(), b.data_ptr(), _lib.ptr(bf), out.data_ptr(), M, N, K,
                                        a.stride(0), b.stride(0), N, int(relu), _lib.stream()), "pda_gemm_anyk")
    return out


class _PointwiseGemmFn(torch.autograd.Function):
    """1x1 stride-1 convolution whose channel counts the implicit-GEMM fprop / dgrad kernels do not step
    through (16, 24, 58, 116, 144, 232 ... in the mobile architectures): forward and input gradient are plain
    ``[pixels, C] x [C, K]`` products on the any-shape MFMA GEMM of this package (:func:`gemm_nt`), the weight
    gradient - a tiny ``[K, C]`` matrix reduced over millions of pixels - on the split-K wgrad kernel (its ragged
    instantiation, which reads rows at element alignment, when the channel counts do not fill 16-byte vectors)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K, Cin = weight.shape[0], weight.shape[1]
        w_lp = weight.detach().reshape(K, Cin).to(x.dtype)
        y = gemm_nt(x.reshape(-1, Cin), w_lp, None if bias is None else bias.detach()).view(*x.shape[:-1], K)
        ctx.save_for_backward(x, w_lp)
        ctx.meta = (weight.dtype, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_lp = ctx.saved_tensors
        wdtype, has_bias = ctx.meta
        K, Cin = w_lp.shape
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_nt(dy.reshape(-1, K), w_lp.t().contiguous()).view(x.shape)
        if ctx.needs_input_grad[1]:
            vec = 16 // x.element_size()
            d = wgrad_gpu(dy, x.contiguous(), 1, 1, (1, 1), (0, 0)) if (K % vec == 0 and Cin % vec == 0) else None
            if d is None:
                # channel counts that do not fill 16-byte vectors (58, 116, 232 ...): the split-K wgrad kernel's
                # ragged instantiation reads the rows at element alignment - no transposed copies of the operands
                x2 = x.reshape(-1, Cin)
                if x2.stride(1) != 1:
                    x2 = x2.contiguous()
                d = torch.zeros((K, Cin), device=dy.device, dtype=torch.float32)
                _lib.check(_lib.lib().pda_conv_wgrad_ragged(_lib.dt(dy), dy.data_ptr(), x2.data_ptr(), d.data_ptr(),
                                                            x2.shape[0], Cin, K, x2.stride(0),
                                                            int(backends.deterministic), _lib.stream()),
                           "pda_conv_wgrad_ragged")
            dw = d.reshape(K, Cin, 1, 1).to(wdtype)
        if has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(K, device=dy.device, dtype=torch.float32)
            _lib.check(_lib.lib().pda_colsum(_lib.dt(dy), dy.data_ptr(), db.data_ptr(), dy.numel() // K, K,
                                             int(backends.deterministic), _lib.stream()), "pda_colsum")
        return dx, dw, db


def conv2d_ref(x: torch.Tensor, weight: torch.Tensor, bias, stride, pad) -> torch.Tensor:
    """CPU / reference path: NHWC in, NHWC out, stock PyTorch math."""
    y = F.conv2d(x.permute(0, 3, 1, 2), weight.to(x.dtype), None if bias is None else bias.to(x.dtype),
                 stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).contiguous()


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias=None, stride=(1, 1), pad=(0, 0),
           cache: Optional[WeightCache] = None, groups: int = 1, relu: bool = False) -> torch.Tensor:
    """NHWC convolution (+ ReLU with ``relu=True``: on the dense GPU path it rides in the bias pass).
    ``weight`` is logical [K,C/groups,R,S]."""
    if relu and (groups != 1 or not x.is_cuda):
        from .elementwise import relu as _relu
        return _relu(conv2d(x, weight, bias, stride, pad, cache, groups))
    if groups != 1:
        from .grouped import grouped_conv2d
        return grouped_conv2d(x, weight, bias, stride, pad, groups)
    if not x.is_cuda:
        return conv2d_ref(x, weight, bias, stride, pad)
    K, Cin, R, S = weight.shape
    # the implicit-GEMM kernels step through 64 bytes of channels: the input channels in fprop / wgrad, the
    # output channels in dgrad
    tiled_in, tiled_out = igemm_tiles(Cin, x.dtype), igemm_tiles(K, x.dtype)
    if (R, S) == (1, 1) and tuple(stride) == (1, 1) and tuple(pad) == (0, 0) and not (tiled_in and tiled_out):
        # a pointwise conv is a plain [pixels, Cin] x [Cin, K] GEMM: channel counts the implicit-GEMM kernels
        # do not tile (16, 24, 144, ... in MobileNet) go to the any-shape MFMA GEMM instead of the scalar fallback
        y = _PointwiseGemmFn.apply(x, weight, bias)
        if relu:
            from .elementwise import relu as _relu
            y = _relu(y)
        return y
    if K % 8 == 0 and not (tiled_in and (tiled_out or not x.requires_grad)):
        # an image (3 channels), a narrow feature map (16, 48 channels in 16-bit) or an odd filter count (208):
        # zero-pad the channels of the activation and of the filter up to the kernels' step and run on the
        # matrix cores - the padded FLOPs are cheaper than the scalar fallback by orders of magnitude (the
        # padding is differentiable: the gradients of the real channels come back through the slices).  A
        # fresh cache: the padded filter is a temporary, nothing about it may be remembered across calls.
        step = 64 // x.element_size()
        ec = -Cin % step
        ek = -K % step if x.requires_grad else 0
        y = _Conv2dFn.apply(F.pad(x, (0, ec)) if ec else x, F.pad(weight, (0, 0, 0, 0, 0, ec, 0, ek)),
                            F.pad(bias, (0, ek)) if (bias is not None and ek) else bias, tuple(stride), tuple(pad),
                            WeightCache(), relu)
        return y[..., :K].contiguous() if ek else y
    return _Conv2dFn.apply(x, weight, bias, tuple(stride), tuple(pad), cache or WeightCache(), relu)


def igemm_tiles(channels: int, dtype: torch.dtype) -> bool:
    """Whether the MFMA implicit-GEMM kernels can step through ``channels`` input channels: their K step is
    64 bytes of one filter tap (32 16-bit or 16 fp32 channels)."""
    return channels % (64 // torch.empty((), dtype=dtype).element_size()) == 0
