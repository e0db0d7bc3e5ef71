"""ctypes binding of ``libpda_kernels.so`` (the gfx950 HIP kernel library).

The library is built in-tree by ``pytorch_distributed_amd._build`` and loaded
lazily the first time a device op runs.  A missing library on a machine that has
a GPU is a hard error (no silent fallback to stock PyTorch kernels): the CPU
reference path in this package exists only for tensors that live on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_LIB = None
_LIB_PATH = Path(os.environ.get("PDA_KERNELS_LIB") or
                 Path(__file__).resolve().parent.parent / "lib" / "libpda_kernels.so")

F32, F16, BF16, U8 = 0, 1, 2, 3
_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16, torch.uint8: U8}

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_d = ctypes.c_double
c_ull = ctypes.c_ulonglong

_SIGS = {
    "pda_conv_igemm_bm": [c_i, c_i, c_i, c_i, c_i],
    "pda_conv_dgrad_stat_rows": [c_i, c_i, c_i, c_i, c_i, c_i, c_i],
    "pda_conv_igemm": [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll,
                       c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_conv_wgrad_ragged": [c_i, c_p, c_p, c_p, c_ll, c_i, c_i, c_ll, c_i, c_p],
    "pda_gemm_anyk": [c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_p],
    "pda_gemm_splitk": [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_conv_igemm_xf": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll,
                          c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_conv_wgrad_xf": [c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i,
                          c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p],
    "pda_conv_wgrad_ring_xf": [c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i,
                          c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p],
    "pda_bn_apply_mask": [c_i, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p],
    "pda_bn_apply_res_affine": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p],
    "pda_conv_igemm_maskbits": [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll,
                                c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_conv_igemm_flags": [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll,
                             c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_bn_lp_tables": [c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "pda_conv_wgrad": [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i,
                       c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p],
    "pda_stem_bwd_fused": [c_i, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_p],
    "pda_grouped_conv3x3": [c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_conv3x3_c64_rows": [c_i, c_i, c_i, c_i],
    "pda_conv3x3_c64": [c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_conv3x3_c64_dgrad_masked": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_grouped_chunk_filters": [c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "pda_grouped_chunk_diag": [c_i, c_p, c_p, c_i, c_i, c_p],
    "pda_grouped_conv3x3_wgrad": [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_bottleneck_tail_bwd_rows": [c_ll],
    "pda_bottleneck_tail_bwd": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p],
    "pda_conv_pw_bm": [c_i],
    "pda_conv_pw_fprop": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_i, c_i,
                          c_i, c_i, c_i, c_p],
    "pda_conv_pw_fprop_mat": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p],
    "pda_conv_wgrad_ring": [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i,
                       c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p],
    "pda_krsc_to_crsk":[c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p],
    "pda_krsc_to_crsk_batched": [c_i, c_i, c_p, c_p, c_i, c_p],
    "pda_direct_conv": [c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                        c_i, c_i, c_p],
    "pda_avgpool": [c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_depthwise_conv": [c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                           c_i, c_i, c_p],
    "pda_bn_num_partials": [c_ll, c_i, c_i],
    "pda_bn_stats_partial": [c_i, c_p, c_ll, c_i, c_p, c_p, c_p],
    "pda_bn_reduce_partials": [c_p, c_p, c_i, c_i, c_p, c_i, c_p],
    "pda_bn_compress_rows": [c_i, c_i],
    "pda_bn_compress_partials": [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p],
    "pda_bn_finalize": [c_p, c_p, c_i, c_i, c_d, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "pda_bn_finalize_lp": [c_p, c_p, c_i, c_i, c_d, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "pda_bn_eval_coeffs": [c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "pda_bn_apply_cap": [c_i, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_f, c_p],
    "pda_bn_bwd_reduce_cap": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_f, c_p, c_p, c_i, c_p],
    "pda_bn_bwd_apply_cap": [c_i, c_p, c_p, c_p, c_p, c_ll, c_i, c_f, c_p, c_p, c_p],
    "pda_bn_apply": [c_i, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p],
    "pda_bn_bwd_reduce": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p, c_i, c_p],
    "pda_bn_bwd_finalize": [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_d, c_p, c_p, c_p, c_i, c_i, c_p],
    "pda_bn_bwd_coefs": [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_d, c_p, c_p],
    "pda_bn_bwd_apply": [c_i, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p, c_p],
    "pda_maxpool_fwd": [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "pda_maxpool_bwd": [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "pda_gap_fwd": [c_i, c_p, c_p, c_i, c_i, c_i, c_p],
    "pda_gap_bwd": [c_i, c_p, c_p, c_i, c_i, c_i, c_p],
    "pda_softmax_ce": [c_i, c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p],
    "pda_nll": [c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "pda_scale_by_devscalar": [c_i, c_p, c_p, c_ll, c_p],
    "pda_mt_chunk": [],
    "pda_mt_scale": [c_i, c_i, c_p, c_p, c_i, c_f, c_p, c_i, c_p, c_p],
    "pda_mt_axpby": [c_i, c_i, c_i, c_p, c_p, c_i, c_f, c_f, c_p, c_p],
    "pda_mt_l2norm": [c_i, c_p, c_p, c_i, c_p, c_i, c_p],
    "pda_sqrt_inplace": [c_p, c_i, c_p],
    "pda_mt_sgd": [c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_f, c_p],
    "pda_scaler_update": [c_p, c_p, c_f, c_f, c_i, c_f, c_f, c_p],
    "pda_nchw_to_nhwc": [c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_p],
    "pda_relu_fwd": [c_i, c_p, c_p, c_ll, c_p],
    "pda_relu_bwd": [c_i, c_p, c_p, c_p, c_ll, c_p],
    "pda_relu_cap_fwd": [c_i, c_p, c_p, c_ll, ctypes.c_float, c_p],
    "pda_relu_cap_bwd": [c_i, c_p, c_p, c_p, c_ll, ctypes.c_float, c_p],
    "pda_add": [c_i, c_p, c_p, c_p, c_ll, c_i, c_p],
    "pda_add_n": [c_i, c_p, c_i, c_p, c_ll, c_p],
    "pda_adasum_dots": [c_i, c_p, c_p, c_ll, c_p, c_p],
    "pda_adasum_combine": [c_i, c_p, c_p, c_p, c_ll, c_p, c_i, c_i, c_p],
    "pda_peer_bcast": [c_p, c_p, c_i, c_ll, c_p],
    "pda_peer_enable": [c_i, c_i],
    "pda_bias_add": [c_i, c_p, c_p, c_p, c_ll, c_i, c_i, c_p],
    "pda_colsum": [c_i, c_p, c_p, c_ll, c_i, c_i, c_p],
    "pda_dropout": [c_i, c_p, c_p, c_ll, c_f, c_ull, c_ull, c_i, c_ll, c_i, c_p, c_p],
}


class KernelLibraryMissing(RuntimeError):
    pass


def available() -> bool:
    return _LIB_PATH.exists()


def lib():
    """Load (once) and return the kernel library."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not _LIB_PATH.exists():
        if os.environ.get("PDA_AUTOBUILD", "0") == "1":
            from .. import _build
            _build.build_kernels()
        else:
            raise KernelLibraryMissing(
                f"{_LIB_PATH} not found: run `python -m pytorch_distributed_amd._build` "
                "(or __graft_entry__.build()) before using device ops")
    l = ctypes.CDLL(str(_LIB_PATH))
    for name, sig in _SIGS.items():
        fn = getattr(l, name)
        fn.argtypes = sig
        fn.restype = c_i
    _LIB = l
    return l


def dt(t) -> int:
    d = t if isinstance(t, torch.dtype) else t.dtype
    try:
        return _DT[d]
    except KeyError:
        raise TypeError(f"unsupported dtype {d} for a pytorch_distributed_amd device op") from None


def ptr(t):
    return 0 if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of the current HIP stream of the current device.  Called once per kernel launch (~2000 times per
    ResNet-50 step): the two C entry points take ~0.3 us, ``torch.cuda.current_stream().cuda_stream`` builds a Python
    Stream object first and took 9 us - a tenth of the step's host time."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, name: str) -> None:
    if rc != 0:
        kind = "unsupported shape/dtype" if rc < 0 else "HIP error"
        raise RuntimeError(f"{name} failed: {kind} (code {rc})")
