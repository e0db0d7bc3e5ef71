#!/usr/bin/env python
"""Backward of the ResNet stem at the benchmark batch: the fused kernel (BatchNorm backward apply + weight gradient,
csrc/kernels/stem_bwd.hip) against the two separate kernels.

    python benchmarks/stem_bwd_bench.py --batch 400
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd.ops import conv as C, fused, norm as Nm
from benchmarks.conv_bench import timeit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=400)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    N, H, W = a.batch, 224, 224
    Hp, Wp, Ho, Wo, geom = fused._stem_geometry(N, H, W, 7, 7, (2, 2), (3, 3))
    dt = torch.float16
    xp = fused.input_to_nhwc(torch.randn(N, 3, H, W, device="cuda"), dt, 4, (3, 3), Wp)
    y = torch.randn(N, Ho, Wo, 64, device="cuda").to(dt)
    dact = torch.randn(N, Ho, Wo, 64, device="cuda").to(dt)
    coef = torch.randn(5, 64, device="cuda")
    t_f = timeit(lambda: C.stem_bwd_fused(dact, y, coef, xp, geom, 2), a.iters)
    dy = Nm.apply_backward_coefs(dact, y, coef, True)
    t_a = timeit(lambda: Nm.apply_backward_coefs(dact, y, coef, True), a.iters)
    t_w = timeit(lambda: C.wgrad_gpu(dy, xp, 7, 1, (2, 1), (0, 0), geom=geom, cfg=66), a.iters)
    print(f"fused {t_f:.0f} us | apply {t_a:.0f} us + wgrad(ring 64x256) {t_w:.0f} us = {t_a + t_w:.0f} us", flush=True)


if __name__ == "__main__":
    main()
