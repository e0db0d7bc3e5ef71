#!/usr/bin/env python
"""What normalise-on-load costs per kernel: fprop and wgrad of the ResNet-50 conv2 / conv3 shapes with and without
``in_affine`` (batch 400, fp16), best tile of the tuner's candidates for each.

    python benchmarks/xf_cost.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd.ops import conv as C


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def best(fn, cfgs):
    ts = []
    for c in cfgs:
        if fn(c) is None:
            continue
        ts.append((timeit(lambda: fn(c)), c))
    return min(ts)


def main():
    B = int(os.environ.get("XF_BATCH", "400"))
    dev = "cuda"
    shapes = [(56, 64, 64, 3), (28, 128, 128, 3), (14, 256, 256, 3), (7, 512, 512, 3),
              (56, 64, 256, 1), (28, 128, 512, 1), (14, 256, 1024, 1), (7, 512, 2048, 1)]
    tot = {"f": 0.0, "fx": 0.0, "w": 0.0, "wx": 0.0}
    mult = {56: 3, 28: 4, 14: 6, 7: 3}
    for h, ci, co, k in shapes:
        pad = k // 2
        x = torch.randn(B, h, h, ci, device=dev).half()
        w = (torch.randn(co, k, k, ci, device=dev) / (k * k * ci) ** 0.5).half()
        dy = torch.randn(B, h, h, co, device=dev).half()
        sc = (torch.rand(ci, device=dev) + 0.5).half()
        sf = (torch.randn(ci, device=dev) * 0.2).half()
        dw = torch.zeros(co, k, k, ci, device=dev)
        f = best(lambda c: C.fprop_gpu(x, w, (1, 1), (pad, pad), stats=True, cfg=c), C._FPROP_CFGS)
        fx = best(lambda c: C.fprop_gpu(x, w, (1, 1), (pad, pad), stats=True, cfg=c, in_affine=(sc, sf)), C._FPROP_XF_CFGS)
        wg = best(lambda c: C.wgrad_gpu(dy, x, k, k, (1, 1), (pad, pad), cfg=c, out=dw), C._WGRAD_CFGS)
        wx = best(lambda c: C.wgrad_gpu(dy, x, k, k, (1, 1), (pad, pad), cfg=c, out=dw, in_affine=(sc, sf)), C._WGRAD_CFGS)
        n = mult[h] - (1 if k == 3 and h != 56 else 0)   # stride-2 3x3 of a stage's first block is not one of these shapes
        tot["f"] += f[0] * n; tot["fx"] += fx[0] * n; tot["w"] += wg[0] * n; tot["wx"] += wx[0] * n
        print(f"{h:3d} {ci:4d}->{co:4d} k{k} x{n}: fprop {f[0]:6.1f} (cfg {f[1]}) -> {fx[0]:6.1f} (cfg {fx[1]})   "
              f"wgrad {wg[0]:6.1f} (cfg {wg[1]}) -> {wx[0]:6.1f} (cfg {wx[1]})", flush=True)
    print("per step (us):", {k: round(v) for k, v in tot.items()})


if __name__ == "__main__":
    main()
