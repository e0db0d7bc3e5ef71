#!/usr/bin/env python
"""Pointwise (1x1) forward layers of ResNet-50 at the benchmark batch: the A-stationary kernel (cfg 33) against the
best tiled configuration, plain and with the input normalised on load, with the statistics epilogue.

    python benchmarks/pw_bench.py --batch 400
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd.ops import conv as C
from benchmarks.conv_bench import timeit

LAYERS = [(56, 64, 256, 1), (28, 128, 512, 1), (14, 256, 1024, 1), (7, 512, 2048, 1), (56, 256, 512, 2),
          (28, 512, 1024, 2), (14, 1024, 2048, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=400)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cfgs", default="0,1,9,12,33")
    a = ap.parse_args()
    for h, ci, co, s in LAYERS:
        x = torch.randn(a.batch, h, h, ci, device="cuda").half()
        w = (torch.randn(co, 1, 1, ci, device="cuda") / ci ** 0.5).half()
        sc = (torch.rand(ci, device="cuda") + 0.5).half()
        sf = (torch.rand(ci, device="cuda") - 0.5).half()
        ho = h // s
        mb = (x.numel() // (s * s) + a.batch * ho * ho * co) * 2 / 1e6
        gf = 2.0 * a.batch * ho * ho * ci * co / 1e9
        out = []
        for aff in (None, (sc, sf)):
            for cfg in [int(c) for c in a.cfgs.split(",")]:
                if C.fprop_gpu(x, w, (s, s), (0, 0), stats=True, cfg=cfg, in_affine=aff) is None:
                    continue
                t = timeit(lambda: C.fprop_gpu(x, w, (s, s), (0, 0), stats=True, cfg=cfg, in_affine=aff), a.iters)
                out.append(f"{'xf' if aff else 'pl'}{cfg}:{t:.0f}")
        print(f"{h:3d} {ci:5d}->{co:5d} s{s}  {mb:6.0f} MB (floor {mb / 5:5.0f} us)  {gf:5.1f} GF | " + " ".join(out), flush=True)


if __name__ == "__main__":
    main()
