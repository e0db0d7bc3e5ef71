#!/usr/bin/env python
"""Per-layer cost of taking the BatchNorm-backward sums in the dgrad epilogue versus the stand-alone
reduce kernel, on the ResNet-50 layer shapes (random operands, device-event timing).

    python benchmarks/dgrad_fusion_bench.py --batch 256
columns: dgrad alone | dgrad + sums (+mask recompute) | separate bn_bwd_reduce over the dgrad output | saving
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from benchmarks.conv_bench import resnet50_shapes, timeit
from pytorch_distributed_amd.ops import _lib, conv as C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--detail", action="store_true", help="also time the epilogue extras one at a time")
    ap.add_argument("--layer", default="", help="only this layer: H,Cin,Cout,k,s")
    ap.add_argument("--cfgs", default="0", help="tile cfgs to time for the fused variant")
    a = ap.parse_args()
    dt, dev = torch.float16, "cuda"
    l = _lib.lib()
    tot = [0.0, 0.0, 0.0]
    for (h, ci, co, k, s), count in resnet50_shapes().items():
        if a.layer and (h, ci, co, k, s) != tuple(int(v) for v in a.layer.split(",")):
            continue
        pad, ho = k // 2, h // s
        w = (torch.randn(co, k, k, ci, device=dev) / (k * k * ci) ** 0.5).to(dt)
        wt = C.krsc_to_crsk(w, dt)
        dy = torch.randn(a.batch, ho, ho, co, device=dev).to(dt)
        c = torch.randn(a.batch, h, h, ci, device=dev).to(dt)
        sc = torch.randn(ci, device=dev)
        sf = torch.randn(ci, device=dev)
        t0 = timeit(lambda: C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad)), a.iters)
        t1s = []
        for cfg in [int(v) for v in a.cfgs.split(",")]:
            t1s.append((timeit(lambda: C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad), bn_input=c, bn_scale=sc,
                                                   bn_shift=sf, bn_sums=True, cfg=cfg), a.iters), cfg))
        t1 = t1s[0][0]
        if a.detail:
            tr = timeit(lambda: C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad), res=c), a.iters)
            tm = timeit(lambda: C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad), bn_input=c, bn_scale=sc, bn_shift=sf),
                        a.iters)
            tms = timeit(lambda: C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad), mask=c, bn_input=c, bn_sums=True),
                         a.iters)
            print(f"      +res only {tr:6.1f}   +bn_input/remask, no sums {tm:6.1f}   +mask tensor +sums {tms:6.1f}")
        dx = C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad))
        M = dx.numel() // ci
        n = l.pda_bn_num_partials(M, ci, _lib.dt(dx))
        part = torch.empty((2, n, ci), device=dev)
        t2 = timeit(lambda: l.pda_bn_bwd_reduce(_lib.dt(dx), dx.data_ptr(), c.data_ptr(), 0, sf.data_ptr(), sf.data_ptr(),
                                                sc.data_ptr(), sf.data_ptr(), M, ci, 1, part[0].data_ptr(),
                                                part[1].data_ptr(), 0, _lib.stream()), a.iters)
        print(f"H{h:3d} {ci:4d}->{co:4d} k{k} s{s} x{count}: dgrad {t0:6.1f}  +sums {t1:6.1f}  reduce {t2:6.1f}  "
              f"saving {t0 + t2 - t1:+6.1f} us  (dX channels {ci}, gemm K {co * k * k})  "
              + " ".join(f"{c_}:{t_:.0f}" for t_, c_ in t1s), flush=True)
        tot[0] += t0 * count
        tot[1] += t1 * count
        tot[2] += t2 * count
    print("totals (ms): dgrad %.2f  dgrad+sums %.2f  reduce %.2f" % tuple(v / 1e3 for v in tot))


if __name__ == "__main__":
    main()
