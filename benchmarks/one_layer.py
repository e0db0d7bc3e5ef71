#!/usr/bin/env python
"""Run ONE implicit-GEMM launch shape a few times (for rocprofv3 --kernel-trace / --pmc runs).

    python benchmarks/one_layer.py --op dgrad --h 56 --cin 256 --cout 512 --k 1 --stride 2 --batch 128 --cfg 0
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd.ops import conv as C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="fprop", choices=["fprop", "dgrad", "wgrad"])
    ap.add_argument("--h", type=int, default=56)
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--nsplit", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--warmup", type=int, default=0, help="untimed launches first (0 for profiler runs)")
    ap.add_argument("--no-stats", action="store_true", help="fprop without the BatchNorm statistics epilogue")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype]
    dev = "cuda"
    s, k, pad = a.stride, a.k, a.k // 2
    ho = (a.h + 2 * pad - k) // s + 1
    x = torch.randn(a.batch, a.h, a.h, a.cin, device=dev).to(dt)
    w = (torch.randn(a.cout, k, k, a.cin, device=dev) / (k * k * a.cin) ** 0.5).to(dt)
    wt = C.krsc_to_crsk(w, dt)
    dy = torch.randn(a.batch, ho, ho, a.cout, device=dev).to(dt)
    res = torch.randn_like(x) if a.res else None
    dw = torch.zeros(a.cout, k, k, a.cin, device=dev)
    def run():
        if a.op == "fprop":
            C.fprop_gpu(x, w, (s, s), (pad, pad), stats=not a.no_stats, cfg=a.cfg)
        elif a.op == "dgrad":
            C.dgrad_gpu(dy, wt, (a.h, a.h), (s, s), (pad, pad), res=res, cfg=a.cfg)
        else:
            C.wgrad_gpu(dy, x, k, k, (s, s), (pad, pad), cfg=a.cfg, nsplit=a.nsplit, out=dw)

    for _ in range(a.warmup):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{a.op} cfg={a.cfg}: {e0.elapsed_time(e1) * 1e3 / a.iters:.1f} us/iter after {a.warmup} warm-up launches")


if __name__ == "__main__":
    main()
