#!/usr/bin/env python
"""Linear layers of the model zoo (small batch, deep K): this package's GEMM kernels vs the library GEMM behind
torch.nn.functional.linear, forward + backward, fp16.

    python benchmarks/linear_bench.py --batch 128
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from pytorch_distributed_amd import backends, nn as pnn


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--frozen", action="store_true",
                    help="weights never change between iterations (default: every iteration starts after an "
                         "optimizer step, so per-weight caches are rebuilt as they are in training)")
    ap.add_argument("--shapes", default="9216x4096,25088x4096,4096x4096,4096x1000,2048x1000,1024x1000")
    ap.add_argument("--direct-grad", action="store_true",
                    help="weight gradients accumulate into a persistent fp32 .grad, as under this package's optimizer")
    a = ap.parse_args()
    from pytorch_distributed_amd.ops import conv as C
    dt = torch.float16 if a.precision == "fp16" else torch.bfloat16
    backends.benchmark = True
    dev = "cuda"
    for k, n in [tuple(int(v) for v in sh.split("x")) for sh in a.shapes.split(",")]:
        lin = pnn.Linear(k, n).to(dev)              # fp32 master weights, 16-bit activations (amp O2)
        x = torch.randn(a.batch, k, device=dev, dtype=dt, requires_grad=True)
        w16 = lin.weight.detach().to(dt).requires_grad_(True)
        b16 = lin.bias.detach().to(dt).requires_grad_(True)
        g = torch.randn(a.batch, n, device=dev, dtype=dt)
        copies = []
        if a.direct_grad:
            # what amp.initialize(O2) + FusedSGD set up: gradients accumulate into a persistent fp32 .grad and the
            # optimizer kernel keeps a 16-bit copy of the weight current
            from pytorch_distributed_amd.ops.linear import LinearCache
            lin.weight.grad = torch.zeros(n, k, device=dev, dtype=torch.float32)
            lin.weight._pda_direct_grad = True
            cache = lin.__dict__["_pda_cache"] = LinearCache()
            half = lin.weight.detach().to(dt)
            cache.pinned_lp = (dt, half, lin.weight._version)
            copies = [half]

        def ours_fwd():
            if not a.frozen:
                C.bump_param_generation(copies)
            with torch.no_grad():
                return lin(x)

        def lib_fwd():
            with torch.no_grad():
                return F.linear(x, w16, b16)

        def ours():
            if not a.frozen:
                C.bump_param_generation(copies)
            if not a.direct_grad:
                lin.weight.grad = None
            y = lin(x)
            y.backward(g)

        def lib():
            w16.grad = None
            y = F.linear(x, w16, b16)
            y.backward(g)
        f_o, f_l, t_o, t_l = timeit(ours_fwd), timeit(lib_fwd), timeit(ours), timeit(lib)
        fl = 3 * 2.0 * a.batch * k * n
        print(f"Linear {k:6d} -> {n:5d}, batch {a.batch}: forward ours {f_o:7.1f} us  library {f_l:7.1f} us | "
              f"forward+backward ours {t_o:7.1f} us ({fl / t_o / 1e6:6.1f} TF)  library {t_l:7.1f} us ({fl / t_l / 1e6:6.1f} TF)",
              flush=True)


if __name__ == "__main__":
    main()
