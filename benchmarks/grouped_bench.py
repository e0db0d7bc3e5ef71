"""ResNeXt-50 (32x4d) grouped 3x3 layers: the 32-channel-chunk kernels against the dense block-diagonal fallback.
Prints, per layer and op, microseconds and the useful TFLOP/s (2 * M * 9 * Cg * C)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch_distributed_amd.ops import grouped as G  # noqa: E402

LAYERS = [(56, 128, 4, 1), (56, 256, 8, 2), (28, 256, 8, 1), (28, 512, 16, 2), (14, 512, 16, 1), (14, 1024, 32, 2),
          (7, 1024, 32, 1)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--dense", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tot = {"chunk": 0.0, "dense": 0.0}
    for hw, ch, cg, stride in LAYERS:
        groups = ch // cg
        x = torch.randn(args.batch, hw, hw, ch, device=dev, dtype=torch.float16)
        w = torch.randn(ch, cg, 3, 3, device=dev) * 0.1
        ho = (hw + 2 - 3) // stride + 1
        dy = torch.randn(args.batch, ho, ho, ch, device=dev, dtype=torch.float16)
        flop = 2.0 * args.batch * ho * ho * 9 * cg * ch
        io = (x.numel() + dy.numel()) * 2
        line = f"{hw:3d} {ch:5d} Cg {cg:2d} s{stride} | io floor {io / 5e12 * 1e6:6.1f} us |"
        for name, fn_cls in (("chunk", G._ChunkedGroupedConvFn), ("dense", G._GroupedConvFn)):
            if name == "dense" and not args.dense:
                continue
            extra = () if name == "chunk" else (groups,)
            xr = x.clone().requires_grad_(True)
            wr = w.clone().requires_grad_(True)
            t_f = timeit(lambda: fn_cls.apply(x, w, None, (stride, stride), (1, 1), *extra))
            y = fn_cls.apply(xr, wr, None, (stride, stride), (1, 1), *extra)
            t_b = timeit(lambda: torch.autograd.grad(y, (xr, wr), dy, retain_graph=True))
            tot[name] += t_f + t_b
            line += f" {name}: fwd {t_f:7.1f} us ({flop / t_f / 1e6:6.1f} TF) bwd {t_b:7.1f} us |"
        print(line, flush=True)
    print(f"totals per step (x blocks not counted): chunk {tot['chunk'] / 1e3:.2f} ms, dense {tot['dense'] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
