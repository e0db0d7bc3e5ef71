#!/usr/bin/env python
"""Per-layer micro-benchmark of the implicit-GEMM kernels on the ResNet-50 layer shapes:
fprop / dgrad / wgrad time, achieved TFLOP/s and the memory-traffic floor, for one or several tile
configurations.  Random (non-zero) operands, device-event timing, many iterations per measurement.

    python benchmarks/conv_bench.py --batch 128 --cfgs 0 --out gpurun_out/conv_bench.csv
"""
import argparse
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd.ops import conv as C


def resnet50_shapes():
    shapes = []
    h, inpl = 56, 64
    for pl, nb, st in [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]:
        for b in range(nb):
            s = st if b == 0 else 1
            shapes.append((h, inpl, pl, 1, 1))
            shapes.append((h, pl, pl, 3, s))
            shapes.append((h // s, pl, pl * 4, 1, 1))
            if b == 0:
                shapes.append((h, inpl, pl * 4, 1, s))
            inpl, h = pl * 4, h // s
    uniq = {}
    for s in shapes:
        uniq[s] = uniq.get(s, 0) + 1
    return uniq


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--cfgs", default="0", help="comma separated igemm tile cfgs (0 auto, 1 128x128, 2 128x64, 3 64x64)")
    ap.add_argument("--wcfgs", default="0", help="wgrad cfgs (0 auto, 1 128x128, 2 64x64, 3 64x128, +8 = 32-pixel steps; 12 256x128, 13 128x256; LDS-DMA ring: 65 128x128, 68 256x128, 69 128x256, 70 128x128 64-pixel steps)")
    ap.add_argument("--nsplits", default="0")
    ap.add_argument("--wblocks", default="", help="wgrad: target workgroup counts (overrides --nsplits)")
    ap.add_argument("--only", default="fdw", help="which of f(prop) d(grad) w(grad) to time")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-stats", action="store_true", help="fprop without the BatchNorm statistics epilogue")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype]
    es = 2 if dt != torch.float32 else 4
    dev = "cuda"
    rows = []
    tot = {"fprop": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for (h, ci, co, k, s), count in resnet50_shapes().items():
        pad = k // 2
        ho = h // s
        x = torch.randn(a.batch, h, h, ci, device=dev).to(dt)
        w = (torch.randn(co, k, k, ci, device=dev) / (k * k * ci) ** 0.5).to(dt)
        wt = C.krsc_to_crsk(w, dt)
        dy = torch.randn(a.batch, ho, ho, co, device=dev).to(dt)
        M = a.batch * ho * ho
        flops = 2.0 * M * co * ci * k * k
        io_f = (x.numel() + M * co) * es
        best = {}
        for cfg in [int(c) for c in a.cfgs.split(",")]:
            # (a configuration that does not cover the shape returns None at once: not a measurement)
            if "f" in a.only and C.fprop_gpu(x, w, (s, s), (pad, pad), stats=not a.no_stats, cfg=cfg) is not None:
                t = timeit(lambda: C.fprop_gpu(x, w, (s, s), (pad, pad), stats=not a.no_stats, cfg=cfg), a.iters)
                best.setdefault("fprop", []).append((t, cfg))
            if "d" in a.only and cfg not in (33, 34) and C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad), cfg=cfg) is not None:
                t = timeit(lambda: C.dgrad_gpu(dy, wt, (h, h), (s, s), (pad, pad), cfg=cfg), a.iters)
                best.setdefault("dgrad", []).append((t, cfg))
        dw = torch.zeros(co, k, k, ci, device=dev)
        for wc in ([int(c) for c in a.wcfgs.split(",")] if "w" in a.only else []):
            ktot = k * k * ci
            tile = wc & 7
            if tile == 0:
                tile = 1 if (co % 128 == 0 and ktot % 128 == 0) else 2
            tlm, tln = {1: (128, 128), 2: (64, 64) if wc < 64 else (64, 256), 3: (64, 128), 4: (256, 128), 5: (128, 256), 6: (128, 128)}[tile]
            tiles = -(-co // tlm) * -(-ktot // tln)
            splits = [int(c) for c in a.nsplits.split(",")]
            if a.wblocks:
                splits = sorted({max(1, round(int(b) / tiles)) for b in a.wblocks.split(",")})
            for ns in splits:
                if C.wgrad_gpu(dy, x, k, k, (s, s), (pad, pad), cfg=wc, nsplit=ns, out=dw) is None:
                    continue
                t = timeit(lambda: C.wgrad_gpu(dy, x, k, k, (s, s), (pad, pad), cfg=wc, nsplit=ns, out=dw), a.iters)
                best.setdefault("wgrad", []).append((t, (wc, ns * tiles if a.wblocks else ns)))
        line = {"H": h, "Cin": ci, "Cout": co, "k": k, "s": s, "count": count, "GFLOP": round(flops / 1e9, 2),
                "io_floor_us": round(io_f / 5e12 * 1e6, 1)}
        for op in ("fprop", "dgrad", "wgrad"):
            if op not in best:
                continue
            t, cfg = min(best[op], key=lambda v: v[0])
            line[f"{op}_us"] = round(t, 1)
            line[f"{op}_TF"] = round(flops / t / 1e6, 1)
            line[f"{op}_cfg"] = str(cfg)
            line[f"{op}_all"] = " ".join(f"{c}:{tt:.0f}" for tt, c in best[op])
            tot[op] += t * count
        rows.append(line)
        print(line, flush=True)
    print("per-step totals (ms):", {k: round(v / 1e3, 2) for k, v in tot.items()}, flush=True)
    if a.out:
        with open(a.out, "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows)


if __name__ == "__main__":
    main()
