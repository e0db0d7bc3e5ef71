#!/usr/bin/env python
"""Training throughput of every model family on one GPU: fp16 O2 + fused SGD, synthetic 224x224 batch,
forward + loss + backward + optimiser step.  One line per architecture (also appended to --out).

    python benchmarks/zoo_bench.py --batch 128 --archs resnet50,vgg16_bn,densenet121 --out gpurun_out/zoo.txt
"""
import argparse
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd import amp, backends, models, ops
from pytorch_distributed_amd.optim import FusedSGD

DEFAULT = "resnet18,resnet50,wide_resnet50_2,resnext50_32x4d,vgg16,vgg16_bn,alexnet,squeezenet1_1,densenet121,mobilenet_v2,shufflenet_v2_x1_0,mnasnet1_0,googlenet,inception_v3"


def run(arch: str, batch: int, steps: int, warmup: int, size: int, graph: bool = False) -> float:
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = models.get_model(arch).to(dev)
    opt = FusedSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    model, opt = amp.initialize(model, opt, opt_level="O2", verbosity=0)
    x = torch.randn(batch, 3, size, size, device=dev)
    t = torch.randint(0, 1000, (batch,), device=dev)

    def step():
        opt.zero_grad()
        loss = ops.cross_entropy(model(x), t)
        with amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        opt.step()
        return loss
    for _ in range(warmup):
        step()
    if graph:
        # the whole step (forward, loss scaling, backward, fused optimizer) recorded once and replayed as a hipGraph
        from pytorch_distributed_amd.utils.graph import GraphedStep
        step = GraphedStep(step, warmup=0, device=dev)
        for _ in range(2):
            step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss).item(), f"{arch}: non-finite loss"
    return batch * steps / dt


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--archs", default=DEFAULT)
    p.add_argument("--batch", type=int, default=128)
    p.add_argument("--size", type=int, default=224)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=6)
    p.add_argument("--out", default="")
    p.add_argument("--graph", action="store_true", help="replay the training step as a hipGraph")
    p.add_argument("--keep-garbage", action="store_true",
                   help="diagnostic: do not collect the previous model before the next one is measured")
    p.add_argument("--no-autotune", action="store_true", help="static kernel choices instead of the per-shape tuner")
    a = p.parse_args()
    backends.benchmark = not a.no_autotune
    for arch in a.archs.split(","):
        ips = run(arch, a.batch, a.steps, a.warmup, a.size, a.graph)
        line = f"{arch:18s} batch {a.batch:4d}  {a.size}x{a.size}  fp16 O2{' graph' if a.graph else ''}  {ips:10.1f} img/s  {1000 * a.batch / ips:8.2f} ms/step"
        print(line, flush=True)
        if a.out:
            with open(a.out, "a") as f:
                f.write(line + "\n")
        if not a.keep_garbage:
            gc.collect()   # the previous model (autograd graphs hold cycles) must be gone, not just unreferenced
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
