#!/usr/bin/env python
"""Host time of the phases of one training step (zero_grad / forward+loss / backward / optimizer), plain model vs
single-rank DistributedDataParallel, for a launch-bound configuration (default: ResNet-18 on 32x32 input, fp32).

    python benchmarks/host_phases.py --arch resnet18 --image-size 32 --batch-size 800
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd import amp, backends, comm as pcomm, models, ops
from pytorch_distributed_amd.optim import FusedSGD


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet18")
    ap.add_argument("--image-size", type=int, default=32)
    ap.add_argument("--batch-size", type=int, default=800)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--precision", default="fp32")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    backends.benchmark = True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    pcomm.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
    x = torch.randn(a.batch_size, 3, a.image_size, a.image_size, device=dev)
    t = torch.randint(0, 1000, (a.batch_size,), device=dev)
    for ddp in (False, True):
        model = models.get_model(a.arch).to(dev)
        opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        level = "O0" if a.precision == "fp32" else "O2"
        model, opt = amp.initialize(model, opt, opt_level=level, verbosity=0)
        if ddp:
            from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP
            model = DDP(model, device_ids=[0], force_comm=True)
        metrics = ops.new_metrics(dev)
        acc = {"zero_grad": 0.0, "forward": 0.0, "backward": 0.0, "step": 0.0}

        def step(record):
            t0 = time.perf_counter()
            model.zero_grad() if ddp else opt.zero_grad(set_to_none=False)
            t1 = time.perf_counter()
            loss = ops.cross_entropy(model(x), t, metrics)
            t2 = time.perf_counter()
            with amp.scale_loss(loss, opt) as scaled:
                scaled.backward()
            t3 = time.perf_counter()
            opt.step()
            t4 = time.perf_counter()
            if record:
                acc["zero_grad"] += t1 - t0
                acc["forward"] += t2 - t1
                acc["backward"] += t3 - t2
                acc["step"] += t4 - t3
        for _ in range(15):
            step(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(True)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print(f"{'DDP  ' if ddp else 'plain'}: host {1e3 * host / a.steps:.3f} ms/step, device-complete {1e3 * total / a.steps:.3f} ms/step | "
              + "  ".join(f"{k} {1e3 * v / a.steps:.3f}" for k, v in acc.items()))
        del model, opt
    pcomm.destroy_process_group()


if __name__ == "__main__":
    main()
