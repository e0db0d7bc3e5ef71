#!/usr/bin/env python
"""Achieved bandwidth of the BatchNorm-backward apply pass (dy = A*g + B*y + K: two reads, one write) at the tensor
shapes of ResNet-50's inner BatchNorms, batch 400, fp16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd.ops import norm as Nm

dev = "cuda"
B = int(os.environ.get("BATCH", "400"))
tot = 0.0
for hw, c, cnt in ((56, 64, 6), (56, 128, 1), (28, 128, 7), (28, 256, 1), (14, 256, 11), (14, 512, 1), (7, 512, 5), (112, 64, 1),
                   (56, 256, 1), (28, 512, 1), (14, 1024, 1), (7, 2048, 1)):
    y = torch.randn(B, hw, hw, c, device=dev).half()
    g = torch.randn(B, hw, hw, c, device=dev).half()
    coef = torch.randn(5, c, device=dev)
    for _ in range(3):
        Nm.apply_backward_coefs(g, y, coef, relu=False)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        Nm.apply_backward_coefs(g, y, coef, relu=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gb = 3 * y.numel() * 2 / 1e9
    tot += dt * cnt
    print(f"{hw:3d}x{hw:<3d} C {c:5d} x{cnt:2d}: {dt * 1e6:7.1f} us  {gb / dt:7.0f} GB/s", flush=True)
print(f"per step (weighted): {tot * 1e3:.3f} ms")
