"""Block-output deferral on / off against the same model in fp32: relative error of logits and per-stage gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd import ops
from pytorch_distributed_amd.ops import fused
from pytorch_distributed_amd.models.resnet import ResNet, Bottleneck

DEV = "cuda"


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def run(m, x, t):
    m.zero_grad(set_to_none=True)
    logits = m(x)
    (ops.cross_entropy(logits, t) * 256.0).backward()
    return logits.detach().float().clone(), {k: p.grad.float().clone() for k, p in m.named_parameters()}


for layers, size, batch in (([2, 2, 1, 1], 64, 32), ([3, 4, 6, 3], 224, 64)):
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        m = ResNet(Bottleneck, layers, num_classes=100).to(DEV)
        x = torch.randn(batch, 3, size, size, device=DEV)
        t = torch.randint(0, 100, (batch,), device=DEV)
        m.set_compute_dtype(torch.float32)
        lo, go = run(m, x, t)
        m.set_compute_dtype(torch.float16)
        for on in (True, False):
            fused._DEFER_BLOCK_OUT = on
            l, g = run(m, x, t)
            keys = ["fc.weight", "layer4.0.conv3.weight", "layer3.0.conv1.weight", "layer2.1.conv1.weight", "layer2.0.conv1.weight",
                    "layer1.1.conv1.weight", "layer1.0.conv3.weight", "layer1.0.conv1.weight", "conv1.weight"]
            print(f"{layers} seed {seed} defer_out {int(on)}: logits {rel(l, lo):.4f} | " +
                  " ".join(f"{rel(g[k], go[k]):.3f}" for k in keys), flush=True)
