"""Which deferred block output changes the forward values?  One deferring block at a time against none."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd import ops
from pytorch_distributed_amd.models.resnet import ResNet, Bottleneck

DEV = "cuda"
torch.manual_seed(1)
m = ResNet(Bottleneck, [3, 4, 6, 3], num_classes=100).to(DEV).set_compute_dtype(torch.float16)
x = torch.randn(64, 3, 224, 224, device=DEV)
blocks = [b for layer in (m.layer1, m.layer2, m.layer3, m.layer4) for b in layer]
names = [f"layer{li + 1}.{bi}" for li, layer in enumerate((m.layer1, m.layer2, m.layer3, m.layer4)) for bi in range(len(layer))]
can = [b.defer_out for b in blocks]
cap = {}
m.layer3[1].register_forward_hook(lambda mod, i, o: cap.__setitem__("o", o.detach().float().clone()))


def fwd(mask):
    for b, c, k in zip(blocks, can, mask):
        b.defer_out = bool(c and k)
    m(x)
    return cap["o"]


base = fwd([False] * len(blocks))
again = fwd([False] * len(blocks))
print("run-to-run (no deferral): differing", int((base != again).sum()), "max", float((base - again).abs().max()))
for i, (n, c) in enumerate(zip(names, can)):
    if not c:
        continue
    o = fwd([j == i for j in range(len(blocks))])
    d = (o - base).abs()
    print(f"defer only {n}: differing {int((o != base).sum())} of {o.numel()}, max {float(d.max()):.4f}, rel {float(d.norm() / base.norm()):.5f}", flush=True)
o = fwd([True] * len(blocks))
d = (o - base).abs()
print(f"defer all: differing {int((o != base).sum())}, max {float(d.max()):.4f}, rel {float(d.norm() / base.norm()):.5f}")
