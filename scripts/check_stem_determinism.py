"""Run-to-run spread of the fused stem backward's weight gradient (float atomics) next to the separate kernels."""
import torch
from pytorch_distributed_amd import models, ops
from pytorch_distributed_amd.ops import fused

DEV = torch.device("cuda:0")
torch.manual_seed(31)
m = models.resnet18().to(DEV).set_compute_dtype(torch.float16)
x = torch.randn(16, 3, 32, 32, device=DEV)
for on in (1, 0):
    fused._FUSE_STEM_BWD = on
    res = []
    for it in range(6):
        m.zero_grad(set_to_none=True)
        out = ops.stem_conv_bn_relu_pool(x, m.conv1.weight, m.bn1.weight, m.bn1.bias, (2, 2), (3, 3), m.bn1.state(), torch.float16)
        g = torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(torch.float16)
        out.backward(g)
        res.append(m.conv1.weight.grad.float().clone())
    ref = res[0]
    print(f"fused={on} |dW| {float(ref.norm()):.4e} max|dW| {float(ref.abs().max()):.3e} run-to-run max abs diff",
          [f"{float((r - ref).abs().max()):.2e}" for r in res[1:]])
    if on:
        keep = ref
    else:
        print(f"fused vs separate: rel {float((keep - ref).norm() / ref.norm()):.3e} max abs {float((keep - ref).abs().max()):.3e}")
