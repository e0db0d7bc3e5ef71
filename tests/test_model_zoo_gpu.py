"""The non-ResNet families on the HIP ops: the model on the GPU (hand-written kernels) against a copy of
itself on the CPU (stock PyTorch math) with the same weights and inputs - loss, gradients, running
statistics - plus one mixed-precision optimiser step."""
import copy

import pytest
import torch
import torch.nn.functional as F

from pytorch_distributed_amd import models, ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _cos(a, b):
    a, b = a.float().flatten().cpu(), b.float().flatten().cpu()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


CASES = [("vgg11", 32, 4), ("vgg13_bn", 32, 8), ("alexnet", 96, 4), ("squeezenet1_0", 64, 4),
         ("squeezenet1_1", 47, 4), ("densenet121", 64, 8), ("wide_resnet50_2", 64, 8), ("resnext50_32x4d", 64, 8),
         ("mobilenet_v2", 64, 8), ("shufflenet_v2_x1_0", 64, 8), ("shufflenet_v2_x0_5", 64, 8),
         ("mnasnet1_0", 64, 8), ("googlenet", 64, 8), ("inception_v3", 160, 8)]


@pytest.mark.parametrize("arch,hw,batch", CASES)
def test_family_matches_cpu_math_fp32(arch, hw, batch):
    torch.manual_seed(5)
    kw = {} if arch.startswith(("densenet", "wide_", "resnext", "shufflenet")) else {"dropout": 0.0}
    cpu = models.get_model(arch, num_classes=16, **kw)
    gpu = copy.deepcopy(cpu).to(DEV)
    x = torch.randn(batch, 3, hw, hw)
    t = torch.randint(0, 16, (batch,))
    lc = F.cross_entropy(cpu(x), t)
    lg = ops.cross_entropy(gpu(x.to(DEV)), t.to(DEV))
    assert abs(lc.item() - lg.item()) < 2e-3 * abs(lc.item()), (lc.item(), lg.item())
    lc.backward()
    lg.backward()
    gc, gg = dict(cpu.named_parameters()), dict(gpu.named_parameters())
    missing = [k for k in gg if gg[k].grad is None]
    assert not missing, missing[:5]
    allc = torch.cat([gc[k].grad.flatten() for k in gc])
    allg = torch.cat([gg[k].grad.float().flatten().cpu() for k in gc])
    cos = sorted((_cos(gg[k].grad, gc[k].grad), k) for k in gc if gc[k].grad.norm() > 1e-12)
    assert _cos(allg, allc) > 0.999, (_cos(allg, allc), cos[:3])
    assert cos[len(cos) // 2][0] > 0.999, cos[:3]
    for (k, bc), (_, bg) in zip(cpu.named_buffers(), gpu.named_buffers()):
        if k.endswith("running_var"):
            assert torch.allclose(bg.cpu(), bc, rtol=2e-3, atol=1e-5), k
            break


@pytest.mark.parametrize("arch,hw,batch", [("vgg11_bn", 32, 8), ("squeezenet1_1", 64, 8), ("densenet121", 64, 8),
                                           ("alexnet", 96, 8), ("resnext50_32x4d", 64, 8), ("mobilenet_v2", 64, 8),
                                           ("shufflenet_v2_x1_0", 64, 8), ("mnasnet1_0", 64, 8),
                                           ("googlenet", 64, 8), ("inception_v3", 96, 8)])
def test_family_amp_step(arch, hw, batch):
    """fp16 O2 + fused SGD: finite loss, every parameter moves, a second step lowers the loss on the same batch."""
    from pytorch_distributed_amd import amp
    from pytorch_distributed_amd.optim import FusedSGD
    torch.manual_seed(6)
    kw = {} if arch.startswith(("densenet", "resnext", "shufflenet")) else {"dropout": 0.0}
    m = models.get_model(arch, num_classes=16, **kw).to(DEV)
    opt = FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    m, opt = amp.initialize(m, opt, opt_level="O2", verbosity=0)
    x = torch.randn(batch, 3, hw, hw, device=DEV)
    t = torch.randint(0, 16, (batch,), device=DEV)
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    losses = []
    for _ in range(16):   # the dynamic loss scale starts at 2**16: the first steps may overflow and be skipped
        opt.zero_grad()
        loss = ops.cross_entropy(m(x), t)
        with amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        opt.step()
        losses.append(loss.item())
    assert all(l == l and abs(l) != float("inf") for l in losses), losses
    assert losses[-1] < losses[0], losses
    # (a zero-initialised conv bias in front of a training-mode BatchNorm has an exactly-zero gradient)
    still = [k for k, p in m.named_parameters() if torch.equal(p.detach(), before[k])
             and not (p.grad is not None and float(p.grad.abs().max()) == 0.0 and float(p.detach().abs().max()) == 0.0)]
    assert not still, still


def test_ceil_mode_and_avg_pools_gpu():
    torch.manual_seed(7)
    for dtype in (torch.float32, torch.float16):
        x = torch.randn(3, 13, 15, 16, device=DEV, dtype=dtype, requires_grad=True)
        y = ops.max_pool2d(x, 3, 2, 0, True)
        xr = x.detach().float().cpu().requires_grad_(True)
        yr = F.max_pool2d(xr.permute(0, 3, 1, 2), 3, 2, ceil_mode=True).permute(0, 2, 3, 1)
        assert y.shape == yr.shape and torch.equal(y.float().cpu(), yr)
        g = torch.randn_like(yr)
        y.backward(g.to(DEV, dtype))
        yr.backward(g)
        assert torch.allclose(x.grad.float().cpu(), xr.grad, atol=2e-2 if dtype == torch.float16 else 1e-6)
    a = torch.randn(2, 8, 8, 32, device=DEV)
    assert torch.allclose(ops.avg_pool2d(a, 2, 2), F.avg_pool2d(a.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1),
                          atol=1e-6)
    assert torch.allclose(ops.adaptive_avg_pool2d(a, (1, 1)).flatten(1), a.mean(dim=(1, 2)), atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,groups,stride,bias", [(128, 128, 32, 1, False), (256, 256, 32, 2, False),
                                                        (64, 128, 8, 1, True)])
def test_grouped_conv_matches_reference(dtype, cin, cout, groups, stride, bias):
    """Grouped convolution (the 32-channel-chunk kernels for the 16-bit square layers, block-diagonal expansion on the
    dense MFMA kernels for the others) == F.conv2d(groups=...) in fp32."""
    torch.manual_seed(8)
    x = torch.randn(4, 14, 14, cin, device=DEV, dtype=dtype, requires_grad=True)
    w = (torch.randn(cout, cin // groups, 3, 3, device=DEV) * 0.1).requires_grad_(True)
    b = torch.randn(cout, device=DEV, requires_grad=True) if bias else None
    y = ops.conv2d(x, w, b, (stride, stride), (1, 1), groups=groups)
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=1, groups=groups)
    tol = 1e-4 if dtype == torch.float32 else (2e-2 if dtype == torch.float16 else 5e-2)
    assert y.shape == yr.permute(0, 2, 3, 1).shape

    def rel(a, r):
        a, r = a.detach().float(), r.detach()
        return float((a - r).norm() / (r.norm() + 1e-30))
    assert rel(y, yr.permute(0, 2, 3, 1)) < tol
    g = torch.randn_like(yr)
    y.backward(g.permute(0, 2, 3, 1).to(dtype).contiguous())
    yr.backward(g.to(dtype).float())
    assert rel(x.grad, xr.grad.permute(0, 2, 3, 1)) < tol
    assert w.grad.shape == wr.grad.shape and rel(w.grad, wr.grad) < tol
    if bias:
        assert rel(b.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("ch,cg,stride,n,h,w", [(128, 4, 1, 3, 56, 56), (256, 8, 2, 2, 56, 56), (512, 16, 1, 5, 14, 14),
                                                (1024, 32, 2, 3, 14, 14), (1024, 32, 1, 5, 7, 7), (160, 8, 1, 2, 13, 20),
                                                (96, 4, 2, 3, 15, 9), (64, 16, 1, 1, 5, 150), (128, 32, 2, 2, 9, 140)])
def test_grouped_conv_chunk_kernels(dtype, ch, cg, stride, n, h, w):
    """The 32-channel-chunk kernels of ResNeXt's grouped 3x3 (fprop / dgrad without LDS, filter in registers; wgrad
    through transposing LDS reads of a halo tile) == F.conv2d(groups=...) in fp32, and == the dense block-diagonal
    fallback; the native entry points must have taken the call."""
    from pytorch_distributed_amd.ops import grouped as G
    torch.manual_seed(18)
    groups = ch // cg
    x = torch.randn(n, h, w, ch, device=DEV, dtype=dtype, requires_grad=True)
    wt = (torch.randn(ch, cg, 3, 3, device=DEV) * 0.1).requires_grad_(True)
    assert G._chunked_ok(x, wt, (stride, stride), (1, 1), groups)
    y = ops.conv2d(x, wt, None, (stride, stride), (1, 1), groups=groups)
    assert y.grad_fn.name().startswith("_ChunkedGroupedConvFn")
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = wt.detach().to(dtype).float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=stride, padding=1, groups=groups)
    tol = 4e-3 if dtype == torch.float16 else 2e-2

    def rel(a, r):
        a, r = a.detach().float(), r.detach()
        return float((a - r).norm() / (r.norm() + 1e-30))
    assert y.shape == yr.permute(0, 2, 3, 1).shape
    assert rel(y, yr.permute(0, 2, 3, 1)) < tol
    g = torch.randn_like(yr)
    y.backward(g.permute(0, 2, 3, 1).to(dtype).contiguous())
    yr.backward(g.to(dtype).float())
    assert rel(x.grad, xr.grad.permute(0, 2, 3, 1)) < tol
    assert wt.grad.shape == wr.grad.shape and rel(wt.grad, wr.grad) < tol
    # element-wise against the dense fallback on the same 16-bit inputs
    x2 = x.detach().clone().requires_grad_(True)
    w2 = wt.detach().clone().requires_grad_(True)
    y2 = G._GroupedConvFn.apply(x2, w2, None, (stride, stride), (1, 1), groups)
    y2.backward(g.permute(0, 2, 3, 1).to(dtype).contiguous())
    atol = 2e-2 if dtype == torch.float16 else 1.5e-1
    assert torch.allclose(y.float(), y2.float(), atol=atol, rtol=2e-2)
    assert torch.allclose(x.grad.float(), x2.grad.float(), atol=atol, rtol=2e-2)
    assert rel(wt.grad, w2.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("ch,k,stride,pad,hw,bias", [(96, 3, 1, 1, 14, False), (144, 3, 2, 1, 15, False),
                                                    (32, 5, 2, 2, 12, True), (20, 3, 1, 1, 9, False),
                                                    (40, 7, 1, 3, 10, False), (58, 3, 2, 1, 13, False),
                                                    (116, 3, 1, 1, 7, True), (21, 3, 1, 1, 8, False)])
def test_depthwise_conv_matches_reference(dtype, ch, k, stride, pad, hw, bias):
    """Depthwise kernels (16-byte vectors for channel counts that fill them, channel pairs for the other even 16-bit
    counts such as ShuffleNetV2's 58 / 116, single channels otherwise)
    == F.conv2d(groups=C) in fp32; also the single-slab (deterministic) weight gradient."""
    from pytorch_distributed_amd import backends
    torch.manual_seed(9)
    x = torch.randn(6, hw, hw + 1, ch, device=DEV, dtype=dtype, requires_grad=True)
    w = (torch.randn(ch, 1, k, k, device=DEV) * 0.3).requires_grad_(True)
    b = torch.randn(ch, device=DEV, requires_grad=True) if bias else None
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad, groups=ch)
    g = torch.randn_like(yr)
    yr.backward(g.to(dtype).float())
    tol = 1e-5 if dtype == torch.float32 else (5e-3 if dtype == torch.float16 else 3e-2)

    def rel(a, r):
        a, r = a.detach().float(), r.detach()
        return float((a - r).norm() / (r.norm() + 1e-30))
    for det in (False, True):
        backends.deterministic = det
        try:
            x.grad = w.grad = None
            if bias:
                b.grad = None
            y = ops.conv2d(x, w, b, (stride, stride), (pad, pad), groups=ch)
            assert y.shape == yr.permute(0, 2, 3, 1).shape
            assert rel(y, yr.permute(0, 2, 3, 1)) < tol
            y.backward(g.permute(0, 2, 3, 1).to(dtype).contiguous())
        finally:
            backends.deterministic = False
        assert rel(x.grad, xr.grad.permute(0, 2, 3, 1)) < tol
        assert w.grad.shape == wr.grad.shape and rel(w.grad, wr.grad) < max(tol, 1e-5)
        if bias:
            assert rel(b.grad, br.grad) < tol


def test_relu6_matches_hardtanh():
    for dtype in (torch.float32, torch.float16):
        x = (torch.randn(5, 7, 7, 24, device=DEV) * 4).to(dtype).requires_grad_(True)
        y = ops.relu6(x)
        xr = x.detach().clone().requires_grad_(True)
        yr = F.hardtanh(xr, 0.0, 6.0)
        assert torch.equal(y, yr)
        g = torch.randn_like(y)
        y.backward(g)
        yr.backward(g)
        assert torch.equal(x.grad, xr.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cin,cout,ks,pad,stride", [(128, 128, (1, 7), (0, 3), 1), (160, 192, (7, 1), (3, 0), 1),
                                                    (384, 384, (1, 3), (0, 1), 1), (48, 64, (5, 5), (2, 2), 1),
                                                    (80, 192, (3, 3), (0, 0), 1), (3, 32, (3, 3), (0, 0), 2),
                                                    (16, 32, (3, 3), (1, 1), 1)])
def test_rectangular_and_padded_channel_convs(dtype, cin, cout, ks, pad, stride):
    """Inception's factorised 1x7 / 7x1 / 1x3 filters on the implicit-GEMM kernels, and the layers whose input
    channels are zero-padded up to the kernels' K step (3, 16, 48, 80 channels), against F.conv2d in fp32."""
    torch.manual_seed(10)
    x = torch.randn(4, 17, 17, cin, device=DEV, dtype=dtype, requires_grad=True)
    w = (torch.randn(cout, cin, *ks, device=DEV) * 0.05).requires_grad_(True)
    y = ops.conv2d(x, w, None, (stride, stride), pad)
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    tol = 1e-4 if dtype == torch.float32 else 1e-2

    def rel(a, r):
        a, r = a.detach().float(), r.detach()
        return float((a - r).norm() / (r.norm() + 1e-30))
    assert y.shape == yr.permute(0, 2, 3, 1).shape and rel(y, yr.permute(0, 2, 3, 1)) < tol
    g = torch.randn_like(yr)
    y.backward(g.permute(0, 2, 3, 1).to(dtype).contiguous())
    yr.backward(g.to(dtype).float())
    assert rel(x.grad, xr.grad.permute(0, 2, 3, 1)) < tol
    assert w.grad.shape == wr.grad.shape and rel(w.grad, wr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape,k,s,p", [((8, 3, 3, 2048), 3, 1, 1), ((4, 17, 17, 288), 3, 1, 1),
                                         ((4, 17, 17, 768), 5, 3, 0), ((4, 14, 14, 128), 2, 2, 0),
                                         ((2, 9, 11, 20), 3, 2, 1)])
def test_avg_pool_kernels_match_reference(dtype, shape, k, s, p):
    """Own average-pool kernels (forward and backward) against F.avg_pool2d on the CPU in fp32."""
    torch.manual_seed(11)
    x = torch.randn(*shape, device=DEV, dtype=dtype, requires_grad=True)
    xr = x.detach().float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = ops.avg_pool2d(x, k, s, p)
    yr = F.avg_pool2d(xr, k, s, p)
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert y.shape == yr.permute(0, 2, 3, 1).shape
    assert torch.allclose(y.float().cpu(), yr.permute(0, 2, 3, 1), atol=tol, rtol=tol)
    g = torch.randn_like(yr)
    y.backward(g.permute(0, 2, 3, 1).to(DEV, dtype).contiguous())
    yr.backward(g.to(dtype).float())
    assert torch.allclose(x.grad.float().cpu(), xr.grad.permute(0, 2, 3, 1), atol=10 * tol, rtol=10 * tol)


def test_adaptive_avg_pool_backward_gpu():
    x = torch.randn(4, 9, 9, 64, device=DEV, requires_grad=True)
    xr = x.detach().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y, yr = ops.adaptive_avg_pool2d(x, (4, 4)), F.adaptive_avg_pool2d(xr, (4, 4))
    g = torch.randn_like(yr)
    y.backward(g.permute(0, 2, 3, 1).to(DEV).contiguous())
    yr.backward(g)
    assert torch.allclose(y.cpu(), yr.permute(0, 2, 3, 1), atol=1e-5)
    assert torch.allclose(x.grad.cpu(), xr.grad.permute(0, 2, 3, 1), atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,offset", [(1, 0), (7, 0), (4099, 0), (1 << 20, 0), (4099, 1), (65536, 3)])
def test_relu_kernels_any_length_and_alignment(dtype, n, offset):
    """Vector body + scalar tail, and the all-scalar path for a view that is not 16-byte aligned."""
    base = (torch.randn(n + offset, device=DEV) * 4).to(dtype)
    for fn, ref in ((ops.relu, torch.relu), (ops.relu6, lambda t: F.hardtanh(t, 0.0, 6.0))):
        x = base[offset:].detach().requires_grad_(True)
        xr = base[offset:].detach().clone().requires_grad_(True)
        y, yr = fn(x), ref(xr)
        assert torch.equal(y, yr)
        g = torch.randn_like(y)
        y.backward(g)
        yr.backward(g)
        assert torch.equal(x.grad, xr.grad)


def test_conv_bias_in_front_of_training_batchnorm_is_folded():
    """VGG-*_bn: conv(bias) + BatchNorm in training mode runs as the bias-free fused op; outputs, gradients
    and EVERY running statistic equal the stock computation, the bias gradient is zero."""
    torch.manual_seed(12)
    cpu = models.get_model("vgg11_bn", num_classes=16, dropout=0.0)
    for m in cpu.modules():   # non-trivial biases and statistics
        if hasattr(m, "kernel_size") and getattr(m, "bias", None) is not None and m.bias.dim() == 1:
            torch.nn.init.normal_(m.bias, 0, 0.5)
    gpu = copy.deepcopy(cpu).to(DEV)
    x = torch.randn(8, 3, 32, 32)
    t = torch.randint(0, 16, (8,))
    for _ in range(2):
        lc = F.cross_entropy(cpu(x), t)
        lg = ops.cross_entropy(gpu(x.to(DEV)), t.to(DEV))
        assert abs(lc.item() - lg.item()) < 2e-3 * abs(lc.item())
    lc.backward()
    lg.backward()
    for (k, bc), (_, bg) in zip(cpu.named_buffers(), gpu.named_buffers()):
        if bc.dtype.is_floating_point:
            assert torch.allclose(bg.cpu(), bc, rtol=2e-3, atol=2e-4), k
    gc, gg = dict(cpu.named_parameters()), dict(gpu.named_parameters())
    # (features.0 reads the 3-channel image and stays on the unfused path; features.4 is folded)
    assert float(gg["features.4.bias"].grad.abs().max()) == 0.0
    assert float(gc["features.4.bias"].grad.abs().max()) < 1e-4     # the stock gradient is rounding noise
    assert _cos(gg["features.4.weight"].grad, gc["features.4.weight"].grad) > 0.999
    gpu.eval(); cpu.eval()
    with torch.no_grad():
        assert torch.allclose(gpu(x.to(DEV)).cpu(), cpu(x), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("arch", [a for a in models.model_names if a != "mnist_net"])
def test_every_arch_one_amp_step_gpu(arch):
    """Every registered architecture runs two fp16 O2 training steps on the HIP ops (finite loss, a gradient
    for every parameter) - channel counts, group counts and window shapes differ a lot between the variants."""
    from pytorch_distributed_amd import amp
    from pytorch_distributed_amd.optim import FusedSGD
    torch.manual_seed(13)
    size = 96 if arch in ("alexnet", "inception_v3") else 64
    m = models.get_model(arch, num_classes=8).to(DEV)
    opt = FusedSGD(m.parameters(), lr=0.01, momentum=0.9)
    m, opt = amp.initialize(m, opt, opt_level="O2", loss_scale=128.0, verbosity=0)
    x = torch.randn(4, 3, size, size, device=DEV)
    t = torch.randint(0, 8, (4,), device=DEV)
    for _ in range(2):
        opt.zero_grad()
        loss = ops.cross_entropy(m(x), t)
        with amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        missing = [k for k, p in m.named_parameters() if p.grad is None]
        assert not missing, missing[:5]
        opt.step()
    assert torch.isfinite(loss).item(), loss


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("relu", [False, True])
def test_filter_count_padding_with_bias_and_relu(dtype, relu):
    """3x3 conv with a filter count the dgrad kernel does not step through (48, 208): the filter, the bias and
    the output are zero-padded and sliced back; bias gradient and the fused ReLU included."""
    torch.manual_seed(14)
    for cin, cout in ((64, 48), (96, 208)):
        x = torch.randn(4, 9, 9, cin, device=DEV, dtype=dtype, requires_grad=True)
        w = (torch.randn(cout, cin, 3, 3, device=DEV) * 0.05).requires_grad_(True)
        b = torch.randn(cout, device=DEV, requires_grad=True)
        y = ops.conv2d(x, w, b, (1, 1), (1, 1), relu=relu)
        xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
        wr = w.detach().to(dtype).float().requires_grad_(True)
        br = b.detach().clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, padding=1)
        if relu:
            # mask from the device result: a pre-activation within rounding of zero may land on either side
            yr = yr * (y.detach().float().permute(0, 3, 1, 2) > 0)
        tol = 1e-4 if dtype == torch.float32 else 1e-2

        def rel(a, r):
            a, r = a.detach().float(), r.detach()
            return float((a - r).norm() / (r.norm() + 1e-30))
        assert y.shape == (4, 9, 9, cout) and rel(y, yr.permute(0, 2, 3, 1)) < tol
        g = torch.randn_like(yr)
        y.backward(g.permute(0, 2, 3, 1).to(dtype).contiguous())
        yr.backward(g.to(dtype).float())
        assert rel(x.grad, xr.grad.permute(0, 2, 3, 1)) < tol
        assert rel(w.grad, wr.grad) < tol and rel(b.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 58, 58), (1000, 116, 24), (257, 10, 50), (64, 232, 116), (129, 130, 33),
                                   (5, 7, 3), (4096, 256, 320)])
def test_any_shape_gemm_matches_fp32(dtype, M, N, K):
    """csrc/kernels/gemm_anyk.hip against an fp32 product: channel counts that are no multiple of anything (the K
    tail and the rows / columns past the last tile multiply exact zeros), vector and element fetch paths, bias
    and ReLU in the write-out, a strided A operand."""
    from pytorch_distributed_amd.ops.conv import gemm_nt
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K + 3, device=DEV).to(dtype)[:, :K]          # row pitch K + 3: never 16-byte aligned rows
    b = torch.randn(N, K, device=DEV).to(dtype)
    bias = torch.randn(N, device=DEV)
    ref = a.float() @ b.float().t()
    tol = 1e-5 if dtype == torch.float32 else (2e-3 if dtype == torch.float16 else 1.5e-2)

    def rel(x, r):
        return float((x.float() - r).norm() / (r.norm() + 1e-30))
    assert rel(gemm_nt(a, b), ref) < tol
    assert rel(gemm_nt(a.contiguous(), b, bias, relu=True), torch.relu(ref + bias)) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cin,cout", [(58, 58), (24, 144), (116, 232), (16, 50)])
def test_odd_channel_pointwise_conv_runs_on_the_own_gemm(dtype, cin, cout, monkeypatch):
    """ShuffleNetV2 / MobileNetV2 pointwise layers (58, 116, 232, 24, 144 channels) forward and backward against
    F.conv2d in fp32 - with the library GEMM entry points made to fail, so a silent fallback cannot pass."""
    def boom(*a, **k):
        raise AssertionError("library GEMM reached from a pointwise conv")
    monkeypatch.setattr(F, "linear", boom)
    monkeypatch.setattr(torch, "matmul", boom)
    torch.manual_seed(3)
    x = torch.randn(3, 9, 11, cin, device=DEV, dtype=dtype, requires_grad=True)
    w = (torch.randn(cout, cin, 1, 1, device=DEV) * 0.1).requires_grad_(True)
    bias = torch.randn(cout, device=DEV, requires_grad=True)
    y = ops.conv2d(x, w, bias, (1, 1), (0, 0))
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(dtype).float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    tol = 1e-4 if dtype == torch.float32 else 1e-2

    def rel(a, r):
        return float((a.detach().float() - r.detach()).norm() / (r.detach().norm() + 1e-30))
    assert rel(y, yr.permute(0, 2, 3, 1)) < tol
    g = torch.randn_like(yr)
    y.backward(g.permute(0, 2, 3, 1).to(dtype).contiguous())
    yr.backward(g.to(dtype).float())
    assert rel(x.grad, xr.grad.permute(0, 2, 3, 1)) < tol
    assert rel(w.grad, wr.grad) < tol and rel(bias.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [(58, 58), (116, 116), (232, 24), (13, 70)])
def test_ragged_wgrad_on_channel_slices(dtype, cin, cout):
    """Weight gradient of a pointwise conv over odd channel counts (conv_igemm.hip, ragged wgrad instantiation):
    many pixels (several split-K workgroups), the input a channel SLICE of a wider tensor as in ShuffleNetV2's
    channel split, against the fp32 product of the same 16-bit values."""
    torch.manual_seed(cin * 7 + cout)
    full = torch.randn(6, 37, 41, 2 * cin, device=DEV).to(dtype)
    x = full[..., cin:].detach().requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device=DEV) * 0.1).requires_grad_(True)
    y = ops.conv2d(x, w, None, (1, 1), (0, 0))
    g = torch.randn(y.shape, device=DEV).to(dtype)
    y.backward(g)
    ref_w = g.float().reshape(-1, cout).t() @ x.detach().float().reshape(-1, cin)
    ref_x = g.float().reshape(-1, cout) @ w.detach().to(dtype).float().reshape(cout, cin)
    ref_y = x.detach().float().reshape(-1, cin) @ w.detach().to(dtype).float().reshape(cout, cin).t()
    tol = 1e-4 if dtype == torch.float32 else (2e-3 if dtype == torch.float16 else 1.5e-2)

    def rel(a, r):
        return float((a.detach().float().reshape(r.shape) - r).norm() / (r.norm() + 1e-30))
    assert rel(y, ref_y) < tol
    assert rel(w.grad, ref_w) < tol
    assert rel(x.grad, ref_x) < tol


@pytest.mark.parametrize("arch", ["resnet50", "resnext50_32x4d", "mobilenet_v2", "alexnet"])
def test_deterministic_mode_is_bit_reproducible(arch):
    """backends.deterministic: three forward + backward passes of a whole model on the same batch (same dropout masks)
    give bit-identical logits and gradients at a size that fills the chip - any race between waves, and any float
    atomic left on the path, shows as a difference.  (scripts/repro_check.py runs the same over every family.)"""
    from pytorch_distributed_amd import backends
    from pytorch_distributed_amd.ops import elementwise as E
    old = backends.deterministic
    backends.deterministic = True
    try:
        torch.manual_seed(0)
        m = models.get_model(arch).to(DEV)
        if hasattr(m, "set_compute_dtype"):
            m.set_compute_dtype(torch.float16)
        x = torch.randn(48, 3, 224, 224, device=DEV)
        t = torch.randint(0, 1000, (48,), device=DEV)
        state = {k: v.clone() for k, v in m.state_dict().items()}
        runs = []
        for _ in range(3):
            m.load_state_dict(state)
            m.zero_grad(set_to_none=True)
            torch.manual_seed(7)
            E._DROPOUT_CALLS = 0
            logits = m(x)
            (ops.cross_entropy(logits, t) * 128.0).backward()
            runs.append((logits.detach().clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]))
        for r in runs[1:]:
            assert torch.equal(r[0], runs[0][0])
            for g0, g1 in zip(runs[0][1], r[1]):
                assert torch.equal(g0, g1)
    finally:
        backends.deterministic = old
