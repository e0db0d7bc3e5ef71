"""The dense-convolution torchvision families behind ``--arch`` (/root/reference/1.dataparallel.py:23-24):
parameter counts and state_dict names against torchvision's published models, and one training step each.
torchvision itself is not installed here, so the counts and key names below are its documented ones."""
import pytest
import torch

from pytorch_distributed_amd import models

PARAMS = {
    "vgg11": 132863336, "vgg13": 133047848, "vgg16": 138357544, "vgg19": 143667240,
    "vgg11_bn": 132868840, "vgg13_bn": 133053736, "vgg16_bn": 138365992, "vgg19_bn": 143678248,
    "alexnet": 61100840, "squeezenet1_0": 1248424, "squeezenet1_1": 1235496,
    "densenet121": 7978856, "densenet161": 28681000, "densenet169": 14149480, "densenet201": 20013928,
    "wide_resnet50_2": 68883240, "wide_resnet101_2": 126886696,
    "resnext50_32x4d": 25028904, "resnext101_32x8d": 88791336, "resnext101_64x4d": 83455272,
    "mobilenet_v2": 3504872, "googlenet": 6624904, "inception_v3": 23834568,
    "shufflenet_v2_x0_5": 1366792, "shufflenet_v2_x1_0": 2278604, "shufflenet_v2_x1_5": 3503624,
    "shufflenet_v2_x2_0": 7393996,
    "mnasnet0_5": 2218512, "mnasnet0_75": 3170208, "mnasnet1_0": 4383312, "mnasnet1_3": 6282256,
    "resnet18": 11689512, "resnet34": 21797672, "resnet50": 25557032, "resnet101": 44549160, "resnet152": 60192808,
}
KEYS = {
    "vgg16": ["features.0.weight", "features.28.bias", "classifier.0.weight", "classifier.6.bias"],
    "vgg16_bn": ["features.1.running_mean", "features.41.num_batches_tracked", "classifier.3.weight"],
    "alexnet": ["features.0.weight", "features.10.bias", "classifier.1.weight", "classifier.6.weight"],
    "squeezenet1_0": ["features.0.weight", "features.3.squeeze.weight", "features.12.expand3x3.bias",
                      "classifier.1.weight"],
    "squeezenet1_1": ["features.3.expand1x1.weight", "features.12.squeeze.bias", "classifier.1.bias"],
    "densenet121": ["features.conv0.weight", "features.norm0.running_var",
                    "features.denseblock1.denselayer1.norm1.weight", "features.denseblock4.denselayer16.conv2.weight",
                    "features.transition3.conv.weight", "features.norm5.bias", "classifier.weight"],
    "wide_resnet50_2": ["layer1.0.conv2.weight", "layer4.2.bn3.running_mean", "fc.weight"],
    "resnext50_32x4d": ["layer1.0.conv2.weight", "layer4.2.conv2.weight", "fc.bias"],
    "googlenet": ["conv1.conv.weight", "inception3a.branch2.1.bn.running_mean", "inception5b.branch4.1.conv.weight",
                  "fc.weight"],
    "inception_v3": ["Conv2d_1a_3x3.conv.weight", "Mixed_5b.branch5x5_2.conv.weight", "Mixed_6b.branch7x7_2.conv.weight",
                     "Mixed_7c.branch3x3dbl_3b.bn.running_var", "fc.bias"],
    "shufflenet_v2_x1_0": ["conv1.0.weight", "stage2.0.branch1.0.weight", "stage2.0.branch1.2.weight",
                           "stage3.1.branch2.3.weight", "stage4.3.branch2.6.running_var", "conv5.1.running_mean",
                           "fc.weight"],
    "mnasnet1_0": ["layers.0.weight", "layers.3.weight", "layers.7.running_mean", "layers.8.0.layers.3.weight",
                   "layers.13.0.layers.6.weight", "layers.14.weight", "layers.15.bias", "classifier.1.weight"],
    "mobilenet_v2": ["features.0.0.weight", "features.1.conv.0.0.weight", "features.1.conv.1.weight",
                     "features.2.conv.1.0.weight", "features.17.conv.3.running_var", "features.18.1.running_mean",
                     "classifier.1.weight"],
}


def test_every_arch_is_constructible_and_listed():
    assert set(PARAMS) | {"mnist_net"} == set(models.model_names)
    for name in models.model_names:
        assert callable(getattr(models, name)) and models.__dict__[name] is getattr(models, name)


@pytest.mark.parametrize("arch", sorted(PARAMS))
def test_parameter_count_matches_torchvision(arch):
    with torch.device("meta"):
        m = models.get_model(arch)
    assert sum(p.numel() for p in m.parameters()) == PARAMS[arch]


@pytest.mark.parametrize("arch", sorted(KEYS))
def test_state_dict_names(arch):
    with torch.device("meta"):
        sd = models.get_model(arch).state_dict()
    for k in KEYS[arch]:
        assert k in sd, (k, list(sd)[:8])
    if arch == "wide_resnet50_2":
        assert tuple(sd["layer1.0.conv2.weight"].shape) == (128, 128, 3, 3)
        assert tuple(sd["layer1.0.conv3.weight"].shape) == (256, 128, 1, 1)
    if arch == "resnext50_32x4d":
        assert tuple(sd["layer1.0.conv2.weight"].shape) == (128, 4, 3, 3)
        assert tuple(sd["layer4.2.conv2.weight"].shape) == (1024, 32, 3, 3)
    if arch == "shufflenet_v2_x1_0":
        assert tuple(sd["stage2.0.branch2.3.weight"].shape) == (58, 1, 3, 3)
    if arch == "mnasnet1_0":
        assert tuple(sd["layers.9.0.layers.3.weight"].shape) == (72, 1, 5, 5)
    if arch == "inception_v3":
        assert tuple(sd["Mixed_6b.branch7x7_2.conv.weight"].shape) == (128, 128, 1, 7)
    if arch == "mobilenet_v2":
        assert tuple(sd["features.2.conv.1.0.weight"].shape) == (96, 1, 3, 3)
    if arch == "densenet121":
        assert tuple(sd["features.denseblock2.denselayer3.conv1.weight"].shape) == (128, 128 + 2 * 32, 1, 1)


@pytest.mark.parametrize("arch,size,kw", [
    ("vgg11_bn", 32, {"dropout": 0.0}), ("alexnet", 64, {}), ("squeezenet1_1", 48, {}), ("squeezenet1_0", 64, {}),
    ("densenet121", 32, {}), ("wide_resnet50_2", 32, {}), ("resnext50_32x4d", 32, {}), ("mobilenet_v2", 64, {}), ("shufflenet_v2_x1_0", 64, {}), ("mnasnet0_5", 64, {}), ("googlenet", 64, {}), ("inception_v3", 96, {})])
def test_one_training_step(arch, size, kw):
    torch.manual_seed(0)
    m = models.get_model(arch, num_classes=10, **kw)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    x, y = torch.randn(2, 3, size, size), torch.tensor([1, 7])
    out = m(x)
    assert out.shape == (2, 10) and torch.isfinite(out).all()
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    missing = [n for n, p in m.named_parameters() if p.grad is None]
    assert not missing, missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    opt.step()
    m.eval()
    with torch.no_grad():
        assert m(x).shape == (2, 10)


def test_ceil_mode_pool_matches_torch():
    from pytorch_distributed_amd import ops
    from pytorch_distributed_amd.ops.pool import pool_out_size
    for h in range(3, 40):
        for k, s, p in ((3, 2, 0), (3, 2, 1), (2, 2, 0)):
            ref = torch.nn.functional.max_pool2d(torch.zeros(1, 1, h, h), k, s, p, ceil_mode=True).shape[-1]
            assert pool_out_size(h, k, s, p, True) == ref, (h, k, s, p)
    x = torch.randn(2, 13, 13, 8)
    ref = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, ceil_mode=True).permute(0, 2, 3, 1)
    assert torch.equal(ops.max_pool2d(x, 3, 2, 0, True), ref)
    a = torch.randn(2, 7, 6, 4)
    assert torch.allclose(ops.avg_pool2d(a, 2, 2),
                          torch.nn.functional.avg_pool2d(a.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1), atol=1e-6)
    assert torch.allclose(ops.adaptive_avg_pool2d(a, (3, 3)),
                          torch.nn.functional.adaptive_avg_pool2d(a.permute(0, 3, 1, 2), 3).permute(0, 2, 3, 1),
                          atol=1e-6)
    assert ops.adaptive_avg_pool2d(a, (7, 6)) is a


def test_channel_shuffle_matches_nchw_definition():
    from pytorch_distributed_amd.models.shufflenet import channel_shuffle
    x = torch.arange(2 * 3 * 3 * 12, dtype=torch.float32).reshape(2, 3, 3, 12)
    nchw = x.permute(0, 3, 1, 2)
    n, c, h, w = nchw.shape
    ref = nchw.reshape(n, 2, c // 2, h, w).transpose(1, 2).reshape(n, c, h, w).permute(0, 2, 3, 1)
    assert torch.equal(channel_shuffle(x, 2), ref)


def test_auxiliary_heads_match_torchvision_counts_and_outputs():
    """With ``aux_logits=True`` (torchvision's default) the parameter counts are the published ones and
    training mode returns the upstream named tuples; eval mode returns the logits alone."""
    from pytorch_distributed_amd.models.googlenet import GoogLeNet, GoogLeNetOutputs
    from pytorch_distributed_amd.models.inception import Inception3, InceptionOutputs
    with torch.device("meta"):
        assert sum(p.numel() for p in GoogLeNet(aux_logits=True).parameters()) == 13004888
        assert sum(p.numel() for p in Inception3(aux_logits=True).parameters()) == 27161264
    g = GoogLeNet(num_classes=7, aux_logits=True)
    out = g(torch.randn(2, 3, 96, 96))
    assert isinstance(out, GoogLeNetOutputs) and all(o.shape == (2, 7) for o in out)
    g.eval()
    assert g(torch.randn(2, 3, 96, 96)).shape == (2, 7)
    i3 = Inception3(num_classes=7, aux_logits=True)
    out = i3(torch.randn(2, 3, 299, 299))
    assert isinstance(out, InceptionOutputs) and out.logits.shape == (2, 7) and out.aux_logits.shape == (2, 7)
