"""Single-process DataParallel on 'cpu' pseudo-devices: scatter / replicate / parallel_apply / gather and
the reduce-add backward, checked against running the module on the whole batch."""
import torch

from pytorch_distributed_amd import models, ops
from pytorch_distributed_amd.parallel.data_parallel import (DataParallel, scatter, gather, broadcast_coalesced,
                                                            reduce_add_coalesced, replicate)


def test_scatter_gather_roundtrip():
    x = torch.arange(10.0).view(5, 2)
    parts = scatter(x, ["cpu", "cpu", "cpu"])
    assert [p.shape[0] for p in parts] == [2, 2, 1]
    assert torch.equal(gather(parts, "cpu"), x)


def test_broadcast_and_reduce_add_coalesced():
    ts = [torch.randn(3, 4), torch.randn(7), torch.randn(2, 3, 2, 2).contiguous(memory_format=torch.channels_last),
          torch.tensor(5, dtype=torch.long)]
    copies = broadcast_coalesced(ts, ["cpu", "cpu"])
    for dev_list in copies:
        for a, b in zip(ts, dev_list):
            assert torch.equal(a, b) and a.stride() == b.stride()
    fl = [t for t in ts if t.is_floating_point()]
    red = reduce_add_coalesced([fl, [2 * t for t in fl], [3 * t for t in fl]], "cpu")
    for a, r in zip(fl, red):
        assert torch.allclose(r, 6 * a)


def test_data_parallel_matches_single_device():
    torch.manual_seed(0)
    m = models.resnet18(num_classes=10)
    # BatchNorm statistics are per replica under DataParallel (as upstream), so compare in eval mode
    m.eval()
    dp = DataParallel(m, device_ids=["cpu", "cpu"], output_device="cpu")
    x = torch.randn(6, 3, 32, 32)
    t = torch.randint(0, 10, (6,))
    out = dp(x)
    ref = m(x)
    assert out.shape == (6, 10) and torch.allclose(out, ref, atol=1e-5)
    ops.cross_entropy(out, t).backward()
    g_dp = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    ops.cross_entropy(m(x), t).backward()
    for a, b in zip(g_dp, m.parameters()):
        assert torch.allclose(a, b.grad, atol=1e-5, rtol=1e-4)
    assert hasattr(dp, "module") and list(dp.state_dict().keys())[0].startswith("module.")


def test_replicas_get_their_own_weight_caches():
    m = models.resnet18(num_classes=10)
    reps = replicate(m, ["cpu", "cpu"])
    assert reps[0] is m and reps[1].conv1._cache is not m.conv1._cache
    assert reps[1].layer1[0].conv1.weight.grad_fn is not None      # autograd-connected copy
