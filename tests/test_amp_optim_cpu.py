"""Loss-scaler state machine, apex-style front end and fused SGD on the CPU path."""
import pytest
import torch

from pytorch_distributed_amd import amp, models, ops
from pytorch_distributed_amd.amp import LossScaler
from pytorch_distributed_amd.ops import multi_tensor as mt
from pytorch_distributed_amd.optim import FusedSGD


def test_loss_scaler_growth_backoff_clamp():
    s = LossScaler("dynamic", init_scale=2.0 ** 16, growth_interval=3)
    for _ in range(3):
        s.after_step()
    assert s.loss_scale() == 2.0 ** 17
    s.overflow_flag.fill_(1)
    s.after_step()
    assert s.loss_scale() == 2.0 ** 16 and s.skipped_steps() == 1 and s.last_step_overflowed()
    assert int(s.overflow_flag) == 0
    s2 = LossScaler("dynamic", init_scale=2.0 ** 24, growth_interval=1)
    s2.after_step()
    assert s2.loss_scale() == 2.0 ** 24          # capped at 2**24
    s3 = LossScaler(128.0)
    s3.overflow_flag.fill_(1)
    s3.after_step()
    assert s3.loss_scale() == 128.0              # static scale never moves
    sd = s.state_dict()
    s4 = LossScaler("dynamic")
    s4.load_state_dict(sd)
    assert s4.loss_scale() == s.loss_scale()


def test_fused_sgd_matches_torch_sgd_including_nesterov():
    torch.manual_seed(0)
    for nesterov in (False, True):
        ps = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5))]
        rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        a = FusedSGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-2, nesterov=nesterov)
        b = torch.optim.SGD(rs, lr=0.1, momentum=0.9, weight_decay=1e-2, nesterov=nesterov)
        for _ in range(4):
            for p, r in zip(ps, rs):
                g = torch.randn_like(p)
                p.grad, r.grad = g.clone(), g.clone()
            a.step()
            b.step()
        for p, r in zip(ps, rs):
            assert torch.allclose(p, r, atol=1e-6)
        assert set(a.state_dict()["state"][0].keys()) == {"momentum_buffer"}
        b.load_state_dict(a.state_dict())        # state_dict layout is torch.optim.SGD's


def test_amp_overflow_skips_step_and_halves_scale(capsys):
    torch.manual_seed(0)
    model = torch.nn.Linear(4, 2)
    opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9)
    model, opt = amp.initialize(model, opt, opt_level="O1", loss_scale="dynamic", verbosity=0)
    scaler = opt._amp_scaler
    before = [p.detach().clone() for p in model.parameters()]
    loss = model(torch.randn(3, 4)).sum()
    with amp.scale_loss(loss, opt) as scaled:
        assert scaled.item() == pytest.approx(loss.item() * 65536.0, rel=1e-5)
        scaled.backward()
    next(model.parameters()).grad[0, 0] = float("inf")      # inject an overflow
    with amp.scale_loss(torch.zeros((), requires_grad=True) * 1.0, opt) as s2:
        s2.backward()
    opt.step()
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p, b), "step must be skipped on overflow"
    assert scaler.loss_scale() == 32768.0 and scaler.skipped_steps() == 1
    opt.zero_grad()
    loss = model(torch.randn(3, 4)).sum()
    with amp.scale_loss(loss, opt) as scaled:
        scaled.backward()
    opt.step()
    assert any(not torch.equal(p, b) for p, b in zip(model.parameters(), before))
    assert "loss_scaler0" in amp.state_dict()


def test_amp_with_stock_optimizer_unscales_and_skips(capsys):
    model = torch.nn.Linear(3, 1)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    model, opt = amp.initialize(model, opt, opt_level="O1", loss_scale=1024.0, verbosity=1)
    x = torch.ones(2, 3)
    loss = model(x).sum()
    with amp.scale_loss(loss, opt) as scaled:
        scaled.backward()
    assert torch.allclose(model.weight.grad, torch.full((1, 3), 2.0))   # grads are unscaled on exit
    w = model.weight.detach().clone()
    model.weight.grad[0, 0] = float("nan")
    opt._amp_scaler_ref.overflow_flag.fill_(1)
    opt.step()
    assert torch.equal(model.weight, w)
    assert "Gradient overflow" in capsys.readouterr().out


def test_opt_levels_set_compute_dtype():
    m = models.resnet18(num_classes=10)
    amp.initialize(m, None, opt_level="O1", dtype=torch.bfloat16, verbosity=0)
    assert m.compute_dtype == torch.bfloat16
    amp.initialize(m, None, opt_level="O0", verbosity=0)
    assert m.compute_dtype == torch.float32
    with pytest.raises(ValueError):
        amp.initialize(m, None, opt_level="O9")


def test_multi_tensor_cpu_paths():
    a = [torch.randn(5), torch.randn(3, 2)]
    b = [torch.empty(5), torch.empty(3, 2)]
    flag = torch.zeros(1, dtype=torch.int32)
    mt.scale(mt.TensorLists([a, b]), 0.5, noop_flag=flag)
    assert torch.allclose(b[1], a[1] * 0.5) and int(flag) == 0
    a[0][2] = float("inf")
    mt.scale(mt.TensorLists([a, b]), 1.0, noop_flag=flag)
    assert int(flag) == 1
    a[0][2] = 0.0
    tot, per = mt.l2norm(mt.TensorLists([a]), per_tensor=True)
    assert tot.item() == pytest.approx(torch.cat([t.flatten() for t in a]).norm().item(), rel=1e-6)
    flat = mt.flatten(a)
    back = [torch.empty_like(t) for t in a]
    mt.unflatten(flat, back)
    assert torch.equal(back[1], a[1])


def test_cross_entropy_metrics_cpu():
    z = torch.tensor([[5.0, 1.0, 0.0], [0.0, 0.1, 3.0], [1.0, 2.0, 0.5]])
    t = torch.tensor([0, 1, 1])
    m = ops.new_metrics("cpu")
    loss = ops.cross_entropy(z, t, m)
    assert loss.item() == pytest.approx(torch.nn.functional.cross_entropy(z, t).item())
    assert m.tolist()[1:] == [2.0, 3.0, 3.0]
