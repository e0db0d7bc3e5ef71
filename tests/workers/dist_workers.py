"""Worker bodies for the multi-process CPU tests (spawned with pytorch_distributed_amd.launch.spawn)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _init(rank, world, port, method="tcp"):
    from pytorch_distributed_amd import comm
    url = f"tcp://127.0.0.1:{port}" if method == "tcp" else f"file://{port}"
    comm.init_process_group("gloo", url, world_size=world, rank=rank)
    return comm


def ddp_equivalence(rank, world, port, wire_fp16):
    import torch.distributed as dist
    comm = _init(rank, world, port)
    from pytorch_distributed_amd import models, ops
    from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP
    torch.manual_seed(rank)                       # replicas start different: the ctor must broadcast rank 0's
    ddp = DDP(models.resnet18(num_classes=10), bucket_cap_mb=4, first_bucket_mb=0.1,
              wire_dtype=torch.bfloat16 if wire_fp16 else None)
    ref = models.resnet18(num_classes=10)
    ref.load_state_dict({k: v.clone() for k, v in ddp.module.state_dict().items()})
    torch.manual_seed(100)
    X, T = torch.randn(4 * world, 3, 32, 32), torch.randint(0, 10, (4 * world,))
    x, t = X[rank * 4:(rank + 1) * 4], T[rank * 4:(rank + 1) * 4]
    nb0 = None
    for it in range(3):
        ddp.zero_grad()
        ops.cross_entropy(ddp(x), t).backward()
        grads = []
        for r in range(world):
            ref.zero_grad()
            ops.cross_entropy(ref(X[r * 4:(r + 1) * 4]), T[r * 4:(r + 1) * 4]).backward()
            grads.append([p.grad.clone() for p in ref.parameters()])
        avg = [sum(g[i] for g in grads) / world for i in range(len(grads[0]))]
        err = max((a - p.grad).abs().max().item() / (a.abs().max().item() + 1e-9)
                  for a, p in zip(avg, ddp.module.parameters()))
        assert err < (2e-2 if wire_fp16 else 1e-4), (it, err)
        if it == 0:
            nb0 = len(ddp.bucket_summary())
    assert len(ddp.bucket_summary()) >= 3 and nb0 >= 3
    assert all(k.startswith("module.") for k in ddp.state_dict())
    # no_sync accumulates locally, the next synced backward reduces the sum
    ddp.zero_grad()
    with ddp.no_sync():
        ops.cross_entropy(ddp(x), t).backward()
    local = [p.grad.clone() for p in ddp.module.parameters()]
    ops.cross_entropy(ddp(x), t).backward()
    chk = ddp.module.fc.bias.grad.clone()
    g = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(g, chk)
    assert all(torch.allclose(g[0], gi, atol=1e-6) for gi in g), "grads must be identical after sync"
    assert not torch.allclose(local[-1], chk)
    comm.destroy_process_group()


def apex_flavour(rank, world, port):
    comm = _init(rank, world, port)
    from pytorch_distributed_amd.parallel import ApexDistributedDataParallel
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    ddp = ApexDistributedDataParallel(m, message_size=64, delay_allreduce=(rank >= 0 and False))
    x = torch.full((2, 8), float(rank + 1))
    ddp(x).sum().backward()
    import torch.distributed as dist
    g = ddp.module[2].bias.grad.clone()
    gs = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gs, g)
    assert all(torch.allclose(gs[0], gi) for gi in gs)
    comm.destroy_process_group()


def syncbn_equivalence(rank, world, port):
    comm = _init(rank, world, port)
    from pytorch_distributed_amd import nn as pnn
    torch.manual_seed(0)
    full = torch.randn(4 * world, 5, 5, 8)
    bn = pnn.SyncBatchNorm(8)
    ref = torch.nn.BatchNorm2d(8)
    x = full[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
    xf = full.clone().requires_grad_(True)
    out = bn(x, None, True)
    o = torch.relu(ref(xf.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    assert torch.allclose(out, o[rank * 4:(rank + 1) * 4], atol=1e-5)
    w = torch.linspace(0.5, 1.5, out.numel()).view_as(out)
    wf = torch.cat([w] * world) if False else None
    (out * w).sum().backward()
    wfull = torch.zeros_like(o)
    for r in range(world):
        wfull[r * 4:(r + 1) * 4] = w
    (o * wfull).sum().backward()
    assert torch.allclose(x.grad, xf.grad[rank * 4:(rank + 1) * 4], atol=1e-4)
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, atol=1e-4)
    comm.destroy_process_group()


def shm_collectives(rank, world, port):
    from pytorch_distributed_amd.comm import TCPStore
    from pytorch_distributed_amd.comm.shm import ShmCommunicator, SUM, MAX, AVG
    store = TCPStore("127.0.0.1", port, world, is_master=(rank == 0))
    c = ShmCommunicator(store, rank, world, slot_bytes=4096)       # tiny slots: exercises chunking
    for dt in (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int64, torch.int32):
        t = (torch.arange(3000) % 7 + rank).to(dt)
        c.all_reduce(t, SUM)
        exp = (torch.arange(3000) % 7).double() * world + sum(range(world))
        assert torch.allclose(t.double(), exp, rtol=2e-2), dt
    t = torch.full((10,), float(rank))
    c.all_reduce(t, MAX)
    assert t[0] == world - 1
    t = torch.full((10,), float(rank))
    w = c.all_reduce(t, AVG, async_op=True)
    w.wait()
    assert torch.allclose(t, torch.full((10,), (world - 1) / 2))
    b = torch.arange(5000.0) if rank == 1 else torch.zeros(5000)
    c.broadcast(b, src=1)
    assert torch.equal(b, torch.arange(5000.0))
    out = torch.empty(world * 3)
    c.all_gather(out, torch.full((3,), float(rank)))
    assert out.view(world, 3)[:, 0].tolist() == [float(r) for r in range(world)]
    got = c.all_gather_bytes(b"x" * (rank + 1))
    assert [len(g) for g in got] == list(range(1, world + 1))
    c.barrier()
    c.close()
    store.close()


def file_rendezvous(rank, world, path):
    import torch.distributed as dist
    comm = _init(rank, world, path, method="file")
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    assert t.item() == sum(range(world))
    assert comm.native_store() is not None and comm.native_store().num_keys() > 0
    comm.destroy_process_group()


def crasher(rank, marker):
    import time
    if rank == 1:
        raise ValueError("boom from rank 1")
    time.sleep(60)
    open(marker, "w").write("survived")


def ddp_grad_semantics(rank, world, port):
    """param.grad is the AVERAGED gradient after backward by default, also next to a fused optimizer and a 16-bit
    wire; with fused_wire=True the reduced gradient lives in the wire buckets (reduced_grad / clip_grad_norm_ find
    it there) and the hand-over does not outlive zero_grad() or step()."""
    _init(rank, world, port)
    from pytorch_distributed_amd import models, ops
    from pytorch_distributed_amd.optim import FusedSGD
    from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP
    torch.manual_seed(100)
    X, T = torch.randn(4 * world, 3, 32, 32), torch.randint(0, 10, (4 * world,))
    x, t = X[rank * 4:(rank + 1) * 4], T[rank * 4:(rank + 1) * 4]
    for fused in (None, True):
        torch.manual_seed(7)
        net = models.resnet18(num_classes=10)
        opt = FusedSGD(net.parameters(), lr=0.1, momentum=0.0)
        ddp = DDP(net, bucket_cap_mb=4, wire_dtype=torch.bfloat16, fused_wire=fused)
        assert ddp.fused_wire == bool(fused)
        ref = models.resnet18(num_classes=10)
        ref.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
        ddp.zero_grad()
        ops.cross_entropy(ddp(x), t).backward()
        acc = None                                       # average of the ranks' gradients (BatchNorm per shard)
        for r in range(world):
            ref.zero_grad()
            ops.cross_entropy(ref(X[r * 4:(r + 1) * 4]), T[r * 4:(r + 1) * 4]).backward()
            gs = [p.grad.clone() for p in ref.parameters()]
            acc = gs if acc is None else [a + g for a, g in zip(acc, gs)]
        for p, a in zip(ref.parameters(), acc):
            p.grad = a / world
        ps, rs = list(net.parameters()), list(ref.parameters())
        big = max(range(len(ps)), key=lambda i: ps[i].numel())
        rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()
        assert rel(ddp.reduced_grad(ps[big]), rs[big].grad) < 2e-2
        if fused:
            assert getattr(ps[big], "_pda_wire_grad", None) is not None
            assert rel(ps[big].grad, rs[big].grad) > 1e-3     # local gradient: another batch
        else:
            assert getattr(ps[big], "_pda_wire_grad", None) is None
            assert rel(ps[big].grad, rs[big].grad) < 2e-2      # what clipping / logging code reads
        ref_norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        norm = ddp.clip_grad_norm_(0.5)
        assert abs(float(norm) - float(ref_norm)) / float(ref_norm) < 2e-2, (float(norm), float(ref_norm))
        assert float(ref_norm) > 0.5                            # the clip was active
        assert rel(ddp.reduced_grad(ps[big]), rs[big].grad) < 3e-2
        w0 = ps[big].detach().clone()
        opt.step()
        assert rel(ps[big].detach() - w0, -0.1 * rs[big].grad) < 3e-2
        assert all(getattr(p, "_pda_wire_grad", None) is None for p in ps)     # consumed by the step
        if fused:
            ops.cross_entropy(ddp(x), t).backward()
            assert getattr(ps[big], "_pda_wire_grad", None) is not None
            ddp.zero_grad()
            assert all(getattr(p, "_pda_wire_grad", None) is None for p in ps)


def ddp_native_hooks(rank, world, port):
    """The C ready hooks (csrc/pyext/ready_hook.cc) against the Python closures: same reduced gradients after a
    plain backward, after zero_grad(set_to_none=True) replaced every .grad, and across a no_sync() accumulation."""
    _init(rank, world, port)
    import os
    from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP, _pyext
    assert _pyext.module() is not None, "the extension is built in-tree by python -m pytorch_distributed_amd._build"
    torch.manual_seed(3)
    X, T = torch.randn(6 * world, 20), torch.randn(6 * world, 4)
    x, t = X[rank * 6:(rank + 1) * 6], T[rank * 6:(rank + 1) * 6]

    def run(native):
        os.environ["PDA_DDP_NATIVE_HOOKS"] = "1" if native else "0"
        torch.manual_seed(11)
        net = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                                  torch.nn.Linear(64, 4))
        ddp = DDP(net, bucket_cap_mb=0.01, first_bucket_mb=0.005)   # several buckets
        assert bool(ddp._native_hooks) == native
        out = []
        ((ddp(x) - t) ** 2).mean().backward()
        out.append([p.grad.clone() for p in net.parameters()])
        net.zero_grad(set_to_none=True)                     # every .grad is replaced by a fresh tensor next time
        ((ddp(x) - t) ** 2).mean().backward()
        out.append([p.grad.clone() for p in net.parameters()])
        assert all(p.grad.data_ptr() == ddp._view_of[i].data_ptr() for i, p in enumerate(net.parameters()))
        ddp.zero_grad()
        with ddp.no_sync():
            ((ddp(x) - t) ** 2).mean().backward()
            local = [p.grad.clone() for p in net.parameters()]
        ((ddp(x) - t) ** 2).mean().backward()               # reduces the sum of both passes
        out.append([p.grad.clone() for p in net.parameters()])
        out.append(local)
        return out

    a, b = run(True), run(False)
    for ga, gb in zip(a, b):
        for u, v in zip(ga, gb):
            assert torch.allclose(u, v, rtol=1e-6, atol=1e-7)
    # the local pass of no_sync differs between the ranks, the reduced ones do not
    import torch.distributed as dist
    gathered = [torch.zeros_like(a[2][0]) for _ in range(world)]
    dist.all_gather(gathered, a[2][0])
    assert all(torch.equal(g, gathered[0]) for g in gathered)
