"""Launch-style examples end to end on one MI355X (one rank each: the multi-rank variants run on the
CPU backends in test_examples_cpu.py), including the hipGraph-replayed training loop."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")
COMMON = ["--arch", "resnet18", "--batch-size", "64", "--epochs", "2", "--steps-per-epoch", "8", "--workers", "0",
          "--num-classes", "10", "-p", "4", "--synthetic"]
LAUNCH = [sys.executable, "-m", "pytorch_distributed_amd.launch"]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    return r.stdout


def test_style2_ddp_one_gpu_graph_replay(tmp_path):
    out = run(LAUNCH + ["--nproc_per_node=1", "--master_port", str(free_port()), "--use_argv_local_rank",
                        os.path.join(EX, "2_distributed.py")] + COMMON + ["--graph"], tmp_path)
    assert out.count(" * Acc@1") == 2 and "Epoch: [1]" in out
    assert (tmp_path / "resnet18.checkpoint.pth.tar").exists()


def test_style4_amp_one_gpu_graph_replay(tmp_path):
    out = run(LAUNCH + ["--nproc_per_node=1", "--master_port", str(free_port()), "--use_argv_local_rank",
                        os.path.join(EX, "4_amp_distributed.py")] + COMMON + ["--graph", "--opt-level", "O2"], tmp_path)
    assert out.count(" * Acc@1") == 2


def test_style3_self_spawn_one_gpu(tmp_path):
    out = run([sys.executable, os.path.join(EX, "3_multiprocessing_distributed.py")] + COMMON, tmp_path)
    assert out.count(" * Acc@1") == 2


def test_style5_horovod_one_gpu(tmp_path):
    out = run(LAUNCH + ["-np", "1", "-H", "localhost:1", "--master_port", str(free_port()),
                        os.path.join(EX, "5_horovod_distributed.py")] + COMMON, tmp_path)
    assert out.count(" * Acc@1") == 2


def test_style5_horovod_mixed_precision_fused_optimizer(tmp_path):
    """--precision fp16: the Horovod wrapper around this package's fused optimizer under amp O2.  The fused conv /
    BatchNorm / Linear ops accumulate their gradients in place and announce them themselves, and autograd still
    runs each parameter's post-accumulate hook afterwards: every gradient must be reduced exactly once per step."""
    out = run(LAUNCH + ["-np", "1", "-H", "localhost:1", "--master_port", str(free_port()),
                        os.path.join(EX, "5_horovod_distributed.py")] + COMMON +
              ["--precision", "fp16", "--opt-level", "O2"], tmp_path)
    assert out.count(" * Acc@1") == 2


def test_hvd_optimizer_combinations_one_gpu(tmp_path, monkeypatch):
    """hvd.DistributedOptimizer around the fused and the stock optimizer, with and without mixed precision, fp16
    compression and local accumulation (backward_passes_per_step): each trains (tests/workers/hvd_combos_gpu.py)."""
    monkeypatch.setenv("MASTER_PORT", str(free_port()))
    out = run([sys.executable, os.path.join(ROOT, "tests", "workers", "hvd_combos_gpu.py")], tmp_path)
    assert out.count(" ok ") == 7 and "FAIL" not in out, out


def test_ddp_combinations_one_gpu(tmp_path, monkeypatch):
    """DistributedDataParallel (one rank, own RCCL communicator) with the fused and the stock optimizer, O0 / O1 / O2,
    no_sync() accumulation, the apex-flavoured wrapper and SyncBatchNorm: each trains (tests/workers/ddp_combos_gpu.py)."""
    monkeypatch.setenv("MASTER_PORT", str(free_port()))   # the worker's rendezvous store listens there
    out = run([sys.executable, os.path.join(ROOT, "tests", "workers", "ddp_combos_gpu.py")], tmp_path)
    assert out.count(" ok ") == 10 and "FAIL" not in out, out


@pytest.mark.parametrize("arch,extra", [("mobilenet_v2", ["--graph"]), ("densenet121", []), ("shufflenet_v2_x1_0", []),
                                        ("resnext50_32x4d", ["--graph"]), ("googlenet", [])])
def test_other_arch_amp_ddp_one_gpu(tmp_path, arch, extra):
    """``--arch`` beyond the ResNets through the AMP + DDP launch style (with and without hipGraph replay)."""
    common = [a if a != "resnet18" else arch for a in COMMON]
    out = run(LAUNCH + ["--nproc_per_node=1", "--master_port", str(free_port()), "--use_argv_local_rank",
                        os.path.join(EX, "4_amp_distributed.py")] + common + extra + ["--opt-level", "O2"], tmp_path)
    assert out.count(" * Acc@1") == 2
    assert (tmp_path / f"{arch}.checkpoint.pth.tar").exists()


def test_other_arch_dataparallel_one_gpu(tmp_path):
    common = [a if a != "resnet18" else "squeezenet1_1" for a in COMMON]
    out = run([sys.executable, os.path.join(EX, "1_dataparallel.py")] + common, tmp_path)
    assert out.count(" * Acc@1") == 2


def test_image_directory_pipeline_on_the_gpu(tmp_path):
    """--dataset imagenet --data DIR on the device path: PNG files -> uint8 batches -> normalisation inside the stem
    kernel.  The uint8 route gives the model output of the ToTensor + Normalize route, and the launch-style example
    trains from the directory (ResNet-50 stem: fused normalise + conv + BatchNorm + ReLU + pool)."""
    PIL = pytest.importorskip("PIL")
    import numpy as np
    import torch
    from PIL import Image
    from pytorch_distributed_amd import data as D, models
    rng = np.random.RandomState(0)
    root = tmp_path / "imgs"
    for split in ("train", "val"):
        for cls in ("n01_cat", "n02_dog"):
            d = root / split / cls
            d.mkdir(parents=True)
            for i, (w, h) in enumerate(((300, 400), (500, 333), (224, 224), (256, 256))):
                Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(d / f"img_{i}.png")
    dev = torch.device("cuda", 0)
    vu, vf = D.imagenet(str(root), False), D.imagenet(str(root), False, uint8=False)
    xu = torch.stack([vu[i][0] for i in range(4)]).to(dev)
    xf = torch.stack([vf[i][0] for i in range(4)]).to(dev)
    assert xu.dtype == torch.uint8
    torch.manual_seed(0)
    m = models.get_model("resnet50", num_classes=8).to(dev).eval()
    with torch.no_grad():
        ref = m(xf)
        m.set_input_normalization(D.IMAGENET_MEAN, D.IMAGENET_STD, 1.0 / 255.0)
        out = m(xu)
    assert torch.allclose(out.float(), ref.float(), atol=2e-3, rtol=2e-3), (out - ref).abs().max().item()
    txt = run([sys.executable, os.path.join(EX, "6_distributed_slurm.py"), "--dataset", "imagenet", "--data", str(root),
               "-a", "resnet50", "-b", "4", "--epochs", "1", "--num-classes", "2", "-j", "2", "-p", "1",
               "--precision", "fp16", "--opt-level", "O2"], tmp_path)
    assert txt.count(" * Acc@1") == 1 and "Epoch: [0][1/2]" in txt


def test_bench_with_stock_comparator_in_a_child_process(tmp_path):
    """`python bench.py` with the comparator on (a tiny configuration): stdout is ONE JSON line carrying the native
    number and `vs_stock`, measured by a child process whose failure could not have cost the native result."""
    import json
    out = run([sys.executable, os.path.join(ROOT, "bench.py"), "--arch", "resnet18", "--image-size", "32",
               "--batch-size", "64", "--num-classes", "10", "--steps", "3", "--warmup", "2", "--compare-stock"], tmp_path)
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 1
    assert isinstance(d["config"]["stock_images_per_sec"], float) and d["config"]["stock_images_per_sec"] > 0
    assert d["vs_stock"] == pytest.approx(d["value"] / d["config"]["stock_images_per_sec"], rel=1e-2)
    assert "child process" in d["config"]["stock_comparator"]
