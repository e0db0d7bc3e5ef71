"""The six launch styles + the MNIST example, end to end on the CPU backends (tiny settings)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")
COMMON = ["--arch", "resnet18", "--batch-size", "16", "--epochs", "1", "--steps-per-epoch", "2", "--cpu", "--workers",
          "0", "--num-classes", "10", "-p", "1"]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


LAUNCH = [sys.executable, "-m", "pytorch_distributed_amd.launch"]


def test_config1_horovod_mnist_single_process_cpu(tmp_path):
    out = run([sys.executable, os.path.join(EX, "5_2_horovod_mnist.py"), "--no-cuda", "--epochs", "1",
               "--max-batches", "12"], tmp_path)
    assert "Train Epoch: 1" in out and "Test set: Average loss" in out


def test_horovod_mnist_two_ranks_adasum(tmp_path):
    out = run(LAUNCH + ["-np", "2", "-H", "localhost:2", "--master_port", str(free_port()),
                        os.path.join(EX, "5_2_horovod_mnist.py"), "--no-cuda", "--epochs", "1", "--max-batches", "6",
                        "--use-adasum", "--no-fp16-allreduce"], tmp_path)
    assert out.count("Test set: Average loss") == 1          # rank 0 only


def test_style1_dataparallel(tmp_path):
    out = run([sys.executable, os.path.join(EX, "1_dataparallel.py")] + COMMON, tmp_path)
    assert " * Acc@1" in out and (tmp_path / "dataparallel.csv").exists()
    assert (tmp_path / "resnet18.checkpoint.pth.tar").exists()


def test_style2_launcher_ddp_and_resume(tmp_path):
    out = run(LAUNCH + ["--nproc_per_node=2", "--master_port", str(free_port()), "--use_argv_local_rank",
                        os.path.join(EX, "2_distributed.py")] + COMMON, tmp_path)
    assert out.count(" * Acc@1") == 1 and "Epoch: [0]" in out
    ck = tmp_path / "resnet18.checkpoint.pth.tar"
    assert ck.exists()
    out = run(LAUNCH + ["--nproc_per_node=2", "--master_port", str(free_port()), os.path.join(EX, "2_distributed.py")]
              + COMMON + ["--resume", str(ck), "--epochs", "2"], tmp_path)
    assert "resumed from" in out and "Epoch: [1]" in out and "Epoch: [0]" not in out


def test_style3_self_spawn(tmp_path):
    out = run([sys.executable, os.path.join(EX, "3_multiprocessing_distributed.py")] + COMMON +
              ["--dist-url", f"tcp://127.0.0.1:{free_port()}"], tmp_path)
    assert " * Acc@1" in out


def test_style4_amp_ddp(tmp_path):
    out = run(LAUNCH + ["--nproc_per_node=2", "--master_port", str(free_port()), "--use_argv_local_rank",
                        os.path.join(EX, "4_amp_distributed.py")] + COMMON, tmp_path)
    assert "[amp] opt_level=O1" in out and " * Acc@1" in out


def test_style5_horovod(tmp_path):
    out = run(LAUNCH + ["-np", "2", "-H", "localhost:2", "--master_port", str(free_port()),
                        os.path.join(EX, "5_horovod_distributed.py")] + COMMON, tmp_path)
    assert " * Acc@1" in out and (tmp_path / "checkpoint.pth.tar").exists()


def test_style6_slurm_spawn(tmp_path):
    out = run([sys.executable, os.path.join(EX, "6_distributed_slurm.py")] + COMMON + ["--dist-file", "rdv"], tmp_path)
    assert " * Acc@1" in out and (tmp_path / "distributed.csv").exists()


@pytest.mark.parametrize("arch", ["squeezenet1_1", "densenet121"])
def test_other_arch_through_ddp_launcher(tmp_path, arch):
    """``--arch`` reaches the other model families (/root/reference/2.distributed.py:28-33)."""
    common = [a if a != "resnet18" else arch for a in COMMON]
    out = run(LAUNCH + ["--nproc_per_node=2", "--master_port", str(free_port()), os.path.join(EX, "2_distributed.py")]
              + common, tmp_path)
    assert " * Acc@1" in out and (tmp_path / f"{arch}.checkpoint.pth.tar").exists()
