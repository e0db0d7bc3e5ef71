"""bench.py's multi-rank control flow (self-launch of N ranks or rendezvous from an external launcher's env,
DDP wrap after amp.initialize, barrier-bracketed timing, MAX over ranks, one JSON line from rank 0, refusal to
report a job of a different size than asked for) rehearsed with two gloo ranks on the CPU."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_bench_two_ranks_cpu(precision):
    env = dict(os.environ, PDA_BENCH_DEVICE="cpu", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--arch", "resnet18", "--batch-size", "2",
           "--image-size", "32", "--num-classes", "10", "--precision", precision]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["value"] == pytest.approx(4 * 2 / (out["ms_per_step"] * 2 / 1000), rel=1e-2)


_SMALL = ["--steps", "2", "--warmup", "1", "--arch", "resnet18", "--batch-size", "2", "--image-size", "32",
          "--num-classes", "10"]


def test_bench_self_launches_two_ranks_cpu():
    """`python bench.py --gpus 2` with no launcher around it: the supervisor starts both ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PDA_BENCH_DEVICE="cpu", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--precision", "fp16"] + _SMALL,
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4
    assert out["config"]["comm_backend"] and out["config"]["buckets"] and out["config"]["wire_dtype"] == "float16"
    assert out["value"] > 0


def test_bench_refuses_rank_count_mismatch_cpu():
    """One rank started by a launcher while --gpus says 2: an error line and a non-zero exit, not a 1-rank number."""
    env = dict(os.environ, PDA_BENCH_DEVICE="cpu", OMP_NUM_THREADS="2", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"] + _SMALL,
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "error" in json.loads(lines[0]) and "metric" not in json.loads(lines[0])


def test_bench_stage_deadline_reports_instead_of_hanging_cpu():
    """Rank 1 never shows up: rank 0's rendezvous stage hits its deadline and the run ends with an error line."""
    env = dict(os.environ, PDA_BENCH_DEVICE="cpu", OMP_NUM_THREADS="2", RANK="0", WORLD_SIZE="2", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--stage-timeout", "6"] + _SMALL,
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines and "error" in json.loads(lines[-1])
