"""bench.py's multi-rank control flow (rendezvous from the launcher's env, DDP wrap after amp.initialize,
barrier-bracketed timing, MAX over ranks, one JSON line from rank 0) rehearsed with two gloo ranks on the CPU."""
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
