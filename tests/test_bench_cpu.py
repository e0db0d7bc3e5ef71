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
    # ... and nothing else: what libraries write to file descriptor 1 (gloo's connection notes here, RCCL's banner on
    # the GPU) is sent to stderr, stdout is the JSON line
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout
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


def _bench(extra, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PDA_BENCH_DEVICE="cpu", OMP_NUM_THREADS="2")
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"] + _SMALL + extra,
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_bench_hvd_engine_with_eval_two_ranks_cpu():
    """--engine hvd: hvd.init / broadcast_parameters / broadcast_optimizer_state / DistributedOptimizer with fp16
    compression + SyncBatchNorm, then the forward-only evaluation whose metric sums cross the ranks in ONE
    all-reduce (BASELINE config 5's shape), under the same supervisor."""
    r, lines = _bench(["--engine", "hvd", "--precision", "fp16", "--sync-bn", "--eval-steps", "2"])
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["engine"] == "hvd" and cfg["sync_bn"] is True
    assert cfg["hvd_engine_stats"]["fused_groups"] > 0 and "hvd" in cfg["optimizer"]
    assert out["eval_images_per_sec"] > 0 and cfg["eval"]["eval_rows"] == 2 * 2 * 2
    assert cfg["eval"]["eval_metric_allreduces"] == 1 and 0.0 <= cfg["eval"]["eval_acc1"] <= 1.0


def test_bench_selftest_and_bucket_sweep_cpu():
    r, lines = _bench(["--selftest", "--bucket-mb", "1,4,25", "--precision", "fp32"],
                      env_extra={"PDA_BENCH_WATCHDOG": "1"})
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1000:], r.stderr[-3000:])
    cfg = json.loads(lines[0])["config"]
    assert cfg["selftest"].startswith("passed") and cfg["watchdog"] is True
    sweep = cfg["bucket_sweep"]
    assert [s["bucket_mb"] for s in sweep] == [1.0, 4.0, 25.0] and all(s["ms_per_step"] > 0 for s in sweep)
    assert cfg["bucket_mb"] in (1.0, 4.0, 25.0)
    assert json.loads(lines[0])["ms_per_step"] == min(s["ms_per_step"] for s in sweep)
    assert sweep[0]["buckets"] > sweep[2]["buckets"]
