"""Peer-memory collectives (one-shot latency kernel, two-shot bandwidth kernel) with several processes on one GPU."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "p2p_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_p2p_allreduce_single_rank():
    import torch
    from pytorch_distributed_amd.comm.p2p import P2PAllReduce
    ar = P2PAllReduce(0, 1, store=None, max_elems=1024)
    t = torch.randn(1000, device="cuda")
    ref = t.clone()
    assert torch.equal(ar.all_reduce(t, "sum"), ref)
    assert torch.equal(ar.all_reduce(t, "avg"), ref)
    with pytest.raises(ValueError):
        ar.all_reduce(torch.zeros(2000, device="cuda"))
    ar.check()
    ar.close()


def _run_ranks(worker, world, attempts=3, extra=()):
    """Start `world` ranks; a rendezvous port that somebody else grabbed between probing and binding is
    the one failure worth a second try."""
    for attempt in range(attempts):
        procs, outs = _start_ranks(worker, world, extra)
        if attempt + 1 < attempts and any("rendezvous store" in o for o in outs):
            continue
        break
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r}: OK" in o, (r, o[-6000:])


def _start_ranks(worker, world, extra=()):
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), *extra], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=100)
            outs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_p2p_allreduce_processes_on_one_gpu(world):
    _run_ranks(WORKER, world)


def test_p2p_twoshot_single_rank():
    import torch
    from pytorch_distributed_amd.comm.p2p import P2PTwoShot
    ar = P2PTwoShot(0, 1, store=None, cap_bytes=1 << 16)
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        t = torch.randn(50001, device="cuda").to(dtype)      # three pieces
        ref = t.clone()
        assert torch.equal(ar.all_reduce(t, "sum"), ref)
        out = torch.empty_like(t)
        assert torch.equal(ar.all_reduce(t, "avg", out=out), ref)
    ar.check()
    ar.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_p2p_twoshot_processes_on_one_gpu(world):
    """Direct reduce-scatter + all-gather kernel: 1 element .. 3 region capacities, fp32 / fp16 / bf16, sum / avg,
    in place / out of place vs an fp64 reference; all ranks end with identical bits."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "p2p2_worker.py"), world)


def test_syncbn_statistics_over_p2p():
    """SyncBatchNorm forward/backward == BatchNorm over the concatenated batch, with the [2, C] statistics
    reduced by the peer-memory kernel (PDA_P2P_SMALL=1) under a gloo process group."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "p2p_syncbn_worker.py"), 2)


def test_syncbn_resnet_two_ranks_matches_full_batch():
    """Fused conv+SyncBatchNorm ResNet-18 on two ranks == plain BatchNorm on the concatenated batch."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "syncbn_resnet_worker.py"), 2)


@pytest.mark.parametrize("world", [2, 8])
def test_ddp_ranks_on_one_gpu_match_full_batch(world):
    """The device-side multi-rank reducer path (2 and 8 processes, peer-memory collectives): DDP +
    SyncBatchNorm + FusedSGD for two steps == one process on the concatenated batch."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "ddp_p2p_worker.py"), world)


@pytest.mark.parametrize("world", [2, 4])
def test_ddp_o2_fused_fp16_wire_ranks_on_one_gpu(world):
    """amp O2 + DDP with per-rank different initial weights: the optimizer-maintained half copies follow the
    parameter broadcast, gradients travel as fp16 (pre-divided by the world size) through the two-shot kernel
    on its own stream and FusedSGD consumes them from the wire buckets; first updates == full-batch reference."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "ddp_p2p_worker.py"), world, extra=("O2",))


@pytest.mark.parametrize("world", [2, 4])
def test_hvd_optimizer_ranks_on_one_gpu(world):
    """hvd.DistributedOptimizer over device tensors, engine collectives through the peer-memory backend: fused
    pack/cast/unpack groups (<= 3 launches each), a real all-gather, and device Adasum (halving / doubling with
    the HIP dot / combine kernels) against the fp64 host tree."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "hvd_p2p_worker.py"), world)


@pytest.mark.parametrize("mode", ["fused", "overflow", "legacy-overflow"])
def test_hvd_fused_optimizer_under_amp_ranks_on_one_gpu(mode):
    """amp.scale_loss + hvd.DistributedOptimizer(FusedSGD, fp16 compression) on a ResNet-18 whose fused ops
    announce their gradients themselves, with the engine thread slowed down so that every reduction is in flight
    when backward returns.  ``fused``: identical parameters after four steps.  ``overflow``: an infinity on one
    rank makes all ranks skip.  ``legacy-overflow``: the same input through the old order (overflow pass first,
    reductions waited for in step()) must show the ranks disagreeing - deterministically, thanks to the delay."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "hvd_amp_p2p_worker.py"), 2, extra=(mode,))


def test_overflow_check_stores_nothing():
    import torch
    from pytorch_distributed_amd.ops import multi_tensor as mt
    src = [torch.randn(70001, device="cuda"), torch.randn(5, device="cuda").half()]
    for dt_group in ([src[0]], [src[1]]):
        dst = [torch.full_like(t, 7.0) for t in dt_group]
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        mt.scale(mt.TensorLists([dt_group, dst]), 3.0, noop_flag=flag, check_only=True)
        assert all(bool((d == 7.0).all()) for d in dst) and int(flag) == 0
        dt_group[0][3] = float("inf")
        mt.scale(mt.TensorLists([dt_group, dst]), 3.0, noop_flag=flag, check_only=True)
        assert all(bool((d == 7.0).all()) for d in dst) and int(flag) == 1


@pytest.mark.parametrize("world", [1, 2])
def test_graphed_ddp_step_matches_eager(world):
    """The DistributedDataParallel step (bucket pack, all-reduce on the communicator's stream, fused-wire SGD,
    side-stream weight gradients) recorded into a hipGraph and replayed == the eager step, bit for bit.  One rank
    on the own RCCL communicator; two ranks sharing the GPU on the peer-memory two-shot kernel."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "ddp_graph_worker.py"), world)


def test_bench_supervisor_with_two_ranks_sharing_the_gpu():
    """``python bench.py --gpus 2`` end to end on the one-GPU box: the supervisor starts two ranks through the
    package's launcher, both use GPU 0 (PDA_BENCH_SHARE_GPU=1) and reduce their buckets through the peer-memory
    two-shot kernel; the JSON line must describe a two-rank job and mark itself as a rehearsal."""
    import json
    env = dict(os.environ, PDA_BENCH_SHARE_GPU="1", PDA_BENCH_STAGE_TIMEOUT="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "p2p2",
                        "--arch", "resnet18", "--batch-size", "32", "--image-size", "64", "--steps", "4", "--warmup", "2"],
                       capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-800:], r.stderr[-1500:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64
    assert d["config"]["comm_backend"] == "p2p-twoshot" and d["config"]["wire_dtype"] == "float16"
    assert "rehearsal" in d["config"]["device"] and d["value"] > 0
    assert len(d["config"]["buckets"]) >= 1


def test_engine_graph_replay_on_two_ranks():
    """engine.train --graph under DistributedDataParallel on two ranks (sharing the GPU, peer-memory all-reduce):
    the step is recorded after three eager steps on every rank and the epoch ends with the eager epoch's
    parameters, bit for bit."""
    _run_ranks(os.path.join(ROOT, "tests", "workers", "engine_graph_worker.py"), 2)
