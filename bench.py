#!/usr/bin/env python
"""Headline benchmark: ResNet-50 training throughput (images/sec, whole job) on
synthetic ImageNet-shaped data, AMP fp16 with dynamic loss scaling, bucketed-all-reduce
DDP over RCCL when run on more than one GPU.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python bench.py --gpus 8 --steps 30 --warmup 10          # starts its own 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 30 --warmup 10   # or under any launcher

Launch: with ``--gpus N`` > 1 and no launcher environment (no ``WORLD_SIZE``), this process becomes a
supervisor that never touches a GPU: it starts N ranks through this package's launcher
(``python -m pytorch_distributed_amd.launch``, the way /root/reference/3.multiprocessing_distributed.py:80-84
spawns its own workers), relays rank 0's JSON line and exits with the first failing rank's code.  If the
ranks fail or stall on this package's own RCCL communicator, the supervisor runs them once more on
``torch.distributed``'s process group and says so in ``comm_backend``.  Under an external launcher the
ranks come from its environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Either way a rank
count that differs from ``--gpus`` is an error, not a silently smaller job.

One timed step = zero_grad + forward + loss + scaled backward (+ gradient all-reduce overlapped with
backward) + overflow check + fused momentum-SGD update of all 25.6 M parameters.  Weak scaling: the
per-GPU batch is fixed.  Rank 0 prints ONE JSON line.  Every blocking stage (rendezvous, communicator
creation, first collective, warm-up, timed loop) runs under a deadline: a stage that exceeds it prints a
JSON ``{"error": ...}`` line and ends the process instead of hanging.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--batch-size", type=int, default=400,
                    help="per-GPU batch (default 400 = the reference's global -b 3200 over 8 GPUs, "
                         "/root/reference/2.distributed.py:41-48,113)")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--num-classes", type=int, default=1000)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--opt-level", default="O2", choices=["O0", "O1", "O2", "O3"])
    ap.add_argument("--engine", default="native", choices=["native", "hvd", "stock"],
                    help="native = this framework's kernels + DistributedDataParallel + AMP; hvd = the same kernels "
                         "under the Horovod-compatible API (hvd.init, broadcast_parameters / optimizer_state, "
                         "hvd.DistributedOptimizer with fp16 compression: /root/reference/5.horovod_distributed.py:"
                         "116-125); stock = plain PyTorch-ROCm comparator")
    ap.add_argument("--bucket-mb", default="25",
                    help="DDP bucket size in MiB; a comma list (8,16,25,32,64) times every value in this one "
                         "invocation, reports them as bucket_sweep and takes the best as the headline")
    ap.add_argument("--eval-steps", type=int, default=0,
                    help="after the training timing: this many forward-only steps in eval mode whose loss / accuracy "
                         "sums stay on the device and cross the ranks in ONE fused all-reduce at the end (the "
                         "distributed evaluation of /root/reference/5.horovod_distributed.py:231-272)")
    ap.add_argument("--selftest", action="store_true",
                    help="before timing, run all_reduce (fp16 and fp32, 1 KB and 32 MB), broadcast, all_gather and "
                         "barrier on the data-plane communicator against closed-form values")
    ap.add_argument("--comm", default="auto", choices=["auto", "native", "torch", "p2p2"],
                    help="gradient all-reduce back-end: native = this package's RCCL communicator with its own "
                         "comm stream (auto picks it on GPUs), torch = torch.distributed's process group, "
                         "p2p2 = the peer-memory two-shot all-reduce kernel")
    ap.add_argument("--wire-dtype", default="auto", choices=["auto", "fp16", "bf16", "fp32"],
                    help="gradient dtype on the links (auto = the AMP compute dtype)")
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--channels-last", action="store_true", help="stock engine only")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=-1)
    ap.add_argument("--graph", action="store_true",
                    help="record the step into a hipGraph after warm-up and replay it (single GPU, native engine)")
    ap.add_argument("--native-comm", action="store_true", help="same as --comm native")
    ap.add_argument("--autotune", action=argparse.BooleanOptionalAction, default=True,
                    help="backends.benchmark: time the candidate conv tile configurations per layer shape during "
                         "warm-up (the reference sets cudnn.benchmark = True, /root/reference/2.distributed.py:122)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="wrap the model in DistributedDataParallel even for one rank (exercises the bucketed "
                         "all-reduce path on a single GPU)")
    ap.add_argument("--compare-stock", action=argparse.BooleanOptionalAction, default=None,
                    help="after the native timing, time the stock PyTorch-ROCm engine (channels-last autocast + "
                         "GradScaler + torch SGD, same model / batch / steps) in a child process right after the native run and report "
                         "vs_stock = native / stock.  Default: on for a single-GPU native run of at most 40 timed "
                         "steps, off otherwise (on a fresh box the comparator adds about 100 s: the stock convs' first-run "
                         "kernel search; a comparator that fails or hangs costs its timeout, never the native result)")
    ap.add_argument("--stage-timeout", type=float, default=float(os.environ.get("PDA_BENCH_STAGE_TIMEOUT", "300")),
                    help="seconds any single blocking stage may take before the run is declared hung")
    ap.add_argument("--master-port", type=int, default=0, help="supervisor mode: rendezvous port (0 = pick a free one)")
    args = ap.parse_args(argv)
    try:
        args.bucket_list = [float(v) for v in str(args.bucket_mb).split(",") if v.strip() != ""]
    except ValueError:
        ap.error(f"--bucket-mb: not a number or comma list of numbers: {args.bucket_mb!r}")
    if not args.bucket_list or any(v <= 0 for v in args.bucket_list):
        ap.error("--bucket-mb needs positive sizes")
    args.bucket_mb = args.bucket_list[0]
    if args.compare_stock is None:
        args.compare_stock = (args.gpus == 1 and args.engine == "native" and args.steps <= 40 and not args.force_ddp
                              and os.environ.get("PDA_BENCH_DEVICE", "") != "cpu")
    return args


# ------------------------------------------------------------------------------------------------
# supervisor: python bench.py --gpus N with no launcher around it
# ------------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(args, argv, comm: str):
    """Start args.gpus ranks through the package's launcher; returns (rc, json line or None, tail of output)."""
    port = args.master_port or _free_port()
    cmd = [sys.executable, "-u", "-m", "pytorch_distributed_amd.launch", "--nproc_per_node", str(args.gpus),
           "--master_addr", "127.0.0.1", "--master_port", str(port), os.path.abspath(__file__)] + argv
    if comm:
        cmd += ["--comm", comm]
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # whole-run bound: every stage of every rank has its own deadline; this one only catches a launcher that
    # itself stops responding
    limit = 6 * args.stage_timeout + 120
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT,
                         start_new_session=True)
    lines = []

    def pump():
        for line in p.stdout:
            line = line.rstrip("\n")
            lines.append(line)
            if not line.startswith("{"):
                print(line, file=sys.stderr, flush=True)     # live: a run that stops shows where
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        rc = p.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, 15)
            time.sleep(5)
            os.killpg(p.pid, 9)
        except OSError:
            pass
        rc = 124
    t.join(timeout=5)
    result = err = None
    for l in lines:
        if l.startswith("{"):
            try:
                d = json.loads(l)
            except ValueError:
                continue
            if "metric" in d and "value" in d:
                result = d
            elif "error" in d and err is None:
                err = d
    return rc, result, err


def supervise(args, argv) -> int:
    # drop the options the supervisor sets itself
    clean, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in ("--comm", "--master-port"):
            skip = True
            continue
        if a.startswith("--comm=") or a.startswith("--master-port=") or a == "--native-comm":
            continue
        clean.append(a)
    first = "native" if args.native_comm and args.comm == "auto" else args.comm
    attempts = [first]
    if first in ("auto", "native") and os.environ.get("PDA_BENCH_DEVICE", "") != "cpu" and args.engine == "native":
        attempts.append("torch")
    last_err = None
    for i, comm in enumerate(attempts):
        rc, result, err = _run_ranks(args, clean, comm)
        if rc == 0 and result is not None:
            if result.get("n_gpus") != args.gpus:
                print(json.dumps({"error": f"asked for {args.gpus} ranks, the job reported {result.get('n_gpus')}"}))
                return 1
            if i > 0:
                result.setdefault("config", {})["comm_fallback_from"] = attempts[0]
                result["config"]["comm_fallback_reason"] = (last_err or {}).get("error", "first attempt failed")
            print(json.dumps(result), flush=True)
            return 0
        last_err = err or {"error": f"ranks exited with code {rc} and no result line"}
        print(f"[bench] attempt with --comm {comm} failed (rc {rc}): {last_err.get('error')}", file=sys.stderr, flush=True)
    print(json.dumps(last_err), flush=True)
    return 1


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
class Deadline:
    """``with deadline.stage("name"):`` - a daemon thread ends the process with an error line if the block
    does not finish within the stage timeout (a hung rendezvous / collective never raises by itself)."""

    def __init__(self, seconds: float, rank: int):
        self.seconds, self.rank = seconds, rank
        self._until, self._name = None, ""
        self._lock = threading.Lock()
        threading.Thread(target=self._watch, daemon=True).start()

    def _watch(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                until, name = self._until, self._name
            if until is not None and time.time() > until:
                msg = {"error": f"rank {self.rank}: stage '{name}' did not finish within {self.seconds:.0f} s "
                                "(hung collective or rendezvous)", "stage": name, "rank": self.rank}
                print(json.dumps(msg), flush=True)
                os._exit(3)

    def stage(self, name: str, factor: float = 1.0):
        dl = self

        class _S:
            def __enter__(self_s):
                with dl._lock:
                    dl._until, dl._name = time.time() + dl.seconds * factor, name
                if dl.rank == 0:   # progress on stderr: a run that stops shows where
                    print(f"[bench] {time.strftime('%H:%M:%S')} stage: {name}", file=sys.stderr, flush=True)

            def __exit__(self_s, *exc):
                with dl._lock:
                    dl._until = None
                return False
        return _S()


def worker(args) -> int:
    import datetime

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(max(args.local_rank, 0))))
    # PDA_BENCH_DEVICE=cpu: control-flow rehearsal of the multi-rank path (gloo, CPU reference ops) for the
    # test-suite; numbers from it are not benchmark results and the JSON says so in "device".
    on_cpu = os.environ.get("PDA_BENCH_DEVICE", "") == "cpu"
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} but the launcher started {world} rank(s) "
                                       "(WORLD_SIZE); refusing to report a job of a different size"}), flush=True)
        return 2
    if not on_cpu and not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; bench.py needs an MI355X"}))
        return 2
    # PDA_BENCH_SHARE_GPU=1: every rank uses GPU 0 - a rehearsal of the multi-rank path (supervisor, launcher,
    # rendezvous, DDP buckets, side stream, fused wire) on a one-GPU box.  RCCL refuses two ranks on one device, so
    # the collectives have to be the peer-memory ones (--comm p2p2); the JSON marks the run as not a result.
    share_gpu = os.environ.get("PDA_BENCH_SHARE_GPU", "") == "1" and not on_cpu
    if share_gpu:
        local_rank = 0
        # SyncBatchNorm statistics and the metric vector: through the one-shot peer-memory kernel as well
        os.environ.setdefault("PDA_P2P_SMALL", "1")
        if world > 1 and args.comm not in ("p2p2",):
            if rank == 0:
                print(json.dumps({"error": "PDA_BENCH_SHARE_GPU=1 needs --comm p2p2 (RCCL cannot put two ranks "
                                           "on one device)"}), flush=True)
            return 2
    if not on_cpu and torch.cuda.device_count() <= local_rank:
        print(json.dumps({"error": f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} "
                                   "GPU(s) visible"}), flush=True)
        return 2
    cpu_binding = None
    if on_cpu:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)      # before anything else touches the device
        dev = torch.device("cuda", local_rank)
        from pytorch_distributed_amd.utils.affinity import bind_to_gpu
        bound = bind_to_gpu(local_rank)        # host thread (and the threads it starts) on the GPU's NUMA node
        cpu_binding = None if bound is None else f"{len(bound)} CPUs local to the GPU"
    deadline = Deadline(args.stage_timeout, rank)
    import torch.distributed as dist
    from pytorch_distributed_amd import comm as pcomm
    ddp = world > 1 or args.force_ddp
    comm_kind = "native" if (args.native_comm and args.comm == "auto") else args.comm
    if comm_kind == "auto":
        comm_kind = "torch" if (on_cpu or args.engine == "stock") else "native"
    if on_cpu and comm_kind != "torch":
        comm_kind = "torch"
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        with deadline.stage("rendezvous"):
            to = datetime.timedelta(seconds=args.stage_timeout)
            if on_cpu:
                pcomm.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
            elif comm_kind == "torch":
                pcomm.init_process_group("nccl", rank=rank, world_size=world, timeout=to, device_id=dev)
            else:
                # control plane on gloo (CPU tensors), data plane on this package's own communicator;
                # torch only creates an RCCL communicator of its own if something hands it a device tensor
                pcomm.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world, timeout=to)

    watchdog = None
    if ddp and not on_cpu or (ddp and os.environ.get("PDA_BENCH_WATCHDOG", "") == "1"):
        # heartbeats through the rendezvous store (+ the communicator's asynchronous error state once it exists): a
        # rank that dies ends the job with a non-zero exit code instead of leaving the others in their next collective
        from pytorch_distributed_amd.comm.watchdog import start_watchdog

        def _peer_failure(reason):
            print(json.dumps({"error": f"rank {rank}: watchdog: {reason}", "stage": "watchdog", "rank": rank}),
                  flush=True)
            os._exit(75)
        watchdog = start_watchdog(on_failure=_peer_failure)

    from pytorch_distributed_amd import models, ops, amp, backends
    backends.benchmark = bool(args.autotune) and args.warmup > 0   # the tuning runs belong to the warm-up steps
    from pytorch_distributed_amd.optim import FusedSGD

    torch.manual_seed(1234 + rank)
    B, S = args.batch_size, args.image_size
    images = torch.randn(B, 3, S, S, device=dev)
    target = torch.randint(0, args.num_classes, (B,), device=dev)
    cdtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.precision]
    comm_obj = None
    comm_info = {"comm_backend": None, "rccl_nranks": None, "wire_dtype": None, "buckets": None,
                 "bucket_mb": args.bucket_mb if ddp else None}

    def make_native_comm():
        """This package's RCCL communicator, created and proven with one bounded collective before the model
        depends on it.  Every rank votes (over the gloo control plane) on whether it works; a single failing or
        stalling rank sends the whole job to torch.distributed's process group instead, and the JSON says so."""
        if "native_kind" in comm_info:
            return comm_info["native_kind"]
        ok, why, c = 1, "", None
        with deadline.stage("RCCL communicator creation", factor=1.0):
            try:
                if os.environ.get("PDA_BENCH_FAIL_NATIVE_COMM", "") == "1":   # test hook: exercise the fallback
                    raise RuntimeError("native communicator disabled by PDA_BENCH_FAIL_NATIVE_COMM")
                c = pcomm.device_communicator(dev)
                if c is None:
                    raise RuntimeError("no communicator")
                if c.nranks() != world:
                    raise RuntimeError(f"ncclCommCount = {c.nranks()}, expected {world}")
                probe = torch.full((1 << 16,), float(rank + 1), device=dev)
                w = c.all_reduce(probe, "sum", async_op=True)
                done = w.synchronize(min(60.0, args.stage_timeout)) if hasattr(w, "synchronize") else True
                if not done:
                    raise TimeoutError("first all-reduce did not complete in time")
                w.wait()
                torch.cuda.synchronize()
                expect = world * (world + 1) / 2
                if abs(float(probe[0].item()) - expect) > 1e-3 or abs(float(probe[-1].item()) - expect) > 1e-3:
                    raise RuntimeError(f"first all-reduce returned {float(probe[0].item())}, expected {expect}")
            except Exception as e:  # noqa: BLE001 - any failure here is a reason to fall back, never to hang
                ok, why = 0, f"{type(e).__name__}: {e}"[:300]
        with deadline.stage("communicator vote"):
            v = torch.tensor([ok], dtype=torch.int32)
            if world > 1:
                dist.all_reduce(v, op=dist.ReduceOp.MIN)
        if int(v.item()) == 1:
            comm_info["native_kind"] = c
            if watchdog is not None:
                watchdog.comm = c
            return c
        comm_info["native_kind"] = "torch"
        comm_info["comm_fallback_reason"] = why or "another rank could not use the native communicator"
        print(f"[bench] rank {rank}: native RCCL communicator unusable ({comm_info['comm_fallback_reason']}); "
              "falling back to torch.distributed", file=sys.stderr, flush=True)
        if c is not None:
            try:
                c.abort()
            except Exception:  # noqa: BLE001
                pass
            pcomm._STATE["rccl"].clear()
        return "torch"

    def build_native(bucket_mb=None):
        nonlocal comm_obj
        bucket_mb = args.bucket_mb if bucket_mb is None else bucket_mb
        model = models.get_model(args.arch, num_classes=args.num_classes).to(dev)
        if args.sync_bn:
            from pytorch_distributed_amd.nn import SyncBatchNorm
            SyncBatchNorm.convert_sync_batchnorm(model)
        opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        level = "O0" if args.precision == "fp32" else args.opt_level
        model, opt = amp.initialize(model, opt, opt_level=level, dtype=cdtype if level != "O0" else torch.float16,
                                    verbosity=0)
        if ddp:
            from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP
            kind = comm_kind
            if comm_obj is not None:
                kind = comm_obj             # a later model of this run (bucket sweep): the same communicator
            elif kind == "native":
                kind = make_native_comm()
            with deadline.stage("parameter broadcast"):
                model = DDP(model, device_ids=None if on_cpu else [local_rank], bucket_cap_mb=bucket_mb,
                            force_comm=args.force_ddp, comm=kind, wire_dtype=args.wire_dtype,
                            fused_wire=True)   # nothing here reads param.grad between backward and step
                comm_obj = model._coll
                if not on_cpu:
                    model.measure_exposed_comm(True)
                comm_info["comm_backend"] = getattr(comm_obj, "backend", type(comm_obj).__name__)
                if hasattr(comm_obj, "nranks"):
                    comm_info["rccl_nranks"] = comm_obj.nranks()
                    if comm_info["rccl_nranks"] != world:
                        raise RuntimeError(f"RCCL communicator has {comm_info['rccl_nranks']} ranks, expected {world}")
                comm_info["wire_dtype"] = str(model.wire_dtype).replace("torch.", "") if model.wire_dtype else \
                    str(model._grad_dtype).replace("torch.", "")
                if not on_cpu:
                    torch.cuda.synchronize()
        metrics = ops.new_metrics(dev)

        def step():
            model.zero_grad() if ddp else opt.zero_grad(set_to_none=False)
            loss = ops.cross_entropy(model(images), target, metrics)
            with amp.scale_loss(loss, opt) as scaled:
                scaled.backward()
            opt.step()
            return loss
        return model, step

    hvd_info = {}

    def build_hvd():
        """The loop of /root/reference/5.horovod_distributed.py:116-125,219-221 on this package's hvd module."""
        nonlocal comm_obj
        if share_gpu and world > 1:
            # ranks sharing one GPU: the engine's device collectives go through the peer-memory backend
            os.environ.update(PDA_HVD_DEVICE_COMM="p2p", PDA_HVD_P2P_DEVICE="0", PDA_HVD_NO_CUDA="1")
        import pytorch_distributed_amd.hvd as hvd
        with deadline.stage("hvd.init"):
            hvd.init()
            if hvd.size() != world:
                raise RuntimeError(f"hvd.size() = {hvd.size()}, expected {world}")
        model = models.get_model(args.arch, num_classes=args.num_classes).to(dev)
        if args.sync_bn:
            from pytorch_distributed_amd.nn import SyncBatchNorm
            SyncBatchNorm.convert_sync_batchnorm(model)
        with deadline.stage("hvd.broadcast_parameters"):
            hvd.broadcast_parameters(model.state_dict(), root_rank=0)
        opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        level = "O0" if args.precision == "fp32" else args.opt_level
        model, opt = amp.initialize(model, opt, opt_level=level, dtype=cdtype if level != "O0" else torch.float16,
                                    verbosity=0)
        with deadline.stage("hvd.broadcast_optimizer_state"):
            hvd.broadcast_optimizer_state(opt, root_rank=0)
        compression = {"fp16": hvd.Compression.fp16, "bf16": hvd.Compression.bf16}.get(
            args.precision if args.wire_dtype == "auto" else args.wire_dtype, hvd.Compression.none)
        opt = hvd.DistributedOptimizer(opt, named_parameters=model.named_parameters(), compression=compression)
        from pytorch_distributed_amd.hvd import mpi_ops as _M
        dc = _M._device_coll(dev) if world > 1 and not on_cpu else None
        comm_obj = dc if (dc is not None and hasattr(dc, "nranks")) else None
        comm_info["comm_backend"] = "hvd engine / " + (getattr(dc, "backend", type(dc).__name__) if dc is not None
                                                       else ("shm" if world > 1 else "single process"))
        comm_info["wire_dtype"] = getattr(compression, "wire", None) and str(compression.wire).replace("torch.", "")
        if watchdog is not None and comm_obj is not None:
            watchdog.comm = comm_obj
        hvd_info["hvd"] = hvd
        metrics = ops.new_metrics(dev)

        def step():
            opt.zero_grad()
            loss = ops.cross_entropy(model(images), target, metrics)
            with amp.scale_loss(loss, opt) as scaled:
                scaled.backward()
            opt.step()
            return loss
        return model, step

    def selftest(coll):
        """Every collective of the data plane once, against closed-form values, before anything is timed."""
        import math
        w = world
        tri = w * (w + 1) / 2
        tests = []
        for dt_, name in ((torch.float32, "fp32"), (torch.float16, "fp16")):
            for n, sz in ((256 if dt_ == torch.float32 else 512, "1KB"), ((8 << 20) if dt_ == torch.float32 else (16 << 20), "32MB")):
                tests.append((f"all_reduce {name} {sz}", dt_, n))
        for label, dt_, n in tests:
            with deadline.stage("selftest " + label):
                t = torch.full((n,), float(rank + 1), device=dev, dtype=dt_)
                t[-1] = float(rank)         # the last element differs from the rest: a shifted slice shows
                coll.all_reduce(t, "sum")
                if not on_cpu:
                    torch.cuda.synchronize()
                got = (float(t[0]), float(t[n // 2]), float(t[-1]), float(t.float().sum() - float(t[-1])))
                want = (tri, tri, w * (w - 1) / 2, tri * (n - 1))
                if not all(math.isclose(g, e, rel_tol=2e-3, abs_tol=1e-3) for g, e in zip(got, want)):
                    return f"{label}: got {got}, expected {want}"
        with deadline.stage("selftest broadcast"):
            t = torch.arange(4099, device=dev, dtype=torch.float32) * (1.0 if rank == 0 else -1.0)
            coll.broadcast(t, 0)
            if not torch.equal(t.cpu(), torch.arange(4099, dtype=torch.float32)):
                return "broadcast fp32 4099 elements from rank 0: wrong values"
        with deadline.stage("selftest all_gather"):
            mine = torch.full((1000,), float(rank), device=dev)
            out = torch.empty(w * 1000, device=dev)
            coll.all_gather(out, mine)
            want = torch.arange(w, dtype=torch.float32).repeat_interleave(1000)
            if not torch.equal(out.cpu(), want):
                return "all_gather fp32 1000 elements per rank: wrong values"
        with deadline.stage("selftest barrier"):
            try:
                coll.barrier(timeout_s=min(60.0, args.stage_timeout)) if hasattr(coll, "nranks") else coll.barrier()
            except Exception as e:  # noqa: BLE001
                return f"barrier: {type(e).__name__}: {e}"
        return None

    def run_eval(model):
        """K forward-only steps in eval mode; the loss / correct@1 / correct@5 / row sums stay on the device and
        cross the ranks in one fused all-reduce at the end."""
        metrics = ops.new_metrics(dev)
        model.eval()
        with torch.no_grad():
            for _ in range(min(3, args.eval_steps)):
                ops.cross_entropy(model(images), target, metrics)
            metrics.zero_()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.eval_steps):
                ops.cross_entropy(model(images), target, metrics)
            if "hvd" in hvd_info:
                red = hvd_info["hvd"].allreduce(metrics, name="eval.metrics", op=hvd_info["hvd"].Sum)
            elif world > 1:
                red = pcomm.small_all_reduce(metrics.clone(), "sum")
            else:
                red = metrics
            vals = [float(v) for v in red.tolist()]      # the one host read of the evaluation
            fence()
            dt = time.perf_counter() - t0
        model.train()
        rows = max(vals[3], 1.0)
        return {"eval_images_per_sec": round(world * B * args.eval_steps / dt, 2), "eval_steps": args.eval_steps,
                "eval_loss": round(vals[0] / rows, 4), "eval_acc1": round(vals[1] / rows, 4),
                "eval_acc5": round(vals[2] / rows, 4), "eval_rows": int(vals[3]),
                "eval_metric_allreduces": 1 if world > 1 else 0}

    def build_stock(channels_last: bool):
        model = models.torch_resnet(args.arch, args.num_classes).to(dev)
        x = images
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
            x = images.contiguous(memory_format=torch.channels_last)
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        scaler = torch.amp.GradScaler("cuda", enabled=(cdtype == torch.float16 and not on_cpu))
        if ddp:
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=None if on_cpu else [local_rank])
            comm_info["comm_backend"] = "torch.distributed:" + dist.get_backend()
        crit = torch.nn.CrossEntropyLoss()

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cpu" if on_cpu else "cuda", dtype=cdtype if not on_cpu else torch.bfloat16,
                                enabled=(cdtype != torch.float32)):
                loss = crit(model(x), target)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            return loss
        return model, step

    def fence():
        if world > 1:
            if comm_obj is not None and hasattr(comm_obj, "nranks"):
                comm_obj.barrier(timeout_s=args.stage_timeout)    # device all-reduce on the data-plane communicator
            elif not on_cpu and comm_kind != "torch":
                # peer-memory data plane: drain the device, then meet on the gloo control plane (a default
                # dist.barrier() would make torch build an RCCL communicator of its own)
                torch.cuda.synchronize()
                dist.all_reduce(torch.zeros(1))
            else:
                dist.barrier()
        if not on_cpu:
            torch.cuda.synchronize()

    host_dt = [0.0]

    def timed(step):
        with deadline.stage("warm-up", factor=2.0):
            loss = None
            for _ in range(args.warmup):
                loss = step()
            fence()
        with deadline.stage("timed loop", factor=2.0):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            host_dt[0] = time.perf_counter() - t0     # launch-path time: the host is done issuing, the device is not
            fence()
            dt = time.perf_counter() - t0
        with deadline.stage("result reduce"):
            tmax = torch.tensor([dt], dtype=torch.float64)
            if world > 1:
                if dist.get_backend() == "nccl":
                    tmax = tmax.to(dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item()), float(loss.item())

    sweep = None

    def buckets_of(m):
        return [{"mb": round(b["bytes"] / 2 ** 20, 2), "wire_mb": round(b["wire_bytes"] / 2 ** 20, 2),
                 "params": b["params"]} for b in m.bucket_summary()]

    if args.engine == "native":
        model, step = build_native(args.bucket_list[0])
    elif args.engine == "hvd":
        model, step = build_hvd()
    else:
        model, step = build_stock(args.channels_last)
    if args.selftest and world > 1:
        coll = comm_obj
        if coll is None and args.engine == "stock":
            from pytorch_distributed_amd.comm.rccl import TorchCollectives
            coll = TorchCollectives()
        if coll is None and args.engine == "hvd":
            from pytorch_distributed_amd.hvd import mpi_ops as _M
            coll = _M._device_coll(dev)
        bad = selftest(coll) if coll is not None else "no data-plane communicator to test"
        flag = torch.tensor([0 if bad is None else 1], dtype=torch.int32)
        with deadline.stage("selftest vote"):
            if dist.is_initialized():
                dist.all_reduce(flag.to(dev) if dist.get_backend() == "nccl" else flag, op=dist.ReduceOp.MAX)
        if bad is not None:
            print(json.dumps({"error": f"rank {rank}: selftest failed: {bad}", "stage": "selftest", "rank": rank,
                              "comm_backend": comm_info["comm_backend"]}), flush=True)
            return 4
        if int(flag.item()) != 0:
            return 4
        comm_info["selftest"] = "passed: all_reduce fp32/fp16 1KB/32MB, broadcast, all_gather, barrier"
    if ddp and args.engine == "native":
        comm_info["buckets"] = buckets_of(model)
    graphed = False
    if args.graph and args.engine == "native" and not ddp:
        from pytorch_distributed_amd.utils.graph import GraphedStep
        step = GraphedStep(step, warmup=min(3, max(args.warmup, 1)))
        graphed = step.captured
    elif args.graph and args.engine == "native" and ddp:
        from pytorch_distributed_amd.utils.graph import GraphedStep
        with deadline.stage("graph capture", factor=2.0):
            step = GraphedStep(step, warmup=max(3, min(args.warmup, 6)), ddp=model)
        graphed = step.captured

    dt, final_loss = timed(step)
    exposed = None
    if ddp and args.engine == "native":
        comm_info["buckets"] = buckets_of(model)      # after the arrival-order rebuild
        if not on_cpu:
            exposed = model.exposed_comm_ms(last=args.steps)
    if ddp and args.engine == "native" and len(args.bucket_list) > 1:
        # the other bucket sizes, same process, same communicator: the headline is the best of them
        sweep = [{"bucket_mb": args.bucket_list[0], "ms_per_step": round(1000 * dt / args.steps, 3),
                  "exposed_comm_ms_per_step": None if exposed is None else round(exposed, 3),
                  "buckets": len(comm_info["buckets"])}]
        best = (dt, final_loss, exposed, comm_info["buckets"], args.bucket_list[0])
        import gc
        for mb in args.bucket_list[1:]:
            del model, step
            gc.collect()
            if not on_cpu:
                torch.cuda.empty_cache()
            model, step = build_native(mb)
            dti, li = timed(step)
            exi = model.exposed_comm_ms(last=args.steps) if not on_cpu else None
            sweep.append({"bucket_mb": mb, "ms_per_step": round(1000 * dti / args.steps, 3),
                          "exposed_comm_ms_per_step": None if exi is None else round(exi, 3),
                          "buckets": len(model.bucket_summary())})
            if dti < best[0]:
                best = (dti, li, exi, buckets_of(model), mb)
        dt, final_loss, exposed, comm_info["buckets"], comm_info["bucket_mb"] = best
    eval_info = None
    if args.eval_steps > 0 and args.engine in ("native", "hvd"):
        with deadline.stage("evaluation", factor=2.0):
            eval_info = run_eval(model)
    ips = world * B * args.steps / dt
    host_ms = 1000 * host_dt[0] / max(args.steps, 1)
    vs_stock = stock_ips = None
    engine_stats = hvd_info["hvd"].engine_stats() if "hvd" in hvd_info else None
    if args.compare_stock and args.engine == "native" and not on_cpu:
        del model, step
        import gc
        gc.collect()               # the native model must be gone, not merely unreferenced, before the comparator runs
        torch.cuda.empty_cache()
        # The comparator runs in a child process on the same GPU right after the native run: whatever happens to it -
        # an exception, a crash inside a library, a hang - the native result below is still printed.
        with deadline.stage("stock comparator (child process)", factor=2.0):
            stock_ips = _stock_in_child(args, B, S, min(deadline.seconds * 1.5, 420.0))
        if isinstance(stock_ips, float) and stock_ips > 0:
            vs_stock = ips / stock_ips
    if rank == 0:
        out = {
            "metric": ("images/sec (whole node) ResNet-50 fp16 training, synthetic ImageNet 224x224"
                       if (args.arch, args.precision, S) == ("resnet50", "fp16", 224) else
                       f"images/sec (whole node) {args.arch} {args.precision} training, synthetic {S}x{S}"),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "vs_stock": None if vs_stock is None else round(vs_stock, 3),
            "config": {"model": args.arch, "global_batch": world * B, "per_gpu_batch": B, "image_size": S,
                       "seq_len": None, "parallelism": f"dp{world}", "engine": args.engine,
                       "amp_opt_level": args.opt_level if args.precision != "fp32" else "O0",
                       "optimizer": ("momentum-SGD (fused)" if args.engine == "native" else
                                     "hvd.DistributedOptimizer(momentum-SGD (fused))" if args.engine == "hvd" else
                                     "torch.optim.SGD"),
                       "loss_scaling": "dynamic" if args.precision == "fp16" else "none",
                       "device": ("cpu (rehearsal, not a result)" if on_cpu else
                                  torch.cuda.get_device_name(dev) + (f" ({world} ranks sharing one GPU: rehearsal, "
                                                                     "not a result)" if share_gpu and world > 1 else "")),
                       "weights": "random-init",
                       "conv_autotune": backends.benchmark if args.engine in ("native", "hvd") else None,
                       "final_loss": round(final_loss, 4), "hip_graph": graphed, "cpu_binding": cpu_binding,
                       "host_issue_ms_per_step": round(host_ms, 3),
                       "exposed_comm_ms_per_step": None if exposed is None else round(exposed, 3),
                       "comm_backend": comm_info["comm_backend"], "rccl_nranks": comm_info["rccl_nranks"],
                       "wire_dtype": comm_info["wire_dtype"], "bucket_mb": comm_info["bucket_mb"],
                       "buckets": comm_info["buckets"],
                       "comm_fallback_reason": comm_info.get("comm_fallback_reason"),
                       "selftest": comm_info.get("selftest"), "bucket_sweep": sweep,
                       "sync_bn": bool(args.sync_bn), "hvd_engine_stats": engine_stats,
                       "watchdog": watchdog is not None,
                       "stock_images_per_sec": (round(stock_ips, 2) if isinstance(stock_ips, float) else stock_ips),
                       "stock_comparator": ("child process on the same GPU right after the native run, same model/batch/"
                                            "steps: torchvision-topology ResNet, channels_last, torch.autocast + "
                                            "GradScaler + torch.optim.SGD"
                                            if args.compare_stock else None),
                       "baseline_note": "reference publishes no number; vs_baseline is null"},
        }
        if eval_info is not None:
            out.update({"eval_images_per_sec": eval_info["eval_images_per_sec"]})
            out["config"]["eval"] = eval_info
        print(json.dumps(out), flush=True)
    # The result is out: nothing after this line may turn the run into a failure (a peer that is a little late,
    # a communicator that takes its time to go away) - a timer ends the process with exit code 0 if the orderly
    # teardown below does not finish
    def _bail():
        time.sleep(45.0)
        sys.stdout.flush()
        os._exit(0)
    threading.Thread(target=_bail, daemon=True).start()
    deadline.seconds = max(deadline.seconds, 3600.0)
    if watchdog is not None:
        from pytorch_distributed_amd.comm.watchdog import stop_watchdog
        stop_watchdog()
    if "hvd" in hvd_info:
        with deadline.stage("hvd shutdown"):
            hvd_info["hvd"].shutdown()
    if ddp:
        with deadline.stage("teardown"):
            pcomm.destroy_process_group()
    return 0


def _stock_in_child(args, batch: int, size: int, timeout_s: float):
    """images/sec of `--engine stock` (same model, batch, steps, precision; channels-last autocast + GradScaler +
    torch.optim.SGD) measured by a child process, or a string saying why there is no number."""
    cmd = [sys.executable, os.path.abspath(__file__), "--engine", "stock", "--channels-last", "--gpus", "1",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--arch", args.arch, "--batch-size", str(batch),
           "--image-size", str(size), "--num-classes", str(args.num_classes), "--precision", args.precision,
           "--opt-level", args.opt_level, "--no-compare-stock"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE",
                                                             "MASTER_PORT", "GROUP_RANK", "ROLE_RANK")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return f"failed: no result within {timeout_s:.0f} s"
    except OSError as e:
        return f"failed: {type(e).__name__}: {e}"[:200]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    try:
        d = json.loads(lines[-1])
        return float(d["value"]) if "value" in d else f"failed: {d.get('error', 'no value')}"[:200]
    except (IndexError, ValueError):
        return f"failed: rc {r.returncode}: {r.stderr.strip()[-160:]}"


def _keep_stdout_for_json() -> None:
    """Native libraries write to file descriptor 1 as well (RCCL's version banner at communicator creation, gloo's
    connection notes): send the descriptor to stderr and keep the real stdout for what this script prints - the JSON
    line."""
    try:
        sys.stdout.flush()
        real = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(real, "w", buffering=1)
    except OSError:
        pass


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return supervise(args, argv)
    _keep_stdout_for_json()
    try:
        return worker(args)
    except Exception as e:  # noqa: BLE001 - a failed run must still say why, in the one place the driver reads
        import traceback
        traceback.print_exc()
        print(json.dumps({"error": f"rank {os.environ.get('RANK', '0')}: {type(e).__name__}: {e}"[:500]}), flush=True)
        return 1


if __name__ == "__main__":
    rc = main()
    sys.stdout.flush()
    sys.stderr.flush()
    # ranks of a failed multi-rank job must not wait in interpreter teardown for peers that are gone
    os._exit(rc) if rc != 0 and "WORLD_SIZE" in os.environ else sys.exit(rc)
