"""Process bootstrap: rank discovery, rendezvous and process-group creation.

One entry point, ``init_process_group``, covers the reference's three rendezvous styles:
``env://`` under a launcher (/root/reference/2.distributed.py:98), ``tcp://127.0.0.1:23456``
with explicit rank/world (/root/reference/3.multiprocessing_distributed.py:102) and
``file://<path>.<jobid>`` on a shared filesystem under SLURM
(/root/reference/6.distributed_slurm_main.py:93-101).  The key-value rendezvous runs on this
package's native stores (``csrc/runtime/store.cc``), wrapped as a c10d ``Store`` so the RCCL
unique-id exchange and gloo's pairing use them; collectives run on ``torch.distributed``
(backend ``"nccl"`` = RCCL over xGMI on GPUs, ``"gloo"`` on CPU tensors).
``backend="rccl"`` is accepted as an alias of ``"nccl"``.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from .store import TCPStore, FileStore, PrefixStore, Store, StoreTimeout, store_barrier, Heartbeat  # noqa: F401

_STATE = {"store": None, "info": None, "heartbeat": None, "p2p": None, "p2p_tried": False, "rccl": {}, "rccl_seq": 0}


@dataclass
class RankInfo:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0
    local_size: int = 1
    node_rank: int = 0
    num_nodes: int = 1
    master_addr: str = "127.0.0.1"
    master_port: int = 29500
    source: str = "single"


def _first(*names, default=None):
    for n in names:
        v = os.environ.get(n)
        if v not in (None, ""):
            return v
    return default


def discover(local_rank_arg: Optional[int] = None, gpus_per_task: Optional[int] = None) -> RankInfo:
    """Work out who we are from whichever launcher started us.

    Order: torchrun/launch env (RANK/WORLD_SIZE/LOCAL_RANK), Horovod-style env
    (HOROVOD_RANK..., OMPI_COMM_WORLD_*), SLURM (SLURM_PROCID/NTASKS/LOCALID - or, when one
    task per node spawns its own workers as the reference's script 6 does, call
    ``slurm_world(gpus_per_task)``), then a single process.
    """
    e = os.environ
    addr = e.get("MASTER_ADDR", "127.0.0.1")
    port = int(e.get("MASTER_PORT", "29500"))
    if "RANK" in e and "WORLD_SIZE" in e:
        lr = int(e.get("LOCAL_RANK", local_rank_arg if local_rank_arg not in (None, -1) else 0))
        ls = int(e.get("LOCAL_WORLD_SIZE", e.get("WORLD_SIZE")))
        return RankInfo(int(e["RANK"]), int(e["WORLD_SIZE"]), lr, ls, int(e.get("GROUP_RANK", 0)),
                        max(1, int(e["WORLD_SIZE"]) // max(ls, 1)), addr, port, "env")
    r = _first("HOROVOD_RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK")
    if r is not None:
        w = int(_first("HOROVOD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", default="1"))
        lr = int(_first("HOROVOD_LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", default=r))
        ls = int(_first("HOROVOD_LOCAL_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", default=str(w)))
        return RankInfo(int(r), w, lr, ls, int(r) // max(ls, 1), max(1, w // max(ls, 1)), addr, port, "horovod")
    if "SLURM_PROCID" in e and gpus_per_task is None:
        w = int(_first("SLURM_NTASKS", "SLURM_NPROCS", default="1"))
        lr = int(e.get("SLURM_LOCALID", 0))
        nn = int(e.get("SLURM_NNODES", 1))
        return RankInfo(int(e["SLURM_PROCID"]), w, lr, max(1, w // max(nn, 1)), int(e.get("SLURM_NODEID", 0)), nn,
                        _slurm_master(addr), port, "slurm")
    if local_rank_arg not in (None, -1):
        return RankInfo(local_rank_arg, int(e.get("WORLD_SIZE", 1)), local_rank_arg, source="argv")
    return RankInfo(master_addr=addr, master_port=port)


def _slurm_master(default: str) -> str:
    nl = os.environ.get("SLURM_STEP_NODELIST") or os.environ.get("SLURM_JOB_NODELIST") or ""
    if not nl:
        return default
    head = nl.split(",")[0]
    if "[" in head:  # prefix[a-b,...] -> prefixa
        pre, rest = head.split("[", 1)
        head = pre + rest.split("-")[0].split(",")[0].rstrip("]")
    return os.environ.get("MASTER_ADDR", head)


def slurm_world(gpus_per_task: int, local_gpu: int) -> RankInfo:
    """One SLURM task per node that spawns ``gpus_per_task`` workers (script 6's scheme) - with the
    world size computed as tasks x gpus (the reference uses the task count alone, which is only
    right for 1 GPU per node: /root/reference/6.distributed_slurm_main.py:89-100)."""
    e = os.environ
    node = int(e.get("SLURM_PROCID", 0))
    ntasks = int(_first("SLURM_NTASKS", "SLURM_NPROCS", default="1"))
    return RankInfo(node * gpus_per_task + local_gpu, ntasks * gpus_per_task, local_gpu, gpus_per_task, node, ntasks,
                    _slurm_master("127.0.0.1"), int(e.get("MASTER_PORT", "29500")), "slurm-spawn")


_ADAPTER_CLONES: list = []


class C10dStoreAdapter(dist.Store):
    """Expose a native store of this package through the c10d ``Store`` interface."""

    def __init__(self, native):
        super().__init__()
        self._n = native
        self._timeout = datetime.timedelta(seconds=native.timeout)

    def set(self, key, value):
        self._n.set(key, value)

    def get(self, key):
        return self._n.get(key, self._timeout.total_seconds())

    def add(self, key, value):
        return self._n.add(key, value)

    def wait(self, keys, timeout=None):
        t = self._timeout if timeout is None else timeout
        self._n.wait(list(keys), t.total_seconds() if hasattr(t, "total_seconds") else float(t))

    def check(self, keys):
        return self._n.check(list(keys))

    def delete_key(self, key):
        return self._n.delete_key(key)

    def num_keys(self):
        return self._n.num_keys()

    def compare_set(self, key, expected, desired):
        return self._n.compare_set(key, expected, desired)

    def set_timeout(self, timeout):
        self._timeout = timeout

    @property
    def timeout(self):
        return self._timeout

    # The rest of the c10d Store surface: ProcessGroupNCCL clones the store when it builds a communicator lazily
    # (first collective on a device tensor), and newer torch releases probe the extended API.
    def clone(self):
        # the C++ side keeps only the C++ half of the returned object: the Python half (which carries these
        # overrides) has to be kept alive here, or every later call lands on the pure-virtual base
        c = C10dStoreAdapter(self._n)
        c._timeout = self._timeout
        _ADAPTER_CLONES.append(c)
        return c

    def has_extended_api(self):
        return True

    def append(self, key, value):
        self._n.append(key, value if isinstance(value, (bytes, bytearray)) else str(value).encode())

    def multi_get(self, keys):
        return [self.get(k) for k in keys]

    def multi_set(self, keys, values):
        for k, v in zip(keys, values):
            self.set(k, v)

    def list_keys(self):
        lk = getattr(self._n, "list_keys", None)
        if lk is None:
            raise NotImplementedError("list_keys is not part of the native store protocol")
        return lk()


def _make_store(init_method: str, rank: int, world_size: int, timeout_s: float):
    if init_method.startswith("tcp://"):
        hostport = init_method[len("tcp://"):].split("?")[0]
        host, port = hostport.rsplit(":", 1)
        return TCPStore(host, int(port), world_size, is_master=(rank == 0), timeout=timeout_s)
    if init_method.startswith("file://"):
        return FileStore(init_method[len("file://"):], world_size, timeout=timeout_s)
    if init_method.startswith("env://"):
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500"))
        return TCPStore(host, port, world_size, is_master=(rank == 0), timeout=timeout_s)
    raise ValueError(f"unsupported init_method {init_method!r} (use env://, tcp://host:port or file://path)")


def _side_store(info: "RankInfo", timeout_s: float):
    """A native TCPStore next to a process group that was created over somebody else's store: rank 0 binds
    a free port and tells the others through the group (CPU object broadcast)."""
    import socket
    port = [0]
    if info.rank == 0:
        s = socket.socket()
        s.bind(("", 0))
        port[0] = s.getsockname()[1]
        s.close()
    try:
        if "gloo" in str(dist.get_backend()):
            dist.broadcast_object_list(port, src=0)
        else:
            t = torch.tensor(port, dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
            dist.broadcast(t, src=0)
            port[0] = int(t.item())
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        return TCPStore(host, int(port[0]), info.world_size, is_master=(info.rank == 0), timeout=timeout_s)
    except Exception:  # noqa: BLE001 - the side store is a convenience; everything has a group-based fallback
        return None


def init_process_group(backend: Optional[str] = None, init_method: Optional[str] = None, world_size: int = -1,
                       rank: int = -1, timeout: Optional[datetime.timedelta] = None, store=None,
                       device_id: Optional[torch.device] = None, heartbeat: bool = False) -> RankInfo:
    """Create the default process group; returns the resolved ``RankInfo``."""
    info = discover()
    if rank >= 0:
        info.rank = rank
    if world_size > 0:
        info.world_size = world_size
    if backend in (None, ""):
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend.lower() in ("rccl", "nccl"):
        backend = "nccl"
    init_method = init_method or "env://"
    timeout = timeout or datetime.timedelta(seconds=float(os.environ.get("PDA_DIST_TIMEOUT_S", "600")))
    kw = {}
    if device_id is not None and backend == "nccl":
        kw["device_id"] = device_id
    use_native = os.environ.get("PDA_NATIVE_STORE", "1") == "1" and store is None
    # Under torchrun the launcher's own agent already serves a c10d store on MASTER_PORT (the workers are
    # told so with TORCHELASTIC_USE_AGENT_STORE): the port is taken and speaks torch's protocol, so the
    # group rendezvous goes through it and this package's store moves to a port rank 0 picks afterwards.
    agent_store = (init_method.startswith("env://") and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True")
    if use_native and not agent_store:
        native = _make_store(init_method, info.rank, info.world_size, timeout.total_seconds())
        _STATE["store"] = native
        store = C10dStoreAdapter(PrefixStore("c10d", native))
        # torch keeps only the C++ half of this object (inside its own PrefixStore); a communicator that is
        # built lazily - at the first device collective, long after this function returned - still has to find
        # the Python overrides
        _ADAPTER_CLONES.append(store)
        if heartbeat:
            _STATE["heartbeat"] = Heartbeat(PrefixStore("pda", native), info.rank)
    if store is not None:
        dist.init_process_group(backend, store=store, rank=info.rank, world_size=info.world_size, timeout=timeout,
                                **kw)
    else:
        dist.init_process_group(backend, init_method=init_method, rank=info.rank, world_size=info.world_size,
                                timeout=timeout, **kw)
    if use_native and agent_store and info.world_size > 1:
        _STATE["store"] = _side_store(info, timeout.total_seconds())
        if heartbeat and _STATE["store"] is not None:
            _STATE["heartbeat"] = Heartbeat(PrefixStore("pda", _STATE["store"]), info.rank)
    _STATE["info"] = info
    return info


def native_store():
    """The package's rendezvous store behind the default group (None if torch's own was used)."""
    return _STATE["store"]


def rank_info() -> Optional[RankInfo]:
    return _STATE["info"]


def device_communicator(device=None, create: bool = True, group=None):
    """This package's own RCCL communicator for ``device`` over the default group (one per process and
    device, shared by DistributedDataParallel, the hvd engine, SyncBatchNorm statistics and the metric
    reduce).  The unique id travels through the native rendezvous store when ``init_process_group`` of
    this package created one, else through one broadcast over the existing ``torch.distributed`` group.
    Creation is collective: every rank calls it at the same point.  ``None`` when there is no GPU, no
    process group, or ``create`` is false and none exists yet."""
    if not torch.cuda.is_available():
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        return None
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if group is not None and group is not dist.group.WORLD:
        key = (dev.index, id(group))
    else:
        key = (dev.index, None)
        group = None
    c = _STATE["rccl"].get(key)
    if c is not None or not create:
        return c
    if not (dist.is_available() and dist.is_initialized()):
        return None
    from .rccl import RcclCommunicator
    seq = _STATE["rccl_seq"]
    _STATE["rccl_seq"] = seq + 1
    st = _STATE["store"] if group is None else None
    c = RcclCommunicator(dist.get_rank(group), dist.get_world_size(group), dev,
                         store=PrefixStore(f"rccl{seq}", st) if st is not None else None,
                         bootstrap_group=group)
    _STATE["rccl"][key] = c
    return c


def destroy_process_group() -> None:
    for c in list(_STATE["rccl"].values()):
        try:
            c.close()
        except Exception:  # noqa: BLE001 - teardown goes on
            pass
    _STATE["rccl"] = {}
    if _STATE["heartbeat"] is not None:
        _STATE["heartbeat"].stop()
        _STATE["heartbeat"] = None
    if _STATE.get("p2p") is not None:
        _STATE["p2p"].close()   # collective: every rank leaves together, nobody frees a region in use
        _STATE["p2p"] = None
    _STATE["p2p_tried"] = False
    if dist.is_initialized():
        dist.destroy_process_group()
    st = _STATE["store"]
    _STATE["store"] = None
    if st is not None:
        st.close()
    _ADAPTER_CLONES.clear()


def enable_p2p_small_allreduce(max_elems: int = 65536, timeout_ms: int = 2000):
    """Route :func:`small_all_reduce` through the one-shot peer-memory kernel (``comm/p2p.py``) instead of
    an RCCL call.  Collective over the default group; single-node jobs with one process per GPU only.
    Also switched on by ``PDA_P2P_SMALL=1`` at the first :func:`small_all_reduce`."""
    from .p2p import P2PAllReduce
    if _STATE.get("p2p") is None and dist.is_initialized() and dist.get_world_size() > 1:
        st = _STATE["store"]
        _STATE["p2p"] = P2PAllReduce(dist.get_rank(), dist.get_world_size(),
                                     store=PrefixStore("p2p", st) if st is not None else None,
                                     max_elems=max_elems, timeout_ms=timeout_ms)
    return _STATE.get("p2p")


def small_all_reduce(t: torch.Tensor, op: str = "sum", group=None) -> torch.Tensor:
    """In-place all-reduce of a small tensor (metric vector, SyncBatchNorm statistics): the peer-memory
    one-shot kernel when it is enabled and applicable (default group, contiguous fp32 device tensor within
    its message bound), ``torch.distributed`` otherwise.  ``op``: ``"sum"`` or ``"avg"``."""
    if (group is None or group is dist.group.WORLD) and t.is_cuda:
        p2p = _STATE.get("p2p")
        if p2p is None and os.environ.get("PDA_P2P_SMALL", "0") == "1" and not _STATE.get("p2p_tried"):
            _STATE["p2p_tried"] = True
            p2p = enable_p2p_small_allreduce()
        if (p2p is not None and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= p2p.max_elems
                and t.device == p2p.device):
            return p2p.all_reduce(t, op)
        nat = device_communicator(t.device, create=False)
        if nat is not None and t.is_contiguous() and t.dtype in (torch.float32, torch.float16, torch.bfloat16,
                                                                torch.float64, torch.int32, torch.int64):
            nat.all_reduce(t, "sum")
            if op == "avg":
                t /= nat.world_size
            return t
    if t.is_cuda and not _has_device_backend(group):
        # a host-only process group (gloo): stage through the host here rather than hand a device tensor to a
        # backend whose device path is an afterthought (seen to stall with two ranks on one GPU)
        h = t.detach().cpu()
        dist.all_reduce(h, group=group)
        if op == "avg":
            h /= dist.get_world_size(group)
        t.copy_(h)
        return t
    dist.all_reduce(t, group=group)
    if op == "avg":
        t /= dist.get_world_size(group)
    return t


def _has_device_backend(group=None) -> bool:
    try:
        return "nccl" in str(dist.get_backend_config(group)).lower()
    except Exception:  # noqa: BLE001
        return "nccl" in str(dist.get_backend(group)).lower()


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


__all__ = ["RankInfo", "discover", "slurm_world", "init_process_group", "destroy_process_group", "native_store",
           "device_communicator",
           "enable_p2p_small_allreduce", "small_all_reduce",
           "rank_info", "get_rank", "get_world_size", "barrier", "TCPStore", "FileStore", "PrefixStore", "Store",
           "StoreTimeout", "store_barrier", "Heartbeat", "C10dStoreAdapter"]
