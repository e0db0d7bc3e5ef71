"""Multi-process tests on the CPU backends (gloo / shared memory): the stand-in for multi-GPU runs."""
import os
import socket
import subprocess
import sys
import threading
import time

import pytest
import torch

from pytorch_distributed_amd import comm
from pytorch_distributed_amd.comm import TCPStore, FileStore, PrefixStore, StoreTimeout, store_barrier, Heartbeat
from pytorch_distributed_amd.launch import spawn, ProcessRaisedException

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "workers"))
import dist_workers as W  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_tcp_store_semantics():
    port = free_port()
    master = TCPStore("127.0.0.1", port, 2, is_master=True, timeout=5)
    client = TCPStore("127.0.0.1", master.port, 2, is_master=False, timeout=5)
    master.set("a", b"1")
    assert client.get("a") == b"1"
    assert client.add("cnt", 2) == 2 and master.add("cnt", 3) == 5
    assert client.check(["a", "cnt"]) and not client.check("missing")
    with pytest.raises(StoreTimeout):
        client.get("missing", timeout=0.2)
    threading.Timer(0.2, lambda: master.set("late", b"x" * 10000)).start()
    assert client.get("late", timeout=5) == b"x" * 10000           # blocking get + large value
    assert client.compare_set("cas", b"", b"v1") == b"v1"
    assert client.compare_set("cas", b"nope", b"v2") == b"v1"
    assert master.delete_key("a") and not master.check("a")
    assert master.num_keys() >= 3
    pre = PrefixStore("p", client)
    pre.set("k", "v")
    assert master.get("p/k") == b"v"
    done = []
    ths = [threading.Thread(target=lambda s=s: (store_barrier(s, 2, "b", 5), done.append(1))) for s in (master, client)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert len(done) == 2
    hb = Heartbeat(master, 0, interval=0.05)
    time.sleep(0.2)
    assert Heartbeat.dead_ranks(client, 2, max_age=5) == [1]
    hb.stop()
    client.close()
    master.close()


def test_file_store_semantics(tmp_path):
    a = FileStore(str(tmp_path / "rdv"), 2, timeout=5)
    b = FileStore(str(tmp_path / "rdv"), 2, timeout=5)
    a.set("x", b"42")
    assert b.get("x") == b"42"
    assert a.add("n", 1) == 1 and b.add("n", 1) == 2
    with pytest.raises(StoreTimeout):
        b.get("nope", timeout=0.1)
    assert b.delete_key("x") and not a.check("x")


def test_rank_discovery(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HOROVOD_RANK", "SLURM_PROCID", "OMPI_COMM_WORLD_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert comm.discover().world_size == 1
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("LOCAL_RANK", "3")
    i = comm.discover()
    assert (i.rank, i.world_size, i.local_rank, i.source) == (3, 8, 3, "env")
    monkeypatch.delenv("RANK"); monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setenv("HOROVOD_RANK", "1"); monkeypatch.setenv("HOROVOD_SIZE", "4")
    monkeypatch.setenv("HOROVOD_LOCAL_RANK", "1"); monkeypatch.setenv("HOROVOD_LOCAL_SIZE", "4")
    assert comm.discover().source == "horovod"
    monkeypatch.delenv("HOROVOD_RANK")
    # simulated 2-node SLURM job, one task per node spawning 4 workers each (script 6's scheme, fixed maths)
    monkeypatch.setenv("SLURM_PROCID", "1"); monkeypatch.setenv("SLURM_NPROCS", "2"); monkeypatch.setenv("SLURM_NNODES", "2")
    monkeypatch.setenv("SLURM_JOB_NODELIST", "node[07-08]")
    i = comm.slurm_world(gpus_per_task=4, local_gpu=2)
    assert (i.rank, i.world_size, i.node_rank, i.master_addr) == (6, 8, 1, "node07")


def test_launcher_env_and_argv_contract(tmp_path):
    script = tmp_path / "probe.py"
    script.write_text(
        "import os,sys\n"
        "open(os.path.join(os.path.dirname(__file__), 'out%s.txt' % os.environ['RANK']),'w').write(' '.join([os.environ[k] for k in "
        "('RANK','WORLD_SIZE','LOCAL_RANK','MASTER_ADDR','MASTER_PORT','HOROVOD_RANK','HOROVOD_SIZE')] + sys.argv[1:]))\n")
    rc = subprocess.run([sys.executable, "-m", "pytorch_distributed_amd.launch", "--nproc_per_node", "3",
                         "--master_port", "29876", "--use_argv_local_rank", str(script), "--foo"], cwd=ROOT).returncode
    assert rc == 0
    for r in range(3):
        assert (tmp_path / f"out{r}.txt").read_text() == f"{r} 3 {r} 127.0.0.1 29876 {r} 3 --foo --local_rank={r}"
    bad = tmp_path / "bad.py"
    bad.write_text("import os,sys,time\nsys.exit(7) if os.environ['RANK']=='1' else time.sleep(30)\n")
    t0 = time.time()
    rc = subprocess.run([sys.executable, "-m", "pytorch_distributed_amd.launch", "-np", "2", "-H", "localhost:2",
                         str(bad)], cwd=ROOT, capture_output=True).returncode
    assert rc == 7 and time.time() - t0 < 20        # first failure tears the job down with its exit code


def test_spawn_propagates_child_exception(tmp_path):
    marker = str(tmp_path / "m")
    t0 = time.time()
    with pytest.raises(ProcessRaisedException) as e:
        spawn(W.crasher, args=(marker,), nprocs=2)
    assert "boom from rank 1" in str(e.value) and e.value.error_index == 1
    assert time.time() - t0 < 30 and not os.path.exists(marker)     # sibling was terminated


def test_shm_collectives_three_ranks():
    spawn(W.shm_collectives, args=(3, free_port()), nprocs=3)


@pytest.mark.parametrize("wire", [False, True])
def test_ddp_matches_averaged_single_process_grads(wire):
    spawn(W.ddp_equivalence, args=(2, free_port(), wire), nprocs=2)


def test_ddp_param_grad_is_the_reduced_gradient_by_default():
    spawn(W.ddp_grad_semantics, args=(2, free_port()), nprocs=2)


def test_ddp_native_ready_hooks_match_python_hooks():
    spawn(W.ddp_native_hooks, args=(2, free_port()), nprocs=2)


def test_apex_flavoured_ddp():
    spawn(W.apex_flavour, args=(2, free_port()), nprocs=2)


def test_syncbn_matches_full_batch_bn():
    spawn(W.syncbn_equivalence, args=(2, free_port()), nprocs=2)


def test_file_rendezvous(tmp_path):
    spawn(W.file_rendezvous, args=(2, str(tmp_path / "rdv.1234")), nprocs=2)


def test_hvd_api_two_ranks():
    r = subprocess.run([sys.executable, "-m", "pytorch_distributed_amd.launch", "-np", "2", "-H", "localhost:2",
                        "--master_port", str(free_port()), os.path.join(ROOT, "tests", "workers", "hvd_worker.py")],
                       cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("hvd OK") == 2


def test_hvd_engine_with_store_control_plane():
    """Ranks on several nodes: the same engine (name negotiation in any order, fusion, broadcast, all-gather, the
    DistributedOptimizer) with its control plane and host tensors on the TCP store instead of shared memory -
    rehearsed with both ranks on this machine."""
    env = dict(os.environ, PDA_HVD_FORCE_MULTI_HOST="1")
    r = subprocess.run([sys.executable, "-m", "pytorch_distributed_amd.launch", "-np", "2", "-H", "localhost:2",
                        "--master_port", str(free_port()), os.path.join(ROOT, "tests", "workers", "hvd_worker.py")],
                       cwd=ROOT, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("hvd OK") == 2
    assert r.stdout.count("'control_plane': 'store'") == 2


@pytest.mark.parametrize("kind", ["fused", "stock"])
def test_hvd_with_amp_skips_overflowed_step_on_every_rank(kind):
    """scale_loss waits for the wrapper's reductions before its overflow pass: an infinity on one rank makes all
    ranks skip; also the amp-patched step() of a stock optimizer keeps going through synchronize()."""
    r = subprocess.run([sys.executable, "-m", "pytorch_distributed_amd.launch", "-np", "2", "-H", "localhost:2",
                        "--master_port", str(free_port()), os.path.join(ROOT, "tests", "workers", "hvd_amp_worker.py"),
                        kind], cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("hvd amp OK") == 2


def _append_records(i, per, kind, arg):
    st = TCPStore("127.0.0.1", arg, 1, is_master=False, timeout=20) if kind == "tcp" else FileStore(arg, 1, timeout=20)
    for j in range(per):
        st.append("log", f"{i}:{j};")
    st.close()


def test_store_append_is_atomic(tmp_path):
    """Eight writers append to ONE key through their own connections: every record arrives, each writer's in
    order (the c10d adapter's append used to be a read-modify-write that lost concurrent writes).  TCP store:
    threads; file store: processes (its lock is a per-process fcntl lock)."""
    import multiprocessing as mp
    master = TCPStore("127.0.0.1", 0, 1, is_master=True, timeout=20)
    fpath = str(tmp_path / "fstore")
    n, per = 8, 100
    ts = [threading.Thread(target=_append_records, args=(i, per, "tcp", master.port)) for i in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    ctx = mp.get_context("fork")
    ps = [ctx.Process(target=_append_records, args=(i, per, "file", fpath)) for i in range(n)]
    [p.start() for p in ps]
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for reader in (TCPStore("127.0.0.1", master.port, 1, is_master=False, timeout=20), FileStore(fpath, 1, timeout=20)):
        recs = [r for r in reader.get("log").decode().split(";") if r]
        reader.close()
        assert len(recs) == n * per and len(set(recs)) == n * per
        for i in range(n):
            assert [int(r.split(":")[1]) for r in recs if r.startswith(f"{i}:")] == list(range(per))
    ad = comm.C10dStoreAdapter(PrefixStore("c", master))
    ad.append("k", b"ab")
    ad.append("k", "cd")
    assert bytes(ad.get("k")) == b"abcd"
    master.close()
