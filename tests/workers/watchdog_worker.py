"""Two ranks over gloo with the job watchdog on (env: RANK / WORLD_SIZE / MASTER_*).  Rank 1 exits hard after a
few steps; rank 0 then waits the way a rank waits inside a device collective whose peer is gone (the reference's
loop would sit in /root/reference/2.distributed.py:219-223 forever) - the watchdog has to end it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

from pytorch_distributed_amd import comm
from pytorch_distributed_amd.comm.watchdog import start_watchdog

info = comm.init_process_group("gloo")
wd = start_watchdog(interval=0.25)
assert wd is not None
for step in range(3):
    t = torch.ones(4)
    dist.all_reduce(t)
    assert float(t[0]) == 2.0
    time.sleep(0.3)
if info.rank == 1:
    print("rank 1: dying", flush=True)
    os._exit(9)
t0 = time.time()
while time.time() - t0 < 60:      # stands for a collective that never completes
    time.sleep(0.1)
print("rank 0: HUNG for 60 s, nobody ended the job", flush=True)
sys.exit(1)
