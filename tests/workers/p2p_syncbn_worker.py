"""SyncBatchNorm whose statistics travel through the peer-memory one-shot all-reduce: ranks share GPU 0,
the process group itself is gloo (RCCL refuses two ranks on one device).  argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PDA_P2P_SMALL"] = "1"
import torch

from pytorch_distributed_amd import comm
from pytorch_distributed_amd import nn as pnn


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    comm.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    C, per = 32, 6
    full = torch.randn(world * per, 5, 5, C, device=dev)
    g_full = torch.randn(world * per, 5, 5, C, device=dev)
    bn = pnn.SyncBatchNorm(C).to(dev)
    ref = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref.load_state_dict(bn.state_dict())
    x = full[rank * per:(rank + 1) * per].clone().requires_grad_(True)
    out = bn(x, None, True)
    out.backward(g_full[rank * per:(rank + 1) * per])
    xr = full.clone().requires_grad_(True)
    o = torch.relu(ref(xr.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    o.backward(g_full)
    sl = slice(rank * per, (rank + 1) * per)
    err_o = (out - o[sl]).abs().max().item()
    err_g = (x.grad - xr.grad[sl]).abs().max().item()
    err_rv = (bn.running_var - ref.running_var).abs().max().item()
    p2p = comm._STATE["p2p"]
    used = p2p is not None and int(p2p._seq_state[0]) >= 2
    if used:
        p2p.check()
    ok = err_o < 1e-4 and err_g < 1e-4 and err_rv < 1e-4 and used
    print(f"rank {rank}: {'OK' if ok else 'FAILED'} out {err_o:.2e} grad {err_g:.2e} rv {err_rv:.2e} p2p_used {used}",
          flush=True)
    comm.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
