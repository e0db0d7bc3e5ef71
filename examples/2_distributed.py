#!/usr/bin/env python
"""Launch style 2 - external launcher + DistributedDataParallel.  Counterpart of
/root/reference/2.distributed.py:

    python -m pytorch_distributed_amd.launch --nproc_per_node=4 --use_argv_local_rank 2_distributed.py
    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 2_distributed.py

env:// rendezvous, one process per GPU, global batch divided by the number of processes, bucketed
RCCL all-reduce overlapped with backward, sharded (distributed) evaluation, rank-0 checkpoints."""
import _common  # noqa: F401
import torch

from pytorch_distributed_amd import comm, engine, utils
from pytorch_distributed_amd.parallel import DistributedDataParallel


def main():
    args = engine.build_parser("DDP training (launcher)", local_rank_flag=True).parse_args()
    utils.seed_everything(args.seed)
    use_cuda = torch.cuda.is_available() and not args.cpu
    info = comm.init_process_group("nccl" if use_cuda else "gloo", "env://")
    local_rank = info.local_rank if args.local_rank < 0 else args.local_rank
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
        utils.bind_to_gpu(device.index)   # host threads on the GPU's NUMA node
    if info.rank == 0:
        print("=> creating model '{}'".format(args.arch))
    model = engine.make_model(args, device)
    optimizer = engine.make_optimizer(args, model)
    model, optimizer = engine.setup_amp(args, model, optimizer)
    per_rank = int(args.batch_size / info.world_size)
    model = DistributedDataParallel(model, device_ids=[local_rank] if use_cuda else None,
                                    bucket_cap_mb=args.bucket_mb)
    train_set, val_set = engine.make_datasets(args)
    engine.fit(args, model, optimizer, device, train_set, val_set, per_rank, distributed=True,
               zero_grad=model.zero_grad)
    comm.destroy_process_group()


if __name__ == "__main__":
    main()
