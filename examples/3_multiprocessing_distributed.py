#!/usr/bin/env python
"""Launch style 3 - the script spawns its own workers (``python 3_multiprocessing_distributed.py``).
Counterpart of /root/reference/3.multiprocessing_distributed.py: ``spawn`` + tcp:// rendezvous
(default tcp://127.0.0.1:23456 as upstream, ``--dist-url`` to change it) + DistributedDataParallel."""
import _common  # noqa: F401
import torch

from pytorch_distributed_amd import comm, engine, utils
from pytorch_distributed_amd.launch import spawn
from pytorch_distributed_amd.parallel import DistributedDataParallel


def main_worker(local_rank, nprocs, args):
    utils.seed_everything(args.seed, warn=(local_rank == 0))       # seeding happens inside each worker
    use_cuda = torch.cuda.is_available() and not args.cpu
    comm.init_process_group("nccl" if use_cuda else "gloo", args.dist_url, world_size=nprocs, rank=local_rank)
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
        utils.bind_to_gpu(device.index)   # host threads on the GPU's NUMA node
    model = engine.make_model(args, device)
    optimizer = engine.make_optimizer(args, model)
    model, optimizer = engine.setup_amp(args, model, optimizer)
    model = DistributedDataParallel(model, device_ids=[local_rank] if use_cuda else None,
                                    bucket_cap_mb=args.bucket_mb)
    train_set, val_set = engine.make_datasets(args)
    engine.fit(args, model, optimizer, device, train_set, val_set, int(args.batch_size / nprocs), distributed=True,
               zero_grad=model.zero_grad)
    comm.destroy_process_group()


def main():
    parser = engine.build_parser("DDP training (self-spawned workers)", epochs=10)
    parser.add_argument("--dist-url", default="tcp://127.0.0.1:23456")
    parser.add_argument("--nprocs", type=int, default=0, help="workers to spawn (default: visible GPUs)")
    args = parser.parse_args()
    nprocs = args.nprocs or (torch.cuda.device_count() if torch.cuda.is_available() and not args.cpu else 2)
    spawn(main_worker, nprocs=nprocs, args=(nprocs, args))


if __name__ == "__main__":
    main()
