#!/usr/bin/env python
"""Launch style 6 - SLURM multi-node.  Counterpart of /root/reference/6.distributed_slurm_main.py:

    srun -N2 --gres gpu:8 python 6_distributed_slurm.py --dist-file dist_file

One SLURM task per node spawns one worker per local GPU; rendezvous is ``file://<dist-file>.<jobid>``
on the shared filesystem.  Unlike upstream, the world size is tasks x GPUs-per-task (upstream uses
the task count alone, which only works with one GPU per node) and only global rank 0 writes
checkpoints and the epoch CSV.  Off-cluster the SLURM variables default to a single node, so the
script also runs stand-alone.

    ... 6_distributed_slurm.py --dataset imagenet --data /path/to/imagenet -a resnet50 -b 3200

reads ``/path/to/imagenet/{train,val}/<class>/*.JPEG`` through this package's ImageFolder with
RandomResizedCrop(224) + flip / Resize(256) + CenterCrop(224) (upstream lines 126-159); the loader ships
uint8 batches and the model's stem kernel normalises them on the GPU."""
import os

import _common  # noqa: F401
import torch

from pytorch_distributed_amd import comm, engine, utils
from pytorch_distributed_amd.launch import spawn
from pytorch_distributed_amd.parallel import DistributedDataParallel


def main_worker(gpu, ngpus_per_node, args):
    info = comm.slurm_world(ngpus_per_node, gpu)
    use_cuda = torch.cuda.is_available() and not args.cpu
    comm.init_process_group("nccl" if use_cuda else "gloo", args.dist_url, world_size=info.world_size,
                            rank=info.rank)
    device = torch.device("cuda", gpu) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
        utils.bind_to_gpu(device.index)   # host threads on the GPU's NUMA node
    model = engine.make_model(args, device)
    optimizer = engine.make_optimizer(args, model)
    model, optimizer = engine.setup_amp(args, model, optimizer)
    model = DistributedDataParallel(model, device_ids=[gpu] if use_cuda else None, bucket_cap_mb=args.bucket_mb)
    train_set, val_set = engine.make_datasets(args)
    engine.fit(args, model, optimizer, device, train_set, val_set, int(args.batch_size / ngpus_per_node),
               distributed=True, zero_grad=model.zero_grad, csv_path="distributed.csv", save_prefix_arch=False)
    comm.destroy_process_group()


def main():
    parser = engine.build_parser("SLURM multi-node DDP training", epochs=90, dist_file_flag=True)
    parser.add_argument("--ngpus", type=int, default=0, help="workers per node (default: visible GPUs)")
    args = parser.parse_args()
    utils.seed_everything(args.seed)
    ngpus = args.ngpus or (torch.cuda.device_count() if torch.cuda.is_available() and not args.cpu else 2)
    dist_file = os.path.realpath(args.dist_file or "dist_file")
    args.dist_url = "file://{}.{}".format(dist_file, os.environ.get("SLURM_JOBID", str(os.getpid())))
    spawn(main_worker, nprocs=ngpus, args=(ngpus, args))


if __name__ == "__main__":
    main()
