#!/usr/bin/env python
"""Launch style 1 - single process, several GPUs: ``DataParallel`` (scatter / replicate /
parallel_apply / gather).  Counterpart of /root/reference/1.dataparallel.py (``python 1_dataparallel.py``).
Defaults follow the reference: resnet101, global batch 3200 split by the scatter, 5 epochs, epoch
times appended to dataparallel.csv.  ``--gpus 0,1,2,3`` replaces the hard-coded ``gpus = [0, 1, 2, 3]``."""
import _common  # noqa: F401
import torch

from pytorch_distributed_amd import engine, utils
from pytorch_distributed_amd.parallel import DataParallel


def main():
    parser = engine.build_parser("DataParallel training", arch="resnet101", epochs=5)
    parser.add_argument("--gpus", default=None, help="comma separated device ids (default: all visible)")
    args = parser.parse_args()
    utils.seed_everything(args.seed)
    if args.cpu or not torch.cuda.is_available():
        devices = ["cpu", "cpu"]
    else:
        devices = [int(g) for g in args.gpus.split(",")] if args.gpus else list(range(torch.cuda.device_count()))
    main_dev = torch.device(devices[0]) if devices[0] == "cpu" else torch.device("cuda", devices[0])
    if main_dev.type == "cuda":
        torch.cuda.set_device(main_dev)
    print("=> creating model '{}'".format(args.arch))
    model = engine.make_model(args, main_dev)
    optimizer = engine.make_optimizer(args, model)
    model, optimizer = engine.setup_amp(args, model, optimizer)
    model = DataParallel(model, device_ids=devices, output_device=devices[0])
    train_set, val_set = engine.make_datasets(args)
    engine.fit(args, model, optimizer, main_dev, train_set, val_set, per_rank_batch=args.batch_size,
               distributed=False, csv_path="dataparallel.csv")


if __name__ == "__main__":
    main()
