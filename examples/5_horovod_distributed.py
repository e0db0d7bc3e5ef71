#!/usr/bin/env python
"""Launch style 5 - Horovod API.  Counterpart of /root/reference/5.horovod_distributed.py:

    python -m pytorch_distributed_amd.launch -np 4 -H localhost:4 5_horovod_distributed.py

``hvd.init`` -> pin GPU to ``hvd.local_rank()`` -> ``broadcast_parameters`` / ``broadcast_optimizer_state``
-> ``DistributedOptimizer(..., compression=Compression.fp16)`` -> samplers with explicit
``num_replicas=hvd.size(), rank=hvd.rank()``.  Checkpoints use the un-prefixed file names and the plain
model's state_dict, as upstream.  Metric averaging uses the value RETURNED by ``hvd.allreduce``
(the upstream helper drops it, /root/reference/5.horovod_distributed.py:70-75)."""
import _common  # noqa: F401
import torch

import pytorch_distributed_amd.hvd as hvd
from pytorch_distributed_amd import engine, utils
from pytorch_distributed_amd.data import DistributedSampler


def main():
    args = engine.build_parser("Horovod-style training", epochs=10).parse_args()
    utils.seed_everything(args.seed)
    hvd.init()
    use_cuda = torch.cuda.is_available() and not args.cpu
    device = torch.device("cuda", hvd.local_rank()) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
        utils.bind_to_gpu(device.index)   # host threads on the GPU's NUMA node
    model = engine.make_model(args, device)
    per_rank = int(args.batch_size / hvd.size())
    hvd.broadcast_parameters(model.state_dict(), root_rank=0)
    # fp32 (the upstream script): the stock optimizer under the Horovod wrapper.  --precision fp16 / bf16: this
    # package's fused optimizer, whose update kernel unscales and checks for overflow AFTER the wrapper has
    # synchronised the (still scaled) gradients - the order the Horovod + apex recipe asks for
    mixed = args.precision != "fp32"
    optimizer = engine.make_optimizer(args, model, fused=mixed)
    if mixed:
        model, optimizer = engine.setup_amp(args, model, optimizer)
    hvd.broadcast_optimizer_state(optimizer, root_rank=0)
    optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters(),
                                         compression=hvd.Compression.fp16)
    train_set, val_set = engine.make_datasets(args)
    samplers = (DistributedSampler(train_set, num_replicas=hvd.size(), rank=hvd.rank()),
                DistributedSampler(val_set, num_replicas=hvd.size(), rank=hvd.rank()))
    engine.fit(args, model, optimizer, device, train_set, val_set, per_rank, distributed=True, samplers=samplers,
               save_prefix_arch=False, is_chief=(hvd.rank() == 0))
    hvd.shutdown()


if __name__ == "__main__":
    main()
