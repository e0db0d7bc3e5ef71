#!/usr/bin/env python
"""The Horovod MNIST example (/root/reference/5.2.horovod_pytorch_mnist.py), CPU-capable:

    python 5_2_horovod_mnist.py --no-cuda --epochs 1
    python -m pytorch_distributed_amd.launch -np 2 -H localhost:2 5_2_horovod_mnist.py --no-cuda

Same flags and flow as upstream (per-rank data directory, LR scaled by ``hvd.size()``,
``broadcast_parameters`` / ``broadcast_optimizer_state``, ``DistributedOptimizer`` with fp16 compression,
``op=Adasum|Average`` and ``gradient_predivide_factor``); ``--fp16-allreduce`` can actually be switched
off here (``--no-fp16-allreduce``).  MNIST idx files are read from ``data-<rank>/`` when present;
without them (this environment has no network) a synthetic set of the same shape is used."""
import argparse

import _common  # noqa: F401
import torch
import torch.nn.functional as F

import pytorch_distributed_amd.hvd as hvd
from pytorch_distributed_amd import models, utils
from pytorch_distributed_amd.data import DistributedSampler, mnist

parser = argparse.ArgumentParser(description="MNIST Example (Horovod-style API)")
parser.add_argument("--batch-size", type=int, default=64, metavar="N")
parser.add_argument("--test-batch-size", type=int, default=1000, metavar="N")
parser.add_argument("--epochs", type=int, default=10, metavar="N")
parser.add_argument("--lr", type=float, default=0.01, metavar="LR")
parser.add_argument("--momentum", type=float, default=0.5, metavar="M")
parser.add_argument("--no-cuda", action="store_true", default=False)
parser.add_argument("--seed", type=int, default=42, metavar="S")
parser.add_argument("--log-interval", type=int, default=10, metavar="N")
parser.add_argument("--fp16-allreduce", action=argparse.BooleanOptionalAction, default=True)
parser.add_argument("--use-adasum", action="store_true", default=False)
parser.add_argument("--gradient-predivide-factor", type=float, default=1.0)
parser.add_argument("--max-batches", type=int, default=0, help="cap batches per epoch (0 = all)")


def metric_average(val, name):
    tensor = torch.tensor(val)
    avg_tensor = hvd.allreduce(tensor, name=name)    # average is performed in 'allreduce', once
    return avg_tensor.item()


def main():
    args = parser.parse_args()
    args.cuda = not args.no_cuda and torch.cuda.is_available()
    hvd.init()
    torch.manual_seed(args.seed)
    if args.cuda:
        torch.cuda.set_device(hvd.local_rank())
        utils.bind_to_gpu(hvd.local_rank())   # host threads on the GPU's NUMA node
        torch.cuda.manual_seed(args.seed)
    torch.set_num_threads(1)
    device = torch.device("cuda", hvd.local_rank()) if args.cuda else torch.device("cpu")
    train_dataset = mnist("data-%d" % hvd.rank(), train=True)
    test_dataset = mnist("data-%d" % hvd.rank(), train=False)
    train_sampler = DistributedSampler(train_dataset, num_replicas=hvd.size(), rank=hvd.rank())
    test_sampler = DistributedSampler(test_dataset, num_replicas=hvd.size(), rank=hvd.rank())
    train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=args.batch_size, sampler=train_sampler)
    test_loader = torch.utils.data.DataLoader(test_dataset, batch_size=args.test_batch_size, sampler=test_sampler)
    model = models.Net().to(device)
    lr_scaler = hvd.size() if not args.use_adasum else 1
    if args.cuda and args.use_adasum and hvd.nccl_built():
        lr_scaler = hvd.local_size()
    optimizer = torch.optim.SGD(model.parameters(), lr=args.lr * lr_scaler, momentum=args.momentum)
    hvd.broadcast_parameters(model.state_dict(), root_rank=0)
    hvd.broadcast_optimizer_state(optimizer, root_rank=0)
    compression = hvd.Compression.fp16 if args.fp16_allreduce else hvd.Compression.none
    optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters(),
                                         compression=compression, op=hvd.Adasum if args.use_adasum else hvd.Average,
                                         gradient_predivide_factor=args.gradient_predivide_factor)

    def train(epoch):
        model.train()
        train_sampler.set_epoch(epoch)
        for batch_idx, (data, target) in enumerate(train_loader):
            if args.max_batches and batch_idx >= args.max_batches:
                break
            data, target = data.to(device), target.to(device)
            optimizer.zero_grad()
            loss = F.nll_loss(model(data), target)
            loss.backward()
            optimizer.step()
            if batch_idx % args.log_interval == 0:
                print("Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {:.6f}".format(
                    epoch, batch_idx * len(data), len(train_sampler), 100. * batch_idx / len(train_loader),
                    loss.item()), flush=True)

    def test():
        model.eval()
        test_loss, test_accuracy = 0., 0.
        with torch.no_grad():
            for i, (data, target) in enumerate(test_loader):
                if args.max_batches and i >= args.max_batches:
                    break
                data, target = data.to(device), target.to(device)
                output = model(data)
                test_loss += F.nll_loss(output, target, reduction="sum").item()
                pred = output.max(1, keepdim=True)[1]
                test_accuracy += pred.eq(target.view_as(pred)).cpu().float().sum().item()
        test_loss /= len(test_sampler)
        test_accuracy /= len(test_sampler)
        test_loss = metric_average(test_loss, "avg_loss")
        test_accuracy = metric_average(test_accuracy, "avg_accuracy")
        if hvd.rank() == 0:
            print("\nTest set: Average loss: {:.4f}, Accuracy: {:.2f}%\n".format(test_loss, 100. * test_accuracy),
                  flush=True)

    for epoch in range(1, args.epochs + 1):
        train(epoch)
        test()
    hvd.shutdown()


if __name__ == "__main__":
    main()
