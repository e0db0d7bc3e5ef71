"""Horovod-compatible API (``import pytorch_distributed_amd.hvd as hvd``)."""
from .mpi_ops import (init, shutdown, is_initialized, size, rank, local_rank, local_size, cross_rank, cross_size,
                      nccl_built, rccl_built, mpi_built, gloo_built, rocm_built, cuda_built, mpi_threads_supported,
                      allreduce, allreduce_, allreduce_async, allreduce_async_, grouped_allreduce, broadcast,
                      broadcast_, broadcast_async, broadcast_async_, allgather, barrier, join, synchronize, poll,
                      Compression, Average, Sum, Adasum, Min, Max, Product, HorovodInternalError, engine_stats)
from .functions import broadcast_parameters, broadcast_optimizer_state, broadcast_object
from .optimizer import DistributedOptimizer
from ..nn import SyncBatchNorm

__all__ = [n for n in dir() if not n.startswith("_")]
