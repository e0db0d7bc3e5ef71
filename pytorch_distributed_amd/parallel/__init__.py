"""Data-parallel engines: multi-process DistributedDataParallel (bucketed RCCL all-reduce
overlapped with backward), its apex-flavoured variant, single-process multi-GPU
DataParallel, and SyncBatchNorm conversion."""
from .distributed import DistributedDataParallel
from .reducer import BucketPlan
from ..nn import SyncBatchNorm


def ApexDistributedDataParallel(module, message_size: int = 10_000_000, delay_allreduce: bool = False,
                                gradient_average: bool = True, gradient_predivide_factor: float = 1.0,
                                allreduce_always_fp32: bool = False, **kw):
    """apex.parallel.DistributedDataParallel(model) flavour (/root/reference/4.apex_distributed2.py:178):
    flat buckets of ``message_size`` elements filled in gradient-arrival order (discovered in the
    first iteration), no per-forward buffer broadcast."""
    kw.setdefault("broadcast_buffers", False)
    return DistributedDataParallel(module, message_size=message_size, delay_allreduce=delay_allreduce,
                                   gradient_average=gradient_average,
                                   gradient_predivide_factor=gradient_predivide_factor, **kw)


def convert_syncbn_model(module, process_group=None):
    """apex.parallel.convert_syncbn_model equivalent."""
    return SyncBatchNorm.convert_sync_batchnorm(module, process_group)


def __getattr__(name):
    if name == "DataParallel":
        from .data_parallel import DataParallel
        return DataParallel
    raise AttributeError(name)


__all__ = ["DistributedDataParallel", "ApexDistributedDataParallel", "DataParallel", "BucketPlan", "SyncBatchNorm",
           "convert_syncbn_model"]
