"""Process launchers: ``spawn`` (mp.spawn semantics) and the ``python -m
pytorch_distributed_amd.launch`` CLI (torch.distributed.launch / horovodrun / srun styles)."""
from .spawn import spawn, start_processes, ProcessContext, ProcessRaisedException, ProcessExitedException

__all__ = ["spawn", "start_processes", "ProcessContext", "ProcessRaisedException", "ProcessExitedException"]
