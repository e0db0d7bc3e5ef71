"""pytorch_distributed_amd: an MI355X-native (gfx950 / CDNA4) data-parallel training
framework - DataParallel, DistributedDataParallel, spawn/launch/SLURM bootstrap,
apex-style mixed precision, a Horovod-compatible API and distributed evaluation -
with the ResNet hot path on hand-written HIP kernels."""
__version__ = "0.1.0"

from . import ops, nn, models  # noqa: F401
