from .meters import AverageMeter, ProgressMeter, ThroughputMeter, DeviceMetrics, StepTimer
from .metrics import accuracy, accuracy_topk, reduce_mean, reduce_mean_many
from .checkpoint import save_checkpoint, load_checkpoint, build_state, strip_module_prefix, add_module_prefix
from .misc import adjust_learning_rate, seed_everything, EpochCSV, JsonlLogger
from .tracing import annotate, range_push, range_pop
from .graph import GraphedStep
from .affinity import bind_to_gpu, gpu_local_cpus, parse_cpulist

__all__ = ["bind_to_gpu", "gpu_local_cpus", "parse_cpulist", "AverageMeter", "ProgressMeter", "ThroughputMeter", "DeviceMetrics", "StepTimer", "accuracy",
           "accuracy_topk", "reduce_mean", "reduce_mean_many", "save_checkpoint", "load_checkpoint", "build_state",
           "strip_module_prefix", "add_module_prefix", "adjust_learning_rate", "seed_everything", "EpochCSV",
           "JsonlLogger", "annotate", "range_push", "range_pop", "GraphedStep"]
