#!/bin/bash
# default benchmark with alternative builds of the kernel library: bash scripts/gpu_ab_bench.sh <variant> [<variant> ...]
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in default "$@"; do
    if [ $lib = default ]; then unset PDA_KERNELS_LIB; else export PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_$lib.so; fi
    timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/abb_$lib.log 2>&1 || { tail -n 20 gpurun_out/abb_$lib.log; exit 1; }
    echo "$lib: $(tail -n 1 gpurun_out/abb_$lib.log | grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": 20, "warmup": 10, "ms_per_step": [0-9.]*')"
  done
done
