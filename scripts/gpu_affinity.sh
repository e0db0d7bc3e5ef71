#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python - <<'PY'
import os, torch
from pytorch_distributed_amd.utils import affinity as A
print("allowed cpus", len(os.sched_getaffinity(0)), "pci", A.gpu_pci_address(0))
loc = A.gpu_local_cpus(0)
print("local cpus", None if loc is None else len(loc))
print("bound", None if (b := A.bind_to_gpu(0)) is None else len(b))
PY
timeout -k 10 300 python -u bench.py > gpurun_out/bench_bind.log 2>&1 || { tail -n 20 gpurun_out/bench_bind.log; exit 1; }
tail -n 1 gpurun_out/bench_bind.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"host_issue_ms_per_step": [0-9.]*\|"cpu_binding": [^,]*'
PDA_BIND_CPU=0 timeout -k 10 300 python -u bench.py > gpurun_out/bench_nobind.log 2>&1 || { tail -n 20 gpurun_out/bench_nobind.log; exit 1; }
tail -n 1 gpurun_out/bench_nobind.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"host_issue_ms_per_step": [0-9.]*\|"cpu_binding": [^,]*'
timeout -k 10 600 python -u -m pytest tests/test_examples_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/examples_gpu.log 2>&1 || { tail -n 30 gpurun_out/examples_gpu.log | cut -c1-300; exit 2; }
tail -n 2 gpurun_out/examples_gpu.log
