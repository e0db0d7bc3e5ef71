#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear or Linear or split_k" --timeout 120 --timeout-method thread > gpurun_out/linear_tests.log 2>&1 || { tail -n 30 gpurun_out/linear_tests.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/linear_tests.log
bash scripts/gpu_ab_bench.sh noksplit
