#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "$1" > gpurun_out/pytest_gpu_k.log 2>&1
rc=$?; tail -n 25 gpurun_out/pytest_gpu_k.log | cut -c1-300; exit $rc
