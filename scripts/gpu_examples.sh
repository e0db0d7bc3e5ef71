#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_examples_gpu.py -m gpu -x -q --tb=short --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu_examples.log 2>&1
rc=$?; tail -n 40 gpurun_out/pytest_gpu_examples.log | cut -c1-1500; exit $rc
