#!/bin/bash
# model-zoo GPU tests (non-ResNet families on the HIP ops)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_model_zoo_gpu.py -x -v --tb=short --timeout 120 --timeout-method thread > gpurun_out/zoo.log 2>&1 || { tail -n 50 gpurun_out/zoo.log | cut -c1-600; exit 1; }
tail -n 25 gpurun_out/zoo.log | cut -c1-300
