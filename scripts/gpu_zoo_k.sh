#!/bin/bash
# a subset of the model-zoo GPU tests: bash scripts/gpu_zoo_k.sh "<pytest -k expression>"
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_model_zoo_gpu.py -v --tb=short --timeout 120 --timeout-method thread -k "$1" > gpurun_out/zoo_k.log 2>&1
rc=$?
grep -E "PASSED|FAILED|Error|assert|passed|failed" gpurun_out/zoo_k.log | cut -c1-420 | tail -n 60
exit $rc
