#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_p2p_gpu.py -m gpu -x -q --tb=short --timeout 300 --timeout-method thread > gpurun_out/pytest_p2p.log 2>&1
rc=$?; tail -n 30 gpurun_out/pytest_p2p.log | cut -c1-600; exit $rc
