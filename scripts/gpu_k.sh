#!/bin/bash
# kernel tests selected with -k "<expr>" (second argument: test file, default tests/test_kernels_gpu.py)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest ${2:-tests/test_kernels_gpu.py} -q --timeout 300 --timeout-method thread -k "$1" > gpurun_out/k_pytest.log 2>&1; rc=$?
tail -${3:-40} gpurun_out/k_pytest.log
exit $rc
