#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "ddp or communicator" > gpurun_out/pytest_ddp.log 2>&1 || { tail -n 30 gpurun_out/pytest_ddp.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/pytest_ddp.log
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 > gpurun_out/bench_plain.log 2>&1 || { tail -n 20 gpurun_out/bench_plain.log; exit 2; }
echo "plain $(tail -n 1 gpurun_out/bench_plain.log | grep -o '"value": [0-9.]*')"
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --force-ddp > gpurun_out/bench_ddp1.log 2>&1 || { tail -n 20 gpurun_out/bench_ddp1.log; exit 2; }
echo "force-ddp $(tail -n 1 gpurun_out/bench_ddp1.log | grep -o '"value": [0-9.]*')"
