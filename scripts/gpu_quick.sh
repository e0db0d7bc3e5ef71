#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
for b in ${BATCHES:-128 256}; do
timeout -k 10 300 python -u bench.py --batch-size $b --steps 30 --warmup 8 > gpurun_out/bench_native_b$b.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b$b.log; exit 1; }
echo "native b$b $(tail -n 1 gpurun_out/bench_native_b$b.log | cut -c95-125) $(tail -n 1 gpurun_out/bench_native_b$b.log | grep -o '"ms_per_step": [0-9.]*')"
done
