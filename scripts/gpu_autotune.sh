#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "wgrad or fprop_dgrad or resnet_fwd" > gpurun_out/pytest_at.log 2>&1 || { tail -n 30 gpurun_out/pytest_at.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/pytest_at.log
for f in "--no-autotune" "--autotune"; do
timeout -k 10 400 python -u bench.py --steps 30 --warmup 10 $f > gpurun_out/bench_at$f.log 2>&1 || { tail -n 20 gpurun_out/bench_at$f.log; exit 2; }
echo "$f $(tail -n 1 gpurun_out/bench_at$f.log | grep -o '"value": [0-9.]*')"
done
