#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "stem or pool or compress or many_tiles or relu_backward or dgrad_mask or dgrad_bn or resnet or conv or ddp or graph or eval or batchnorm or sync" > gpurun_out/pytest_relufuse.log 2>&1 || { tail -n 30 gpurun_out/pytest_relufuse.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/pytest_relufuse.log
for b in 256 400; do
timeout -k 10 300 python -u bench.py --batch-size $b --steps 30 --warmup 8 > gpurun_out/bench_rf_b$b.log 2>&1 || { tail -n 20 gpurun_out/bench_rf_b$b.log; exit 2; }
echo "native b$b $(tail -n 1 gpurun_out/bench_rf_b$b.log | grep -o '"value": [0-9.]*')"
done
