#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "conv or dgrad or wgrad" > gpurun_out/pytest_gpu_conv.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_gpu_conv.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --only dw --cfgs 0,16 --wcfgs 0,3,8 --out gpurun_out/conv_bench_b128_v3.csv > gpurun_out/conv_bench_b128_v3.log 2>&1 || { tail -n 20 gpurun_out/conv_bench_b128_v3.log; exit 1; }
tail -n 1 gpurun_out/conv_bench_b128_v3.log
timeout -k 10 300 python -u bench.py --batch-size 128 --steps 30 --warmup 8 > gpurun_out/bench_native_b128.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b128.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b128.log | cut -c1-200
for b in 192 256 384; do
timeout -k 10 300 python -u bench.py --batch-size $b --steps 20 --warmup 6 > gpurun_out/bench_native_b$b.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b$b.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b$b.log | cut -c1-200
done
