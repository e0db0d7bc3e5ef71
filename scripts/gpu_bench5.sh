#!/bin/bash
# GPU tests + wgrad sweep (tile / pixel-step / split count) + end-to-end bench.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 15 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --only w --wcfgs 1,2,9,10 --wblocks 256,512,768,1024 --out gpurun_out/wgrad_sweep_b128.csv > gpurun_out/wgrad_sweep_b128.log 2>&1 || { tail -n 20 gpurun_out/wgrad_sweep_b128.log; exit 1; }
tail -n 1 gpurun_out/wgrad_sweep_b128.log
timeout -k 10 300 python -u bench.py --batch-size 128 --steps 30 --warmup 8 > gpurun_out/bench_native_b128.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b128.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b128.log | cut -c1-200
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 > gpurun_out/bench_native_default.log 2>&1 || { tail -n 20 gpurun_out/bench_native_default.log; exit 1; }
tail -n 1 gpurun_out/bench_native_default.log | cut -c1-200
