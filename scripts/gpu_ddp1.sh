#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "ddp or wgrad" > gpurun_out/pytest_gpu_ddp.log 2>&1
rc=$?; tail -n 25 gpurun_out/pytest_gpu_ddp.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --only w --wcfgs 0 > gpurun_out/wgrad_auto.log 2>&1 || { tail -n 20 gpurun_out/wgrad_auto.log; exit 1; }
tail -n 1 gpurun_out/wgrad_auto.log
timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 > gpurun_out/bench_native_b256.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b256.log; exit 1; }
echo "plain     $(tail -n 1 gpurun_out/bench_native_b256.log | cut -c95-125)"
timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 --force-ddp > gpurun_out/bench_native_b256_ddp.log 2>&1 || { tail -n 30 gpurun_out/bench_native_b256_ddp.log; exit 1; }
echo "force-ddp $(tail -n 1 gpurun_out/bench_native_b256_ddp.log | cut -c95-125)"
timeout -k 10 300 python -u bench.py --steps 20 --warmup 6 > gpurun_out/bench_native_default.log 2>&1 || { tail -n 20 gpurun_out/bench_native_default.log; exit 1; }
echo "default   $(tail -n 1 gpurun_out/bench_native_default.log | cut -c95-125)"
