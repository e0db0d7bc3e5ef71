#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "conv or dgrad or wgrad or resnet or arena" > gpurun_out/pytest_gpu_conv.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_gpu_conv.log; [ $rc -ne 0 ] && exit $rc
for ws in 64 0; do
PDA_WGRAD_WORKSPACE_MB=$ws timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --only w --wcfgs 0 --wblocks 512,768,1024 > gpurun_out/wgrad_ws$ws.log 2>&1 || { tail -n 20 gpurun_out/wgrad_ws$ws.log; exit 1; }
echo "ws=$ws $(tail -n 1 gpurun_out/wgrad_ws$ws.log)"
done
for rep in 1 2; do
for ws in 64 0; do
PDA_WGRAD_WORKSPACE_MB=$ws timeout -k 10 300 python -u bench.py --batch-size 128 --steps 30 --warmup 8 > gpurun_out/bench_ab_ws$ws.log 2>&1 || { tail -n 20 gpurun_out/bench_ab_ws$ws.log; exit 1; }
echo "b128 ws=$ws $(tail -n 1 gpurun_out/bench_ab_ws$ws.log | cut -c95-125)"
done
done
for ws in 64 0; do
PDA_WGRAD_WORKSPACE_MB=$ws timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 > gpurun_out/bench_ab_ws$ws.log 2>&1 || { tail -n 20 gpurun_out/bench_ab_ws$ws.log; exit 1; }
echo "b256 ws=$ws $(tail -n 1 gpurun_out/bench_ab_ws$ws.log | cut -c95-125)"
done
