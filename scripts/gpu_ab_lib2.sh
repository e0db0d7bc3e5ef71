#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V=$1; shift
PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_$V.so timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "conv or dgrad or wgrad or resnet" > gpurun_out/pytest_gpu_variant.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_gpu_variant.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for lib in default $V; do
if [ $lib = default ]; then unset PDA_KERNELS_LIB; else export PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_$lib.so; fi
timeout -k 10 300 python -u benchmarks/conv_bench.py --batch 128 "$@" > gpurun_out/ab_$lib.log 2>&1 || { tail -n 20 gpurun_out/ab_$lib.log; exit 1; }
echo "$lib: $(tail -n 1 gpurun_out/ab_$lib.log)"
done
done
for lib in default $V; do
if [ $lib = default ]; then unset PDA_KERNELS_LIB; else export PDA_KERNELS_LIB=$GRAFT_REPO_ROOT/pytorch_distributed_amd/lib/libpda_kernels_$lib.so; fi
timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 > gpurun_out/ab_bench_$lib.log 2>&1 || { tail -n 20 gpurun_out/ab_bench_$lib.log; exit 1; }
echo "$lib b256: $(tail -n 1 gpurun_out/ab_bench_$lib.log | grep -o '"value": [0-9.]*')"
done
