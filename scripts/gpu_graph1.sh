#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "graphed" > gpurun_out/pytest_gpu_graph.log 2>&1
rc=$?; tail -n 30 gpurun_out/pytest_gpu_graph.log; [ $rc -ne 0 ] && exit $rc
for prec in fp32 fp16; do
for g in "" "--graph"; do
timeout -k 10 300 python -u bench.py $g --arch resnet18 --image-size 32 --batch-size 800 --precision $prec --steps 40 --warmup 10 > gpurun_out/bench_r18c_native_$prec$g.log 2>&1 || { tail -n 20 gpurun_out/bench_r18c_native_$prec$g.log; exit 1; }
echo "r18 cifar b800 native $prec $g $(tail -n 1 gpurun_out/bench_r18c_native_$prec$g.log | cut -c95-125) $(tail -n 1 gpurun_out/bench_r18c_native_$prec$g.log | grep -o '"ms_per_step": [0-9.]*')"
done
done
for g in "" "--graph"; do
timeout -k 10 300 python -u bench.py $g --batch-size 128 --steps 30 --warmup 8 > gpurun_out/bench_native_b128$g.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b128$g.log; exit 1; }
echo "r50 b128 $g $(tail -n 1 gpurun_out/bench_native_b128$g.log | cut -c95-125)"
done
