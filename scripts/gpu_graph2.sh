#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 6 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
for prec in fp32 fp16; do
for g in "" "--graph"; do
timeout -k 10 300 python -u bench.py $g --arch resnet18 --image-size 32 --batch-size 800 --precision $prec --steps 40 --warmup 10 > gpurun_out/bench_r18c_native_$prec$g.log 2>&1 || { tail -n 20 gpurun_out/bench_r18c_native_$prec$g.log; exit 1; }
echo "r18 cifar b800 native $prec $g $(tail -n 1 gpurun_out/bench_r18c_native_$prec$g.log | grep -o '"value": [0-9.]*') $(tail -n 1 gpurun_out/bench_r18c_native_$prec$g.log | grep -o '"ms_per_step": [0-9.]*')"
done
timeout -k 10 300 python -u bench.py --engine stock --arch resnet18 --image-size 32 --batch-size 800 --precision $prec --steps 40 --warmup 10 > gpurun_out/bench_r18c_stock_nchw_$prec.log 2>&1
echo "r18 cifar b800 stock nchw $prec $(tail -n 1 gpurun_out/bench_r18c_stock_nchw_$prec.log | grep -o '"value": [0-9.]*')"
done
timeout -k 10 300 python -u bench.py --steps 20 --warmup 6 > gpurun_out/bench_native_default.log 2>&1 || { tail -n 20 gpurun_out/bench_native_default.log; exit 1; }
echo "r50 default $(tail -n 1 gpurun_out/bench_native_default.log | grep -o '"value": [0-9.]*')"
