#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for eng in native stock; do
for prec in fp32 fp16; do
extra=""; [ $eng = stock ] && extra="--channels-last"
timeout -k 10 300 python -u bench.py --engine $eng $extra --arch resnet18 --image-size 32 --batch-size 800 --precision $prec --steps 40 --warmup 10 > gpurun_out/bench_r18c_${eng}_$prec.log 2>&1 || { tail -n 20 gpurun_out/bench_r18c_${eng}_$prec.log; exit 1; }
echo "r18 cifar b800 $eng $prec $(tail -n 1 gpurun_out/bench_r18c_${eng}_$prec.log | cut -c95-125) $(tail -n 1 gpurun_out/bench_r18c_${eng}_$prec.log | grep -o '"ms_per_step": [0-9.]*')"
done
done
timeout -k 10 300 python -u bench.py --engine stock --arch resnet18 --image-size 32 --batch-size 800 --precision fp32 --steps 40 --warmup 10 > gpurun_out/bench_r18c_stock_nchw_fp32.log 2>&1
echo "r18 cifar b800 stock nchw fp32 $(tail -n 1 gpurun_out/bench_r18c_stock_nchw_fp32.log | cut -c95-125)"
