#!/bin/bash
# ResNet-18, CIFAR-shaped 32x32 input, batch 800 (the reference's default workload), native engine
set -o pipefail
mkdir -p gpurun_out
for cfg in "fp32" "fp32 --graph" "fp16" "fp16 --graph"; do
tag=$(echo $cfg | tr -d ' -')
timeout -k 10 300 python -u bench.py --arch resnet18 --image-size 32 --batch-size 800 --precision $cfg --steps 60 --warmup 15 > gpurun_out/bench_r18c_$tag.log 2>&1 || { tail -n 20 gpurun_out/bench_r18c_$tag.log; exit 2; }
echo "r18 cifar b800 native $cfg: $(tail -n 1 gpurun_out/bench_r18c_$tag.log | grep -o '"value": [0-9.]*') $(tail -n 1 gpurun_out/bench_r18c_$tag.log | grep -o '"ms_per_step": [0-9.]*')"
done
